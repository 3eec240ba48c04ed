#!/usr/bin/env python3
"""Container-only: cuts the data the figure tests need out of the reference author's published figures
(/root/reference/output_images/, README.md:98, :114, :120) into tests/golden/ref_figures/.  Only pixels are kept (the
image area inside the matplotlib axes), never source text.  See tests/golden/ref_figures/PROVENANCE.md."""
import os
import numpy as np
from PIL import Image

REF = "/root/reference/output_images"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_figures")

# the two thresholded bird's-eye views of test4 (README.md:120): image area 1090 x 1110 at (40, 10); the figures are
# anti-aliased renderings of a {0, 255} image -> stored as 1-bit
for name in ("test4_thresh_bilat", "test4_thresh_cv2adapt"):
    a = np.asarray(Image.open(os.path.join(REF, name + ".png")).convert("L"))[10:10 + 1110, 40:40 + 1090]
    Image.fromarray(np.where(a > 127, 255, 0).astype(np.uint8)).convert("1").save(os.path.join(OUT, name + "_axes.png"), optimize=True)

# the colour-channel comparison (README.md:98): column 1 is test_images/test4.jpg; rows: original, RGB R, RGB G, ..., LAB B
a = np.asarray(Image.open(os.path.join(REF, "color_channels10.png")).convert("RGB"))
x0, x1 = 462, 849
for tag, (y0, y1) in (("original", (27, 245)), ("rgb_r", (285, 503)), ("lab_b", (1318, 1535))):
    Image.fromarray(a[y0:y1, x0:x1]).save(os.path.join(OUT, "color_channels10_test4_%s_panel.png" % tag), optimize=True)
print(sorted(os.listdir(OUT)))

# the demo-3 pair (README.md:142-148): search_lane_result01.png = the annotated camera frame of the first frame of the third
# demo video, drawn at 0.70x (image area 894 x 503 at (33, 11)); search_lane_vis01.png = its bird's-eye mask with the search
# drawn over it (image area 855 x 871 at (40, 10)).  Stored as cut: the pixels inside the axes.
a = np.asarray(Image.open(os.path.join(REF, "search_lane_result01.png")).convert("RGB"))[11:514, 33:927]
Image.fromarray(a).save(os.path.join(OUT, "search_lane_result01_axes.png"), optimize=True)
a = np.asarray(Image.open(os.path.join(REF, "search_lane_vis01.png")).convert("RGB"))[10:881, 40:895]
Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=16).save(os.path.join(OUT, "search_lane_vis01_axes.png"), optimize=True)
print(sorted(os.listdir(OUT)))

# the second pair of the same example (README.md:146-148): the SECOND frame of that video and its band-search visualisation --
# same axes boxes, same kind of data (round 5: the last unused figure pair)
a = np.asarray(Image.open(os.path.join(REF, "search_lane_result02.png")).convert("RGB"))[11:514, 33:927]
Image.fromarray(a).save(os.path.join(OUT, "search_lane_result02_axes.png"), optimize=True)
a = np.asarray(Image.open(os.path.join(REF, "search_lane_vis02.png")).convert("RGB"))[10:881, 40:895]
Image.fromarray(a).convert("P", palette=Image.ADAPTIVE, colors=16).save(os.path.join(OUT, "search_lane_vis02_axes.png"), optimize=True)
print(sorted(os.listdir(OUT)))
