"""DESIGN.md's measured tables are generated from the bench line kept under profiles/ (tools/design_tables.py): the document and
the profile file it cites cannot disagree."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_tables_match_the_kept_bench_line():
    bench = os.path.join(ROOT, "profiles", "r06_bench.json")
    assert os.path.exists(bench), "profiles/r06_bench.json is missing"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), bench, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    for name in ("status", "stream", "settings", "kernels", "process"):
        assert "<!-- bench:%s -->" % name in txt and "<!-- /bench:%s -->" % name in txt, name
