"""Builds tests/fake_rccl.c (the TEST-ONLY stand-in for librccl, see its header) and returns the environment that
makes lt_gather.cpp load it: LT_RCCL_LIB=<the .so>, LT_DEVICE_MODULO=1 (local rank r uses GPU r % visible GPUs, so
two rank processes share GPU 0 on the one-GPU test box)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.c")
OUT = os.path.join(HERE, "libfake_rccl.so")


def build():
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__",
                               "-I", os.path.join(rocm, "include"), SRC, "-L", os.path.join(rocm, "lib"),
                               "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lamdhip64", "-lrt", "-o", OUT])
    return OUT


def env(base=None):
    e = dict(os.environ if base is None else base)
    e.update(LT_RCCL_LIB=build(), LT_DEVICE_MODULO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return e
