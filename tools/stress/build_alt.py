#!/usr/bin/env python
"""Experiment builds of libgsgen_hip.so for tools/stress (the matrix-core chain with different wait disciplines):

    python tools/stress/build_alt.py [name ...]        -> gsgen_amd/lib_alt/<name>/libgsgen_hip.so

    plain   -DGSGEN_MFMA_PLAIN        the chain exactly as hipcc schedules and pads it (no asm statements)
    short   -DGSGEN_MFMA_SHORT_WAITS  operands up front + 16 wait states after the chain (the round-1 build that
                                      failed on the first launches of a process on some boxes)
    fixed   -DGSGEN_MFMA_FIXED_WAITS  16 before + 64 after
    waves3  -DGSGEN_BWD_WAVES3        packed SH backward forced to three wavefronts per SIMD (A/B)
Run a variant with LD_LIBRARY_PATH=gsgen_amd/lib_alt/<name> tools/stress/bwd_stress ... (the tool's RUNPATH comes after
LD_LIBRARY_PATH) or GSGEN_HIP_LIB=... for the Python binding.  Only composite.hip is recompiled."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gsgen_amd import build as B  # noqa: E402

ALT = {"waves3": ["-DGSGEN_BWD_WAVES3"], "plain": ["-DGSGEN_MFMA_PLAIN"], "short": ["-DGSGEN_MFMA_SHORT_WAITS"], "fixed": ["-DGSGEN_MFMA_FIXED_WAITS"]}


def build(name):
    B.build()
    d = os.path.join(ROOT, "gsgen_amd", "lib_alt", name)
    os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, "composite.o")
    src = os.path.join(B.CSRC, "composite.hip")
    out = os.path.join(d, "libgsgen_hip.so")
    if not os.path.exists(out) or B._newer(src, out):
        subprocess.check_call([B.HIPCC, *B.COMMON, *ALT[name], "-c", src, "-o", obj])
        others = [os.path.join(B.OBJDIR, n.replace(".hip", ".o")) for n in B.SOURCES if n != "composite.hip"]
        subprocess.check_call([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out, obj, *others])
    return out


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(ALT)):
        print(build(n))
