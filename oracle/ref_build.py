"""Compiles the REFERENCE's own CUDA path for the CPU -> oracle/_ref/libgs_ref.so.

The reference's rasterizer (gs/src/include/*.h under /root/reference) is CUDA-only and there is
no nvcc, CUDA toolkit or NVIDIA GPU here.  Its kernels are plain SIMT C++ though, so they are
compiled with g++ against this repo's own stand-in headers (oracle/emu/cuda/*: vector types,
threadIdx/blockIdx, __syncthreads, atomics, cudaMalloc/..., cub::DeviceRadixSort) on top of the
fiber-based SIMT executor oracle/emu/simt_core.h.  The only thing g++ cannot parse is the
`kernel<<<grid, block>>>(...)` launch syntax: this script reads each header from where it lies,
rewrites exactly those launch expressions into a macro call, writes the result to a scratch
directory under oracle/_ref/ (git-ignored), compiles, and DELETES the scratch copies again --
no reference source enters the repository or travels to the GPU box, only the built .so does.

    python -m oracle.ref_build          # needs /root/reference; a no-op elsewhere
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_INC = "/root/reference/gs/src/include"
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libgs_ref.so")
HEADERS = ["common.h", "data_spec.h", "helper_math.h", "kernels.h", "culling.h", "aabb_culling.h", "vol_render.h",
           "vol_render_scalar.h", "shencoder.h", "vol_render_sh.h", "vol_render_bg.h", "tile_ops.h"]
# `extern __shared__ float name[];` (dynamic shared memory, tile_ops.h) -> a per-block buffer of the emulator
_DYN_SMEM = re.compile(r"extern\s+__shared__\s+(\w+)\s+(\w+)\s*\[\s*\]\s*;")

_LAUNCH = re.compile(r"([A-Za-z_]\w*(?:\s*<[^<>;(){}]*>)?)\s*<<<(.*?)>>>\s*\(", re.S)


def available():
    return os.path.isdir(REF_INC)


def build(force=False):
    if not available():
        raise RuntimeError("/root/reference is not present: oracle/_ref can only be (re)built in the authoring "
                           "container; the prebuilt libgs_ref.so travels with the repo snapshot")
    srcs = [os.path.join(REF_INC, h) for h in HEADERS]
    deps = srcs + [os.path.join(HERE, "emu", f) for f in ("simt_core.h", "ref_driver.cpp", "cuda/cuda_runtime.h",
                                                          "cuda/cub/cub.cuh")] + [__file__]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    gen = os.path.join(OUT_DIR, "gen")
    shutil.rmtree(gen, ignore_errors=True)
    os.makedirs(gen)
    try:
        for h, src in zip(HEADERS, srcs):
            text = open(src).read()
            text = _LAUNCH.sub(lambda m: f"SIMT_LAUNCH(({m.group(1)}), {m.group(2)})(", text)
            text = _DYN_SMEM.sub(lambda m: f"{m.group(1)} *{m.group(2)} = reinterpret_cast<{m.group(1)} *>(simt_dyn_smem());", text)
            open(os.path.join(gen, h), "w").write(text)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w",
               "-I", gen, "-I", os.path.join(HERE, "emu", "cuda"), os.path.join(HERE, "emu", "ref_driver.cpp"),
               "-o", LIB, "-lm"]
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(gen, ignore_errors=True)
    return LIB


def build_if_possible():
    if available():
        return build()
    if os.path.exists(LIB):
        return LIB
    raise RuntimeError("oracle/_ref/libgs_ref.so missing and /root/reference not present")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
