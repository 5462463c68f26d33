"""Builds libgsgen_hip.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m gsgen_amd.build            # incremental
    python -m gsgen_amd.build --force

hipcc cross-compiles without a GPU.  geometry.hip is compiled with -ffp-contract=off (its
fp32 results must be bit-identical to the reference's torch ops, see the file header); the
compositing kernels are allowed to contract.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgsgen_hip.so")
OBJDIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = {
    "composite.hip": [],
    "composite_bwd.hip": [],
    "geometry.hip": ["-ffp-contract=off"],
    "binning.hip": [],
    "legacy.hip": ["-ffp-contract=off"],
}
# -fvisibility=hidden: the dynamic symbol table holds the entry points of include/gsgen_hip.h and nothing else (three
# cross-file helpers used to be exported beside them)
COMMON = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", f"-I{CSRC}"]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src, os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "composite_common.hpp"),
            
            os.path.join(HERE, "..", "include", "gsgen_hip.h"), __file__, *extra]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    objs, rebuilt = [], False
    for name, flags in SOURCES.items():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJDIR, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj):
            cmd = [HIPCC, *COMMON, *flags, *extra_flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


EXT_DIR = os.path.join(HERE, "ext")


def _build_ext(name, source, force=False, verbose=False):
    import sysconfig
    import torch
    lib = build()
    os.makedirs(EXT_DIR, exist_ok=True)
    src = os.path.join(CSRC, source)
    out = os.path.join(EXT_DIR, name + sysconfig.get_config_var("EXT_SUFFIX"))
    if force or _newer(src, out, extra=(lib,)):
        tdir = os.path.dirname(torch.__file__)
        rocm = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME") or "/opt/rocm"
        inc = [os.path.join(tdir, "include"), os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
               os.path.join(rocm, "include"), sysconfig.get_paths()["include"]]
        tlib = os.path.join(tdir, "lib")
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__=1",
               "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-DTORCH_EXTENSION_NAME={name}",
               *[f"-I{i}" for i in inc], src, "-o", out, f"-L{LIBDIR}", "-lgsgen_hip", f"-L{tlib}", "-ltorch", "-ltorch_cpu",
               "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN/../lib", f"-Wl,-rpath,{tlib}"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_torch_ext(force=False, verbose=False):
    """gsgen_amd/ext/_gs.<abi>.so: the `_gs` CPython module (torch::Tensor in, C ABI underneath, gsgen_amd/csrc/
    torch_gs.cpp) -- what the reference builds from gs/src/bindings.cpp.  Plain host C++: compiled with g++ against
    torch's headers and linked to the in-tree HIP library (relative rpath) and torch's own libraries."""
    return _build_ext("_gs", "torch_gs.cpp", force, verbose)


def build_batch_ext(force=False, verbose=False):
    """gsgen_amd/ext/_gsbatch.<abi>.so: the camera batch as one C++ autograd node (gsgen_amd/csrc/torch_batch.cpp), the host side
    of BatchRenderer's fast path.  Optional: without it BatchRenderer runs its Python autograd Functions (same launches)."""
    return _build_ext("_gsbatch", "torch_batch.cpp", force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--ext" in sys.argv:
        print(build_torch_ext(force="--force" in sys.argv, verbose=True))
        print(build_batch_ext(force="--force" in sys.argv, verbose=True))
