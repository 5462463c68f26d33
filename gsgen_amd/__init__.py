"""gsgen_amd -- MI355X (gfx950) differentiable Gaussian-splatting rasterizer, a drop-in for the
`_gs` extension of gsgen3d/gsgen.

    gsgen_amd/ext/_gs    the compiled `_gs` module (csrc/torch_gs.cpp over the C ABI): the reference's pybind surface
    gsgen_amd._gs        ctypes mirror of the same 23 names (no compiler needed at install time)
    gsgen_amd.renderer   HIP projection, fused RGB + heads, the fused `render_frame`
    gsgen_amd.batch      camera batches: one enqueue per stage for all cameras of a batch
    gsgen_amd.dist       camera sharding across GPUs (one process per GPU, RCCL all_gather)
    gsgen_amd.optim      one flat Adam step over the replicated parameters
    gsgen_amd.densify    the reference's densify / prune bookkeeping, identical on every rank (pure torch)
    gsgen_amd.io         the reference's checkpoint / .ply / .splat formats
    gsgen_amd.build      hipcc build of gsgen_amd/lib/libgsgen_hip.so (C ABI: include/gsgen_hip.h)

There is no CPU implementation in this package: every entry point needs the HIP library and
a GPU, and raises if either is missing.
"""
import sys

__version__ = "0.1.0"


def __getattr__(name):  # (lazy: importing the package must not import torch)
    if name == "PairListOverflow":
        from .renderer import PairListOverflow
        return PairListOverflow
    raise AttributeError(name)


def compiled_gs():
    """The compiled `_gs` CPython module (gsgen_amd/ext/_gs.*.so, built by gsgen_amd.build.build_torch_ext from
    csrc/torch_gs.cpp: torch::Tensor in, C ABI underneath -- what the reference builds from gs/src/bindings.cpp), or
    None when it has not been built."""
    import glob
    import importlib.util
    import os
    hits = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ext", "_gs.*.so"))
    if not hits:
        return None
    import torch  # noqa: F401  (its HIP runtime and libtorch go first: one libamdhip64 per process)
    spec = importlib.util.spec_from_file_location("_gs", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_as_gs(compiled=True):
    """Register this package's `_gs` as the top-level module `_gs`, which is what the reference's Python imports
    (`import _gs as _backend`, gs/renderer.py:20-24): the compiled extension when it is built (compiled=True, ~5 us of
    host time per call), else the ctypes mirror gsgen_amd._gs (same names, argument orders and errors)."""
    mod = compiled_gs() if compiled else None
    if mod is None:
        from . import _gs as mod
    sys.modules["_gs"] = mod
    return mod
