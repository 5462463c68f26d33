"""gsgen_amd -- MI355X (gfx950) differentiable Gaussian-splatting rasterizer, a drop-in for the
`_gs` extension of gsgen3d/gsgen.

    gsgen_amd._gs        mirror of the reference's pybind module (same names / argument orders)
    gsgen_amd.renderer   the autograd.Function surface of gs/renderer.py + HIP projection +
                         the fused `render_frame`
    gsgen_amd.dist       camera sharding across GPUs (one process per GPU, RCCL all_gather)
    gsgen_amd.build      hipcc build of gsgen_amd/lib/libgsgen_hip.so (C ABI: include/gsgen_hip.h)

There is no CPU implementation in this package: every entry point needs the HIP library and
a GPU, and raises if either is missing.
"""
import sys

__version__ = "0.1.0"


def install_as_gs():
    """Register gsgen_amd._gs as the top-level module `_gs`, which is what the reference's
    Python imports (`import _gs as _backend`, gs/renderer.py:20-24)."""
    from . import _gs
    sys.modules["_gs"] = _gs
    return _gs
