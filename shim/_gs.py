"""`import _gs` resolves here when this directory is on PYTHONPATH -- the zero-line-change way to
put the MI355X rasterizer behind the reference's `try: import _gs as _backend`
(gs/renderer.py:20-24, gs/gaussian_splatting.py:42-46, gs/sh_renderer.py:26-29):

    PYTHONPATH=/path/to/repo:/path/to/repo/shim python main.py ...

The module object that ends up in sys.modules["_gs"] is what gsgen_amd.install_as_gs() registers: the compiled
extension gsgen_amd/ext/_gs.*.so when it has been built, else the ctypes mirror gsgen_amd._gs (same functions)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import gsgen_amd  # noqa: E402

sys.modules[__name__] = gsgen_amd.install_as_gs()
