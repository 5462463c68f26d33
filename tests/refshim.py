"""Import the *reference's own Python* for the hot path from /root/reference (this
container only) behind stand-ins for the third-party modules that are absent here.

Used by tests/golden/make_golden.py to produce the committed golden vectors and by
tests that are skipped when /root/reference is missing (it does not exist on the GPU
box).  Nothing is copied: the reference modules are imported from where they lie.

Stand-ins:
  kornia.geometry.conversions  quaternion_to_rotation_matrix(order=WXYZ) restated from the
                               published kornia 0.6.0 formula (normalise, then the standard
                               rotation matrix) -- the one piece of arithmetic on the path
                               that is not under /root/reference.
  torchtyping / jaxtyping / typeguard / omegaconf / cv2   annotation-only or unused here.
  _gs                          an empty module: the CUDA extension cannot exist here.
"""
import enum
import os
import sys
import types

import torch

REF = "/root/reference"
STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_refpy.zip")  # tests/stage_refpy.py: what travels to the GPU box (zipimport)


def available():
    return os.path.isdir(os.path.join(REF, "gs"))


def staged_available():
    return os.path.isfile(STAGED)


def root():
    """where the reference's Python is imported from: /root/reference in the authoring container, the staged copy elsewhere"""
    if os.environ.get("GSGEN_TEST_REFPY") == "staged":  # (checks that the archive alone serves the tests, in the container that has both)
        return STAGED if staged_available() else None
    return REF if available() else (STAGED if staged_available() else None)


def _kornia_quat_to_rotmat(quaternion, order=None):
    q = torch.nn.functional.normalize(quaternion, p=2.0, dim=-1, eps=1e-12)
    w, x, y, z = torch.chunk(q, chunks=4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.tensor(1.0)
    m = torch.stack(
        (one - (tyy + tzz), txy - twz, txz + twy,
         txy + twz, one - (txx + tzz), tyz - twx,
         txz - twy, tyz + twx, one - (txx + tyy)), dim=-1).view(-1, 3, 3)
    if len(quaternion.shape) == 1:
        m = torch.squeeze(m, dim=0)
    return m


class _Sub:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return a[0] if a and callable(a[0]) else self


def install(force_root=None):
    base = force_root or root()
    if base is None:
        raise RuntimeError("/root/reference is not present (and tests/_refpy has not been staged)")
    if "_refshim_installed" in sys.modules:
        return
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class QuaternionCoeffOrder(enum.Enum):
        XYZW = "xyzw"
        WXYZ = "wxyz"

    mod("kornia"); mod("kornia.geometry")
    mod("kornia.geometry.conversions", QuaternionCoeffOrder=QuaternionCoeffOrder,
        quaternion_to_rotation_matrix=_kornia_quat_to_rotmat,
        rotation_matrix_to_quaternion=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    mod("torchtyping", TensorType=_Sub())
    sub = _Sub()
    mod("jaxtyping", Bool=sub, Complex=sub, Float=sub, Inexact=sub, Int=sub, Integer=sub, Num=sub,
        Shaped=sub, UInt=sub)
    mod("typeguard", typechecked=lambda f=None, **k: f)
    mod("omegaconf", OmegaConf=type("OmegaConf", (), {}), DictConfig=dict)
    mod("cv2")
    mod("_gs")
    for name in ("matplotlib", "matplotlib.pyplot"):
        try:
            __import__(name)
        except Exception:
            mod(name)
    mod("_refshim_installed")
    if base not in sys.path:
        sys.path.insert(0, base)


def reference_api():
    """-> (project_gaussians, tile_culling_aabb_count, CameraInfo, get_c2w_from_up_and_look_at)"""
    install()
    from gs.renderer import project_gaussians  # noqa
    from gs.culling import tile_culling_aabb_count  # noqa
    from utils.camera import CameraInfo  # noqa
    return project_gaussians, tile_culling_aabb_count, CameraInfo
