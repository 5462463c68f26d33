"""Stages the reference's own Python for the hot path into ONE archive, tests/_refpy.zip, so that it can run ON THE MI355X BOX,
where /root/reference does not exist (VERDICT r4 #2b).  Exactly how oracle/_ref travels: the archive is git-ignored (no reference
source enters the history or the tree) but not gpurun-ignored (it rides along with the snapshot of the working tree like the
built .so files); Python imports straight from it (zipimport), nothing is unpacked.

    python tests/stage_refpy.py          # in the authoring container; __graft_entry__.build() calls it when /root/reference exists

What is staged is the import closure of gs/gaussian_splatting.py (+ gs/renderer.py, gs/culling.py) -- found by importing them
from /root/reference behind tests/refshim.py's stand-ins and collecting every module whose file lies under /root/reference --
and the default config conf/base.yaml the model class reads its fields from.  Nothing is edited.  tests/refshim.py then serves the archive to
tests/test_gpu_reference_python.py when /root/reference is absent."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "_refpy.zip")
REF = "/root/reference"


def stage(verbose=True):
    if not os.path.isdir(os.path.join(REF, "gs")):
        raise RuntimeError(f"{REF} is not present: tests/_refpy.zip can only be staged in the authoring container")
    sys.path.insert(0, HERE)
    import types
    import refshim
    refshim.install(force_root=REF)
    dm = types.ModuleType("kornia.geometry.depth")  # utils/ops.py:5 imports depth_to_3d (unused on this path)
    dm.depth_to_3d = None
    sys.modules["kornia.geometry.depth"] = dm
    sys.modules["kornia"].__path__ = []
    sys.modules["kornia.geometry"].__path__ = []
    import gs.gaussian_splatting  # noqa: F401  (pulls gs.renderer, gs.culling, gs.backgrounds, utils.*)
    files = set()
    for m in list(sys.modules.values()):
        f = getattr(m, "__file__", None)
        if f and os.path.abspath(f).startswith(REF + os.sep) and f.endswith(".py"):
            files.add(os.path.abspath(f))
    for f in list(files):  # every package on the way needs its __init__.py
        d = os.path.dirname(f)
        while d != REF:
            init = os.path.join(d, "__init__.py")
            if os.path.exists(init):
                files.add(init)
            d = os.path.dirname(d)
    files.add(os.path.join(REF, "conf", "base.yaml"))
    import zipfile
    if os.path.exists(DST):
        os.remove(DST)
    with zipfile.ZipFile(DST, "w", zipfile.ZIP_DEFLATED) as z:
        for f in sorted(files):
            rel = os.path.relpath(f, REF)
            z.write(f, rel)
            if verbose:
                print("staged", rel)
        z.writestr("STAGED_FROM", f"{REF} (gsgen3d/gsgen): unmodified copies made by tests/stage_refpy.py; git-ignored, test "
                                  "infrastructure only\n")
    return sorted(os.path.relpath(f, REF) for f in files)


if __name__ == "__main__":
    stage()
