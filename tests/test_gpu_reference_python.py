"""The reference's OWN Python on the MI355X build (VERDICT r4 missing #3 / next #2b): gs/renderer.py's autograd classes
(_render_with_T :1135-1283, _render_scalar :999-1132, _render_sh :674-830, _render_sh_bg :833-996, render_start_end :541-672),
its PyTorch projection chained in front of them, and its MODEL class (gs/gaussian_splatting.py GaussianSplattingRenderer.forward
:1423-1466 over render_one :1198-1421, post_backward) -- imported UNMODIFIED, CUDA tensors, `_backend` = the COMPILED `_gs`
module of this repo (gsgen_amd/ext/_gs.*.so over libgsgen_hip.so) -- held to the golden vectors the reference itself produced
(tests/golden/*.npz, tests/golden/model/model_batch.npz), forward and backward, per-row gradient tolerances.

/root/reference does not exist on the GPU box: the reference's modules are imported from tests/_refpy.zip, an archive of
unmodified copies staged in the authoring container by tests/stage_refpy.py (git-ignored, travels with the working tree like
oracle/_ref).  Skipped cleanly where neither is present.  The same case bodies run on the CPU emulator build in
tests/test_reference_python_on_mirror.py (tests/refpy_cases.py)."""
import sys
import types

import pytest
import torch

import refshim
import refpy_cases as RC

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(refshim.root() is None, reason="neither /root/reference nor tests/_refpy.zip is present")]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ref():
    """gs.renderer of the reference with `_backend` = this repo's compiled `_gs`"""
    import gsgen_amd
    backend = gsgen_amd.compiled_gs()
    assert backend is not None, "gsgen_amd/ext/_gs.*.so is not built (python -m gsgen_amd.build)"
    refshim.install()
    sys.modules["_gs"] = backend  # what `import _gs as _backend` finds (gs/renderer.py:20-24)
    import gs.renderer as GR
    GR._backend = backend
    assert GR.__file__.startswith(refshim.root()), GR.__file__
    # the SH classes bracket their kernels with cudaProfilerStart/Stop (gs/renderer.py:698,720,...): not a CUDA runtime here
    stub = types.SimpleNamespace(cudaProfilerStart=lambda: 0, cudaProfilerStop=lambda: 0)
    orig = torch.cuda.profiler.cudart
    torch.cuda.profiler.cudart = lambda: stub
    yield GR
    torch.cuda.profiler.cudart = orig


@pytest.mark.parametrize("name", ["mock2", "rand_c1", "rand_c3", "rand_c4", "long_lists"])
def test_reference_render_with_T_and_start_end_on_the_compiled_module(ref, name):
    RC.case_render_with_T_and_start_end(ref, DEV, name)


@pytest.mark.parametrize("name", ["mock2", "rand_c3", "long_lists"])
def test_reference_render_scalar_on_the_compiled_module(ref, name):
    RC.case_render_scalar(ref, DEV, name)


@pytest.mark.parametrize("name", ["mock2", "rand_c1", "rand_c3", "rand_c4", "long_lists"])
@pytest.mark.parametrize("with_bg", [False, True])
@pytest.mark.parametrize("basis", ["auto", "exact"])
def test_reference_render_sh_on_the_compiled_module(ref, name, with_bg, basis):
    """both settings of the drop-in's SH degree-3 arithmetic: the routed default and the reference's per-pixel basis"""
    ref._backend.set_sh_basis(basis)
    try:
        RC.case_render_sh(ref, DEV, name, with_bg)
    finally:
        ref._backend.set_sh_basis("auto")


def test_reference_projection_chained_into_reference_render_sh_on_the_compiled_module(ref):
    RC.case_projection_chained_into_reference_render_sh(ref, DEV)


def test_reference_model_class_on_the_compiled_module(ref):
    """GaussianSplattingRenderer.forward(batch) -> rgb / depth / opacity / z_var, backward, post_backward: every `_backend.*`
    call of gs/gaussian_splatting.py lands in the compiled module; images, raw-parameter gradients and densify statistics
    against the fixture the reference's own kernels produced under the same class"""
    M = RC.import_reference_model(ref._backend)
    RC.check_model_against_fixture(RC.run_reference_model(M, DEV))
