"""The compositing kernels exist in several compiled shapes (pixels per lane of the forward / backward, packed vs unpacked
per-pixel arithmetic, block order of the batched grids), selected through gsgen_debug_set_variant.  Only one combination is
the default; these tests run parts of the parity suite in a subprocess for the others (GSGEN_TEST_VARIANTS, applied by
tests/conftest.py to every library handle) so that none of them rots.  CPU: on the SIMT emulator; GPU: on the real library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

VARIANTS = [
    {"ppl_fwd": 4, "ppl_bwd": 2},
    {"ppl_fwd": 2, "ppl_bwd": 1},
    {"sh_chred": 0},  # packed SH backward with ONE 64-component gradient reduction (2 wavefronts per SIMD)
    {"chan_packed": 0},  # RGB / scalar / RGB + heads backward on the unpacked k_composite_bwd_pixel
]


def _ids(v):
    return ",".join(f"{k}={x}" for k, x in v.items())


def _run(variant, args):
    env = dict(os.environ)
    env["GSGEN_TEST_VARIANTS"] = _ids(variant)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("variant", VARIANTS, ids=_ids)
def test_variant_on_emulator(variant):
    # two SH degrees (padded and unpadded coefficient rows) + the fused RGB heads: ~20 s per variant
    _run(variant, ["tests/test_cpu_host.py", "-k",
                   "emulated_kernels_match_oracle and (4-33-20 or 3-48-32) or emulated_fused_rgb_heads"])


# the batched launches (cameras of a batch in one launch) have their own switches: wavefronts per tile and the
# order of the (camera, tile) blocks in the grid.  GPU only: the emulator runs the default of each in
# tests/test_cpu_host.py, and one emulated batch costs ~30 s.
BATCH_VARIANTS = [
    {"batch_map": 0},                                   # interleaved cameras
    # interleaved + rotated; 2 wavefronts per tile forward (the per-camera launches the images are compared with bit
    # for bit get the same split: different pixels-per-lane builds round a few pixels differently)
    {"batch_map": 1, "ppl_fwd_batch": 2, "ppl_fwd": 2},
    {"ppl_bwd_sh_batch": 2, "ppl_bwd_batch": 4},  # SH backward 2 wavefronts per tile, heads 1
    {"ppl_bwd_sh_batch": 1, "ppl_bwd_batch": 1, "ppl_fwd_batch": 4, "ppl_fwd": 4, "ppl_fwd_poly": 4},
    {"sh_chred": 0},
    {"chan_packed": 0},
]

def test_batch_block_order_variant_on_emulator():
    # the rotated interleaving exercises every branch of batch_view(); one emulated batch, ~25 s
    _run({"batch_map": 1}, ["tests/test_cpu_host.py", "-k", "emulated_batched_views_match_per_view_launches and 4-0"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", BATCH_VARIANTS, ids=_ids)
def test_batch_variant_on_gpu(variant):
    _run(variant, ["tests/test_gpu_api.py", "-m", "gpu", "-k",
                   "batched_cameras_match_one_at_a_time and (3-4-1 or 2-2-3) or batched_fused_heads_match_oracle"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS, ids=_ids)
def test_variant_on_gpu(variant):
    _run(variant, ["tests/test_gpu_parity.py", "tests/test_gpu_golden.py", "-m", "gpu", "-k",
                   "forward_backward or golden"])
