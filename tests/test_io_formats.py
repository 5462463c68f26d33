"""gsgen_amd/io.py against files written by the reference's own exporters
(tests/golden/io/export.npz, made by tests/golden/make_golden_io.py) and round trips."""
import os

import numpy as np
import torch

from gsgen_amd import io as gio

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    z = np.load(os.path.join(HERE, "golden", "io", "export.npz"))
    params = {k: z[k] for k in ("mean", "qvec", "svec", "color", "alpha")}
    return params, z["ply"].tobytes(), z["splat"].tobytes()


def test_ply_bytes_equal_reference_export(tmp_path):
    params, ply, _ = _golden()
    path = tmp_path / "a.ply"
    gio.write_ply(path, params)
    assert open(path, "rb").read() == ply
    back = gio.read_ply(path)
    for k in ("mean", "svec", "qvec", "alpha"):
        assert np.array_equal(back[k], params[k])
    assert np.allclose(back["color"], params["color"], rtol=1e-6, atol=0)


def test_splat_bytes_equal_reference_export(tmp_path):
    params, _, splat = _golden()
    path = tmp_path / "a.splat"
    gio.write_splat(path, {k: torch.from_numpy(v) for k, v in params.items()})  # tensors or arrays
    assert open(path, "rb").read() == splat
    rec = gio.read_splat(path)
    assert rec.dtype.itemsize == 32 and len(rec) == params["mean"].shape[0]
    vol = rec["scale"].prod(1) * rec["rgba"][:, 3]
    assert np.all(vol[:-1] >= vol[1:])                      # sorted by volume * opacity, descending
    assert 0 in rec["rot"][:, 0] and 0 in rec["rot"][:, 1]  # q = (1,0,0,0) wraps to 0; q_x = -1 maps to 0


def test_empty_and_checkpoint_roundtrip(tmp_path):
    empty = {"mean": np.zeros((0, 3), np.float32), "qvec": np.zeros((0, 4), np.float32), "svec": np.zeros((0, 3), np.float32),
             "color": np.zeros((0, 3), np.float32), "alpha": np.zeros((0,), np.float32)}
    gio.write_splat(tmp_path / "e.splat", empty)
    assert os.path.getsize(tmp_path / "e.splat") == 0
    gio.write_ply(tmp_path / "e.ply", empty)
    assert gio.read_ply(tmp_path / "e.ply")["mean"].shape == (0, 3)
    params, _, _ = _golden()
    gio.save_checkpoint(tmp_path / "step_7.pt", params, cfg={"prompt": {"prompt": "x"}}, step=7)
    p2, cfg, step = gio.load_checkpoint(tmp_path / "step_7.pt")
    assert step == 7 and cfg["prompt"]["prompt"] == "x"
    for k, v in params.items():
        assert np.array_equal(p2[k].numpy(), v)
    torch.save(p2, tmp_path / "bare.pt")  # the bare-params layout the exporters also accept
    p3, cfg3, _ = gio.load_checkpoint(tmp_path / "bare.pt")
    assert cfg3 is None and np.array_equal(p3["mean"].numpy(), params["mean"])
