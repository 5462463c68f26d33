"""On-disk formats of the reference: the trainer's checkpoint
(trainer.py:255-266), the 17-field float32 `.ply` (utils/export.py:157-203) and the 32-byte-record
`.splat` of the web viewer sorted by volume * opacity (utils/export.py:206-283).  Host code only,
vectorised (the reference packs record by record with struct.pack); byte-identical output.

`params` is the dict the reference saves under "params": raw (pre-activation) tensors or arrays
mean [N,3], qvec [N,4] (w,x,y,z), svec [N,3] (log scale), color [N,3] (logit), alpha [N] (logit).
"""
import numpy as np

PLY_FIELDS = ("x", "y", "z", "nx", "ny", "nz", "red", "green", "blue", "opacity", "scale_0", "scale_1", "scale_2",
              "rot_0", "rot_1", "rot_2", "rot_3")
SPLAT_DTYPE = np.dtype([("pos", "<f4", 3), ("scale", "<f4", 3), ("rgba", "u1", 4), ("rot", "u1", 4)])


def _np(params, key):
    v = params[key]
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, np.float32)


def _sigmoid(x):
    """torch.sigmoid, as utils/export.py uses it: the u8 truncation that follows is sensitive to the last bit, and
    a numpy 1 / (1 + exp(-x)) differs from torch's kernel on rare values"""
    import torch
    return torch.sigmoid(torch.from_numpy(np.ascontiguousarray(x, np.float32))).numpy()


# ---- checkpoint ---------------------------------------------------------------------------------
def save_checkpoint(path, params, cfg=None, step=0):
    """{"params", "cfg", "step"} through torch.save, as Trainer.save (trainer.py:255-266)."""
    import torch
    state = {"params": {k: (v if hasattr(v, "detach") else torch.as_tensor(np.asarray(v))) for k, v in params.items()},
             "cfg": cfg if cfg is not None else {}, "step": int(step)}
    torch.save(state, path)


def load_checkpoint(path, map_location="cpu"):
    """-> (params, cfg, step); accepts both the wrapped and the bare-params layout the reference's
    exporters accept (utils/export.py:163-165)."""
    import torch
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if "params" in ckpt:
        return ckpt["params"], ckpt.get("cfg"), ckpt.get("step", 0)
    return ckpt, None, 0


# ---- .ply -----------------------------------------------------------------------------------------
def ply_table(params):
    """[N,17] float32 in PLY_FIELDS order; raw parameter values, colour scaled by 255
    (utils/export.py:185-195: no activations applied, normals zero)."""
    pos = _np(params, "mean")
    rgb = _np(params, "color") * np.float32(255.0)
    opacity = _np(params, "alpha").reshape(-1, 1)
    return np.concatenate((pos, np.zeros_like(pos), rgb, opacity, _np(params, "svec"), _np(params, "qvec")), axis=1)


def write_ply(path, params):
    """binary_little_endian PLY, one `vertex` element of 17 float properties (what
    plyfile.PlyData([PlyElement.describe(elements, "vertex")]).write() produces)."""
    tab = np.ascontiguousarray(ply_table(params), "<f4")
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {tab.shape[0]}"]
    header += [f"property float {name}" for name in PLY_FIELDS] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(tab.tobytes())


def read_ply(path):
    """-> dict of raw parameter arrays (inverse of write_ply)."""
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.index(b"end_header\n") + len(b"end_header\n")
    lines = blob[:end].decode("ascii").split("\n")
    if lines[0] != "ply" or "binary_little_endian" not in lines[1]:
        raise ValueError("not a binary little-endian PLY")
    n = int([ln for ln in lines if ln.startswith("element vertex")][0].split()[2])
    names = [ln.split()[2] for ln in lines if ln.startswith("property float")]
    if tuple(names) != PLY_FIELDS:
        raise ValueError(f"unexpected properties {names}")
    tab = np.frombuffer(blob, "<f4", count=n * 17, offset=end).reshape(n, 17)
    return {"mean": tab[:, 0:3].copy(), "color": tab[:, 6:9] / np.float32(255.0), "alpha": tab[:, 9].copy(),
            "svec": tab[:, 10:13].copy(), "qvec": tab[:, 13:17].copy()}


# ---- .splat -------------------------------------------------------------------------------------
def splat_records(params):
    """The records of utils/export.py:241-281 as a structured array, in file order:
    position f32x3 | exp(svec) f32x3 | sigmoid(color), sigmoid(alpha) * 255 truncated to u8 |
    normalised quaternion * 128 + 128 truncated to u8 (a component of exactly 1.0 wraps to 0, as in
    the reference); sorted by prod(scale) * opacity_u8 descending, ties in index order."""
    pos = _np(params, "mean")
    rgb = (_sigmoid(_np(params, "color")) * 255.0).astype(np.uint8).clip(0, 255)
    opacity = (_sigmoid(_np(params, "alpha")).reshape(-1, 1) * 255.0).astype(np.uint8).clip(0, 255)
    svec = np.exp(_np(params, "svec"))
    qvec = _np(params, "qvec")
    qvec = qvec / np.linalg.norm(qvec, axis=1, keepdims=True)
    qvec = (qvec * 128 + 128).astype(np.uint8).clip(0, 255)
    volume = np.prod(svec, axis=1) * opacity[..., 0]
    # sorted(range(n), key=volume, reverse=True) is stable: equal keys keep ascending index
    order = np.argsort(-volume.astype(np.float64), kind="stable")
    rec = np.empty(pos.shape[0], SPLAT_DTYPE)
    rec["pos"], rec["scale"] = pos[order], svec[order]
    rec["rgba"][:, :3], rec["rgba"][:, 3] = rgb[order], opacity[order, 0]
    rec["rot"] = qvec[order]
    return rec


def write_splat(path, params):
    splat_records(params).tofile(path)


def read_splat(path):
    return np.fromfile(path, SPLAT_DTYPE)
