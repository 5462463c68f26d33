"""Golden files for gsgen_amd/io.py from the REFERENCE's exporters (authoring container only).

utils/export.py cannot be imported here (it pulls the whole trainer and `plyfile`, absent), so the
two functions to_ply / to_splat are compiled from the reference's own source text, as it lies under
/root/reference, into a namespace that supplies their globals.  The only stand-in with behaviour is
plyfile (third party, not under /root/reference): PlyElement.describe / PlyData.write restated from
plyfile's documented output -- ASCII header, `format binary_little_endian 1.0`, one `property float`
line per f4 field, `end_header`, then the packed records.

    python tests/golden/make_golden_io.py
"""
import ast
import os
import struct
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_EXPORT = "/root/reference/utils/export.py"


class PlyElement:
    def __init__(self, data, name):
        self.data, self.name = data, name

    @staticmethod
    def describe(data, name):
        return PlyElement(data, name)


class PlyData:
    _types = {"f4": "float", "u1": "uchar", "i4": "int"}

    def __init__(self, elements):
        self.elements = elements

    def write(self, path):
        lines = ["ply", "format binary_little_endian 1.0"]
        for el in self.elements:
            lines.append(f"element {el.name} {len(el.data)}")
            for fname in el.data.dtype.names:
                lines.append(f"property {self._types[el.data.dtype[fname].str[1:]]} {fname}")
        lines.append("end_header")
        with open(path, "wb") as f:
            f.write(("\n".join(lines) + "\n").encode("ascii"))
            for el in self.elements:
                f.write(el.data.astype(el.data.dtype.newbyteorder("<")).tobytes())


class _Console:
    def print(self, *a, **k):
        pass


def reference_exporters():
    tree = ast.parse(open(REF_EXPORT).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("to_ply", "to_splat")]
    ns = {"np": np, "torch": torch, "struct": struct, "Path": Path, "PlyData": PlyData, "PlyElement": PlyElement,
          "console": _Console(), "get_ckpt_path": lambda p: Path(p)}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF_EXPORT, "exec"), ns)
    return ns["to_ply"], ns["to_splat"]


def params(n=64, seed=5):
    g = torch.Generator().manual_seed(seed)
    p = {"mean": torch.randn(n, 3, generator=g) * 0.5,
         "qvec": torch.randn(n, 4, generator=g),
         "svec": torch.log(torch.rand(n, 3, generator=g) * 0.05 + 0.005),
         "color": torch.randn(n, 3, generator=g) * 2.0,
         "alpha": torch.randn(n, generator=g) * 3.0}
    p["qvec"][0] = torch.tensor([1.0, 0.0, 0.0, 0.0])   # a component of exactly 1 -> 256 -> u8 wrap
    p["qvec"][1] = torch.tensor([0.0, -2.0, 0.0, 0.0])  # -1 after normalisation -> 0
    p["svec"][5] = p["svec"][4]; p["alpha"][5] = p["alpha"][4]  # equal sort keys: index order decides
    return p


def main():
    to_ply, to_splat = reference_exporters()
    p = params()
    with tempfile.TemporaryDirectory() as d:
        ck = os.path.join(d, "step_1.pt")
        torch.save({"params": p, "cfg": {"prompt": {"prompt": "a golden"}}, "step": 1}, ck)
        to_ply(ck, d)
        to_splat(ck, d)
        ply = open(os.path.join(d, "ply", "a_golden.ply"), "rb").read()
        splat = open(os.path.join(d, "splat", "a_golden.splat"), "rb").read()
    np.savez_compressed(os.path.join(HERE, "io", "export.npz"), ply=np.frombuffer(ply, np.uint8), splat=np.frombuffer(splat, np.uint8),
                        **{k: v.numpy() for k, v in p.items()})
    print("ply", len(ply), "bytes; splat", len(splat), "bytes")


if __name__ == "__main__":
    main()
