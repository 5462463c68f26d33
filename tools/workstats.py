"""Per-wavefront work statistics of the compositing kernels on the bench workload (CPU, oracle):
how many (wavefront, list entry) pairs are walked, how many contribute, pixels per contributing pair,
for the forward's 4 strips, the backward's 2 interleaved halves and a whole-tile wavefront."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402

sc = scenes.pointe_scene(100_000, seed=0, C=4)
cam = scenes.Camera(800, 800, fx=800.0, c2w=scenes.orbit(2.5, 15, 30))
g = scenes.oracle_geometry(sc, cam)
m = g["mask"]
lib = O.lib()
f = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
mean2d, cov2d, alpha, tl = f(g["mean2d"]), f(g["cov2d"]), f(sc["alpha"][m]), f(cam.topleft)
st, en, ids = g["start"], g["end"], g["ids"]
nth, ntw = cam.tiles
print("pairs D =", g["D"])
for name, rows in (("forward, 4 strips of 4 rows", [r // 4 for r in range(16)]),
                   ("backward, 2 interleaved halves", [(r // 4) % 2 for r in range(16)]),
                   ("one wavefront per tile", [0] * 16)):
    rows = np.array(rows, np.int32)
    w, c, pp = C.c_longlong(), C.c_longlong(), C.c_longlong()
    lib.gso_part_workstats(vp(mean2d), vp(cov2d), vp(alpha), vp(st), vp(en), vp(ids), vp(tl), nth, ntw, C.c_float(1 / 800),
                           C.c_float(1 / 800), 800, 800, C.c_float(1e-4), vp(rows), int(rows.max()) + 1, C.byref(w),
                           C.byref(c), C.byref(pp))
    print(f"{name}: walked {w.value}, contributing {c.value} ({c.value / w.value:.1%}), "
          f"{pp.value / c.value:.1f} contributing pixels per contributing pair ({pp.value} pixel pairs)")
