"""One example of the per-camera chain fuzz (tests/tile_chain.other_tile_size_chain) on the GPU with every comparison's deviation printed:
python tools/dbg_chain_example.py ts C W H n seed svec opaque [repeats]"""
import inspect, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import tile_chain
import test_gpu_parity as TP

ts, C, W, H, n, seed = (int(x) for x in sys.argv[1:7])
svec, opaque = float(sys.argv[7]), sys.argv[8] in ("1", "True", "true")
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 3
code = inspect.getsource(tile_chain.other_tile_size_chain).replace(
    "        assert np.abs(a - b).max() <= rtol * (np.abs(b).max() + 1e-12) + atol",
    "        dev = np.abs(a - b).max(); lim = rtol * (np.abs(b).max() + 1e-12) + atol\n"
    "        print('   dev %.3e  max %.3e  limit %.3e  %s %s' % (dev, np.abs(b).max(), lim, np.shape(b), 'ok' if dev <= lim else '<<<<<< beyond'))")
ns = dict(tile_chain.__dict__)
exec(code, ns)
for r in range(reps):
    print("repeat", r)
    ns["other_tile_size_chain"](TP._DeviceArrays(), ts, C, W, H, sync=torch.cuda.synchronize, n=n, seed=seed, svec=svec, opaque=opaque,
                                rtol=1e-3, atol=tile_chain.FUZZ_ATOL, ftol=1e-4)
