"""The entry loop of one kernel of the built library: the smallest backward-branch loop that encloses the first occurrence of an
anchor opcode (default v_exp_f32: the Gaussian of an entry), with an opcode histogram and issue-slot estimate.
    python tools/kloop.py <kernel substring> [--anchor OPCODE] [--nth K] [--dump] [--lib path]"""
import collections, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
QUARTER = ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")


def kernels(lib):
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "lib.so")
        with open(lib, "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f], cwd=tmp,
                                 capture_output=True, text=True).stdout
            for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
                m = re.match(r"[0-9a-f]+ <(\S+)>:", blk)
                if m:
                    yield m.group(1), blk


def main():
    argv = sys.argv[1:]
    def opt(name, default=None):
        if name in argv:
            i = argv.index(name); v = argv[i + 1]; del argv[i:i + 2]; return v
        return default
    anchor = opt("--anchor", "v_exp_f32")
    nth = int(opt("--nth", "0"))  # which occurrence of the anchor (kernels with several loops that hold one)
    lib = opt("--lib") or os.environ.get("GSGEN_HIP_LIB") or os.path.join(os.path.dirname(__file__), "..", "gsgen_amd", "lib", "libgsgen_hip.so")
    dump = "--dump" in argv
    if dump:
        argv.remove("--dump")
    sub = argv[0]
    for name, blk in kernels(lib):
        if sub in name and name.startswith("_ZN2gs"):
            break
    else:
        raise SystemExit("no such kernel")
    ins = [(int(a, 16), t.strip()) for t, a in re.findall(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):", blk, flags=re.M)]
    anchors = [a for a, t in ins if t.split()[0].startswith(anchor)]
    if not anchors:
        raise SystemExit(f"{name}: no {anchor}")
    first = anchors[min(nth, len(anchors) - 1)]
    loops = []
    for a, t in ins:
        b = re.match(r"s_c?branch\S*\s+(\d+)", t)
        if b:
            off = int(b.group(1)); off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt <= first <= a:
                loops.append((a - tgt, tgt, a))
    if not loops:
        raise SystemExit("no enclosing loop")
    _, lo, hi = min(loops)
    body = [(a, t) for a, t in ins if lo <= a <= hi]
    c = collections.Counter(t.split()[0] for a, t in body)
    valu = sum(n for k, n in c.items() if k.startswith("v_"))
    slots = sum(n * (4 if k.startswith(QUARTER) else 1) for k, n in c.items() if k.startswith("v_"))
    print(name)
    print(f"loop {hex(lo)}..{hex(hi)}: {len(body)} instructions, {valu} vector ({slots} issue slots with transcendentals at a quarter), "
          f"{sum(n for k, n in c.items() if k.startswith('s_'))} scalar, {sum(n for k, n in c.items() if k.startswith('ds_'))} LDS, "
          f"{sum(n for k, n in c.items() if k.startswith(('global_', 'flat_', 'buffer_', 'scratch_')))} memory")
    if dump:
        for a, t in body:
            print(hex(a), t)
    print(sorted(c.items(), key=lambda x: -x[1]))


main()
