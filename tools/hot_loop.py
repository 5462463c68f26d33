"""Disassembles one kernel of the built library and prints its per-entry loop (the smallest backward branch that spans
2 * PPL calls) with an opcode histogram.      python tools/hot_loop.py <kernel substring> [--dump] [--calls N]"""
import collections, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_text(lib, sub):
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
                if m and sub in m.group(1) and m.group(1).startswith("_ZN2gs"):
                    return m.group(1), blk
    raise SystemExit("no such kernel")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ncalls = int(sys.argv[sys.argv.index("--calls") + 1]) if "--calls" in sys.argv else 8
    if "--calls" in sys.argv:
        args.remove(str(ncalls))
    lib = os.environ.get("GSGEN_HIP_LIB") or os.path.join(os.path.dirname(__file__), "..", "gsgen_amd", "lib", "libgsgen_hip.so")
    name, blk = kernel_text(lib, args[0])
    ins = [(int(a, 16), t.strip()) for t, a in re.findall(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):", blk, flags=re.M)]
    calls = [a for a, t in ins if t.startswith("s_swappc")]
    loops = []
    for a, t in ins:
        b = re.match(r"s_c?branch\S*\s+(\d+)", t)
        if b:
            off = int(b.group(1)); off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt <= a and sum(tgt <= c <= a for c in calls) == ncalls:
                loops.append((a - tgt, tgt, a))
    _, lo, hi = min(loops)
    body = [(a, t) for a, t in ins if lo <= a <= hi]
    print(name, "loop", hex(lo), hex(hi), len(body), "instructions")
    if "--dump" in sys.argv:
        for a, t in body:
            print(hex(a), t)
    c = collections.Counter(t.split()[0] for a, t in body)
    print(sorted(c.items(), key=lambda x: -x[1]))


main()
