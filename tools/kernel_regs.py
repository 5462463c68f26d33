"""Per-kernel register / LDS / scratch figures of a built libgsgen_hip.so (the code object's notes).

    python tools/kernel_regs.py [lib] [substring ...]
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def notes(lib):
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "lib.so")
        with open(lib, "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={dst}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/k.co"], check=False, capture_output=True)
        co = os.path.join(tmp, "k.co")
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, capture_output=True)
            co = next(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = {}
    for blk in txt.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                     for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(
        os.path.dirname(__file__), "..", "gsgen_amd", "lib", "libgsgen_hip.so")
    subs = [a for a in sys.argv[1:] if not os.path.exists(a)]
    for k, v in sorted(notes(lib).items()):
        if not subs or any(s in k for s in subs):
            print(f"{v['vgpr_count']:4d} v {v['sgpr_count']:4d} s {v['private_segment_fixed_size']:5d} scratch "
                  f"{v['group_segment_fixed_size']:6d} lds  {k}")
