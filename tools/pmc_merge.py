"""Merges the per-pass outputs of tools/pmc.sh (gpurun_out/pmc_<tag>.json: FETCH_SIZE | WRITE_SIZE | SQ_* passes of a short
bench.py run) into profiles/<round>_traffic.json, keyed the way bench.py looks entries up: "<config>|<kernel variant>|views=<B>".
    python tools/pmc_merge.py r03 cfg2 8 fetch write sq [sq2 ...]
traffic = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE
uncalibrated), both in KB in rocprofv3's output.  valu_floor = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / 1024 SIMDs / 2.4 GHz."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
rnd, cfg, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
tags = sys.argv[4:]
src_dir = os.environ.get("PMC_DIR", os.path.join(ROOT, "gpurun_out"))
from gsgen_amd import _capi  # noqa: E402
lib = _capi.load()  # (names only: works without a GPU)
# raw (demangled) kernel name fragments -> the library's descriptive variant names
names = {
    "k_composite_bwd_sh_vec<4, 4, true, 6, true>": lib.kernel_variant("sh_bwd_batch_poly", 4, 1),   # (round 6: the moment form is what bench.py runs)
    "k_composite_bwd_sh_vec<4, 4, true, 6, false>": lib.kernel_variant("sh_bwd_batch_poly", 4, 1) + " (plain gradients)",
    "k_composite_fwd_sh_vec<4, 4, true, 6, false>": lib.kernel_variant("sh_fwd_batch_poly", 4, 1),
    "k_composite_fwd_sh_vec<4, 4, true, 6, true>": lib.kernel_variant("sh_fwd_batch_poly", 4, 2),
    "k_composite_bwd_sh_vec<4, 4, true, -2, true>": "persistent exact fallback of the bounded backward (normally leaves at once)",
    "k_composite_bwd_sh_vec<4, 4, true, -2, false>": "persistent exact fallback of the bounded backward (normally leaves at once; plain gradients)",
    "k_composite_fwd_sh_vec<4, 2, true, -2, true>": "persistent exact fallback of the bounded forward (normally leaves at once)",
    "k_composite_bwd_sh_vec<4, 4, true, 0, true>": lib.kernel_variant("sh_bwd_batch", 4, 1),
    "k_composite_bwd_sh_vec<4, 4, true, 0, false>": lib.kernel_variant("sh_bwd_batch", 4, 1) + " (plain gradients)",
    "k_composite_fwd_sh_vec<4, 2, true, 0, true>": lib.kernel_variant("sh_fwd_batch", 4, 1),
    "k_composite_bwd_chan_vec<3, true, true>": lib.kernel_variant("rgbd_bwd_batch_moments", 1, 1),
    "k_composite_bwd_chan_vec<3, true, false>": lib.kernel_variant("rgbd_bwd_batch", 1, 1),
    "k_composite_fwd_chan_vec<3, true>": lib.kernel_variant("rgbd_fwd_batch", 1, 1),
    # the geometry chain of a batch (same kernels behind the SH and the RGB + heads step)
    "k_frame_project_views": "k_frame_project_views", "k_bin_pull_views<false>": "k_bin_pull_views<count>",
    "k_bin_pull_views<true>": "k_bin_pull_views<emit>", "k_bin_push_views<false>": "k_bin_push_views<count>",
    "k_bin_push_views<true>": "k_bin_push_views<emit>", "k_sort_tiles_views": "k_sort_tiles_views",
    "k_scan_chunks_views": "k_scan_chunks_views", "k_scan_order_tiles_views": "k_scan_order_tiles_views",
    "k_project_bwd_views": "k_project_bwd_views",
}
out_path = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
out["note"] = ("rocprofv3 --pmc, separate passes (tools/pmc.sh), bench.py --only-timed at the driver's launch shape, per launch, MI355X. "
               "FETCH_SIZE / WRITE_SIZE in KB; traffic = 2*FETCH + WRITE (MI355X_MICROARCH.md gfx950 correction; WRITE_SIZE uncalibrated). "
               "valu_floor = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / 1024 SIMDs / 2.4 GHz.")
for tag in tags:
    d = json.load(open(os.path.join(src_dir, f"pmc_{tag}.json")))
    for raw, vals in d.items():
        for frag, nice in names.items():
            if frag in raw.replace("gs::", ""):
                e = out.setdefault(f"{cfg}|{nice}|views={B}", {"raw_kernel": frag})
                for k, v in vals.items():
                    if k == "launches":
                        e.setdefault("launches_per_pass", {})[tag] = v
                    elif k in ("FETCH_SIZE", "WRITE_SIZE"):
                        e[k + "_KB"] = v
                    else:
                        e[k] = v
for k, e in out.items():
    if not isinstance(e, dict):
        continue
    if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
        e["traffic_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024)
    if "SQ_ACTIVE_INST_VALU" in e:
        e["valu_floor_ms_per_launch"] = e["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e9 * 1e3
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
