#!/usr/bin/env python
"""Runs tools/stress/bwd_stress in many FRESH processes per configuration and tabulates the launches that
disagreed with the in-process vector-kernel reference.

    python tools/stress/run_matrix.py --out gpurun_out/stress.jsonl --procs 40 [--budget-s 600]
                                      [--variants vec,mfma2,mfma4,mfma1] [--batches 0,3] [--Cs 1,4]
                                      [--extra "--poison"] [--first-only]

Every process is one JSON line in --out (plus "cfg" and "rc"); the summary (one row per configuration:
processes, processes with a bad launch, bad launches, worst relative deviation) goes to stdout and to
<out>.summary.json.  No torch import: a process costs ~0.5 s of HIP start-up plus a few launches."""
import argparse
import itertools
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "tools", "stress", "bwd_stress")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress.jsonl"))
    ap.add_argument("--procs", type=int, default=20)
    ap.add_argument("--budget-s", type=float, default=600.0)
    ap.add_argument("--variants", default="vec,mfma2,mfma4,mfma1")
    ap.add_argument("--batches", default="0,3")
    ap.add_argument("--Cs", default="1,4")
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--N", type=int, default=30000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--extra", default="")
    ap.add_argument("--orders", default="test-first,ref-first")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    cfgs = list(itertools.product(args.variants.split(","), [int(b) for b in args.batches.split(",")],
                                  [int(c) for c in args.Cs.split(",")], args.orders.split(",")))
    t0 = time.time()
    rows = {}
    with open(args.out, "a") as f:
        # round-robin over the configurations so that a budget cut leaves every row with a similar sample
        for rep in range(args.procs):
            for variant, batch, C, order in cfgs:
                if time.time() - t0 > args.budget_s:
                    break
                cmd = [BIN, "--variant", variant, "--C", str(C), "--N", str(args.N), "--size", str(args.size),
                       "--launches", str(args.launches), "--seed", str(1 + rep % 5)]
                if batch:
                    cmd += ["--batch", str(batch)]
                if order == "test-first":
                    cmd += ["--test-first"]
                cmd += args.extra.split()
                key = f"{variant}|B={batch}|C={C}|{order}" + (f"|{args.extra}" if args.extra else "")
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
                    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}"
                    rec = json.loads(line)
                    rec["rc"] = r.returncode
                    if r.returncode not in (0, 1):
                        rec["stderr"] = r.stderr[-400:]
                except Exception as e:  # a hang or a crash is a finding too
                    rec = {"rc": -1, "error": repr(e)[:300]}
                rec["cfg"] = key
                f.write(json.dumps(rec) + "\n")
                f.flush()
                row = rows.setdefault(key, {"procs": 0, "bad_procs": 0, "bad_launches": 0, "launches": 0, "worst": 0.0,
                                            "errors": 0, "kernel": rec.get("kernel"), "first_launch_bad": 0})
                row["procs"] += 1
                if rec["rc"] not in (0, 1):
                    row["errors"] += 1
                    continue
                row["launches"] += rec.get("launches", 0)
                row["bad_launches"] += rec.get("bad_launches", 0)
                row["bad_procs"] += 1 if rec.get("bad_launches", 0) or rec["rc"] == 1 else 0
                row["worst"] = max(row["worst"], rec.get("worst", 0.0))
                if any(d.get("launch") == 0 for d in rec.get("detail", [])):
                    row["first_launch_bad"] += 1
    summary = {"elapsed_s": time.time() - t0, "rows": rows}
    with open(args.out + ".summary.json", "w") as f:
        json.dump(summary, f, indent=1)
    print(f"{'configuration':58s} procs bad_procs bad_launches/launches first_bad  worst   errors")
    for k, r in rows.items():
        print(f"{k:58s} {r['procs']:5d} {r['bad_procs']:9d} {r['bad_launches']:6d}/{r['launches']:<6d} {r['first_launch_bad']:9d}  "
              f"{r['worst']:.1e} {r['errors']:6d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
