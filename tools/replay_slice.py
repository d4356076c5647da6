#!/usr/bin/env python3
"""Per-kernel statistics of the REPLAYED steps only (runs on the GPU box, right after the rocprofv3 --kernel-trace pass):

    python tools/replay_slice.py <trace_dir> <out_csv> [skip_steps=3]

The kernel trace of `bench.py` mixes the engine's two eager warm-up steps and the capture with the replayed steps.  One
flat_adamw launch ends every training step, so everything after the `skip_steps`-th flat_adamw launch belongs to replays:
those kernels are aggregated per name (calls, total / average ns, share) and into a glue summary -- library kernels
(mphsir::*), RCCL, and everything else (at::native, copies, fills), each with ms per replayed step."""
import csv
import glob
import os
import sys
from collections import defaultdict

trace_dir, out_csv = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = (glob.glob(os.path.join(trace_dir, "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(trace_dir, "*_kernel_trace.csv")))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [int(r["End_Timestamp"]) for r in rows if "flat_adamw" in r["Kernel_Name"] and "scaled" not in r["Kernel_Name"]] or \
       [int(r["End_Timestamp"]) for r in rows if "flat_adamw" in r["Kernel_Name"]]
if len(ends) <= skip:
    sys.exit("replay_slice: only %d optimizer launches in the trace" % len(ends))
t0, steps = ends[skip - 1], len(ends) - skip
agg = defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0 and e <= ends[-1]:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += e - s
tot = sum(v[1] for v in agg.values())
span = ends[-1] - t0
with open(out_csv, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls_per_step", "Total_ms_per_step", "Avg_us", "Percent"])
    for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, round(n / steps, 2), round(ns / steps / 1e6, 4), round(ns / n / 1e3, 2), round(100.0 * ns / tot, 2)])
    lib = sum(v[1] for k, v in agg.items() if "mphsir" in k)
    rccl = sum(v[1] for k, v in agg.items() if "ccl" in k.lower())
    nlib = sum(v[0] for k, v in agg.items() if "mphsir" in k)
    w.writerow(["# replayed steps", steps, "", "", ""])
    w.writerow(["# wall ms per replayed step (first replayed launch to last optimizer launch)", "", round(span / steps / 1e6, 4), "", ""])
    w.writerow(["# library kernels (mphsir::*)", round(nlib / steps, 1), round(lib / steps / 1e6, 4), "", ""])
    w.writerow(["# RCCL kernels", "", round(rccl / steps / 1e6, 4), "", ""])
    w.writerow(["# everything else (at::native, copies, fills)", round((sum(v[0] for v in agg.values()) - nlib) / steps, 1),
                round((tot - lib - rccl) / steps / 1e6, 4), "", ""])
print(open(out_csv).read()[-600:])
