"""diagnostic (GPU box, after `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`): where the last replayed
training step leaves the GPU idle.  Prints, for the span between the last two optimizer launches: wall time, the union of kernel
intervals (time with at least one kernel running), time with exactly one / two or more kernels running, and the longest idle gaps with
the kernels on either side.          python tools/diag/step_gaps.py <trace_dir> [top] [mark]
(mark: substring of a kernel that runs exactly once per step / forward; default flat_adamw = the optimizer of a training step)"""
import csv, glob, os, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
mark = sys.argv[3] if len(sys.argv) > 3 else "flat_adamw"
f = (glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(d, "*_kernel_trace.csv")))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
lo, hi = ad[-2] + 1, ad[-1] + 1
step = rows[lo:hi]
t0, t1 = int(rows[ad[-2]]["End_Timestamp"]), int(step[-1]["End_Timestamp"])
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, busy = 0, t0, [0, 0, 0]
for t, dlt in ev:
    busy[min(depth, 2)] += t - last
    depth, last = depth + dlt, t
print("launches %d   wall %.3f ms   idle %.3f ms   one kernel %.3f ms   two or more %.3f ms   sum of kernel time %.3f ms" % (
    len(step), (t1 - t0) / 1e6, busy[0] / 1e6, busy[1] / 1e6, busy[2] / 1e6, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6))
# idle gaps: intervals with no kernel running
gaps, end_max, prev = [], t0, None
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end_max:
        gaps.append((s - end_max, prev, r))
    if e > end_max:
        end_max, prev = e, r
hist = {}
for g, _, _ in gaps:
    k = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else ">=8us"
    h = hist.setdefault(k, [0, 0])
    h[0] += 1
    h[1] += g
print("idle gaps:", {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
q = "Queue_Id"
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:top]:
    print("%7.1f us   after q=%s %-48s before q=%s %s" % (g / 1e3, a[q] if a else "-", (a["Kernel_Name"] if a else "-")[8:56], b[q], b["Kernel_Name"][8:60]))
# optional 4th argument: write every launch of that step in issue order (kernel, duration, grid, workgroup, LDS) to a CSV -- with the
# serial trace (tools/lab/step_sweeps.sh serial) this is the per-launch cost table
if len(sys.argv) > 4:
    import re
    with open(sys.argv[4], "w") as out:
        out.write("index,kernel,us,grid,workgroup,lds_bytes\n")
        for i, r in enumerate(step):
            n = re.sub(r"^_ZN6mphsir\d+|^void mphsir::|^mphsir::", "", r["Kernel_Name"])[:70].replace(",", ";")
            out.write("%d,%s,%.2f,%s,%s,%s\n" % (i, n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", ""),
                                                r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", ""))))
