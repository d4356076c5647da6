"""diagnostic (GPU box, after `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`): what runs in the last
~0.6 ms before the gradient hand-over (multi_copy) of the last replayed step, per queue -- is the weight-gradient branch the long pole
at the final join?      python tools/diag/step_tail.py <trace_dir>"""
import csv, glob, os, sys
d = sys.argv[1]
f = (glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(d, "*_kernel_trace.csv")))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
mc = [i for i, r in enumerate(rows) if "multi_copy" in r["Kernel_Name"]]
i = mc[-1]
t_mc = int(rows[i]["Start_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", list(rows[0].keys()))
print("kernels ending within 600 us before the hand-over (start_us, end_us relative to it, queue, name):")
for r in rows[max(0, i - 60):i + 2]:
    s, e = int(r["Start_Timestamp"]) - t_mc, int(r["End_Timestamp"]) - t_mc
    if e > -600000:
        print("%9.1f %9.1f  q=%s  %s" % (s / 1e3, e / 1e3, r.get(qkey, "?") if qkey else "?", r["Kernel_Name"][:70]))
