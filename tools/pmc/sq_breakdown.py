"""Diagnostic (runs on the GPU box): per-kernel wait / issue breakdown from two rocprofv3 --pmc passes.
    python tools/sq_breakdown.py <dir_pass_A> <dir_pass_B>
pass A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass B: SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU"""
import csv
import glob
import os
import sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarize_profiles import short


def load(d):
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    agg = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
print("%-18s %7s %7s %7s %7s | %7s %7s %7s | %8s %8s %8s %8s" % ("kernel", "parked", "w_inst", "w_lds", "active", "a_lds", "a_valu", "a_vmem",
                                                               "ldsidx/b", "conf/idx", "mfma/lds", "valu/mfma"))
for k in sorted(a, key=lambda k: -a[k]["SQ_WAVE_CYCLES"])[:20]:
    x, y = a[k], b.get(k, defaultdict(float))
    wc, bc = max(x["SQ_WAVE_CYCLES"], 1), max(y["SQ_BUSY_CYCLES"], 1)
    print("%-18s %7.3f %7.3f %7.3f %7.3f | %7.3f %7.3f %7.3f | %8.3f %8.3f %8.2f %8.2f" % (
        k[:18], x["SQ_WAIT_ANY"] / wc, x["SQ_WAIT_INST_ANY"] / wc, x["SQ_WAIT_INST_LDS"] / wc, x["SQ_ACTIVE_INST_ANY"] / wc,
        x["SQ_ACTIVE_INST_LDS"] / wc, x["SQ_ACTIVE_INST_VALU"] / wc, x["SQ_ACTIVE_INST_VMEM"] / wc,
        y["SQ_LDS_IDX_ACTIVE"] / bc, y["SQ_LDS_BANK_CONFLICT"] / max(y["SQ_LDS_IDX_ACTIVE"], 1),
        y["SQ_INSTS_MFMA"] / max(y["SQ_INSTS_LDS"], 1), y["SQ_INSTS_VALU"] / max(y["SQ_INSTS_MFMA"], 1)))
