# SQ counters of ONE kernel on its own: bash tools/pmc_kernel.sh <kernel-name substring> <python tool + args ...>   (GPU box, through gpurun)
#   bash tools/pmc_kernel.sh win_attn_kernel tools/pmc/run_win_once.py 128 2 512
#   bash tools/pmc_kernel.sh gated_mlp_bwd tools/pmc/run_mlp_bwd_once.py 128 340 131072
# Three passes (the counters do not fit one); prints the LAST launch's value of every counter.  Values are quad-cycles / instruction counts
# summed over the chip; per wave = value / SQ_WAVES.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=$1; shift
mkdir -p gpurun_out/pmck
A="python3 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmck/a -- $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmck/b -- $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmck/c -- $A > /dev/null 2>&1
KERNEL_SUBSTR=$K python3 - <<'PY'
import csv, glob, collections, os
k = os.environ["KERNEL_SUBSTR"]
vals = {}
for d in "abc":
    for f in glob.glob("gpurun_out/pmck/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            vals[c] = v[-1]
            print(d, c, "n=%d" % len(v), "last=%.4g" % v[-1])
w = vals.get("SQ_WAVES", 0)
if w:
    print("per wave:", {c: round(v / w, 1) for c, v in vals.items() if c.startswith("SQ_INSTS") or c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS")})
PY
find gpurun_out/pmck -name "*.csv" -delete
