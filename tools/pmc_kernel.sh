# per-wave issue / wait breakdown of ONE kernel from three rocprofv3 --pmc passes (GPU box):
#     bash tools/pmc_kernel.sh <kernel-name substring> <python script> [args...]        e.g. gated_mlp_lds_kernel tools/pmc/run_once.py mlp 128 340 131072
# (counter values are quad-cycles / instruction counts summed over the sampled waves; the last launch of the kernel is reported)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KN="$1"; shift
rm -rf gpurun_out/pmck && mkdir -p gpurun_out/pmck
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmck/a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmck/b -- python3 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmck/c -- python3 "$@" > /dev/null 2>&1
KN="$KN" python3 - <<'PY'
import csv, glob, collections, os
kn = os.environ["KN"]
val = {}
for d in "abc":
    for f in glob.glob("gpurun_out/pmck/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kn in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            val[k] = v[-1]
            print(d, k, "n=%d" % len(v), "last=%.4g" % v[-1])
w, wc = val.get("SQ_WAVES", 1), val.get("SQ_WAVE_CYCLES", 1)
print("per wave: cycles %.0f  VALU %.0f  SALU %.0f  LDS %.0f  MFMA %.0f  VMEM %.0f instructions" % tuple(
    val.get(k, 0) / w for k in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM")))
print("of a wave's cycles: parked %.2f  stalled at issue %.2f  issuing %.2f (VALU %.2f  LDS %.2f  scalar %.2f  VMEM %.2f)   waiting on LDS %.2f" % tuple(
    val.get(k, 0) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
                                 "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS")))
bc = val.get("SQ_BUSY_CYCLES", 1)
print("per SQ busy cycle: MFMA pipe busy %.3f  LDS index active %.3f  bank-conflict share of it %.3f" % (
    val.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / bc, val.get("SQ_LDS_IDX_ACTIVE", 0) / bc, val.get("SQ_LDS_BANK_CONFLICT", 0) / max(val.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
rm -rf gpurun_out/pmck
