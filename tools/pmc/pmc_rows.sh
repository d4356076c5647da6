cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc/a -- python3 tools/pmc/run_rows_once.py 64 2 512 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmc/b -- python3 tools/pmc/run_rows_once.py 64 2 512 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/c -- python3 tools/pmc/run_rows_once.py 64 2 512 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in "abc":
    for f in glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rows_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "n=%d" % len(v), "last=%.4g" % v[-1])
PY
