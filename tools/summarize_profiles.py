#!/usr/bin/env python3
"""Turn rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

    python tools/summarize_profiles.py <tag> <trace_dir> [<pmc_fetch_dir> <pmc_write_dir> [<pmc_sq_dir>]]

* <trace_dir>: `rocprofv3 --kernel-trace --stats --output-format csv` of `bench.py` -> profiles/<tag>_kernel_stats.csv
* PMC dirs: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command (MI355X_MICROARCH.md
  §HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts exactly half the bytes of wide coalesced
  reads -> doubled here) -> profiles/<tag>_pmc_summary.json with per-kernel mean HBM bytes per launch.
* <pmc_sq_dir>: a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE
  GRBM_GUI_ACTIVE` pass -> per kernel: mfma_util = MFMA-busy cycles / (GRBM_GUI_ACTIVE/8 XCDs x 256 CUs x 4 SIMDs) (the gfx94x
  MfmaUtil formula: ROCm 7.2 ships no gfx950 derived-counter section, MI355X_MICROARCH.md), wave_parked = SQ_WAIT_ANY /
  SQ_WAVE_CYCLES, lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"mphsir\d*(\w+?)_kernel", name) or re.search(r"mphsir::(\w+?)_kernel", name)
    if m:
        n = re.sub(r"^\d+", "", m.group(1))
        n = {"qkv_dwconv_gram_rows": "qkv_dwconv_gram"}.get(n, n)
        # kernel forms that share one C-ABI entry point / library kernel id
        return {"gemm_tn_tr": "gemm_tn", "gemm_tn_tr_group": "gemm_tn", "gemm_tn_ring": "gemm_tn", "gemm_tn_ring_group": "gemm_tn", "gemm_tok_ring": "gemm_tok", "mlp_combine": "gated_mlp", "mlp_bwd_combine": "gated_mlp_bwd", "gated_mlp_bwd2": "gated_mlp_bwd", "gated_mlp_lds": "gated_mlp", "dwconv_gram2": "dwconv_gram",
                "pg_gate_fwd": "pg_gate", "dwconv3x3_tile": "dwconv3x3", "dwconv3x3_wgrad_tile": "dwconv3x3_wgrad", "dwconv_gate_tile": "dwconv_gate", "tvsp_text_map": "resample", "tvsp_text_map_bwd": "resample", "resize_bilinear": "resample",
                "resize_bilinear_bwd": "resample", "grad_check": "flat_adamw", "scaler_update": "flat_adamw"}.get(n, n)
    return name[:60]


def counter_per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv"))
    agg = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        k = short(row["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(row["Counter_Value"])
    return agg


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    out = os.environ.get("MPHSIR_PROFILE_OUT", os.path.join(ROOT, "profiles"))      # (on the GPU box: a directory under gpurun_out/)
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(trace, "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
    summ = {}
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        e = summ.setdefault(k, {"calls": 0, "total_ms": 0.0})
        e["calls"] += int(r["Calls"])
        e["total_ms"] += float(r["TotalDurationNs"]) / 1e6
    for e in summ.values():
        e["avg_us"] = round(e["total_ms"] * 1e3 / e["calls"], 2)
        e["total_ms"] = round(e["total_ms"], 3)
    if len(sys.argv) >= 5:
        fe, wr = counter_per_kernel(sys.argv[3], "FETCH_SIZE"), counter_per_kernel(sys.argv[4], "WRITE_SIZE")
        for k in summ:
            if k in fe and k in wr and fe[k][0] and wr[k][0]:
                rd = fe[k][1] / fe[k][0] * 1024 * 2       # KiB -> bytes, gfx950 half-count correction
                wb = wr[k][1] / wr[k][0] * 1024
                summ[k]["hbm_read_bytes_per_launch"] = round(rd)
                summ[k]["hbm_write_bytes_per_launch"] = round(wb)
    if len(sys.argv) >= 6:
        names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ACTIVE",
                 "GRBM_GUI_ACTIVE"]
        cnt = {n: counter_per_kernel(sys.argv[5], n) for n in names}
        for k in summ:
            v = {n: (cnt[n][k][1] / cnt[n][k][0] if k in cnt[n] and cnt[n][k][0] else None) for n in names}
            if v["GRBM_GUI_ACTIVE"]:
                summ[k]["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 4) if v["SQ_VALU_MFMA_BUSY_CYCLES"] is not None else None
            if v["SQ_WAVE_CYCLES"]:
                summ[k]["wave_parked"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4) if v["SQ_WAIT_ANY"] is not None else None
            if v["SQ_LDS_ACTIVE"]:
                summ[k]["lds_conflict"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_ACTIVE"], 4) if v["SQ_LDS_BANK_CONFLICT"] is not None else None
    top = dict(sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:40])
    with open(os.path.join(out, tag + "_pmc_summary.json"), "w") as f:
        json.dump(top, f, indent=1, sort_keys=True)
    for k, e in list(top.items())[:22]:
        print("%-28s calls %5d avg %9.2f us  rd %s wr %s  mfma %s parked %s ldsconf %s" % (
            k, e["calls"], e["avg_us"], e.get("hbm_read_bytes_per_launch"), e.get("hbm_write_bytes_per_launch"), e.get("mfma_util"),
            e.get("wave_parked"), e.get("lds_conflict")))


if __name__ == "__main__":
    main()
