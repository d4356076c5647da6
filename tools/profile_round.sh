#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{trace,fetch,write,sq}/ (+ .log); summarise with tools/summarize_profiles.py
# Counters go in their own passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-spectral"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- $B --steps 8 > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $B --steps 2 > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $B --steps 2 > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- $B --steps 2 > gpurun_out/${tag}_sq.log 2>&1
# forward-only legs: batch 16 64x64 (BASELINE configs[1]) and one 512x512 cube
F="python3 bench.py --forward-only --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_fwd_b16_trace -- $F --batch 16 --steps 8 > gpurun_out/${tag}_fwd_b16_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_fwd512_trace -- $F --batch 1 --patch 512 --steps 8 > gpurun_out/${tag}_fwd512_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fwd512_fetch -- $F --batch 1 --patch 512 --steps 2 > gpurun_out/${tag}_fwd512_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_fwd512_write -- $F --batch 1 --patch 512 --steps 2 > gpurun_out/${tag}_fwd512_write.log 2>&1
# keep the merged-back payload small: per-kernel stats + counter CSVs only
find gpurun_out/${tag}_* -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh gpurun_out/${tag}_* | tail -12
