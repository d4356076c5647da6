#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{trace,fetch,write,sq}/ (+ .log); summarise with tools/summarize_profiles.py
# Counters go in their own passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-spectral --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- $B --steps 8 > gpurun_out/${tag}_trace.log 2>&1
python3 tools/replay_slice.py gpurun_out/${tag}_trace gpurun_out/${tag}_train_replay_only.csv 5 > gpurun_out/${tag}_replay.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $B --steps 2 > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $B --steps 2 > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- $B --steps 2 > gpurun_out/${tag}_sq.log 2>&1
# forward-only legs: batch 16 64x64 (BASELINE configs[1]) and one 512x512 cube
F="python3 bench.py --forward-only --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_fwd_b16_trace -- $F --batch 16 --steps 8 > gpurun_out/${tag}_fwd_b16_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_fwd512_trace -- $F --batch 1 --patch 512 --steps 8 > gpurun_out/${tag}_fwd512_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fwd512_fetch -- $F --batch 1 --patch 512 --steps 2 > gpurun_out/${tag}_fwd512_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_fwd512_write -- $F --batch 1 --patch 512 --steps 2 > gpurun_out/${tag}_fwd512_write.log 2>&1
# BASELINE configs[4]: remote-sensing training, batch 16, fp16 + loss scaling (the same four passes)
R="python3 bench.py --model remote_sensing --dtype f16 --batch 16 --warmup 3 --no-cpu-baseline --no-roofline --no-spectral --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_rs_trace -- $R --steps 8 > gpurun_out/${tag}_rs_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_rs_fetch -- $R --steps 2 > gpurun_out/${tag}_rs_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_rs_write -- $R --steps 2 > gpurun_out/${tag}_rs_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_rs_sq -- $R --steps 2 > gpurun_out/${tag}_rs_sq.log 2>&1
# summarise on the box (the SQ counter CSVs are ~40 MB each; gpurun merges back at most 64 MiB): gpurun_out/<tag>_summaries/
export MPHSIR_PROFILE_OUT=gpurun_out/${tag}_summaries
python3 tools/summarize_profiles.py ${tag}_train_b32_bf16_graph gpurun_out/${tag}_trace gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq > gpurun_out/${tag}_summaries_train.log 2>&1
python3 tools/summarize_profiles.py ${tag}_fwd_b16_bf16_graph gpurun_out/${tag}_fwd_b16_trace > gpurun_out/${tag}_summaries_fwd16.log 2>&1
python3 tools/summarize_profiles.py ${tag}_fwd512_b1_bf16_graph gpurun_out/${tag}_fwd512_trace gpurun_out/${tag}_fwd512_fetch gpurun_out/${tag}_fwd512_write > gpurun_out/${tag}_summaries_fwd512.log 2>&1
python3 tools/summarize_profiles.py ${tag}_rs_train_b16_f16_graph gpurun_out/${tag}_rs_trace gpurun_out/${tag}_rs_fetch gpurun_out/${tag}_rs_write gpurun_out/${tag}_rs_sq > gpurun_out/${tag}_summaries_rs.log 2>&1
cp gpurun_out/${tag}_train_replay_only.csv gpurun_out/${tag}_summaries/${tag}_train_b32_bf16_graph_replay_only.csv
find gpurun_out/${tag}_* -name "*.csv" ! -path "*_summaries/*" -delete 2>/dev/null
cat gpurun_out/${tag}_summaries_train.log | tail -24
