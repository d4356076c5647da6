# lab: kernel trace of the replayed training step with EVERY side branch off (one stream, one queue): each kernel's time is then
# its own cost, not its co-scheduled span.  -> gpurun_out/<tag>_serial_replay_only.csv
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0 MPHSIR_PROMPT_SIDE=0
B="python3 bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-spectral --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_serial_trace -- $B --steps 8 > gpurun_out/${tag}_serial_trace.log 2>&1
python3 tools/replay_slice.py gpurun_out/${tag}_serial_trace gpurun_out/${tag}_serial_replay_only.csv 5 > gpurun_out/${tag}_serial_replay.log 2>&1
python3 tools/diag/step_gaps.py gpurun_out/${tag}_serial_trace 12 > gpurun_out/${tag}_serial_gaps.log 2>&1
find gpurun_out/${tag}_serial_trace -name "*.csv" -delete 2>/dev/null
tail -3 gpurun_out/${tag}_serial_trace.log | cut -c1-200; head -20 gpurun_out/${tag}_serial_gaps.log
