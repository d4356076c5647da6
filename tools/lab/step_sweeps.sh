#!/bin/bash
# lab (GPU box): whole-step experiments of round 5, one mode per call.  Each prints the bench line's value / ms_per_step per setting.
#   tools/lab/step_sweeps.sh ab NAME A B [bench flags]   A/B of one environment knob, twice each, on the training step
#   tools/lab/step_sweeps.sh sweep "K=V [K=V]" ...          each environment setting (plus the default, first and last) twice round-robin: ms per step
#   tools/lab/step_sweeps.sh side                         what each captured side branch buys (step, 512x512 forward, batch-16 forward)
#   tools/lab/step_sweeps.sh graphenv                     HIP runtime knobs for graph replay (branch queues, packet capture), with lab chains
#   tools/lab/step_sweeps.sh procs                        two / four bench processes on one GPU against one
#   tools/lab/step_sweeps.sh serial TAG                   kernel trace of the step with EVERY side branch off (one stream): per-kernel own cost
#                                                         -> gpurun_out/TAG_serial_replay_only.csv, TAG_serial_gaps.log
mode=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 40 --warmup 5"
line() { env "$@" python3 bench.py $F $X 2>&1 | tail -1 | cut -c50-150; }
case $mode in
ab)
  N=$1; A=$2; B=$3; shift 3; X="$*"
  for i in 1 2; do for v in $A $B; do echo "$N=$v: $(line $N=$v)"; done; done ;;
sweep)
  ms() { env "$@" python3 bench.py $F 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
  for i in 1 2; do echo "default: $(ms X=1)"; for e in "$@"; do echo "$e: $(ms $e)"; done; echo "default: $(ms X=1)"; done ;;
side)
  for X in "" "--forward-only --patch 512 --batch 1" "--forward-only --batch 16"; do
    echo "### $X"
    for e in "X=1" "MPHSIR_SIDE_BRANCH=0" "MPHSIR_DW_SIDE=0" "MPHSIR_DW_SIDE=1" "MPHSIR_PROMPT_SIDE=0" "MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0" \
             "MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0 MPHSIR_PROMPT_SIDE=0" "MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0 MPHSIR_PROMPT_SIDE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "X=1"; do
      echo "$e: $(line $e)"
    done
  done ;;
graphenv)
  for e in "X=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" \
           "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "GPU_MAX_HW_QUEUES=8"; do
    echo "=== $e"; env $e python3 tools/lab/stream_overlap.py 2>&1 | grep numel | cut -c1-260; echo "step: $(line $e)"
  done ;;
procs)
  echo "batch 32 alone: $(line X=1)"; X="--batch 16"; echo "batch 16 alone: $(line X=1)"
  for n in 2 4; do
    X="--batch $((32 / n)) --steps 60"; pids=""
    for k in $(seq $n); do (echo "$n processes, batch $((32 / n)): $(line X=1)") & pids="$pids $!"; done
    for p in $pids; do wait $p; done
  done ;;
serial)
  tag=$1
  export MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0 MPHSIR_PROMPT_SIDE=0
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_serial_trace -- python3 bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-spectral --no-extra --steps 8 > gpurun_out/${tag}_serial_trace.log 2>&1
  python3 tools/replay_slice.py gpurun_out/${tag}_serial_trace gpurun_out/${tag}_serial_replay_only.csv 5 > gpurun_out/${tag}_serial_replay.log 2>&1
  python3 tools/diag/step_gaps.py gpurun_out/${tag}_serial_trace 12 flat_adamw gpurun_out/${tag}_serial_step_launches.csv > gpurun_out/${tag}_serial_gaps.log 2>&1
  find gpurun_out/${tag}_serial_trace -name "*.csv" -delete 2>/dev/null
  head -3 gpurun_out/${tag}_serial_gaps.log ;;
*) echo "modes: ab side graphenv procs serial"; exit 2 ;;
esac
