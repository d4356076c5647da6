# lab: HIP runtime knobs for graph replay (branch queues, packet capture) -- lab chains and the training step
cd $GRAFT_REPO_ROOT
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 30 --warmup 5"
run() {
  echo "=== $*"
  env "$@" python tools/lab/stream_overlap.py 2>&1 | grep numel | cut -c1-260
  env "$@" python bench.py $F 2>&1 | tail -1 | cut -c1-140
}
run X=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run GPU_MAX_HW_QUEUES=8
