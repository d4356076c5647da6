# lab: what each captured side branch buys on the training step and the forward legs
cd $GRAFT_REPO_ROOT
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 40 --warmup 5"
run() { echo "$*: $(env "$@" python bench.py $F $X 2>&1 | tail -1 | cut -c50-150)"; }
for X in "" "--forward-only --patch 512 --batch 1" "--forward-only --batch 16"; do
echo "### $X"
run X=1
run MPHSIR_SIDE_BRANCH=0
run MPHSIR_DW_SIDE=0
run MPHSIR_DW_SIDE=1
run MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0
run MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0 MPHSIR_PROMPT_SIDE=0
run MPHSIR_SIDE_BRANCH=0 MPHSIR_DW_SIDE=0 MPHSIR_PROMPT_SIDE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run MPHSIR_PROMPT_SIDE=0
run X=1
done
