set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_branch_sum" 2>&1 | tail -2
python -m pytest tests/test_gpu_model.py -x -q -k "fused_block" 2>&1 | tail -2
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 40 --warmup 5"
for i in 1 2; do
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F 2>&1 | tail -1 | cut -c50-150
python bench.py $F 2>&1 | tail -1 | cut -c50-150
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F --forward-only --patch 512 --batch 1 2>&1 | tail -1 | cut -c50-150
python bench.py $F --forward-only --patch 512 --batch 1 2>&1 | tail -1 | cut -c50-150
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F --forward-only --batch 16 2>&1 | tail -1 | cut -c50-150
python bench.py $F --forward-only --batch 16 2>&1 | tail -1 | cut -c50-150
done
