set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_branch_sum or test_gated_mlp" 2>&1 | tail -5
python -m pytest tests/test_gpu_model.py -x -q -k "fused_block or block_gradients or tiny" 2>&1 | tail -5
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 40 --warmup 5"
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F 2>&1 | tail -1 | cut -c1-120
python bench.py $F 2>&1 | tail -1 | cut -c1-120
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F --forward-only --patch 512 --batch 1 2>&1 | tail -1 | cut -c1-120
python bench.py $F --forward-only --patch 512 --batch 1 2>&1 | tail -1 | cut -c1-120
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F --forward-only --batch 16 2>&1 | tail -1 | cut -c1-120
python bench.py $F --forward-only --batch 16 2>&1 | tail -1 | cut -c1-120
MPHSIR_MLP_FUSE_SUM=0 python bench.py $F 2>&1 | tail -1 | cut -c1-120
python bench.py $F 2>&1 | tail -1 | cut -c1-120
