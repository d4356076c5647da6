cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "pgsstb_backward or block_gradients or whole_net or tiny_net or tiny_adamw or fused_block or fold_bwd or spectral_dqkv or graph_replay or deferred or batch32" > gpurun_out/r06n_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06n_tests.log
tail -3 gpurun_out/r06n_tests.log
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2 3; do
  for e in "MPHSIR_COMBINE_SIDE=1" "MPHSIR_COMBINE_SIDE=0"; do
    echo "$e: $(env $e python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
  done
done
