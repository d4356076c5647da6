cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu -k "block_gradients or whole_net or tiny or full_width_forward or b16" > gpurun_out/r06i_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06i_tests.log
tail -3 gpurun_out/r06i_tests.log
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2; do
  echo "default:        $(python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
  echo "dm tokens 4096: $(MPHSIR_FOLD_BWD_DM_TOKENS=4096 python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
done
