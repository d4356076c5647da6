cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fold_bwd or channel_attention_bwd" > gpurun_out/r06c_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06c_tests.log
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2; do
  python bench.py $F 2>/dev/null | tail -1 | cut -c50-140
  MPHSIR_LIB_AB=ab/libmphsir_base.so python bench.py $F 2>/dev/null | tail -1 | cut -c50-140
done > gpurun_out/r06c_ab.log 2>&1
tail -2 gpurun_out/r06c_tests.log; cat gpurun_out/r06c_ab.log
