cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2; do
  for e in "MPHSIR_DW_DEFER=1" "MPHSIR_DW_DEFER=0" "MPHSIR_DW_DEFER=1 MPHSIR_DW_DEFER_ROWS=30000"; do
    echo "$e: $(env $e python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
  done
done > gpurun_out/r06e_defer.log 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "deferred or tiny_adamw or graph_replay or no_parameter_gradient or data_parallel_world2_on_one" > gpurun_out/r06e_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06e_tests.log
cat gpurun_out/r06e_defer.log; tail -3 gpurun_out/r06e_tests.log
