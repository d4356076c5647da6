cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "heads or dxn or conv" > gpurun_out/r06c_tests.log 2>&1; tail -3 gpurun_out/r06c_tests.log
timeout 2400 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/r06c_model.log 2>&1; tail -3 gpurun_out/r06c_model.log
B="python3 bench.py --warmup 5 --no-cpu-baseline --no-roofline --no-spectral --no-extra --steps 60"
for i in 1 2 3; do
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  MPHSIR_BASE_SKIP_BWD=0 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/noskipbwd /'
done
python3 tools/diag/glue_sites.py > gpurun_out/r06c_glue_sites.log 2>&1; head -3 gpurun_out/r06c_glue_sites.log
