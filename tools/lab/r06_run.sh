cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dxn" 2>&1 | tail -2
B="python3 bench.py --warmup 5 --no-cpu-baseline --no-roofline --no-spectral --no-extra --steps 60"
for i in 1 2 3; do
  MPHSIR_LIB_AB=ab/libmphsir_old.so $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/old /'
  $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new /'
done
