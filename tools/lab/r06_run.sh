cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gated_mlp" > gpurun_out/r06q_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06q_tests.log
tail -2 gpurun_out/r06q_tests.log
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2 3; do echo "$(python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"; done
