cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "ln_bwd_tok_with" > gpurun_out/r06l_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06l_tests.log
tail -3 gpurun_out/r06l_tests.log
