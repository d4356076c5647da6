cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "fold_bwd or channel_attention_bwd or pgsstb_backward or block_gradients or whole_net or tiny" > gpurun_out/r06h_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06h_tests.log
tail -3 gpurun_out/r06h_tests.log
