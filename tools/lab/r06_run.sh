cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -m gpu -k "channel_attention_bwd or block_gradients or tiny_net_gradients or whole_net" > gpurun_out/r06f_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06f_tests.log
tail -3 gpurun_out/r06f_tests.log
bash tools/profile_round.sh r06a > gpurun_out/r06a_profile.log 2>&1
tail -30 gpurun_out/r06a_profile.log
