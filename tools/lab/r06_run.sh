cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "b16_forward or block_gradients_bf16 or eval_forward_with_prompt or tiny_adamw or fused_block_equals" -s > gpurun_out/r06a_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06a_tests.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gated_mlp" > gpurun_out/r06a_mlp.log 2>&1; echo "rc=$?" >> gpurun_out/r06a_mlp.log
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 5 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -3 gpurun_out/r06a_tests.log; tail -2 gpurun_out/r06a_mlp.log; cut -c1-400 gpurun_out/r06a_bench.json
