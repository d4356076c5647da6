cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "spectral_dqkv or channel_attention_bwd or pgsstb_backward" > gpurun_out/r06b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06b_tests.log
timeout 600 python tools/bench/bench_spectral_bwd.py > gpurun_out/r06b_bench_sb.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-spectral --steps 40 --warmup 5 > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err
MPHSIR_SPECTRAL_BWD_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5 > gpurun_out/r06b_bench_off.json 2> gpurun_out/r06b_bench_off.err
tail -3 gpurun_out/r06b_tests.log; cat gpurun_out/r06b_bench_sb.log; cut -c1-200 gpurun_out/r06b_bench.json; cut -c1-200 gpurun_out/r06b_bench_off.json
