cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_tok" 2>&1 | tail -3
MPHSIR_TOK_FORM=1 python tools/bench_gemm_shapes.py 2>&1 | grep -v amdgpu > gpurun_out/gemm_shapes_form1.log
MPHSIR_TOK_FORM=2 python tools/bench_gemm_shapes.py 2>&1 | grep -v amdgpu > gpurun_out/gemm_shapes_form2.log
tail -1 gpurun_out/gemm_shapes_form1.log; tail -1 gpurun_out/gemm_shapes_form2.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra --no-spectral"
for i in 1 2; do
for f in 1 0 2; do
MPHSIR_TOK_FORM=$f $B 2>gpurun_out/err_$f.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TOK_FORM=$f', d['value'], d['ms_per_step'])"
done
done
