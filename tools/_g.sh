cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra --no-spectral"
for i in 1 2; do
for f in 0 1 2; do
MPHSIR_DW_SIDE=$f $B 2>gpurun_out/err_$f.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DW_SIDE=$f', d['value'], d['ms_per_step'])"
done
done
MPHSIR_DW_SIDE=2 python -m pytest tests/test_gpu_model.py -x -q -k "deferred or adamw or determinism or world2_on_one or checkpoint" 2>&1 | tail -3
MPHSIR_DW_SIDE=1 python -m pytest tests/test_gpu_model.py -x -q -k "deferred or adamw or determinism or world2_on_one or checkpoint" 2>&1 | tail -3
