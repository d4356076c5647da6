cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r06d_fulltests.log 2>&1; echo "rc=$?" >> gpurun_out/r06d_fulltests.log
tail -5 gpurun_out/r06d_fulltests.log
R="python3 bench.py --model remote_sensing --dtype f16 --batch 16 --warmup 5 --no-cpu-baseline --no-roofline --no-spectral --no-extra --steps 40"
$R 2>/dev/null | tail -1 | cut -c50-140 > gpurun_out/r06d_rs.log
MPHSIR_SPECTRAL_BWD_FUSED=0 $R 2>/dev/null | tail -1 | cut -c50-140 >> gpurun_out/r06d_rs.log
cat gpurun_out/r06d_rs.log
