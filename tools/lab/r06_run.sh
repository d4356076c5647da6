cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r06c_model.log 2>&1; tail -2 gpurun_out/r06c_model.log
B="python3 bench.py --warmup 5 --no-cpu-baseline --no-roofline --no-spectral --no-extra --steps 60"
for i in 1 2 3; do $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
bash tools/profile_round.sh r06c > gpurun_out/r06c_profile.log 2>&1
bash tools/lab/step_sweeps.sh serial r06c > gpurun_out/r06c_serial.log 2>&1; cat gpurun_out/r06c_serial.log
python3 tools/diag/glue_sites.py > gpurun_out/r06c_glue_sites.log 2>&1; head -3 gpurun_out/r06c_glue_sites.log
cat gpurun_out/r06c_replay.log | tail -6
