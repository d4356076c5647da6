# usage (through gpurun, from the repo root): bash tools/lab/final_run.sh   -- tag of the outputs: see the r06e below
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06e_fulltests.log 2>&1; echo rc=$? >> gpurun_out/r06e_fulltests.log; tail -3 gpurun_out/r06e_fulltests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r06e > gpurun_out/r06e_profile.log 2>&1
bash tools/lab/step_sweeps.sh serial r06e > gpurun_out/r06e_serial.log 2>&1; cat gpurun_out/r06e_serial.log | head -2
python3 tools/diag/glue_sites.py > gpurun_out/r06e_glue_sites.log 2>&1
tail -5 gpurun_out/r06e_replay.log
python3 bench.py > gpurun_out/r06e_bench.json 2> gpurun_out/r06e_bench.err; cut -c1-200 gpurun_out/r06e_bench.json
