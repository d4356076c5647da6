cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r04e.json 2> gpurun_out/bench_r04e.err; tail -c 600 gpurun_out/bench_r04e.json
bash tools/profile_round.sh r04e > gpurun_out/profile_r04e_inner.log 2>&1
tail -3 gpurun_out/profile_r04e_inner.log
