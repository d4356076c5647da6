set -x
cd $GRAFT_REPO_ROOT
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 40 --warmup 5"
python bench.py $F --batch 32 > gpurun_out/conc_b32.log 2>&1
python bench.py $F --batch 16 > gpurun_out/conc_b16_solo.log 2>&1
python bench.py $F --batch 16 > gpurun_out/conc_b16_a.log 2>&1 &
P1=$!
python bench.py $F --batch 16 > gpurun_out/conc_b16_b.log 2>&1 &
P2=$!
wait $P1; wait $P2
python bench.py $F --batch 8 --steps 60 > gpurun_out/conc_b8_a.log 2>&1 &
P1=$!
python bench.py $F --batch 8 --steps 60 > gpurun_out/conc_b8_b.log 2>&1 &
P2=$!
python bench.py $F --batch 8 --steps 60 > gpurun_out/conc_b8_c.log 2>&1 &
P3=$!
python bench.py $F --batch 8 --steps 60 > gpurun_out/conc_b8_d.log 2>&1 &
P4=$!
wait $P1; wait $P2; wait $P3; wait $P4
tail -n 1 gpurun_out/conc_*.log
