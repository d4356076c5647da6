# lab: A/B of one environment knob on the training step: bash tools/lab/ab.sh NAME VALUE_A VALUE_B [bench flags]
cd $GRAFT_REPO_ROOT
N=$1; A=$2; B=$3; shift 3
F="--no-extra --no-cpu-baseline --no-roofline --no-spectral --steps 40 --warmup 5 $*"
for i in 1 2; do
  for v in $A $B; do echo "$N=$v: $(env $N=$v python bench.py $F 2>&1 | tail -1 | cut -c1-150)"; done
done
