cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2 3; do
  echo "default: $(python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
  echo "max-ilp: $(MPHSIR_LIB_AB=ab/libmphsir_ilp.so python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
done
