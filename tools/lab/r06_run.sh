cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
F="--no-cpu-baseline --no-extra --no-spectral --no-roofline --steps 40 --warmup 5"
for i in 1 2 3; do
  for e in "MPHSIR_GDFN_DW_BWD=1" "MPHSIR_GDFN_DW_BWD=0"; do
    echo "$e: $(env $e python bench.py $F 2>/dev/null | tail -1 | cut -c50-140)"
  done
done
