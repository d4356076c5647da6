cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
B="python3 bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-spectral --no-extra --steps 8"
for v in base per4; do
  if [ $v = per4 ]; then export MPHSIR_PACK_PER4=1; fi
  rm -rf gpurun_out/pk_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk_$v -- $B > /dev/null 2>&1
  echo "$v: $(grep -h 'pack_gather' gpurun_out/pk_$v/*/*kernel_stats.csv | head -3 | cut -c1-140)"
  rm -rf gpurun_out/pk_$v
done
