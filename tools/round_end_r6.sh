# Round-6 evidence run on the GPU box (the -m gpu suite is run separately): tools/round_end_r6.sh main | cfg3
# main: smoke, counter passes of the headline bench, the full bench line.  cfg3: FETCH / WRITE passes + kernel table of cfg #3.
# Everything lands under gpurun_out/ (trimmed to what tools/make_profile_r6.py reads: the merge back is capped at 64 MiB).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
if [ "${1:-main}" = main ]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  bash tools/pmc_passes.sh r6 > gpurun_out/r6_passes.log 2>&1; tail -3 gpurun_out/r6_passes.log
  find gpurun_out/r6_stats -name "*.db" -delete
  for d in sq1 sq2 fetch write; do find gpurun_out/r6_$d -type f ! -name "*.db" -delete; done
  du -sh gpurun_out/*
  cd $R && timeout 1200 python bench.py > gpurun_out/r6_bench.log 2>&1; tail -c 400 gpurun_out/r6_bench.log
else
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/cfg3_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/cfg3_$c -o r -- python $R/tools/cfg3_bench.py > $R/gpurun_out/cfg3_$c.log 2>&1
    python $R/tools/traffic.py $(find $R/gpurun_out/cfg3_$c -name "*.db" | head -1) $c 8 > $R/gpurun_out/cfg3_$c.json
    rm -rf $R/gpurun_out/cfg3_$c
  done
  cd $R && bash tools/cfg3_prof.sh > gpurun_out/r6_cfg3_prof.txt 2>&1; tail -3 gpurun_out/r6_cfg3_prof.txt
  rm -rf gpurun_out/cfg3_prof
fi
du -sh $R/gpurun_out
