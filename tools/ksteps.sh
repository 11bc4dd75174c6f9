#!/bin/bash
# Per-kernel listing of one bench step at several batch sizes (GPU box): tools/ksteps.sh 2 8 32
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for bs in "$@"; do
  out=$R/gpurun_out/kstep_bs$bs; rm -rf $out
  rocprofv3 --kernel-trace -d $out -o r -- python $R/tools/step_bs.py $bs > $out.log 2>&1
  tail -1 $out.log
  python $R/tools/kstep.py $out/r_results.db "k_prep(" 6
done
