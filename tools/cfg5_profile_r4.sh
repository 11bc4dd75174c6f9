#!/bin/bash
# cfg #5 step at both encoder precisions: kernel-family table + idle-gap attribution -> gpurun_out/r4_cfg5_{bf16,fp32}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for prec in bf16 fp32; do
  export MST_ENCODER_PRECISION=$prec
  cd $R && bash tools/cfg5_prof.sh > gpurun_out/r4_cfg5_$prec.txt 2>&1
  cd $R && python tools/cfg5_gaps.py gpurun_out/cfg5_prof/r_kernel_trace.csv >> gpurun_out/r4_cfg5_$prec.txt 2>&1
  echo "== $prec"; head -4 gpurun_out/r4_cfg5_$prec.txt; grep -A4 "^one step" gpurun_out/r4_cfg5_$prec.txt | head -8
done
