#!/bin/bash
# cfg #5 step per encoder precision: kernel-family table + idle-gap attribution of a PIPELINED step (the window tools/cfg5_gaps.py takes
# lies in the un-synchronised loop at the end of tools/cfg5_step.py) -> gpurun_out/r5_cfg5_<precision>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
for prec in bf16 bf16x3 bf16x6 fp32; do
  export MST_ENCODER_PRECISION=$prec
  cd $R && timeout 600 bash tools/cfg5_prof.sh > gpurun_out/r5_cfg5_$prec.txt 2>&1 < /dev/null
  cd $R && python tools/cfg5_gaps.py gpurun_out/cfg5_prof/r_kernel_trace.csv >> gpurun_out/r5_cfg5_$prec.txt 2>&1
  echo "== $prec"; grep "pipelined" gpurun_out/cfg5_prof.log; grep -A3 "^one step" gpurun_out/r5_cfg5_$prec.txt | head -5
done
