#!/bin/bash
# counts __amd_rocclr_copyBuffer / fill launches per variant of tools/copy_hunt.py (GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in fwd loss full full_direct lossbwd; do
  out=$R/gpurun_out/copy_hunt_$m; rm -rf $out
  rocprofv3 --kernel-trace -d $out -o r --output-format csv -- python $R/tools/copy_hunt.py $m > /dev/null 2>&1
  f=$(find $out -name r_kernel_trace.csv | head -1)
  echo "$m: copyBuffer $(grep -c rocclr_copyBuffer $f) fillBuffer $(grep -c rocclr_fillBuffer $f) at::native $(grep -c 'at::native' $f) of $(wc -l < $f) kernels in 20 iterations"
done
