#!/bin/bash
# Developer A/B on the GPU box: per-kernel time of one console fwd+bwd step for several builds of the library.
# usage: tools/ab.sh "<grep pattern>" variant1 variant2 ...   (variants = basenames under diff-mst_amd/lib, without .so)
pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  out=/root/repo/gpurun_out/ab_$v
  rm -rf $out
  MST_HIP_LIB=/root/repo/diff-mst_amd/lib/$v.so rocprofv3 --kernel-trace -d $out -o r -- python /root/repo/tools/${AB_SCRIPT:-quick_bench.py} ${AB_ARGS:-8 8 262144 10} > $out.log 2>&1
  echo "== $v: $(grep -m1 'ms/step\|mixes' $out.log | cut -c1-110)"
  python /root/repo/tools/kstep.py $out/r_results.db "${AB_ANCHOR:-k_prep(}" 14 | grep -E "$pat|wall"
done
