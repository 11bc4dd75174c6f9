#!/bin/bash
# Developer A/B on the GPU box with ENV switches: tools/abenv.sh "<grep pattern>" "NAME1 ENV=..." "NAME2 ENV=..."
pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; name=$1; shift
  out=/root/repo/gpurun_out/ab_$name
  rm -rf $out
  env "$@" rocprofv3 --kernel-trace -d $out -o r -- python /root/repo/tools/${AB_SCRIPT:-quick_bench.py} ${AB_ARGS:-8 8 262144 10} > $out.log 2>&1
  echo "== $name ($*): $(grep -m1 'ms/step\|mixes' $out.log | cut -c1-110)"
  python /root/repo/tools/kstep.py $out/r_results.db "${AB_ANCHOR:-k_prep(}" 14 | grep -E "$pat|wall"
done
