#!/bin/bash
# Per-kernel table of the encoder step on the GPU box: tools/encoder_prof.sh [signals] [samples] [precision]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/enc_prof; rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o r --output-format csv -- python $R/tools/encoder_bench.py ${1:-16} ${2:-262144} ${3:-bf16} 3 > $out.log 2>&1
cat $out.log | grep encoder
python - <<PY
import csv, glob
f = glob.glob("$out/**/r_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
