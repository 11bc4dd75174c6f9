#!/bin/bash
# Kernel-family table of the cfg #5 step (GPU box): bash tools/cfg5_prof.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/cfg5_prof; rm -rf $out
rocprofv3 --kernel-trace -d $out -o r --output-format csv -- python $R/tools/cfg5_step.py 3 > $out.log 2>&1
grep "cfg5 step\|pipelined" $out.log
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/r_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last k_spectrogram pair start: take the final third of the trace by time
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
last = [r for r in rows if int(r["Start_Timestamp"]) > t1 - (t1 - t0) / 4.2]
fam = collections.defaultdict(float)
for r in last:
    n = r["Kernel_Name"].replace("void ", "").replace("mst::", "")
    fam[n.split("<")[0].split("(")[0][:48]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(fam.values())
print(f"kernels in the window: {len(last)}, busy {tot/1e3:.2f} ms of {(int(last[-1]['End_Timestamp'])-int(last[0]['Start_Timestamp']))/1e6:.2f} ms")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:28]: print(f"  {k:48s} {v:9.1f} us {100*v/tot:5.1f} %")
PY
