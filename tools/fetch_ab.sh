cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base prev; do
  if [ "$v" != base ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/f_$v -o r --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline > /tmp/f_$v.log 2>&1 < /dev/null
  f=$(find /tmp/f_$v -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r["Counter_Name"] != "FETCH_SIZE": continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mst::", "")[:40]
    agg[n][0] += 1; agg[n][1] += float(r["Counter_Value"])
print("== $v")
for n, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]: print(f"  {n:40s} calls {c:4d}  FETCH_SIZE per call {v / c / 1024:.1f} MB (raw, KB units)")
PY
done
