# A/B of kernel libraries on the encoder step: tools/ab_encoder.sh <lib basename> ...   (libs in diff-mst_amd/lib; "base" = default)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in base "$@"; do
  if [ "$v" != base ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  out=$R/gpurun_out/abenc_$v; rm -rf $out
  timeout 600 rocprofv3 --kernel-trace --stats -d $out -o r --output-format csv -- python $R/tools/encoder_bench.py ${NS:-16} ${NSAMP:-262144} bf16 3 > $out.log 2>&1
  echo "== $v: $(grep encoder $out.log | tr '\n' ' ')"
  python - <<PY
import csv, glob
f = glob.glob("$out/**/r_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:${TOP:-10}]:
    print(f"   {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
done
