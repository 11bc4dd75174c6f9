# LDS bank-conflict share of one kernel for several library builds: tools/lds_conflict_ab.sh "<kernel regex>" lib1 lib2 ...
pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -n "$v" ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  rm -rf $R/gpurun_out/lds_$v
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/lds_$v -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python - <<PY
import sqlite3, re
cur = sqlite3.connect("$R/gpurun_out/lds_$v/r_results.db").cursor()
acc = {}
for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    n = re.sub(r"\(.*", "", k).replace("void ", "").replace("mst::", "")
    if re.search(r"$pat", n):
        acc.setdefault(n, {}).setdefault(c, 0.0); acc[n][c] += v
for n, d in acc.items():
    print(f"[${v:-default}] {n[:44]:44s} insts {d.get('SQ_INSTS_LDS',0):.3e} active {d.get('SQ_LDS_IDX_ACTIVE',0):.3e} conflict {d.get('SQ_LDS_BANK_CONFLICT',0):.3e} = {100*d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):.0f} %")
PY
done
