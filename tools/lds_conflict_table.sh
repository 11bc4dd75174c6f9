cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/lds
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/lds -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python - <<PY
import sqlite3, re
cur = sqlite3.connect("$R/gpurun_out/lds/r_results.db").cursor()
acc = {}
for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    n = re.sub(r"\(.*", "", k).replace("void ", "").replace("mst::", "")
    acc.setdefault(n, {}).setdefault(c, 0.0)
    acc[n][c] += v
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:14]:
    a = d.get("SQ_LDS_IDX_ACTIVE", 0); b = d.get("SQ_LDS_BANK_CONFLICT", 0)
    print(f"{n[:48]:48s} LDS active {a:.3e}  conflict {b:.3e}  = {100*b/max(a,1):.0f} %")
PY
