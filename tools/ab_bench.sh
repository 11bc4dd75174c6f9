# A/B of kernel libraries on the bench step: tools/ab_bench.sh <lib basename> ...   (libs in diff-mst_amd/lib, "" = default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ "$v" != base ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/ab_$v -o r -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/ab_$v.log 2>&1
  python - <<PY
import json
for line in open("$R/gpurun_out/ab_$v.log"):
    if line.startswith("{"):
        b = json.loads(line); print("== $v", round(b["value"]), round(b["ms_per_step"], 4), {k: round(x, 4) for k, x in b["roofline"]["stages"].items()})
PY
done
cd $R && python tools/kavg.py "${KPAT:-cascade|coefgrad|allpole}" $(find gpurun_out/ab_* -name "*.db" | sort) 2>&1 | head -80
