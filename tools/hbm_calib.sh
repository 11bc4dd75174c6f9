#!/bin/bash
# Known-bytes calibration of FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel-trace only): writes gpurun_out/hbm_calib.json
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_counters "$R/tools/ubench/hbm_counters.hip" || exit 1
for mib in 1024 64; do
  rm -rf "$O/calib_fetch" "$O/calib_write" "$O/calib_trace"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/calib_fetch" -o r -- /tmp/hbm_counters $mib > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/calib_write" -o r -- /tmp/hbm_counters $mib > /dev/null 2>&1
  rocprofv3 --kernel-trace -d "$O/calib_trace" -o r -- /tmp/hbm_counters $mib > /dev/null 2>&1
  CALIB_MIB=$mib python "$R/tools/hbm_calib.py" $(find "$O/calib_fetch" -name "*.db") $(find "$O/calib_write" -name "*.db") $(find "$O/calib_trace" -name "*.db") > "$O/hbm_calib_${mib}MiB.json"
  echo "== $mib MiB"; python - <<PY
import json
d = json.load(open("$O/hbm_calib_${mib}MiB.json"))
for k, v in d.items():
    if "us" in v: print(f"{k:10s} FETCH/true {v['FETCH_SIZE_per_true_byte']:.3f}  WRITE/true {v['WRITE_SIZE_per_true_byte']:.3f}  {v['us']:.0f} us  {v['true_GBs']:.0f} GB/s")
PY
done
