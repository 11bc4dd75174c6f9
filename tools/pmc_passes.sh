#!/bin/bash
# Counter passes of one bench command on the GPU box (each --pmc set in its own run, kernel-trace only):
#   tools/pmc_passes.sh <tag> [bench args...]          (PMC_SCRIPT=tools/loss_bench.py: another script of the repo)
# writes gpurun_out/<tag>_{stats,sq1,sq2,fetch,write}/ and gpurun_out/<tag>_counters.txt (rocprofv3 -L excerpt)
tag="$1"; shift
args="${@:---steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|TCC_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE|VALUBusy|OccupancyPercent|MeanOccupancy[A-Za-z]*)\b" | sort -u > "$O/${tag}_counters.txt"
run() {  # name, rocprof options...
  name="$1"; shift
  rm -rf "$O/${tag}_$name"
  timeout 600 rocprofv3 "$@" -d "$O/${tag}_$name" -o r -- python "$R/${PMC_SCRIPT:-bench.py}" $args > "$O/${tag}_$name.log" 2>&1
  echo "$name rc=$?"
}
run stats --kernel-trace --stats --output-format csv
run sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
ls "$O/${tag}_sq1" "$O/${tag}_stats" | head -20
