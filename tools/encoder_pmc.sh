#!/bin/bash
# Counter passes of the encoder step on the GPU box (MFMA utilisation, LDS): tools/encoder_pmc.sh [signals] [samples] [precision]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*|SQ_INSTS_MFMA|SQ_VALU_MFMA_BUSY_CYCLES)\b" | sort -u > $O/enc_mfma_counters.txt
cat $O/enc_mfma_counters.txt | tr '\n' ' '; echo
run() { name="$1"; shift; rm -rf $O/enc_$name; timeout 600 rocprofv3 --kernel-trace "$@" -d $O/enc_$name -o r --output-format csv -- python $R/tools/encoder_bench.py ${SIG:-16} ${SAMP:-262144} ${PREC:-bf16} 2 > $O/enc_$name.log 2>&1; echo "$name rc=$?"; }
run pmc1 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVES
run pmc2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY
python - <<PY
import csv, glob, collections
for name in ("pmc1", "pmc2"):
    fs = glob.glob("$O/enc_%s/**/r_counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void mst::", "")[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0)))[:8]:
        print(name, k, {c: "%.3g" % v for c, v in d.items()})
PY
