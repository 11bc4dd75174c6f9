cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
KPAT="." bash tools/_ab.sh 2>&1 | grep -v "torch\|at::\|rocclr\|tables" | head -40
timeout 300 python tools/af_bench.py 8 32 2>&1 | grep "bs="
timeout 300 python tools/fx_bench.py 2>&1 | grep use_fx
