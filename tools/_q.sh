cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/c3prof -o r -- python $R/tools/cfg3_bench.py 2>&1 | grep cfg3
cd $R && python tools/kavg.py "." $(find gpurun_out/c3prof -name "*.db") | grep -v "torch\|at::\|rocclr\|tables" | head -40
