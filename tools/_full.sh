cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; tail -c 3000 gpurun_out/bench_full.log
