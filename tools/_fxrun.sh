cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_r02_gpu.py -m gpu -q -k "fx or default_flags" 2>&1 | tail -8
timeout 300 python tools/fx_bench.py 2>&1 | tail -3
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/fxprof -o r -- python $GRAFT_REPO_ROOT/tools/fx_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/kavg.py "fx|prep" $(find gpurun_out/fxprof -name "*.db") 2>&1 | head -60
