cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_parity_r02_gpu.py -m gpu -q -k "afloss or cfg3" 2>&1 | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/afq_8 -o r -- python $R/tools/af_bench.py 8 2>&1 | grep "bs="
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/afq_32 -o r -- python $R/tools/af_bench.py 32 2>&1 | grep "bs="
cd $R && python tools/kavg.py "af2_" $(find gpurun_out/afq_* -name "*.db" | sort) 2>&1 | head -12
