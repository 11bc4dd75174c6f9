cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_parity_r02_gpu.py -m gpu -q -k "afloss or fx" 2>&1 | tail -3
cd /tmp
for v in "" aw4; do
  if [ -n "$v" ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  echo "== variant ${v:-base}"
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/afp_${v:-base}_8 -o r -- python $R/tools/af_bench.py 8 2>&1 | grep "bs="
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/afp_${v:-base}_32 -o r -- python $R/tools/af_bench.py 32 2>&1 | grep "bs="
done
unset MST_HIP_LIB
timeout 300 python $R/tools/fx_bench.py 2>&1 | grep use_fx
cd $R && python tools/kavg.py "af2_" $(find gpurun_out/afp_* -name "*.db" | sort) 2>&1 | head -30
