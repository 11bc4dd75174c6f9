cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_loss_gpu.py -m gpu -q -k "afloss" 2>&1 | tail -5
cd /tmp
for v in "" afw2 afw22; do
  if [ -n "$v" ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  echo "== variant ${v:-base}"
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/afp_${v:-base}_8 -o r -- python $R/tools/af_bench.py 8 2>&1 | grep "bs="
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/afp_${v:-base}_32 -o r -- python $R/tools/af_bench.py 32 2>&1 | grep "bs="
done
cd $R && python tools/kavg.py "af2_|gather" $(find gpurun_out/afp_* -name "*.db" | sort) 2>&1 | head -80
