cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_parity_r02_gpu.py -m gpu -q -k "fx or default_flags" 2>&1 | tail -4
for v in "" fxw4; do
  if [ -n "$v" ]; then export MST_HIP_LIB=$R/diff-mst_amd/lib/$v.so; fi
  echo "== ${v:-base}"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/fxp_${v:-base} -o r -- python $R/tools/fx_bench.py 2>&1 | grep use_fx
done
cd $R && python tools/kavg.py "fx" $(find gpurun_out/fxp_* -name "*.db" | sort) 2>&1 | head -60
