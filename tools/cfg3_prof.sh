# Per-kernel table of the cfg #3 step (bs 32, 16 tracks, AudioFeatureLoss): rocprofv3 --kernel-trace of tools/cfg3_bench.py, averaged by tools/kavg.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/cfg3_prof -o r -- python $R/tools/cfg3_bench.py > $R/gpurun_out/cfg3_prof.log 2>&1
tail -2 $R/gpurun_out/cfg3_prof.log
cd $R && python tools/kavg.py "." $(find gpurun_out/cfg3_prof -name "*.db") 2>&1 | head -60
