# Round-end evidence run on the GPU box: full -m gpu suite, smoke, the counter passes, the full bench line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/pmc_passes.sh ${1:-r3} > gpurun_out/${1:-r3}_passes.log 2>&1; tail -3 gpurun_out/${1:-r3}_passes.log
cd $R && timeout 900 python bench.py > gpurun_out/${1:-r3}_bench.log 2>&1; tail -c 300 gpurun_out/${1:-r3}_bench.log
# the spectrogram encoder: per-kernel table and MFMA / LDS counters (16 signals x 262144, bf16)
cd $R && bash tools/encoder_prof.sh 16 262144 bf16 > gpurun_out/${1:-r3}_encoder_prof.txt 2>&1
cd $R && bash tools/encoder_pmc.sh > gpurun_out/${1:-r3}_encoder_pmc.txt 2>&1; tail -3 gpurun_out/${1:-r3}_encoder_pmc.txt
# cfg #5 step: kernel trace for the idle-gap attribution / controller kernel table (tools/cfg5_gaps.py, tools/ctrl_prof.py), controller alone
cd $R && bash tools/cfg5_prof.sh > gpurun_out/${1:-r3}_cfg5_prof.txt 2>&1; tail -3 gpurun_out/${1:-r3}_cfg5_prof.txt
cd $R && timeout 300 python tools/ctrl_bench.py 2>&1 | grep controller | tee gpurun_out/${1:-r3}_ctrl_bench.txt
