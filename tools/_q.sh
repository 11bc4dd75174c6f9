cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_console_gpu.py -m gpu -q -x 2>&1 | tail -4
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/q_prof -o r -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/q_bench.log 2>&1
cd $R && python - <<'PY'
import json
for line in open("gpurun_out/q_bench.log"):
    if line.startswith("{"):
        b = json.loads(line); print(b["value"], b["ms_per_step"], b["roofline"]["stages"])
PY
python tools/kavg.py "mrstft|stft2|prep|fill|copy|Fill" $(find gpurun_out/q_prof -name "*.db") | head -30
