cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 1500 python -m pytest tests/test_loss_gpu.py -m gpu -q -x 2>&1 | tail -3
KPAT="stft2" bash tools/_ab.sh 2>&1 | tail -8
