cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
KPAT="coefgrad|comp_bwd|apply|stft2" bash tools/_ab.sh 2>&1 | tail -16
