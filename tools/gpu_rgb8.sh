cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rgb8 or emitter or demo_driver" 2>&1 | tail -4 &&
bash tools/demo_1080p.sh 2>&1 | tail -6
