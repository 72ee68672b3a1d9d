set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1 || (tail -40 gpurun_out/final/pytest.log; false)
tail -3 gpurun_out/final/pytest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 400 python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/final/bench.json'))
print('bench value %.4e ms/step %.3f verified %s cpu %s'%(d['value'], d['ms_per_step'], d['verified'], d['cpu_baseline']['value']))"
