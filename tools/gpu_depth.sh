cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for depth in 1 2 3 4; do
  for mode in "" "--animation 60"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --depth $depth $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('depth $depth $mode', 'ms/step %.3f value %.3e verified %s'%(d['ms_per_step'], d['value'], d['verified']))"
  done
done
python3 tools/host_path_bench.py 2>&1 | tail -6
bash tools/demo_1080p.sh 2>&1 | tail -6
