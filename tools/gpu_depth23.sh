cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for depth in 2 3; do
  for mode in "" "--animation 60"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --depth $depth $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('depth $depth $mode', 'ms/step %.3f value %.3e verified %s'%(d['ms_per_step'], d['value'], d['verified']))"
  done
done
done
