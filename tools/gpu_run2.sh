set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1 || (tail -40 gpurun_out/r2b/pytest.log; false)
tail -3 gpurun_out/r2b/pytest.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err || (tail -20 gpurun_out/r2b/bench.err; false)
timeout -k 10 300 python3 bench.py --no-cpu-baseline --animation 60 > gpurun_out/r2b/bench_anim.json 2> gpurun_out/r2b/bench_anim.err || (tail -20 gpurun_out/r2b/bench_anim.err; false)
python3 - <<'PY'
import json
for f in ("bench","bench_anim"):
    d=json.load(open(f"gpurun_out/r2b/{f}.json"))
    print(f, "value %.3e ms/step %.3f verified %s d1 %s diag %s"%(d["value"], d["ms_per_step"], d["verified"], d["one_frame_at_a_time"], d["diagnostics"]))
PY
