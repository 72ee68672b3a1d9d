#!/bin/bash
# Config 3 through the plain and the decoupled instantiation: bench lines (one frame at a time and three in flight) and the SQ counters
# of one launch each.   usage: gpurun -- bash tools/gpu_plain_vs_decoupled.sh      -> gpurun_out/pvd/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pvd; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-verify --no-configs --no-moving-camera"
S="--depth 1 --steps 4 --warmup 1"
for c in 0 1; do
  export TRT_COMPACTION=$c
  for depth in 1 3; do
    timeout -k 10 200 $B --depth $depth > $O/bench_c${c}_d$depth.json 2> $O/err.log || { tail $O/err.log; exit 1; }
  done
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $O/pmc_sq_c$c -o p -- $B $S > /dev/null 2>> $O/err.log || exit 1
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq2_c$c -o p -- $B $S > /dev/null 2>> $O/err.log || exit 1
  timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH -d $O/pmc_sq3_c$c -o p -- $B $S > /dev/null 2>> $O/err.log || echo "sq3 pass failed (counter names?)"
  python3 tools/pmc_summary.py $O/pmc_sq_c$c $O/pmc_sq2_c$c $O/pmc_sq3_c$c > $O/summary_c$c.txt 2>> $O/err.log
done
python3 - <<'PY'
import json
for c in (0,1):
    for d in (1,3):
        j=json.loads(open('gpurun_out/pvd/bench_c%d_d%d.json'%(c,d)).read().strip().splitlines()[-1])
        print('compaction',c,'depth',d,'%.3f G'%(j['value']/1e9),'ms/step %.4f'%j['ms_per_step'],'kernel d1 %.4f'%j['one_frame_at_a_time']['render_kernel_ms'],j['roofline']['kernel'])
PY
grep -h "render_rounds" $O/summary_c0.txt $O/summary_c1.txt
