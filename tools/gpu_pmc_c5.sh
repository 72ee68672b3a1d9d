cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/pmc_c5
rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-verify --animation 60 --depth 1 --steps 4 --warmup 1"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $O/sq -o p -- $B > /dev/null &&
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq2 -o p -- $B > /dev/null &&
python3 tools/pmc_summary.py $O/sq $O/sq2 > $O/summary.txt
grep "rounds_kernel<false" $O/summary.txt
