# The round's evidence in one go (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats at one and two
# frames in flight, PMC passes (HBM traffic in separate FETCH_SIZE / WRITE_SIZE passes as MI355X_MICROARCH.md prescribes).
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/$tag
mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --animation 60 --no-cpu-baseline > $O/bench_anim.json 2> $O/bench_anim.err
B="python3 bench.py --no-cpu-baseline --no-verify --no-configs --no-moving-camera"
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_d1 -o p -- $B --depth 1 > $O/bench_d1_prof.json
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_d2 -o p -- $B > $O/bench_d2_prof.json
S="--depth 1 --steps 4 --warmup 1"
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $B $S > /dev/null
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p -- $B $S > /dev/null
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $O/pmc_sq -o p -- $B $S > /dev/null
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p -- $B $S > /dev/null
# executed FP64 work: wave-level instruction counts by class (the roofline of SURVEY 8d's "bounding roofline")
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32"
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc $F64 -d $O/pmc_f64 -o p -- $B $S > /dev/null
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 $O/pmc_f64 --json $O/pmc_summary.json > $O/pmc_summary.txt
# config 5 (256 spheres, 12 bounces, orbit): the same counters
A="$B --animation 60 $S"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/c5_stats_d1 -o p -- $B --animation 60 --depth 1 > $O/c5_bench_d1_prof.json
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d $O/c5_pmc_sq -o p -- $A > /dev/null
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/c5_pmc_sq2 -o p -- $A > /dev/null
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc $F64 -d $O/c5_pmc_f64 -o p -- $A > /dev/null
# config 5's memory side: its tables (hundreds of MB of list cells) leave every cache -- HBM bytes in separate passes, as for config 3
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/c5_pmc_fetch -o p -- $A > /dev/null
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/c5_pmc_write -o p -- $A > /dev/null
python3 tools/pmc_summary.py $O/c5_pmc_fetch $O/c5_pmc_write $O/c5_pmc_sq $O/c5_pmc_sq2 $O/c5_pmc_f64 --json $O/c5_pmc_summary.json > $O/c5_pmc_summary.txt
find $O -name "*kernel_stats.csv" | head
cat $O/pmc_summary.txt | grep -v "<true>"
cat $O/c5_pmc_summary.txt | grep -v "<true>"
