set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 bench.py > gpurun_out/h_bench.json
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/h_stats_d1 -o h -- python3 bench.py --no-cpu-baseline --depth 1 > gpurun_out/h_bench_d1_prof.json
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d gpurun_out/h_fetch -o h -- python3 bench.py --no-cpu-baseline --depth 1 --steps 4 --warmup 1 > /dev/null
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d gpurun_out/h_write -o h -- python3 bench.py --no-cpu-baseline --depth 1 --steps 4 --warmup 1 > /dev/null
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d gpurun_out/h_sq -o h -- python3 bench.py --no-cpu-baseline --depth 1 --steps 4 --warmup 1 > /dev/null
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/h_sq2 -o h -- python3 bench.py --no-cpu-baseline --depth 1 --steps 4 --warmup 1 > /dev/null
find gpurun_out/h_* -name "*.csv" | head -40
