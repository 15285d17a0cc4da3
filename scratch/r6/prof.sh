#!/bin/bash
# rocprofv3 kernel trace + counters for the bench command (run on the GPU box from the repo root)
export TMPDIR=/tmp
ROOT=$(pwd)
TAG=${1:-run}
shift
BENCH_ARGS="$@"   # e.g. --workload lmpc
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-batch1 --no-pmc --no-others --min-window 0 --streams 1 $BENCH_ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $OUT/pmc1 -o run -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 --no-pmc --no-others --min-window 0 --no-latency --streams 1 $BENCH_ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc2 -o run -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 --no-pmc --no-others --min-window 0 --no-latency --streams 1 $BENCH_ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o run -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 --no-pmc --no-others --min-window 0 --no-latency --streams 1 $BENCH_ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o run -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 --no-pmc --no-others --min-window 0 --no-latency --streams 1 $BENCH_ARGS > $OUT/pmc4.log 2>&1
cd $ROOT
python scratch/prof_summary.py $OUT > $OUT/summary.md
cat $OUT/summary.md
# keep the summaries only (the raw rocprofv3 databases are tens of MB and gpurun_out/ is capped at 64 MiB)
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
