#!/bin/bash
# Collects the profiles committed under profiles/ for the default bench workload, in one GPU
# session: kernel trace + stats of the default bench.py run, then one counter pass each for
# HBM reads, HBM writes and the SQ instruction counters.  Outputs land in gpurun_out/prof/.
# usage: tools/profile_round.sh <tag>      (e.g. r01)
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
SHORT="--steps 2 --warmup 1 --no-cpu-baseline --no-flat-probe"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/trace.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 bench.py $SHORT > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 bench.py $SHORT > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/sq -o sq --output-format csv -- python3 bench.py $SHORT > $OUT/sq.log 2>&1 || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_default_kernel_stats.csv
cp $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_fetch_size.csv
cp $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_write_size.csv
cp $(find $OUT/sq -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_sq.csv
tail -1 $OUT/${TAG}_bench_default.json | cut -c1-300
cat $OUT/${TAG}_bench_default_kernel_stats.csv
