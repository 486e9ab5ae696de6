#!/bin/bash
# Collects what is committed under profiles/ in one GPU session:
#   - per workload (default: all five) four counter passes of a short bench.py run: two SQ passes
#     (issue and lane counters; instruction types), HBM reads, HBM writes; tools/collect_counters.py turns
#     them into profiles/sq_counters.json and profiles/hbm_traffic.json (copied to gpurun_out/prof/ as well),
#   - then kernel trace + stats of the default bench.py run (the line the driver records), which joins its
#     timings with the counter records just taken.
# Outputs land in gpurun_out/prof/; copy the two .json files, <tag>_bench_default* and <workload>/pmc_*.csv
# (as <tag>_<workload>_pmc_*.csv) into profiles/.
# usage: tools/profile_round.sh <tag> [workloads...]      (e.g. r03 c4 c2)
set -u
TAG=${1:-r03}; shift
WLS=${*:-c1 c2 c3 c4 c5}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
for W in $WLS; do
  D=$OUT/$W; mkdir -p $D
  # (a step is the whole frame, up to 512 samples per launch: c5 takes two launches per step, one step is enough)
  STEPS=2; [ $W = c5 ] && STEPS=1
  SHORT="--workload $W --steps $STEPS --warmup 1 --no-cpu-baseline --no-flat-probe --others="
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $D/sq_a -o sq_a --output-format csv -- python3 bench.py $SHORT > $D/bench_sq_a.json 2> $D/sq_a.err || exit 1
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $D/sq_b -o sq_b --output-format csv -- python3 bench.py $SHORT > $D/bench_sq_b.json 2> $D/sq_b.err || exit 1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $D/fetch -o fetch --output-format csv -- python3 bench.py $SHORT > $D/bench_fetch.json 2> $D/fetch.err || exit 1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $D/write -o write --output-format csv -- python3 bench.py $SHORT > $D/bench_write.json 2> $D/write.err || exit 1
  cp $(find $D/sq_a -name "*counter_collection.csv" | head -1) $D/pmc_sq_a.csv
  cp $(find $D/sq_b -name "*counter_collection.csv" | head -1) $D/pmc_sq_b.csv
  cp $(find $D/fetch -name "*counter_collection.csv" | head -1) $D/pmc_fetch_size.csv
  cp $(find $D/write -name "*counter_collection.csv" | head -1) $D/pmc_write_size.csv
  rm -rf $D/sq_a $D/sq_b $D/fetch $D/write
  python3 tools/collect_counters.py $W $D $TAG > $D/collected.json || exit 1
  echo "$W: counter passes done"
done
cp profiles/sq_counters.json profiles/hbm_traffic.json $OUT/
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/trace.err || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_default_kernel_stats.csv
tail -1 $OUT/${TAG}_bench_default.json | cut -c1-300
cat $OUT/${TAG}_bench_default_kernel_stats.csv
