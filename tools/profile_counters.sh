#!/bin/bash
# Counter passes (HBM reads, HBM writes, SQ instruction counters) of bench.py for one
# workload; outputs in gpurun_out/prof_<workload>/.   usage: tools/profile_counters.sh <workload>
set -u
W=${1:-c5}
OUT=gpurun_out/prof_$W
mkdir -p $OUT
export TMPDIR=/tmp
SHORT="--workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-flat-probe"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 bench.py $SHORT > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 bench.py $SHORT > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/sq -o sq --output-format csv -- python3 bench.py $SHORT > $OUT/sq.log 2>&1 || exit 1
cp $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch_size.csv
cp $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_write_size.csv
cp $(find $OUT/sq -name "*counter_collection.csv" | head -1) $OUT/pmc_sq.csv
tail -1 $OUT/sq.log | cut -c1-250
