#!/bin/bash
# SQ counter pass of bench.py for one library build; prints VALU instructions and activity
# usage: tools/sq_quick.sh <lib.so> [workload]
export TMPDIR=/tmp
LIB=$1; W=${2:-c4}
OUT=gpurun_out/sq_quick_$(basename $LIB .so)_$W
rm -rf $OUT; mkdir -p $OUT
export RT_HIP_LIBRARY=$LIB
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT -o sq --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-flat-probe > $OUT/log 2>&1 || exit 1
python3 tools/collect_sq.py $(find $OUT -name "*counter_collection.csv" | head -1) | grep -E "INSTS_VALU|ACTIVE_INST_VALU|WAVE_CYCLES|WAIT_INST_ANY|VMEM|LDS"
tail -1 $OUT/log | cut -c90-200
