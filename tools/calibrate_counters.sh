#!/bin/bash
# Units of the SQ counters behind bench.py's roofline figure: tools/ub_counters issues known
# numbers of VALU instructions with known lane masks; two counter passes, CSVs in gpurun_out/calib/.
# Build first (here, no GPU needed): hipcc --offload-arch=gfx950 -O2 -o tools/ub_counters tools/ub_counters.hip
set -u
OUT=gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
./tools/ub_counters > $OUT/plain.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_BUSY_CU_CYCLES -d $OUT/a -o a --output-format csv -- ./tools/ub_counters > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU -d $OUT/b -o b --output-format csv -- ./tools/ub_counters > $OUT/b.log 2>&1 || exit 1
cp $(find $OUT/a -name "*counter_collection.csv" | head -1) $OUT/calib_a.csv
cp $(find $OUT/b -name "*counter_collection.csv" | head -1) $OUT/calib_b.csv
cat $OUT/plain.log
