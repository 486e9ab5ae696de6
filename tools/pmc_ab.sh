#!/bin/bash
# Hardware counters of the render kernel for several builds of the library on one workload.
# usage: [COUNTERS="SQ_.. SQ_.."] tools/pmc_ab.sh <workload> lib1.so lib2.so ...
# (per-launch averages on stdout, CSVs in gpurun_out/pmc_ab/; at most 8 SQ counters per call)
export TMPDIR=/tmp
W=$1; shift
C=${COUNTERS:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY}
OUT=gpurun_out/pmc_ab
mkdir -p $OUT
for LIB in "$@"; do
  N=$(basename $LIB .so)
  rm -rf $OUT/$N
  timeout -s KILL 200 rocprofv3 --pmc $C -d $OUT/$N -o p --output-format csv -- python3 tools/ab.py --child --workloads $W --launches 3 $LIB > $OUT/$N.log 2>&1 || { tail -5 $OUT/$N.log; exit 1; }
  python3 - "$N" $(find $OUT/$N -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys
from collections import defaultdict
name, path = sys.argv[1], sys.argv[2]
s = defaultdict(float); n = set()
for r in csv.DictReader(open(path)):
    if "render_tiles" in r["Kernel_Name"]:
        s[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
k = len(n)
print("%-16s launches %d  " % (name, k) + "  ".join("%s %.4gG" % (a.replace("SQ_", "").replace("SQC_", "C_"), b / k / 1e9) for a, b in sorted(s.items())))
PY
  grep "Mrays" $OUT/$N.log | tail -1
done
