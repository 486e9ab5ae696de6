#!/bin/bash
# Pixels of a tile per work unit (RT_HIP_UNIT_PX; 0 = chosen by launch size): whole frame and the
# shares of an 8-rank job, per workload.   usage: tools/unit_sweep.sh [workloads...]
for w in "${@:-c4}"; do
  for s in 0 2 4 8 16; do
    echo "== $w unit_px $s"
    RT_HIP_UNIT_PX=$s timeout -k 10 200 python tools/band_balance.py $w 8 || exit 1
  done
done
