#!/bin/bash
# A/B: time the same workloads with two builds of the library in one GPU session
# usage: tools/ab_bench.sh libA.so libB.so [workloads...]
A=$1; B=$2; shift 2
for w in "${@:-c4}"; do
  for rep in 1 2; do
    for lib in "$A" "$B"; do
      RT_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --workload $w --steps 4 --no-cpu-baseline --no-flat-probe 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w', '$lib', d['value'], d['ms_per_step'])"
    done
  done
done
