#!/bin/bash
# A/B two builds of libplssvm_amd.so on the same GPU box: tests/tools/ab.sh <libA.so> <libB.so> <workload> [steps] [repeats]
A=$1; B=$2; WL=${3:-c5}; STEPS=${4:-5}; REP=${5:-2}
for i in $(seq $REP); do
  for L in "$A" "$B"; do
    PLSSVM_AMD_LIBRARY=$(realpath "$L") python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', '$WL', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  done
done
