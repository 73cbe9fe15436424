#!/bin/bash
# A/B of several builds on one box, printing the residuum after the timed steps too (equal digits = bit-identical iterations):
#   tests/tools/ab_residuum.sh <workload> <steps> <rounds> <lib dir> ...
WL=$1; ST=$2; RD=$3; shift 3
for round in $(seq $RD); do
for lib in "$@"; do
  PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 bench.py --workload $WL --steps $ST --warmup 2 --no-cpu-baseline --no-ceiling --no-native-reference --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; bp=r.get('board_power') or {}
print('%-12s %s round $round: ms/step %.3f kernel %.3f frac %.4f  residuum %.9g  power %s W clock %s GHz' % ('$lib', '$WL', j['ms_per_step'], r['avg_launch_ms'], r['frac'], j['config']['residuum_after_timed_steps'], bp.get('median_w'), bp.get('shader_clock_ghz_median')))"
done
done
