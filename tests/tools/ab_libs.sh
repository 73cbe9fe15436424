#!/bin/bash
# A/B of several builds of the library on one box: tests/tools/ab_libs.sh <workload> <steps> <extra bench options> -- <lib dir> ...
WL=$1; ST=$2; shift 2
EXTRA=""
while [ "$1" != "--" ]; do EXTRA="$EXTRA $1"; shift; done
shift
for round in 1 2; do
for lib in "$@"; do
  PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 bench.py --workload $WL --steps $ST --warmup 2 $EXTRA --no-cpu-baseline --no-ceiling --no-native-reference 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-22s %s round $round: ms/step %.2f kernel %.2f frac %.3f' % ('$lib', '$WL', j['ms_per_step'], r['avg_launch_ms'], r['frac']))"
done
done
