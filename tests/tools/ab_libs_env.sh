#!/bin/bash
# same-box A/B of builds and environment switches, interleaved:   tests/tools/ab_libs_env.sh <rounds> <workload> <steps> "<lib dir>[;VAR=value...]" ...
RD=$1; WL=$2; ST=$3; shift 3
for round in $(seq $RD); do
  for spec in "$@"; do
    lib=${spec%%;*}; envs=$(echo "$spec" | cut -s -d";" -f2- | tr ";" " ")
    env $envs PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 bench.py --workload $WL --steps $ST --warmup 2 --no-cpu-baseline --no-ceiling --no-native-reference --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; bp=r.get('board_power') or {}
print('%-28s %s round $round: ms/step %.4f kernel %.4f frac %.4f  residuum %.9g  power %s W clock %s GHz' % ('$spec', '$WL', j['ms_per_step'], r['avg_launch_ms'], r['frac'], j['config']['residuum_after_timed_steps'], bp.get('median_w'), bp.get('shader_clock_ghz_median')))"
  done
done
