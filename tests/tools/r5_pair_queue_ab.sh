#!/bin/bash
# same-box A/B of the persistent (queue) launches of the 256-row kernel against one workgroup per item: LSSVM_MI355_PAIR_QUEUE=0/1, interleaved
#   tests/tools/r5_pair_queue_ab.sh <rounds> <workload> <steps> [<workload> <steps> ...]
RD=$1; shift
while [ $# -ge 2 ]; do
  WL=$1; ST=$2; shift 2
  for round in $(seq $RD); do
    for q in 0 1; do
      LSSVM_MI355_PAIR_QUEUE=$q python3 bench.py --workload $WL --steps $ST --warmup 2 --no-cpu-baseline --no-ceiling --no-native-reference --no-other-workloads 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; bp=r.get('board_power') or {}
print('queue $q %s round $round: ms/step %.4f kernel %.4f frac %.4f  residuum %.9g  power %s W clock %s GHz' % ('$WL', j['ms_per_step'], r['avg_launch_ms'], r['frac'], j['config']['residuum_after_timed_steps'], bp.get('median_w'), bp.get('shader_clock_ghz_median')))"
    done
  done
done
