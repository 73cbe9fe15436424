#!/bin/bash
# round 3: the three split Gram modes on the same box (f16x3 default / bf16x6 / native), then the GPU test suite
TAG=${1:-r03a}
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$TAG
mkdir -p $O
for wl in c5 c2 c3; do
  for gm in 3 1; do
    st=5; [ $wl = c2 ] && st=20
    python3 bench.py --workload $wl --steps $st --warmup 2 --gram-mode $gm --no-cpu-baseline --no-ceiling --no-native-reference > $O/bench_${wl}_gm${gm}.json 2> $O/bench_${wl}_gm${gm}.err
    python3 -c "
import json,sys
j=json.loads(open('$O/bench_${wl}_gm${gm}.json').read().strip().splitlines()[-1])
r=j['roofline']
print('$wl gm$gm', r['gram_mode'], 'ms/step %.2f' % j['ms_per_step'], 'kernel ms %.2f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'value %.0f' % j['value'])
" || tail -3 $O/bench_${wl}_gm${gm}.err
  done
done
timeout 1500 python3 -m pytest tests -m gpu -q --tb=short 2>&1 | tail -400 > $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
