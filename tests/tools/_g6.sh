#!/bin/bash
O=gpurun_out; mkdir -p $O
for wl in c2 c3 c4 c5; do
  st=10; [ $wl = c5 ] && st=5
  python3 bench.py --workload $wl --steps $st --warmup 2 > $O/r02_bench_${wl}_n1.json 2> $O/r02_bench_${wl}.err
done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err; cut -c1-400 $O/r02_bench_default.json
