#!/bin/bash
# Round 5: chunk choice of the 256-row kernel under the persistent (queue) launches: uniform chunks / a searched head, and the former launches (the run kept under profiles/ also varied the replay's item cost, 2.6 / 1.2 / 0.6, through a
# development hook that is gone).  -> gpurun_out/r05_queue_chunk_sweep.log
out=gpurun_out/r05_queue_chunk_sweep.log; : > $out
for n in 12000 20000 30000 40000 50000 70000 100000 150000; do
  LSSVM_MI355_DEBUG=1 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 100 --warmup 10 --repeat 2 \
     --variant "ENV:LSSVM_MI355_PAIR_QUEUE=0" --variant "j_chunk_head=0" --variant "" 2>&1 | grep -v "^\[plssvm_amd\] f16" >> $out
done
