#!/bin/bash
# Round 5: the replayed dispatch with a head of short column chunks (automatic) against uniform chunks, same box, interleaved.  -> gpurun_out/r05_ab_chunk_head.log
out=gpurun_out/r05_ab_chunk_head.log; : > $out
for n in 12000 20000 30000 40000 50000 70000 100000; do
  LSSVM_MI355_DEBUG=1 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 100 --warmup 10 --repeat 2 \
     --variant "" --variant "j_chunk_tiles=16" --variant "j_chunk_tiles=24" --variant "j_chunk_tiles=32" --variant "j_chunk_tiles=40" --variant "j_chunk_tiles=48" --variant "j_chunk_tiles=64" 2>&1 | grep -v "^\[plssvm_amd\] f16" >> $out
done
