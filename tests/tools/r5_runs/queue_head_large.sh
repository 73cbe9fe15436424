out=gpurun_out/r05_queue_head_large.log; : > $out
for n in 70000 100000 150000 250000; do
  st=60; [ $n -ge 150000 ] && st=20
  LSSVM_MI355_DEBUG=1 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps $st --warmup 5 --repeat 2 \
     --variant "" --variant "j_chunk_head=2064" --variant "j_chunk_head=4104" --variant "j_chunk_head=2072" --variant "j_chunk_head=1044" --variant "j_chunk_head=4112" --variant "j_chunk_head=3096" 2>&1 | grep -v "^\[plssvm_amd\] f16" >> $out
done
