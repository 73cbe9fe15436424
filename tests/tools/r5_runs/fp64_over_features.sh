#!/bin/bash
# fp64 tile kernels over the feature count (the table of profiles/r04_fp64_over_features.log, re-taken after the degree-10 exp2 of round 5) -> gpurun_out/r05_fp64_over_features.log
out=gpurun_out/r05_fp64_over_features.log; : > $out
for k in rbf polynomial linear; do for d in 64 128 256; do
timeout 300 python3 tests/tools/ab_options.py --points 60000 --features $d --kernel $k --dtype float64 --steps 6 --repeat 1 2>&1 | grep "rep 0" | awk -v d=$d -v k=$k '{ms=$6; printf "fp64 60000 x %4d %-10s tile kernel %8.3f ms  -> %5.1f TFLOP/s useful (n^2 d / t), %.3f of 78.6\n", d, k, ms, 60000.0*60000.0*d/ms/1e9, 60000.0*60000.0*d/ms/1e9/78.6}' >> $out
done; done
timeout 300 python3 tests/tools/ab_options.py --points 100000 --features 64 --kernel rbf --dtype float64 --steps 10 --repeat 2 2>&1 | grep "rep" >> $out
cat $out
