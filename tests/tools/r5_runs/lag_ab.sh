#!/bin/bash
# Round 5: the halves of a 256-row workgroup staggered (development build, option pair_lag: 0 lock step + priority for waves 4-7 (shipped), 1 / 3 steps of lag, 4 lock step
# without the priority, 5 priority 3, 6 one step + priority, 7 priority for waves 0-3) -- re-measured under the persistent launches, c2 (not at the power cap) and c5.
for spec in "50000 300" "1000000 4"; do set -- $spec
  PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_v_dev/libplssvm_amd.so python3 tests/tools/ab_options.py --points $1 --features 128 --kernel rbf --steps $2 --warmup 3 --repeat 2 \
    --variant "pair_lag=0" --variant "pair_lag=1" --variant "pair_lag=3" --variant "pair_lag=4" --variant "pair_lag=6" --variant "pair_lag=7" 2>&1 | grep "^rep\|^#"
done
