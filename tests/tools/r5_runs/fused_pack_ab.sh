#!/bin/bash
# Round 5: k_pack_dc folded into k_update_d (lib) against the build before (lib_v_prev): iteration time at small and medium sizes, fp32 and fp64, same box, interleaved.
for spec in "3000 128 rbf float32 400" "10000 128 rbf float32 400" "20000 128 rbf float32 300" "50000 128 rbf float32 300" "20000 64 polynomial float64 100" "100000 128 rbf float32 40"; do
  set -- $spec
  for round in 1 2; do for lib in lib_v_prev lib; do
    PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --dtype $4 --steps $5 --warmup 10 --repeat 1 2>&1 | grep "^rep" | sed "s/^rep 0/$lib $1 x $2 $3 $4/"
  done; done
done
