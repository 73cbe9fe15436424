#!/bin/bash
# Round 5: the persistent launches for EVERY kernel of the symmetric variant (for_each_work_item) against the build before (256-row kernel only): same box, interleaved.
# lib_v_prev = the library of commit e17a0f1.  -> gpurun_out/r05_ab_generic_queue.log
out=gpurun_out/r05_ab_generic_queue.log; : > $out
run() {  # points features kernel dtype steps [options]
  for round in 1 2; do for lib in lib_v_prev lib; do
    PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --dtype $4 --steps $5 --warmup 3 --repeat 1 $6 --variant "$7" 2>&1 | grep "^rep" | sed "s/^rep 0/$lib $1 x $2 $3 $4/" >> $out
  done; done
}
run 100000 64 polynomial float64 20 "" ""
run 60000 64 rbf float64 20 "" ""
run 100000 128 linear float64 10 "" ""
run 50000 256 rbf float32 60 "" ""
run 50000 384 rbf float32 40 "" ""
run 50000 128 rbf float32 100 "--gamma 4" ""
run 100000 128 rbf float32 30 "--gamma 4" ""
run 50000 128 rbf float32 100 "" "gram_mode=1"
run 50000 128 rbf float32 40 "" "gram_mode=0"
run 100000 128 polynomial float32 10 "" "gram_mode=0"
run 40000 1024 rbf float32 10 "" ""
run 7000 128 rbf float32 200 "" ""
cat $out
