for n in 9000 12000 20000 30000; do for round in 1 2; do for lib in lib_v_prev lib; do
 PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 200 --warmup 10 --repeat 1 2>&1 | grep "^rep" | sed "s/^rep 0/$lib $n/"
done; done; done
