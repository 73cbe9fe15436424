for rep in 1 2; do for lib in lib_dev lib_dev_s1 lib_dev_s2 lib_dev_s3; do
PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so python3 tests/tools/ab_options.py --points 400000 --features 128 --kernel rbf --steps 10 --repeat 1 --variant mfma_shape=3 2>&1 | grep "rep 0" | sed "s/^/$lib /"
done; done
