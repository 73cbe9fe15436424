mkdir -p gpurun_out/r4l
PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev_new/libplssvm_amd.so timeout 600 python3 tests/tools/pair_check.py 0 > gpurun_out/r4l/pair_check.log 2>&1; tail -1 gpurun_out/r4l/pair_check.log
for n in 1000000 50000 20000; do
st=4; [ $n -lt 100000 ] && st=100
for rep in 1 2; do
for lib in lib_dev lib_dev_new; do
PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so timeout 600 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps $st --repeat 1 --variant mfma_shape=3 2>&1 | grep "rep 0" | sed "s/^/$lib $n /" | tee -a gpurun_out/r4l/ab_prologue.log
done; done; done
