mkdir -p gpurun_out/r4a
for lib in lib_dev lib_dev2; do
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so
echo "== $lib"
timeout 900 python3 tests/tools/ab_options.py --points 400000 --features 128 --kernel rbf --steps 6 --repeat 2 --variant mfma_shape=2 --variant mfma_shape=3,pair_lag=0 --variant mfma_shape=3,pair_lag=1 --variant mfma_shape=3,pair_lag=3 2>&1 | tee -a gpurun_out/r4a/setprio_400k.log
done
