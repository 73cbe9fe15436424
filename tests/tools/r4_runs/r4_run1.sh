export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
mkdir -p gpurun_out/r4a
timeout 600 python3 tests/tools/pair_check.py 0,1,3 > gpurun_out/r4a/pair_check.log 2>&1; echo "pair_check rc=$?"; tail -3 gpurun_out/r4a/pair_check.log; grep -c BAD gpurun_out/r4a/pair_check.log
timeout 900 python3 tests/tools/ab_options.py --points 1000000 --features 128 --kernel rbf --steps 4 --repeat 2 --variant mfma_shape=2 --variant mfma_shape=3,pair_lag=0 --variant mfma_shape=3,pair_lag=1 --variant mfma_shape=3,pair_lag=3 > gpurun_out/r4a/ab_c5.log 2>&1; cat gpurun_out/r4a/ab_c5.log
timeout 300 python3 tests/tools/ab_options.py --points 50000 --features 128 --kernel rbf --steps 50 --repeat 2 --variant mfma_shape=2 --variant mfma_shape=3,pair_lag=0 --variant mfma_shape=3,pair_lag=1 --variant mfma_shape=3,pair_lag=3 > gpurun_out/r4a/ab_c2.log 2>&1; cat gpurun_out/r4a/ab_c2.log
timeout 300 python3 tests/tools/ab_options.py --points 200000 --features 256 --kernel linear --steps 6 --repeat 2 --variant mfma_shape=2 --variant mfma_shape=3,pair_lag=0 --variant mfma_shape=3,pair_lag=1 --variant mfma_shape=3,pair_lag=3 > gpurun_out/r4a/ab_c3.log 2>&1; cat gpurun_out/r4a/ab_c3.log
