export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_abl/libplssvm_amd.so
mkdir -p gpurun_out/r4a
V=""
for d in 0 4 8 16 12 20 24 28; do V="$V --variant mfma_shape=3,pair_lag=0,debug_ablate=$d"; done
for d in 0 4 16 20; do V="$V --variant mfma_shape=2,debug_ablate=$d"; done
timeout 900 python3 tests/tools/ab_options.py --points 400000 --features 128 --kernel rbf --steps 6 --repeat 2 $V > gpurun_out/r4a/ablate_pair_400k.log 2>&1; cat gpurun_out/r4a/ablate_pair_400k.log
V=""
for d in 0 4 8 16 28; do V="$V --variant mfma_shape=3,pair_lag=0,debug_ablate=$d"; done
timeout 900 python3 tests/tools/ab_options.py --points 400000 --features 128 --kernel linear --steps 6 --repeat 1 $V > gpurun_out/r4a/ablate_pair_400k_linear.log 2>&1; cat gpurun_out/r4a/ablate_pair_400k_linear.log
