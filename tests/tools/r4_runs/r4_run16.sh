# persistent workgroups: per-lane heads (pair_lag 0) / one head (8) against one workgroup per item (2): development build, same box
mkdir -p gpurun_out/r4q
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
L=gpurun_out/r4q/ab_persistent2.log
for cfg in "1000000 6 rbf" "400000 10 rbf" "50000 200 rbf" "20000 300 rbf"; do
set -- $cfg
timeout 900 python3 tests/tools/ab_options.py --points $1 --features 128 --kernel $3 --steps $2 --repeat 2 --check --variant pair_lag=2 --variant pair_lag=0 --variant pair_lag=8 2>&1 | grep -v "f16 planes" | tee -a $L
done
