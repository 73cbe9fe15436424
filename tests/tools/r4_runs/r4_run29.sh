# product build with the row-group major order as the default of the wide symmetric kernels: same-box A/B against lib_A (= before the fragment-major row side
# AND the new order), chunk lengths around the default, the fp64 wide kernels, the suite
mkdir -p gpurun_out/r4z
L=gpurun_out/r4z/ab_wide_final.log
for cfg in "60000 640 rbf float32 10" "60000 640 polynomial float32 10" "100000 640 rbf float32 6" "100000 385 rbf float32 6" "40000 2000 rbf float32 6" "40000 2000 polynomial float32 6" "20000 1025 rbf float32 20" "9000 600 linear float32 40" "60000 320 rbf float64 6" "60000 512 polynomial float64 6" "60000 512 rbf float64 6"; do
set -- $cfg
for lib in lib_A lib; do
PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so timeout 600 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --dtype $4 --steps $5 --repeat 1 2>&1 | grep "rep 0" | sed "s/^/$lib $1 x $2 $3 $4: /" | tee -a $L
done
done
timeout 600 python3 tests/tools/ab_options.py --points 60000 --features 640 --kernel rbf --steps 10 --repeat 1 --variant j_chunk_tiles=2 --variant j_chunk_tiles=3 --variant j_chunk_tiles=4 --variant j_chunk_tiles=6 2>&1 | grep "rep 0" | tee -a $L
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
