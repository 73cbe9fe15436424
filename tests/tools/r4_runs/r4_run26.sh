# same-box A/B of the panels-inside-a-tile kernel: row side row-major (lib_A = the committed build) against fragment-major (lib)
mkdir -p gpurun_out/r4z
L=gpurun_out/r4z/ab_wide_fragment_major.log
for cfg in "60000 640 rbf 20" "60000 640 polynomial 20" "100000 640 rbf 10" "100000 385 rbf 10" "40000 2000 rbf 10" "40000 2000 polynomial 10" "20000 1025 rbf 40" "30000 1025 polynomial 20"; do
set -- $cfg
for rep in 1 2; do
for lib in lib_A lib; do
PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/$lib/libplssvm_amd.so timeout 600 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --steps $4 --repeat 1 2>&1 | grep "rep 0" | sed "s/^/$lib $1 x $2 $3: /" | tee -a $L
done; done; done
