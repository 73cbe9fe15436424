# timing ablations of the panels-inside-a-tile kernel (development build with -DLSSVM_ENABLE_ABLATION): 1 = no row-panel re-loads, 4 = no epilogue,
# 16 = no LDS-DMA after the prologue (results wrong, timing only)
mkdir -p gpurun_out/r4z
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_abl/libplssvm_amd.so
for cfg in "60000 640 rbf" "60000 640 polynomial" "40000 2000 rbf"; do
set -- $cfg
V=""
for dbg in 0 1 4 16 5 17 21; do V="$V --variant debug_ablate=$dbg"; done
timeout 900 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --steps 8 --repeat 1 $V 2>&1 | grep -v "f16 planes" | tee -a gpurun_out/r4z/ablation_wide.log
done
