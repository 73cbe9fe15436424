# row-group major item order of the panels-inside-a-tile kernels: group size (item_order_dev 7 / 5 / 4 / 6 = 1 / 2 / 4 / 8 row blocks) x chunk length; development build
mkdir -p gpurun_out/r4z
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
for cfg in "60000 640 rbf 10" "60000 640 polynomial 10" "40000 2000 rbf 6" "40000 2000 polynomial 6" "100000 385 rbf 6" "20000 1025 rbf 20"; do
set -- $cfg
V="--variant item_order_dev=0"
for o in 7 5 4 6; do for j in 4 8 12; do V="$V --variant item_order_dev=$o,j_chunk_tiles=$j"; done; done
timeout 900 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --steps $4 --repeat 1 $V 2>&1 | grep -v "f16 planes" | tee -a gpurun_out/r4z/ab_wide_item_order_groups.log
done
