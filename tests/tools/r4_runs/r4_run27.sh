# item order of the panels-inside-a-tile kernels: column-chunk major in XCD lanes (3, shipped) against row-group major (4: four row blocks x their
# column chunks per XCD at a time), several chunk lengths; development build, same box
mkdir -p gpurun_out/r4z
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
for cfg in "60000 640 rbf 10" "60000 640 polynomial 10" "40000 2000 rbf 6" "100000 385 rbf 6"; do
set -- $cfg
timeout 900 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --steps $4 --repeat 1 --check --variant item_order_dev=0 --variant item_order_dev=4 --variant item_order_dev=4,j_chunk_tiles=32 --variant item_order_dev=4,j_chunk_tiles=16 --variant item_order_dev=4,j_chunk_tiles=8 --variant item_order_dev=0,j_chunk_tiles=16 2>&1 | grep -v "f16 planes" | tee -a gpurun_out/r4z/ab_wide_item_order.log
done
