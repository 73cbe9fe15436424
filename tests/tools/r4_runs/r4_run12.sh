export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
mkdir -p gpurun_out/r4k
V=""
for j in 64 128 256 512 1024; do V="$V --variant item_order_dev=3,j_chunk_tiles=$j"; done
timeout 900 python3 tests/tools/ab_options.py --points 1000000 --features 128 --kernel rbf --steps 3 --repeat 2 $V 2>&1 | tee gpurun_out/r4k/ab_long_chunks_c5.log
timeout 900 python3 tests/tools/ab_options.py --points 300000 --features 128 --kernel rbf --steps 8 --repeat 2 $V 2>&1 | tee gpurun_out/r4k/ab_long_chunks_300k.log
V=""
for j in 32 64 128 256; do V="$V --variant item_order_dev=3,j_chunk_tiles=$j"; done
timeout 900 python3 tests/tools/ab_options.py --points 200000 --features 256 --kernel linear --steps 6 --repeat 2 $V 2>&1 | tee gpurun_out/r4k/ab_long_chunks_c3.log
timeout 900 python3 tests/tools/ab_options.py --points 100000 --features 128 --kernel rbf --steps 20 --repeat 2 $V 2>&1 | tee gpurun_out/r4k/ab_long_chunks_100k.log
