mkdir -p gpurun_out/r4b
V=""
for j in 0 4 6 8 10 12 16 24; do V="$V --variant j_chunk_tiles=$j"; done
timeout 600 python3 tests/tools/ab_options.py --points 50000 --features 128 --kernel rbf --steps 100 --repeat 2 $V 2>&1 | tee gpurun_out/r4b/ab_c2_chunks.log
V=""
for j in 0 8 16 32; do V="$V --variant j_chunk_tiles=$j"; done
timeout 600 python3 tests/tools/ab_options.py --points 20000 --features 128 --kernel rbf --steps 100 --repeat 1 $V 2>&1 | tee gpurun_out/r4b/ab_20k_chunks.log
timeout 600 python3 tests/tools/ab_options.py --points 100000 --features 128 --kernel rbf --steps 30 --repeat 1 $V --variant j_chunk_tiles=48 2>&1 | tee gpurun_out/r4b/ab_100k_chunks.log
