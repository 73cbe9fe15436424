mkdir -p gpurun_out/r4b
for n in 5000 10000 20000 30000 50000 70000; do
V="--variant mfma_shape=2"
for j in 0 8 12 16 20 24 32 48 64; do V="$V --variant j_chunk_tiles=$j"; done
timeout 600 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 100 --repeat 1 $V 2>&1 | tee -a gpurun_out/r4b/chunk_sweep.log
done
