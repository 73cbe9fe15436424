mkdir -p gpurun_out/r4b
for n in 3000 5000 10000 20000 30000 50000 70000 100000; do
LSSVM_MI355_DEBUG=1 timeout 600 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 100 --repeat 1 --variant mfma_shape=2 --variant mfma_shape=3 2>&1 | grep -v "f16 planes" | tee -a gpurun_out/r4b/chunk_auto.log
done
