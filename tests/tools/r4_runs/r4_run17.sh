# CG iteration time of small and mid-size problems after folding the single-block finish kernels and the K*v memset into their neighbours
mkdir -p gpurun_out/r4u
L=gpurun_out/r4u/cg_iteration_small.log
for cfg in "10000 300" "20000 300" "50000 200" "100000 50"; do
set -- $cfg
timeout 600 python3 tests/tools/ab_options.py --points $1 --features 128 --kernel rbf --steps $2 --repeat 2 --variant mfma_shape=3 2>&1 | grep -v "f16 planes" | tee -a $L
done
timeout 600 python3 tests/tools/ab_options.py --points 100000 --features 64 --kernel polynomial --dtype float64 --steps 50 --repeat 2 --variant mfma_shape=3 2>&1 | tee -a $L
