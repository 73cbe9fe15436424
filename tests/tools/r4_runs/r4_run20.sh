# the panels-inside-a-tile kernels: 128-row workgroups (mfma_shape 2) against 256-row workgroups on block pairs (3, new), same box
mkdir -p gpurun_out/r4x
L=gpurun_out/r4x/ab_wide_pair.log
for cfg in "60000 640 rbf 20" "60000 640 polynomial 20" "100000 640 rbf 10" "40000 2000 rbf 10" "40000 2000 polynomial 10" "20000 1025 rbf 40" "100000 385 rbf 10"; do
set -- $cfg
timeout 900 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --steps $4 --repeat 2 --check --variant mfma_shape=2 --variant mfma_shape=3 2>&1 | grep -v "f16 planes" | tee -a $L
done
timeout 900 python3 tests/tools/wide_stress.py 40 31 2>&1 | tail -42 | tee gpurun_out/r4x/wide_stress_f32_seed31.log | tail -3
