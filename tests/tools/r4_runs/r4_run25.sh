# the panels-inside-a-tile kernel with the row side fragment-major (product build): timing at the shapes of r04_ab_wide_pair_workgroups.log (128-row column there:
# 8.54 / 7.00 / 26.0 / 18.7 / 12.2 / 9.8 / 1.72 ms), 40 randomised wide cases, the wide tests
mkdir -p gpurun_out/r4z
for cfg in "60000 640 rbf 20" "60000 640 polynomial 20" "100000 640 rbf 10" "100000 385 rbf 10" "40000 2000 rbf 10" "40000 2000 polynomial 10" "20000 1025 rbf 40"; do
set -- $cfg
timeout 900 python3 tests/tools/ab_options.py --points $1 --features $2 --kernel $3 --steps $4 --repeat 2 2>&1 | grep -v "f16 planes" | tee -a gpurun_out/r4z/wide_fragment_major.log
done
timeout 900 python3 tests/tools/wide_stress.py 40 41 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_random_cross_check.py tests/test_gpu_parity.py -x -q -m gpu -k "wide or cross or linear_kernel_beyond or bf16_split" 2>&1 | tail -3
