# fp64 tile kernels over the feature count (one-pass kernels up to 256 features, panels inside a sub-tile beyond): fraction of the fp64 matrix-core peak
mkdir -p gpurun_out/r4z
for k in rbf polynomial linear; do
for d in 64 128 192 256 320 512; do
timeout 300 python3 tests/tools/ab_options.py --points 60000 --features $d --kernel $k --dtype float64 --steps 6 --repeat 1 2>&1 | grep "rep 0" | awk -v d=$d -v k=$k '{ms=$6; printf "fp64 60000 x %4d %-10s tile kernel %8.3f ms  -> %5.1f TFLOP/s useful (n^2 d / t), %.3f of 78.6\n", d, k, ms, 60000.0*60000.0*d/ms/1e9, 60000.0*60000.0*d/ms/1e9/78.6}'
done; done | tee gpurun_out/r4z/fp64_over_features.log
