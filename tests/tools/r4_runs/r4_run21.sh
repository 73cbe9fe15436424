# linear kernel on few points and many features: the tile kernels as the polynomial kernel of degree 1 (new default below 10 000 points) -- the probe, the tests
mkdir -p gpurun_out/r4z
timeout 900 python3 tests/tools/very_wide_probe.py 3000 16384 1500 65536 6000 1025 9000 600 2>&1 | grep "linear" | tee gpurun_out/r4z/very_wide_probe_linear_in_tile.log
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
