mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_random_cross_check.py -m gpu -q --durations=5 > gpurun_out/r4c/pytest_random.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r4c/pytest_random.log
timeout 1200 python3 tests/tools/narrow_stress.py 200 11 > gpurun_out/r4c/narrow_stress.log 2>&1; echo "narrow rc=$?"; tail -2 gpurun_out/r4c/narrow_stress.log; grep CHECK gpurun_out/r4c/narrow_stress.log | head
timeout 1200 python3 tests/tools/narrow_stress.py 120 12 pair > gpurun_out/r4c/pair_stress.log 2>&1; echo "pair rc=$?"; tail -2 gpurun_out/r4c/pair_stress.log; grep CHECK gpurun_out/r4c/pair_stress.log | head
