mkdir -p gpurun_out/r4h
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "acceptance_edge" > gpurun_out/r4h/f16_edge.log 2>&1; tail -3 gpurun_out/r4h/f16_edge.log; grep "statistic" gpurun_out/r4h/f16_edge.log | head -40
timeout 900 python3 tests/tools/rho_study.py 8192 128 1e-6 8 > gpurun_out/r4h/rho_study_8192.log 2>&1; tail -3 gpurun_out/r4h/rho_study_8192.log
timeout 1500 python3 tests/tools/rho_study.py 16384 128 1e-6 8 > gpurun_out/r4h/rho_study_16384.log 2>&1; tail -3 gpurun_out/r4h/rho_study_16384.log
