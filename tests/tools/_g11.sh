cd $GRAFT_REPO_ROOT; O=gpurun_out/r02n; mkdir -p $O
python tests/tools/ab_options.py --points 1000000 --features 128 --kernel rbf --steps 3 --repeat 2 --variant "" --variant mfma_shape=0 2>&1 | tee $O/ab_c5_civ.log
python tests/tools/ab_options.py --points 50000 --features 128 --kernel rbf --steps 30 --repeat 2 --check --variant mfma_shape=0 --variant mfma_shape=2 2>&1 | tee $O/ab_c2_civ.log
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1) | tail -3
