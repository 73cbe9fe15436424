cd $GRAFT_REPO_ROOT; O=gpurun_out/r02i; mkdir -p $O
(python -m pytest tests -m gpu -q 2>&1) | tail -8
python tests/tools/ab_options.py --points 60000 --features 384 --kernel rbf --steps 10 --repeat 2 --check --variant gram_mode=0 --variant gram_mode=1 2>&1 | tee $O/ab_384.log
python tests/tools/ab_options.py --points 60000 --features 320 --kernel linear --steps 10 --repeat 2 --check --variant gram_mode=0 --variant gram_mode=1 2>&1 | tee $O/ab_320.log
