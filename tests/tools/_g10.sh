cd $GRAFT_REPO_ROOT; O=gpurun_out/r02m; mkdir -p $O
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_A/libplssvm_amd.so
python tests/tools/ab_options.py --points 400000 --features 128 --kernel rbf --steps 5 --repeat 2 \
  --variant "" --variant debug_ablate=4 --variant debug_ablate=2 --variant debug_ablate=256 --variant debug_ablate=258 2>&1 | tee $O/ablate_epilogue_400k.log
