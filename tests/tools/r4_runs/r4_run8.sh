mkdir -p gpurun_out/r4d
for v in "$@"; do
echo "=== $v"
PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_mc_$v/libplssvm_amd.so timeout 300 python3 ${MC_SCRIPT:-tests/tools/miscompile_f64_wide.py} 2>&1 | tee -a gpurun_out/r4d/miscompile_$v.log
done
