# what the two event records around every tile launch cost (development build with an environment switch that skips them; not in the product)
export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
for n in 10000 20000 50000; do
for rep in 1 2; do
python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 300 --repeat 1 2>&1 | grep "rep 0" | sed "s/^/events    $n /"
LSSVM_MI355_NO_EVENTS=1 python3 tests/tools/ab_options.py --points $n --features 128 --kernel rbf --steps 300 --repeat 1 2>&1 | grep "rep 0" | sed "s/^/NO events $n /"
done; done
