export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
mkdir -p gpurun_out/r4k
timeout 900 python3 tests/tools/ab_options.py --points 1000000 --features 128 --kernel rbf --steps 4 --repeat 2 --check --variant item_order_dev=0 --variant item_order_dev=3 2>&1 | tee gpurun_out/r4k/ab_xcd_order_c5.log
timeout 300 python3 tests/tools/ab_options.py --points 50000 --features 128 --kernel rbf --steps 50 --repeat 2 --check --variant item_order_dev=0 --variant item_order_dev=3 2>&1 | tee gpurun_out/r4k/ab_xcd_order_c2.log
timeout 300 python3 tests/tools/ab_options.py --points 200000 --features 256 --kernel linear --steps 6 --repeat 2 --check --variant item_order_dev=0 --variant item_order_dev=3 2>&1 | tee gpurun_out/r4k/ab_xcd_order_c3.log
export TMPDIR=/tmp
for o in 0 3; do
PMC_SQ_ONLY= bash -c "timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4k/fetch_$o -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-native-reference --no-ceiling --option item_order_dev=$o > gpurun_out/r4k/fetch_$o.log 2>&1"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r4k/tcc_$o -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-native-reference --no-ceiling --option item_order_dev=$o > gpurun_out/r4k/tcc_$o.log 2>&1
python3 - <<PY
import csv, glob, collections
for name in ("fetch_$o", "tcc_$o"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r4k/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            if "tile_matvec" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("order $o", k, "launches", len(v), "mean %.4g" % (sum(v) / len(v)))
PY
rm -rf gpurun_out/r4k/fetch_$o gpurun_out/r4k/tcc_$o
done
