export PLSSVM_AMD_LIBRARY=$PWD/plssvm_amd/lib_dev/libplssvm_amd.so
mkdir -p gpurun_out/r4k
V=""
for j in 64 32 16; do for o in 0 3; do V="$V --variant item_order_dev=$o,j_chunk_tiles=$j"; done; done
timeout 900 python3 tests/tools/ab_options.py --points 1000000 --features 128 --kernel rbf --steps 3 --repeat 2 $V 2>&1 | tee gpurun_out/r4k/ab_xcd_chunks_c5.log
export TMPDIR=/tmp
for j in 64 32 16; do for o in 0 3; do
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4k/fetch_${o}_$j -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --no-native-reference --no-ceiling --option item_order_dev=$o --option j_chunk_tiles=$j > gpurun_out/r4k/fetch_${o}_$j.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r4k/fetch_${o}_$j/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tile_matvec" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("order $o chunk $j", k, "launches", len(v), "mean %.4g -> %.1f GB per matvec" % (sum(v) / len(v), sum(v) / len(v) * 1024 * 2 * 8 / 1e9))
PY
rm -rf gpurun_out/r4k/fetch_${o}_$j
done; done
