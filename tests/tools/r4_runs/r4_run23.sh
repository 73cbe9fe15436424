# size of the column-slab bands at c5 (option colslab_band_mb; default 2048 = 8 bands): same box
mkdir -p gpurun_out/r4z
timeout 900 python3 tests/tools/ab_options.py --points 1000000 --features 128 --kernel rbf --steps 5 --repeat 2 --variant colslab_band_mb=2048 --variant colslab_band_mb=1024 --variant colslab_band_mb=4096 --variant colslab_band_mb=512 2>&1 | grep -v "f16 planes" | tee gpurun_out/r4z/ab_band_size_c5.log
