# chunk length of the 256-row workgroups chosen over EVERY length 2..64 (j_chunk_tiles = 0) against the best of the old candidate list (forced)
mkdir -p gpurun_out/r4w
L=gpurun_out/r4w/chunk_fine.log
for cfg in "12000 10 300" "15000 16 300" "20000 32 300" "25000 40 300" "30000 64 200" "40000 48 200" "50000 40 200" "60000 32 100" "70000 48 100" "100000 64 50"; do
set -- $cfg
LSSVM_MI355_DEBUG=1 timeout 600 python3 tests/tools/ab_options.py --points $1 --features 128 --kernel rbf --steps $3 --repeat 2 --variant j_chunk_tiles=$2 --variant j_chunk_tiles=0 2>&1 | grep -v "f16 planes" | grep "rep\|tiles per work item\|^#" | tee -a $L
done
