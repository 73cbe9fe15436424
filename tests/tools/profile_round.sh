#!/bin/bash
# One GPU-box call that produces everything profiles/ holds for a round: bench lines for the four GPU workloads, the
# rocprofv3 kernel-trace summaries of the same commands, and the PMC passes (separate runs, counters only).
# usage: tests/tools/profile_round.sh <tag> [workloads of the profiled passes, default "c2 c3 c4 c5"]     (outputs under gpurun_out/<tag>_*)
#        PROFILE_ONLY=1: skip the plain bench lines and the microbenchmark (re-take of the rocprofv3 / PMC passes alone)
TAG=${1:-rXX}
PWLS=${2:-"c2 c3 c4 c5"}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
if [ -z "$PROFILE_ONLY" ]; then
for wl in c2 c3 c4 c5; do
  # (short launches need enough timed steps for the chip to settle: 10 steps of the 0.7 ms c2 iteration measure 0.79 ms per launch, 200 steps 0.705)
  st=20; [ $wl = c5 ] && st=5; [ $wl = c2 ] && st=200
  python3 bench.py --workload $wl --steps $st --warmup 2 > $O/${TAG}_bench_${wl}_n1.json 2> $O/${TAG}_bench_${wl}.err
done
# what a BARE bf16 MFMA loop sustains on THIS box (the chip lowers its clock under matrix-core load): the practical ceiling beside the bench lines
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tests/tools/microbench_bf16.hip -o /tmp/microbench_bf16 && timeout 300 /tmp/microbench_bf16 > $O/${TAG}_microbench_bf16_same_box.log 2>&1
python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-native-reference > $O/${TAG}_bench_c5_n1_after_microbench.json 2>/dev/null
# the native v_mfma_f32 path and the bf16x6 split of the fp32 workloads, same box, for reference (the library default is f16x3 where the data allows)
for wl in c2 c3 c5; do
  st=5; [ $wl = c2 ] && st=100
  python3 bench.py --workload $wl --steps $st --warmup 2 --gram-mode 0 --no-cpu-baseline > $O/${TAG}_bench_${wl}_n1_native_f32_mfma.json 2> $O/${TAG}_bench_${wl}_native.err
  python3 bench.py --workload $wl --steps $st --warmup 2 --gram-mode 1 --no-cpu-baseline --no-native-reference > $O/${TAG}_bench_${wl}_n1_bf16x6.json 2> $O/${TAG}_bench_${wl}_bf16x6.err
done
fi
for wl in $PWLS; do
  rm -rf $O/prof_$wl
  st=5; [ $wl = c2 ] && st=100
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -- python3 bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline --no-native-reference --no-ceiling --no-other-workloads > $O/prof_$wl.log 2>&1
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${TAG}_rocprofv3_kernel_stats_bench_${wl}.csv
  grep "^{" $O/prof_$wl.log | tail -1 > $O/${TAG}_rocprofv3_bench_line_${wl}.json
done
# predict_values (other_workloads.predict): the same two kinds of passes around the predict leg alone
rm -rf $O/prof_predict
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_predict -- python3 tests/tools/predict_bench.py > $O/prof_predict.log 2>&1
f=$(find $O/prof_predict -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/${TAG}_rocprofv3_kernel_stats_predict.csv
grep "^{" $O/prof_predict.log | tail -1 > $O/${TAG}_rocprofv3_bench_line_predict.json
rm -rf $O/pmc_predict
bash tests/tools/pmc_passes.sh predict 2 $O/pmc_predict
python3 tests/tools/pmc_summarize.py predict $O/pmc_predict > $O/${TAG}_pmc_predict.txt 2>&1
rm -rf $O/pmc_predict
[ -z "$PROFILE_ONLY" ] && : > $O/${TAG}_pmc_tile_matvec.txt
for wl in $PWLS; do
  rm -rf $O/pmc_$wl
  bash tests/tools/pmc_passes.sh $wl 3 $O/pmc_$wl
  python3 tests/tools/pmc_summarize.py $wl $O/pmc_$wl --json $O/${TAG}_hbm_traffic.json --key ${wl}_n1 --profile profiles/${TAG}_pmc_tile_matvec.txt >> $O/${TAG}_pmc_tile_matvec.txt 2>&1
  echo >> $O/${TAG}_pmc_tile_matvec.txt
  rm -rf $O/pmc_$wl   # raw per-dispatch CSVs are large; the summary is what is kept
done
cat $O/${TAG}_pmc_tile_matvec.txt | grep "=>"
grep "normal(0,1)" $O/${TAG}_microbench_bf16_same_box.log
for wl in c2 c3 c4 c5; do cat $O/${TAG}_bench_${wl}_n1.json; done
