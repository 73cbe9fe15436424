#!/bin/bash
# What the box tells an ordinary user about throttling while the c5 tile kernel runs: the raw gpu_metrics table (layout by its header), the hwmon files,
# amd-smi's reading of the same.  Output: gpurun_out/r05_throttle_probe.log
out=gpurun_out/r05_throttle_probe.log
mkdir -p gpurun_out
dev=$(ls -d /sys/class/drm/card*/device | head -1)
for d in /sys/class/drm/card*/device; do if [ -e $d/gpu_metrics ]; then dev=$d; break; fi; done
echo "device dir: $dev" > $out
ls $dev | tr '\n' ' ' >> $out; echo >> $out
ls $dev/hwmon/hwmon*/ | tr '\n' ' ' >> $out; echo >> $out
for f in power1_cap power1_cap_max power1_average power1_input energy1_input freq1_input temp1_input temp2_input temp3_input; do
  [ -e $dev/hwmon/hwmon*/$f ] && echo "$f $(cat $dev/hwmon/hwmon*/$f)" >> $out
done
echo "--- idle gpu_metrics" >> $out
xxd -l 16 $dev/gpu_metrics >> $out
python3 bench.py --workload c5 --steps 40 --warmup 2 --no-cpu-baseline --no-native-reference --no-ceiling --no-other-workloads > gpurun_out/r05_throttle_probe_bench.json 2>/dev/null &
pid=$!
sleep 25
for i in 1 2 3; do
  echo "--- under load, sample $i" >> $out
  xxd $dev/gpu_metrics >> $out
  for f in power1_average power1_input energy1_input freq1_input; do [ -e $dev/hwmon/hwmon*/$f ] && echo "$f $(cat $dev/hwmon/hwmon*/$f)" >> $out; done
  sleep 1
done
echo "--- amd-smi metric under load" >> $out
timeout 30 amd-smi metric --json >> $out 2>&1
echo "--- amd-smi static (limits)" >> $out
timeout 30 amd-smi static --limit --json >> $out 2>&1
echo "--- rocm-smi" >> $out
timeout 30 rocm-smi --showpower --showclocks --showperflevel >> $out 2>&1
wait $pid
echo "--- after" >> $out
xxd $dev/gpu_metrics | head -12 >> $out
