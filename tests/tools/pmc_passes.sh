#!/bin/bash
# Collect PMC counters of the bench in separate passes (rocprofv3 --pmc only; no trace domains), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tests/tools/pmc_passes.sh <workload> <steps> <outdir> [extra bench.py arguments, e.g. --option mfma_shape=1]
WL=${1:-c2}; STEPS=${2:-5}; OUT=${3:-gpurun_out/pmc_$WL}; if [ $# -ge 3 ]; then shift 3; else shift $#; fi; EXTRA="$@"
export TMPDIR=/tmp
mkdir -p "$OUT"
# WL = predict: the predict leg on its own (tests/tools/predict_bench.py) instead of a CG workload of bench.py
if [ "$WL" = predict ]; then PROG="tests/tools/predict_bench.py 200000 50000 2"; else PROG="bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-native-reference --no-ceiling --no-other-workloads $EXTRA"; fi
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $PROG > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
[ -n "$PMC_SQ_ONLY" ] || run fetch FETCH_SIZE
[ -n "$PMC_SQ_ONLY" ] || run write WRITE_SIZE
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
[ -n "$PMC_SQ_ONLY" ] || run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum
