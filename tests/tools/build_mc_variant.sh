#!/bin/bash
# builds plssvm_amd/lib_mc_<name>/libplssvm_amd.so = the shipped objects with tile_launch_f64x.o recompiled from a patched copy of the sources:
#   <name> <PEELED|ROLLED> <patch: none|nop_before|nop_after|wait_all|...> [extra hipcc flags]
set -e
NAME=$1; MODE=$2; PATCH=$3; shift 3
SRC=$PWD/plssvm_amd/csrc; OUT=$PWD/plssvm_amd/lib_mc_$NAME; TOP=/tmp/mc_$NAME; TMP=$TOP/plssvm_amd/csrc
rm -rf $TOP; mkdir -p $OUT $TMP; ln -s $PWD/include $TOP/include   # (the sources include "../../include/plssvm_amd.h")
cp $SRC/*.hpp $SRC/*.inc $SRC/tile_launch_f64x.hip $TMP/
sed -i 's/#pragma clang loop unroll(disable).*$/LSSVM_ROLLED/' $TMP/lssvm_tile_f64_wide.hip.hpp
case $PATCH in
  none) ;;
  nop_before) sed -i 's/asm("s_nop 1\\n\\tv_permlane32_swap_b32 %0, %2\\n\\tv_permlane32_swap_b32 %1, %3"/asm("s_nop 7\\n\\tv_permlane32_swap_b32 %0, %2\\n\\tv_permlane32_swap_b32 %1, %3"/; s/asm("s_nop 1\\n\\tv_permlane16_swap_b32 %0, %2\\n\\tv_permlane16_swap_b32 %1, %3"/asm("s_nop 7\\n\\tv_permlane16_swap_b32 %0, %2\\n\\tv_permlane16_swap_b32 %1, %3"/' $TMP/lssvm_device_common.hip.hpp ;;
  nop_after) sed -i 's/v_permlane32_swap_b32 %0, %2\\n\\tv_permlane32_swap_b32 %1, %3"/v_permlane32_swap_b32 %0, %2\\n\\tv_permlane32_swap_b32 %1, %3\\n\\ts_nop 7"/; s/v_permlane16_swap_b32 %0, %2\\n\\tv_permlane16_swap_b32 %1, %3"/v_permlane16_swap_b32 %0, %2\\n\\tv_permlane16_swap_b32 %1, %3\\n\\ts_nop 7"/' $TMP/lssvm_device_common.hip.hpp ;;
  wait_all) sed -i 's/if (kc == 0 \&\& p == 0 \&\& t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");/asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");/' $TMP/lssvm_tile_f64_wide.hip.hpp ;;
  volatile_swaps) sed -i 's/    asm("s_nop 1\\n\\tv_permlane/    asm volatile("s_nop 1\\n\\tv_permlane/' $TMP/lssvm_device_common.hip.hpp ;;
  opaque_init) python3 - "$TMP/lssvm_tile_f64_wide.hip.hpp" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "LSSVM_ROLLED\n        for (int p = 0; p < panels; ++p) {"
new = """#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+v"(acc[rb][cb]));  // the start values live in the accumulators' OWN registers
LSSVM_ROLLED
        for (int p = 0; p < panels; ++p) {"""
assert old in s
open(p, "w").write(s.replace(old, new, 1))
PY
  ;;
  *) echo "unknown patch $PATCH"; exit 1 ;;
esac
if [ "$MODE" = "ROLLED" ]; then D='-DLSSVM_ROLLED=_Pragma("clang loop unroll(disable)")'; else D='-DLSSVM_ROLLED='; fi
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I$TMP -I$PWD/include "$D" "$@" -save-temps=obj -c $TMP/tile_launch_f64x.hip -o $OUT/tile_launch_f64x.o 2>/dev/null
cp $OUT/tile_launch_f64x-hip-amdgcn-amd-amdhsa-gfx950.s $OUT/kernel.s 2>/dev/null || true
OBJS=$(ls plssvm_amd/lib/*.o | grep -v tile_launch_f64x.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -ldl -lpthread -o $OUT/libplssvm_amd.so $OBJS $OUT/tile_launch_f64x.o
rm -f $OUT/*.bc $OUT/*.hipi $OUT/*.out $OUT/*.hipfb $OUT/*.txt $OUT/tile_launch_f64x-*
echo "built $OUT ($PATCH)"
