#!/bin/bash
# builds plssvm_amd/lib_v_<name>/libplssvm_amd.so = the shipped objects with ONE translation unit recompiled with extra flags (A/B of a kernel variant):
#   tests/tools/build_unit_variant.sh <name> <unit, e.g. tile_launch_f32x> [extra hipcc flags]
set -e
NAME=$1; UNIT=$2; shift 2
OUT=$PWD/plssvm_amd/lib_v_$NAME; mkdir -p $OUT
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 --offload-compress -Wno-inline-asm -I$PWD/include "$@" -c plssvm_amd/csrc/$UNIT.hip -o $OUT/$UNIT.o
OBJS=$(ls plssvm_amd/lib/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -ldl -lpthread -o $OUT/libplssvm_amd.so $OBJS $OUT/$UNIT.o
rm -f $OUT/$UNIT.o
echo "built $OUT ($UNIT $@)"
