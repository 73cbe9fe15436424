#!/bin/bash
# builds plssvm_amd/lib_v_<name>/libplssvm_amd.so (git-ignored, travels to the GPU box) = the shipped objects with tile_launch_f32d.o (the 256-row pair kernels) recompiled from a copy of the sources whose
# MFMA groups were regenerated with other options:   tests/tools/build_pair_variant.sh <name> "<gen_s6w_groups.py options>" [extra hipcc flags]
set -e
NAME=$1; GENOPTS=$2; shift 2
SRC=$PWD/plssvm_amd/csrc; OUT=$PWD/plssvm_amd/lib_v_$NAME; TOP=/tmp/pv_$NAME; TMP=$TOP/plssvm_amd/csrc
rm -rf $TOP; mkdir -p $OUT $TMP; ln -s $PWD/include $TOP/include
cp $SRC/*.hpp $SRC/*.inc $SRC/tile_launch_f32d.hip $SRC/gen_s6w_groups.py $TMP/
(cd $TMP && python3 gen_s6w_groups.py $GENOPTS lssvm_s6w_groups.inc)
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 --offload-compress -Wno-inline-asm -I$TMP -I$PWD/include "$@" -save-temps=obj -c $TMP/tile_launch_f32d.hip -o $OUT/tile_launch_f32d.o 2>/dev/null
mkdir -p $OUT/asm; mv $OUT/tile_launch_f32d-hip-amdgcn-amd-amdhsa-gfx950.s $OUT/asm/tile_launch_f32d.s
python3 tests/tools/audit_hand_asm.py $OUT/asm/tile_launch_f32d.s | tail -1
OBJS=$(ls plssvm_amd/lib/*.o | grep -v tile_launch_f32d.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -ldl -lpthread -o $OUT/libplssvm_amd.so $OBJS $OUT/tile_launch_f32d.o
rm -f $OUT/*.bc $OUT/*.hipi $OUT/*.out $OUT/*.hipfb $OUT/*.txt $OUT/tile_launch_f32d-*
echo "built $OUT ($GENOPTS $@)"
