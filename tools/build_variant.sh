#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DFLAG ...]: libsavfi_hip with ONE translation unit rebuilt with extra flags ->
# tools/variants/libsavfi_NAME.so (objects of the other units cached under /tmp/savfi_objs; use with SAVFI_HIP_LIB=...)
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/meta-interpolation_amd/csrc
OBJ=/tmp/savfi_objs; mkdir -p $OBJ $ROOT/tools/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -Wall -Wno-unused-function -I $ROOT/include -I $CS"
for f in $CS/*.hip; do
  b=$(basename $f .hip)
  if [ "$b.hip" == "$UNIT" ]; then continue; fi
  if [ ! -f $OBJ/$b.o ] || [ $f -nt $OBJ/$b.o ] || [ $CS/common.h -nt $OBJ/$b.o ] || [ $CS/sepconv_x6_shared.h -nt $OBJ/$b.o ] || [ $ROOT/include/savfi_hip.h -nt $OBJ/$b.o ]; then
    hipcc $FLAGS -c $f -o $OBJ/$b.o &
  fi
done
hipcc $FLAGS "$@" -c $CS/$UNIT -o $OBJ/variant_$NAME.o
wait
OBJS=""; for f in $CS/*.hip; do b=$(basename $f .hip); if [ "$b.hip" != "$UNIT" ]; then OBJS="$OBJS $OBJ/$b.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $OBJS $OBJ/variant_$NAME.o -o $ROOT/tools/variants/libsavfi_$NAME.so
echo $ROOT/tools/variants/libsavfi_$NAME.so
