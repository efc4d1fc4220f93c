#!/bin/bash
# Experiment builds: libmrirt_<name>.so under build_exp/ with extra -D flags applied to ONE source file
# (objects of the other sources are cached).  Select at run time with MRIRT_LIB=build_exp/libmrirt_<name>.so.
#   bash tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]
set -e
NAME=$1; SRC=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/mri-raytracer_amd/csrc
OUT=$REPO/build_exp
mkdir -p $OUT/obj
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -I$REPO/include"
OBJS=""
for f in brats_march brats_slab brats_ring volume_march grid_ops inr_mlp; do
  if [ "$f.hip" == "$SRC" ]; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/$f.hip -o $OUT/obj/${f}_$NAME.o
    OBJS="$OBJS $OUT/obj/${f}_$NAME.o"
  else
    if [ ! -f $OUT/obj/$f.o ] || [ $CS/$f.hip -nt $OUT/obj/$f.o ] || [ $CS/mrirt_device.h -nt $OUT/obj/$f.o ] || [ $CS/mrirt_host.h -nt $OUT/obj/$f.o ] || [ $CS/brats_device.h -nt $OUT/obj/$f.o ]; then
      /opt/rocm/bin/hipcc $FLAGS -c $CS/$f.hip -o $OUT/obj/$f.o
    fi
    OBJS="$OBJS $OUT/obj/$f.o"
  fi
done
if [ ! -f $OUT/obj/abort_trace.o ] || [ $CS/abort_trace.cpp -nt $OUT/obj/abort_trace.o ]; then /opt/rocm/bin/hipcc $FLAGS -c $CS/abort_trace.cpp -o $OUT/obj/abort_trace.o; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/obj/abort_trace.o -o $OUT/libmrirt_$NAME.so
echo built $OUT/libmrirt_$NAME.so
