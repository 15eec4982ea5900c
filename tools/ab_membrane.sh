#!/bin/bash
# A/B of the membrane plan's cell side (PSX_MEMBRANE_CELL): whole libraries under tools/ab/ (git-ignored; they travel to the GPU box).
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for v in 8 16 32; do
  $HIPCC $FLAGS -DPSX_MEMBRANE_CELL=$v -c membrane.hip -o build/ab/membrane_c$v.o &
done
wait
for v in 8 16 32; do
  objs=$(ls build/*.o | grep -v membrane.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_mc$v.so $objs build/ab/membrane_c$v.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls ../../tools/ab/ | grep mc
