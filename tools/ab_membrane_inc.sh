#!/bin/bash
# A/B of the membrane splat's row loop (PSX_MEMBRANE_SPLAT 0 / 1 / 2: see k_membrane_layers):
# whole libraries under tools/ab/ (git-ignored; they travel to the GPU box).
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for v in 0 1 2; do
  $HIPCC $FLAGS -DPSX_MEMBRANE_SPLAT=$v -c membrane.hip -o build/ab/membrane_inc$v.o &
done
wait
for v in 0 1 2; do
  objs=$(ls build/*.o | grep -v membrane.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_minc$v.so $objs build/ab/membrane_inc$v.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls ../../tools/ab/ | grep minc
