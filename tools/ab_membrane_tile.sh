#!/bin/bash
# A/B of the membrane layers kernel's tile (PSX_ML_TX x PSX_ML_TY): whole libraries under tools/ab/ (git-ignored; they travel to the GPU box).
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
VARIANTS=${VARIANTS:-"32x32x256 32x64x256 64x64x512 32x64x512 64x32x512 32x128x512"}      # rows x columns x threads
for v in $VARIANTS; do
  $HIPCC $FLAGS -DPSX_MEMBRANE_SPLAT=${SPLAT:-2} $(echo $v | awk -Fx '{printf "-DPSX_ML_TX=%s -DPSX_ML_TY=%s -DPSX_ML_THREADS=%s", $1, $2, $3}') -c membrane.hip -o build/ab/membrane_t$v.o &
done
wait
for v in $VARIANTS; do
  objs=$(ls build/*.o | grep -v membrane.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_mt$v.so $objs build/ab/membrane_t$v.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls ../../tools/ab/ | grep "_mt"
