#!/bin/bash
# A/B of the lanes per far-ray list in k_refract_far (PSX_FAR_SUB): whole libraries under tools/ab/.
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for v in 64 32 8 4; do
  $HIPCC $FLAGS -DPSX_FAR_SUB=$v -c refract.hip -o build/ab/refract_sub$v.o &
done
wait
for v in 64 32 8 4; do
  objs=$(ls build/*.o | grep -v refract.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_sub$v.so $objs build/ab/refract_sub$v.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls ../../tools/ab/ | grep sub
