#!/bin/bash
# Builds the A/B variants of the refraction tile kernel's accumulator layout as whole libraries under
# tools/ab/ (git-ignored .so files; they travel to the GPU box).  On the box:
#   for v in p58_m0 p58_m1 p64_m1 p64_m2; do cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so; python bench.py ...; done
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for v in "58 0" "58 1" "64 1" "64 2"; do
  set -- $v
  tag=p$1_m$2
  $HIPCC $FLAGS -DPSX_ACC_PITCH=$1 -DPSX_MISS=$2 -c refract.hip -o build/ab/refract_$tag.o &
done
wait
for v in "58 0" "58 1" "64 1" "64 2"; do
  set -- $v
  tag=p$1_m$2
  objs=$(ls build/*.o | grep -v refract.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_$tag.so $objs build/ab/refract_$tag.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls -la ../../tools/ab/
