#!/bin/bash
# A/B libraries for one -D switch of refract.hip:  tools/ab_refract3.sh NAME V1 V2 ...  -> tools/ab/libparesis_hip_NAME<V>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for v in "$@"; do $HIPCC $FLAGS -D$name=$v -c refract.hip -o build/ab/refract_$name$v.o & done
wait
objs=$(ls build/*.o | grep -v refract.o)
for v in "$@"; do
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_$name$v.so $objs build/ab/refract_$name$v.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls -la ../../tools/ab/
