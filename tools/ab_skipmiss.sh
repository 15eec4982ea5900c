#!/bin/bash
# A/B: waves of the refraction tile kernel whose 64 sources all miss the tile skip their deposits (halo >= 8 only).
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
$HIPCC $FLAGS -DPSX_SKIP_MISS=1 -c refract.hip -o build/ab/refract_skip1.o
objs=$(ls build/*.o | grep -v refract.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_skip1.so $objs build/ab/refract_skip1.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
cp ../libparesis_hip.so ../../tools/ab/libparesis_hip_skip0.so
ls ../../tools/ab | grep skip
