#!/bin/bash
# A/B of the intermediate's block shape [Nx/IB][Ny][IB]: whole libraries with IB = 4 / 16 under tools/ab/.
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for ib in 4 16; do
  $HIPCC $FLAGS -DPSX_IB=$ib -c fresnel_lds.hip -o build/ab/fresnel_lds_ib$ib.o &
done
wait
for ib in 4 16; do
  objs=$(ls build/*.o | grep -v fresnel_lds.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_ib$ib.so $objs build/ab/fresnel_lds_ib$ib.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls -la ../../tools/ab/
