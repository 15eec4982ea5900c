#!/bin/bash
# Builds A/B variants of the power-of-two line kernels (fresnel_p2.hip) as whole libraries under tools/ab/ (git-ignored .so
# files; they travel to the GPU box).     tools/ab_p2.sh "tag1:-DPSX_P2_X=1" "tag2:-DPSX_P2_Y=1 -DPSX_P2_Z=0" ...
# On the box (tools/sessions/*.sh): cp tools/ab/libparesis_hip_<tag>.so paresis_amd/libparesis_hip.so; python bench.py ...
set -e
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  $HIPCC $FLAGS $defs -c fresnel_p2.hip -o build/ab/fresnel_p2_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%:*}
  objs=$(ls build/*.o | grep -v fresnel_p2.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_$tag.so $objs build/ab/fresnel_p2_$tag.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls -la ../../tools/ab/ | grep "$(date +%b)" | tail -12
