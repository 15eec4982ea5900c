#!/bin/bash
# Builds A/B variants of ONE source file of csrc/ (SRC, default refract) as whole libraries under tools/ab/ (git-ignored .so
# files; they travel to the GPU box).     SRC=refract tools/ab_src.sh "tag1:-DPSX_X=1" "tag2:-DPSX_Y=1 -DPSX_Z=0" ...
# On the box: tools/ab_run.sh OUTDIR tag1 tag2 ...
set -e
SRC=${SRC:-refract}
cd "$(dirname "$0")/../paresis_amd/csrc"
make -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I/opt/rocm/include -Wall -Wno-unused-function"
mkdir -p ../../tools/ab build/ab
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  $HIPCC $FLAGS $defs -c $SRC.hip -o build/ab/${SRC}_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%:*}
  objs=$(ls build/*.o | grep -v "build/$SRC.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libparesis_hip_$tag.so $objs build/ab/${SRC}_$tag.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
ls -la ../../tools/ab/ | grep "$(date +%b)" | tail -12
