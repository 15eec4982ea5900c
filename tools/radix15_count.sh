#!/bin/bash
# Packed-instruction count of a 15-point vs a 16-point register DFT (compile only, no GPU): see radix15_count.hip.
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S radix15_count.hip -o /tmp/radix15_count.s 2>/dev/null
for k in k_dft15 k_dft16; do
    awk -v k="$k" '$0 ~ "^_Z[0-9]+" k "P" {on=1} on && /v_pk_/ {n++} on && /s_endpgm/ {print k ": " n " packed instructions"; on=0; n=0}' /tmp/radix15_count.s
done
