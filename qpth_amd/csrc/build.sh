#!/bin/bash
# Builds qpth_amd/libqpx_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libqpx_hip.so
if [ -f "$OUT" ] && [ -z "$(find . ../../include -newer "$OUT" \( -name '*.h' -o -name '*.hip' -o -name '*.inc' -o -name build.sh \) -print -quit)" ]; then
  exit 0   # up to date
fi
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wno-unused-variable -o "$OUT.tmp" qpx_hip.hip
mv "$OUT.tmp" "$OUT"
