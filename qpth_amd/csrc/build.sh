#!/bin/bash
# Builds qpth_amd/libqpx_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
make -s -j"$(nproc)" all
