#!/bin/bash
# usage: scripts/mk.sh <name> [extra hipcc flags]  ->  scripts/abl/<name>.so (CO + toy instantiations only)
set -e
cd "$(dirname "$0")/../radex_emcee_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000 \
  -mllvm -disable-machine-licm -DRX_NL_LIST=8,41 '-DRX_NL_CASES=RX_CASE(8) RX_CASE(41)' "$@" -shared -o ../../scripts/abl/$name.so rx_api.hip
ls -la ../../scripts/abl/$name.so
