#!/bin/bash
# Build libdccn.so of another git revision (for A/B timing against the working tree):
#   tools/build_rev.sh HEAD abl/libdccn_head.so [extra hipcc flags]
set -e
REV=$1; OUT=$(realpath -m $2); shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
(cd $ROOT && git archive $REV dl_ofdm_amd/csrc include) | tar -x -C $T
mkdir -p $(dirname $OUT)
(cd $T/dl_ofdm_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function "$@" $(ls dccn_abi*.hip) -o $OUT)
rm -rf $T
echo built $OUT
