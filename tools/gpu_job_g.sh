#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
run() { echo "== $1"; shift; env "$@" timeout 150 python $ROOT/tools/bench_sk.py 30 "ffn1 fwd dense" 2>&1 | grep -v amdgpu.ids | cut -c36-200; }
run "64x64 MFMA-only W=128 (4/SIMD)" CTTS_SK_DEBUG=29
run "64x64 MFMA-only W=64 (2/SIMD)" CTTS_SK_DEBUG=29 CTTS_SK_W=64
run "64x64 MFMA-only W=32 (1/SIMD)" CTTS_SK_DEBUG=29 CTTS_SK_W=32
run "128x128 MFMA-only W=64 (2/SIMD)" CTTS_SK_DEBUG=29 CTTS_SK_TILE=22
run "128x128 MFMA-only W=32 (1/SIMD)" CTTS_SK_DEBUG=29 CTTS_SK_TILE=22 CTTS_SK_W=32
run "128x128 no-DMA W=32" CTTS_SK_DEBUG=17 CTTS_SK_TILE=22 CTTS_SK_W=32
run "128x128 no-DMA W=64" CTTS_SK_DEBUG=17 CTTS_SK_TILE=22
run "128x128 full W=64" CTTS_SK_DEBUG=16 CTTS_SK_TILE=22
