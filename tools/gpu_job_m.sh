#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
run() { echo "== $1"; shift; env "$@" timeout 150 python $ROOT/tools/bench_sk.py 50 "ffn1 fwd dense" 2>&1 | grep -v amdgpu.ids | cut -c36-120; }
run "128x128 base" X=1
run "delay odd wj, 1x8k cycles" CTTS_SK_DEBUG=$((32+256))
run "delay upper half, 1x8k" CTTS_SK_DEBUG=$((64+256))
run "delay odd wj, 2x" CTTS_SK_DEBUG=$((32+512))
run "delay upper half, 2x" CTTS_SK_DEBUG=$((64+512))
run "64x64 base" CTTS_SK_TILE=11
run "64x64 delay odd 1x" CTTS_SK_TILE=11 CTTS_SK_DEBUG=$((32+256))
run "64x64 delay upper 1x" CTTS_SK_TILE=11 CTTS_SK_DEBUG=$((64+256))
run "128x128 S3 W32" CTTS_SK_STAGES=3 CTTS_SK_W=32
