#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
run() { echo "== $1"; shift; env "$@" timeout 100 python $ROOT/tools/bench_sk.py 30 "ffn1 fwd dense" 2>&1 | grep -v amdgpu.ids; }
run "baseline" X=1
run "no DMA (1)" CTTS_SK_DEBUG=1
run "no DMA no reads (5)" CTTS_SK_DEBUG=5
run "no DMA no reads no barrier (13)" CTTS_SK_DEBUG=13
run "no barrier only (8) [racy]" CTTS_SK_DEBUG=8
run "no reads only (4)" CTTS_SK_DEBUG=4
run "no DMA no barrier (9)" CTTS_SK_DEBUG=9
