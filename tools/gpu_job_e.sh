#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/ms $ROOT/tools/ubench/mfma_sustained.hip && for w in 1 4; do /tmp/ms $w 2 1; done
run() { echo "== $1"; shift; env "$@" timeout 150 python $ROOT/tools/bench_sk.py 30 "ffn1 fwd dense" 2>&1 | grep -v amdgpu.ids; }
run "sk clock" CTTS_SK_DEBUG=16
run "sk clock, no DMA" CTTS_SK_DEBUG=17
run "sk clock, MFMA only" CTTS_SK_DEBUG=29
run "sk clock 128x128" CTTS_SK_DEBUG=16 CTTS_SK_TILE=22
