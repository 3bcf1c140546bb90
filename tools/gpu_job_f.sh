#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
run() { echo "== $1"; shift; env "$@" timeout 150 python $ROOT/tools/bench_sk.py 30 "ffn1 fwd" 2>&1 | grep -v amdgpu.ids; }
run "base" CTTS_SK_DEBUG=16
run "prio by round" CTTS_SK_DEBUG=48
run "prio wj%4" CTTS_SK_DEBUG=80
run "prio mixed" CTTS_SK_DEBUG=112
run "base again" CTTS_SK_DEBUG=16
run "128x128 prio by round (W=64: round = wj/32 in 0..1)" CTTS_SK_DEBUG=48 CTTS_SK_TILE=22
run "128x128 base" CTTS_SK_DEBUG=16 CTTS_SK_TILE=22
