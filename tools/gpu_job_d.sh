#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
L=$ROOT/comprehensive-transformer-tts_amd/csrc
run() { echo "== $1"; shift; env "$@" timeout 150 python $ROOT/tools/bench_sk.py 20 "$SEL" 2>&1 | grep -v amdgpu.ids; }
SEL="ffn1"
run "64x64" CTTS_SK_TILE=11
run "64x128" CTTS_SK_TILE=12
run "128x128" CTTS_SK_TILE=22
run "64x128 reads-first" CTTS_SK_TILE=12 CTTS_LIB=$L/libctts_hip_rf.so
run "128x128 reads-first" CTTS_SK_TILE=22 CTTS_LIB=$L/libctts_hip_rf.so
run "128x128 W=32" CTTS_SK_TILE=22 CTTS_SK_W=32
run "64x128 W=128" CTTS_SK_TILE=12 CTTS_SK_W=128
SEL="sq"
run "sq 64x128" CTTS_SK_TILE=12
run "sq 128x128" CTTS_SK_TILE=22
SEL="NT"
run "linears 64x128" CTTS_SK_TILE=12 CTTS_SK_MIN_NKB=8
run "linears 128x128" CTTS_SK_TILE=22 CTTS_SK_MIN_NKB=8
