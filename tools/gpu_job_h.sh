#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
echo "== 64x64"; CTTS_SK_MIN_NKB=8 timeout 200 python $ROOT/tools/bench_sk.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-140
echo "== 128x128"; CTTS_SK_TILE=22 CTTS_SK_MIN_NKB=8 timeout 200 python $ROOT/tools/bench_sk.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-140
echo "== 64x128"; CTTS_SK_TILE=12 CTTS_SK_MIN_NKB=8 timeout 200 python $ROOT/tools/bench_sk.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-140
