#!/bin/bash
# Build libctts_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=libctts_hip.so
SRCS="gemm.hip lr.hip norm.hip elementwise.hip"
newest=$(ls -t $SRCS ctts_common.h ../../include/ctts.h | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then echo "libctts_hip.so up to date"; exit 0; fi
objs=""
for s in $SRCS; do
  o="${s%.hip}.o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ ctts_common.h -nt "$o" ] || [ ../../include/ctts.h -nt "$o" ]; then
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$s" -o "$o" &
  fi
  objs="$objs $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $(pwd)/$OUT"
