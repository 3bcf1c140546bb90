#!/bin/bash
# Build libctts_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   CTTS_VARIANT=name CTTS_CXXFLAGS="-DCTTS_BK=64 -DCTTS_GEMM_WAVES=2" build.sh  -> libctts_hip_name.so (tuning A/B builds)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SUF=${CTTS_VARIANT:+_$CTTS_VARIANT}
OUT=libctts_hip${SUF}.so
OBJ=.obj${SUF}
mkdir -p $OBJ
SRCS="gemm.hip gemm_sk.hip gemm_ws.hip gemm_pl.hip gemm_plw.hip attn.hip lr.hip norm.hip elementwise.hip conformer.hip align.hip prosody.hip optim.hip loss.hip mel.hip pitch.hip comm.hip"
newest=$(ls -t $SRCS ctts_common.h gemm_common.h gemm_pl_common.h sk_plan.h ../../include/ctts.h build.sh | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then echo "$OUT up to date"; exit 0; fi
objs=""
for s in $SRCS; do
  o="$OBJ/${s%.hip}.o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ ctts_common.h -nt "$o" ] || [ gemm_common.h -nt "$o" ] || [ sk_plan.h -nt "$o" ] || [ gemm_pl_common.h -nt "$o" ] || [ ../../include/ctts.h -nt "$o" ] || [ build.sh -nt "$o" ]; then
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC $CTTS_CXXFLAGS -c "$s" -o "$o" &
  fi
  objs="$objs $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $OUT
echo "built $(pwd)/$OUT"
