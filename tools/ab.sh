#!/bin/bash
# Same-box A/B of the working tree against an earlier commit (box-to-box differences are smaller than most tuning effects, but only
# a same-box pair is conclusive):
#   bash tools/ab.sh <commit> [bench.py flags]      e.g.  bash tools/ab.sh HEAD~1 --block conformer
# Step 1 (here, no GPU): exports <commit> into _prev/ and builds its library.  Step 2: prints the gpurun command that alternates
# both builds twice on one box.  Remove _prev/ afterwards (it travels with every gpurun snapshot).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:?usage: tools/ab.sh <commit> [bench flags]}; shift || true
FLAGS="$* --no-cpu-baseline --no-pcie"
rm -rf "$ROOT/_prev" && mkdir "$ROOT/_prev"
git -C "$ROOT" archive "$REV" | tar -x -C "$ROOT/_prev"
bash "$ROOT/_prev/comprehensive-transformer-tts_amd/csrc/build.sh" | tail -1
bash "$ROOT/comprehensive-transformer-tts_amd/csrc/build.sh" | tail -1
cat <<MSG
now run:
  gpurun --timeout 900 -- 'for i in 1 2; do echo new; timeout 250 python bench.py $FLAGS 2>/dev/null | tail -1 | cut -c80-200; echo prev; (cd _prev && timeout 250 python bench.py $FLAGS 2>/dev/null | tail -1 | cut -c80-200); done'
MSG
