#!/bin/bash
# The reference's own harness (main.cpp) twice on the same 8192x8192 raw plane: once with the
# reference's CPU code, once relinked against the MI355X engine (host pointers, so PCIe-inclusive).
# Needs oracle/_ref/simd_dct_original, simd_dct_relinked and simd_dct_relinked_warm (make -C oracle original relink relink_warm);
# the last one is the same link plus tools/shim_warmup_ctor.cpp (mdct_shim_warmup() before main()).
set -e
cd "$(dirname "$0")/.."
RAW=${TMPDIR:-/tmp}/plane8192.raw
python3 - "$RAW" <<'PY'
import sys
sys.path.insert(0, ".")
from simd_dct_amd import synth
synth.plane_u8_np(8192, 8192, "photo").tofile(sys.argv[1])
PY
for bin in simd_dct_original simd_dct_relinked simd_dct_relinked_warm; do
  echo "== $bin"
  ./oracle/_ref/$bin "$RAW" 8192 8192 --quality 2000 --runs 16 --mode enc-quant32 --mode enc-quant-stereo --mode enc-quant 2>&1 | tr "\r" "\n" | grep -v "^Features\|^$"
done
rm -f "$RAW"
