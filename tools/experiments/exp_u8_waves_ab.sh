#!/bin/bash
# k_u8_batch steered to 3 / 4 (product) / 5 / 6 waves per SIMD after round 6's instruction diet (944 -> 793 per wave), A/B in ONE box session:
#   for w in 3 5 6; do hipcc <product flags> -DMDCT_U8B_WAVES=$w ... -o build_variants/libmdct_u8w$w.so; done   (see the round-6 entry of profiles/README.md)
cd "$(dirname "$0")/../.."
for round in 1 2; do
  for v in product u8w3 u8w5 u8w6; do
    if [ $v = product ]; then unset MDCT_LIB_PATH; else export MDCT_LIB_PATH=$PWD/build_variants/libmdct_$v.so; fi
    echo "== $v (round $round)"
    python3 tools/time_u8_roundtrip.py --quick 2>&1 | grep "8K 4:2:0 frame u8"
  done
done
