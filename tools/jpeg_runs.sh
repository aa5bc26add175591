set -e
cd $GRAFT_REPO_ROOT
# default (grey: one launch; colour: fused kernel + pack per plane), both forms forced either way, round 2's staged path; every file is opened by libjpeg and the three must be identical
for mode in "" "MDCT_JPEG_TWO_LAUNCH=1" "MDCT_JPEG_ONE_LAUNCH=1" "MDCT_JPEG_STAGED=1"; do
for args in "synthetic 1920 1072" "synthetic-color 1920 1072" "synthetic 3840 2160" "synthetic-color 3840 2160" "synthetic 8192 8192" "synthetic-color 8192 8192"; do
  tag=$(echo "$args" | tr ' ' '_')
  env $mode python3 tools/gpu_jpeg.py gpurun_out/t.jpg $args
  python3 -c "
from PIL import Image; im=Image.open('gpurun_out/t.jpg'); im.load(); print('   libjpeg opens it:', im.size, im.mode)"
  if [ -z "$mode" ]; then cp gpurun_out/t.jpg /tmp/ref_$tag.jpg; else cmp gpurun_out/t.jpg /tmp/ref_$tag.jpg && echo "   identical to the default file"; fi
done
done
rm -f /tmp/ref_*.jpg
