set -e
cd $GRAFT_REPO_ROOT
for mode in 0 1; do
for args in "synthetic 1920 1072" "synthetic-color 1920 1072" "synthetic 3840 2160" "synthetic-color 3840 2160" "synthetic 8192 8192" "synthetic-color 8192 8192"; do
  MDCT_JPEG_STAGED=$mode python3 tools/gpu_jpeg.py gpurun_out/t.jpg $args
  python3 -c "
from PIL import Image; im=Image.open('gpurun_out/t.jpg'); im.load(); print('   libjpeg opens it:', im.size, im.mode)"
done
done
