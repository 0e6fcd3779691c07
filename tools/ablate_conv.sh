#!/bin/bash
# What bounds ifx_conv3d_cl: rebuild the library with the run-time ablation mask enabled, time the three dominant decoder
# shapes with parts of the kernel switched off (results are garbage, only the time matters), rebuild the normal library.
# mask bits: 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue stores.     usage (GPU box): bash tools/ablate_conv.sh
cd "$(dirname "$0")/../inferix_amd/csrc" || exit 1
touch ifx_conv.hip; make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIFX_CONV_ABLATE_RT=1" > /dev/null 2>&1
cd ../..
for m in 0 8 4 2 6 1 7; do
  echo "== ablate $m"
  IFX_CONV_ABLATE=$m timeout 300 python tools/bench_vae.py --detail 2>&1 | grep -E "k3x3x3 96->96 @480x832 t12|k3x3x3 192->192 @240x416 t12|k3x3x3 384->384 @120x208 t6 " | cut -c1-100
done
cd inferix_amd/csrc; touch ifx_conv.hip; make > /dev/null 2>&1
