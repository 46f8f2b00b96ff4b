#!/bin/bash
# Ablation of the LDS-DMA convolution kernels: what does the kernel cost without its DMAs / its MFMAs?
# Build the variants first (results are garbage, the timing is the point):
#   for v in 'base' 'nodma -DZS_EXP_CONV_NO_DMA' 'noa -DZS_EXP_CONV_NO_DMA_A' 'nob -DZS_EXP_CONV_NO_DMA_B' 'nomfma -DZS_EXP_CONV_NO_MFMA' 'nosplit -DZS_EXP_CONV_NO_SPLIT'; do set -- $v; n=$1; shift;
#     python tools/build_variant_lib.py cv_$n nn_conv.hip zeroshape_amd/csrc/nn_conv.hip "$@"; done
for f in tools/_timing/cv_*.so; do
  echo "== $f"
  for shape in "--B 28 --H 14 --Cin 768 --Cout 3072 --k 1" "--B 28 --H 14 --Cin 3072 --Cout 768 --k 1" "--B 28 --H 14 --Cin 768 --Cout 768 --k 1" "--B 28 --H 56 --Cin 256 --Cout 256 --k 3"; do
    ZS_LIB_PATH=$PWD/$f python tools/bench_conv.py $shape --engine ops --iters 20 2>&1 | grep TFLOP
  done
done
