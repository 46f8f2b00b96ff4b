#!/bin/bash
# 128x128 DMA kernel (no stream-K) vs the 256x256 stream-K kernel on the large layer shapes of the batch-28 encoder
for cfg in "ZS_CONV_STREAM_K=0" "ZS_CONV_STREAM_K=1 ZS_CONV_256_MIN_KSTEPS=0"; do
  echo "== $cfg"
  for shape in "--B 28 --H 14 --Cin 768 --Cout 3072 --k 1" "--B 28 --H 14 --Cin 3072 --Cout 768 --k 1" "--B 28 --H 14 --Cin 768 --Cout 768 --k 1" "--B 28 --H 14 --Cin 768 --Cout 2304 --k 1" "--B 28 --H 56 --Cin 256 --Cout 256 --k 3" "--B 28 --H 28 --Cin 256 --Cout 256 --k 3" "--B 28 --H 14 --Cin 256 --Cout 1024 --k 1"; do
    env $cfg python tools/bench_conv.py $shape --engine ops --iters 20 2>&1 | grep TFLOP
  done
done
