#!/bin/bash
for f in tools/_timing/cv_*.so; do
  echo "== $f"
  for shape in "--B 28 --H 14 --Cin 768 --Cout 3072 --k 1" "--B 28 --H 14 --Cin 3072 --Cout 768 --k 1" "--B 28 --H 14 --Cin 768 --Cout 768 --k 1" "--B 28 --H 56 --Cin 256 --Cout 256 --k 3"; do
    ZS_LIB_PATH=$PWD/$f python tools/bench_conv.py $shape --engine ops --iters 20 2>&1 | grep TFLOP
  done
done
