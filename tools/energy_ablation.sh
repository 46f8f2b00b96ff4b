#!/bin/bash
# Energy ablation of the split decoder: the kernel runs at the socket power limit, so time tracks
# energy; each variant removes one consumer (results are garbage, the guard's fp32 re-evaluation is switched off).
#   python tools/build_variant_lib.py abl_<name> sdf_decoder_split.hip zeroshape_amd/csrc/sdf_decoder_split.hip -DZS_EXP_...
for rep in 1 2; do
for f in tools/_timing/abl_*.so; do
  ZS_NO_GUARD=1 ZS_LIB_PATH=$PWD/$f python tools/power_trace.py 2>&1 | grep loaded
done
done
