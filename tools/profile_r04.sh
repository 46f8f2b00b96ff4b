#!/bin/bash
# round-4 evidence in one gpurun call: decoder + evaluation passes (tools/profile_round.sh), the LDS-DMA GEMM kernel's trace and
# SQ / LDS counters (tools/prof_conv.sh), encoder traces per kernel and per (kernel, grid) at batch 1 / 28 for both seen-surface
# encoders (tools/prof_encoder.sh), and the per-layer-shape tables (tools/conv_shapes.py).  Summaries -> gpurun_out/prof/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_round.sh r04 > gpurun_out/prof_round_r04.log 2>&1
bash tools/prof_conv.sh r04_conv > gpurun_out/prof_conv_r04.log 2>&1
bash tools/prof_encoder.sh 1 r04_encoder_b1 > /dev/null 2>&1
bash tools/prof_encoder.sh 28 r04_encoder_b28 > /dev/null 2>&1
bash tools/prof_encoder.sh 1 r04_encoder_att_b1 att > /dev/null 2>&1
bash tools/prof_encoder.sh 28 r04_encoder_att_b28 att > /dev/null 2>&1
python3 tools/conv_shapes.py 1 > gpurun_out/prof/r04_conv_shapes_b1.txt 2>&1
python3 tools/conv_shapes.py 28 > gpurun_out/prof/r04_conv_shapes_b28.txt 2>&1
ls gpurun_out/prof | grep r04
