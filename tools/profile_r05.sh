#!/bin/bash
# round-5 evidence in one gpurun call: decoder + evaluation passes (tools/profile_round.sh), the 256 x 256 ping-pong GEMM kernel's
# trace and TCC / TCP / SQ / LDS counters on ViT fc1 / fc2 / qkv at batch 28 with the 128 x 128 kernel beside it on fc1
# (tools/prof_pp256.sh), encoder traces per kernel and per (kernel, grid) at batch 1 / 28 (tools/prof_encoder.sh), the
# per-layer-shape table at batch 28 (tools/conv_shapes.py).  Summaries -> gpurun_out/prof/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_round.sh r05 > gpurun_out/prof_round_r05.log 2>&1
bash tools/prof_pp256.sh r05_pp256_fc1 5516 768 3072 > /dev/null 2>&1
bash tools/prof_pp256.sh r05_pp256_fc2 5516 3072 768 > /dev/null 2>&1
bash tools/prof_pp256.sh r05_pp256_qkv 5516 768 2304 > /dev/null 2>&1
ZS_CONV_PP256=0 bash tools/prof_pp256.sh r05_old128_fc1 5516 768 3072 > /dev/null 2>&1
bash tools/prof_encoder.sh 1 r05_encoder_b1 > /dev/null 2>&1
bash tools/prof_encoder.sh 28 r05_encoder_b28 > /dev/null 2>&1
python3 tools/conv_shapes.py 28 > gpurun_out/prof/r05_conv_shapes_b28.txt 2>&1
ls gpurun_out/prof | grep r05
