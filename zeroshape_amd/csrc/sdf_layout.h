// Decoder-program layout constants; mirror of zeroshape_amd/program.py (the Python
// packer is the source of truth, tests compare zs_sdf_program_bytes() with it).
#pragma once

namespace zs {
namespace lay {

constexpr int C = 256, NT = 8, HEADS = 8, HD = 32, L = 197, LT = 7, HID = 1024, HT = 32;
constexpr int BLOCKS = 2;
constexpr int GROUP_FLOATS = 256;  // 4 records x 64 lanes
constexpr int RING = 8;

constexpr int G_QKV_HEAD = 3 * NT * 4;                        // 96
constexpr int G_KV_HEAD = LT * 8;                             // 56
constexpr int G_PROJ_HEAD = NT * 4;                           // 32
constexpr int G_HEAD = G_QKV_HEAD + G_KV_HEAD + G_PROJ_HEAD;  // 184
constexpr int G_MLP_TILE = 64;
constexpr int G_BLOCK = HEADS * G_HEAD + HT * G_MLP_TILE;     // 3520
constexpr int G_IMPL = 256 * 5 + 512 * 3;                     // 2816
constexpr int G_TOTAL = BLOCKS * G_BLOCK + G_IMPL;            // 9856
constexpr int REC_FLOATS = (G_TOTAL + RING) * GROUP_FLOATS;
constexpr int PARAM_FLOATS = 13824;
constexpr int PROGRAM_FLOATS = REC_FLOATS + PARAM_FLOATS;

// params section (floats), row-param order [tile][hi][r]
constexpr int P_PP = 0;                     // [8][2][16][4]
constexpr int P_BLK0 = 1024;
constexpr int P_BLK_STRIDE = 3328;
constexpr int PB_LN1G = 0, PB_LN1B = 256, PB_BPROJ = 512, PB_BQKV = 768, PB_LN2G = 1536,
              PB_LN2B = 1792, PB_B2 = 2048, PB_B1 = 2304;
constexpr int P_LNFG = 7680, P_LNFB = 7936;
constexpr int P_IMPL0 = 8192;   // xyz4 table (1024)
constexpr int P_IMPL1 = 9216;   // bias (256)
constexpr int P_IMPL_PAIR = 9472;  // layers (2,3), (4,5), (6,7): [xyz4 table 1024][bias 256]
constexpr int P_IMPL_PAIR_STRIDE = 1280;
constexpr int P_W8 = 13312, P_B8 = 13568;
constexpr int P_USED = 13584;
// split program only (written by zs_sdf_split_programs, read through scalar loads):
constexpr int P_FLAG = 13600;   // uint32, non-zero: an operand is not finite or outside the fp16 range
constexpr int P_KMAX = 13616;   // [BLOCKS][HEADS] largest |k_l| over the latent rows of the image
constexpr int P_PHASE_B = 8192;  // params [0, 8192) serve the attention blocks, the rest impl_mlp

// latent-path parameter block (prologue), weights transposed to [K][N]
constexpr int LQ_WLP = 0, LQ_BLP = 65536, LQ_POS = 65792, LQ_LN1G0 = 116224, LQ_LN1B0 = 116480,
              LQ_WQKV0 = 116736, LQ_BQKV0 = 313344, LQ_WPROJ0 = 314112, LQ_BPROJ0 = 379648,
              LQ_LN2G0 = 379904, LQ_LN2B0 = 380160, LQ_W1 = 380416, LQ_B1 = 642560,
              LQ_W2 = 643584, LQ_B2 = 905728, LQ_LN1G1 = 905984, LQ_LN1B1 = 906240,
              LQ_WKV1 = 906496, LQ_BKV1 = 1037568, LQ_TOTAL = 1038080;

constexpr int LPAD = 200;  // latent rows padded to the prologue's row block
// scratch per image (floats)
constexpr int S_LAT = 0;                      // [LPAD][256]
constexpr int S_X1 = S_LAT + LPAD * C;        // [LPAD][256]
constexpr int S_X2 = S_X1 + LPAD * C;         // [LPAD][256]
constexpr int S_QKV0 = S_X2 + LPAD * C;       // [LPAD][768]
constexpr int S_ATT = S_QKV0 + LPAD * 3 * C;  // [LPAD][256]
constexpr int S_HID = S_ATT + LPAD * C;       // [LPAD][1024]
constexpr int S_KV1 = S_HID + LPAD * HID;     // [LPAD][512]
constexpr int SCRATCH_FLOATS = S_KV1 + LPAD * 2 * C;  // 665600

}  // namespace lay
}  // namespace zs
