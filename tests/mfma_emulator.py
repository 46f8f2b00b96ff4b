"""Numpy emulation of the fused decoder kernel's DATAFLOW (csrc/sdf_decoder.hip):
one wave = 64 lanes x registers, v_mfma_f32_32x32x2_f32 semantics, the packed
record stream and params buffer of zeroshape_amd/program.py consumed in exactly the
order the kernel consumes them.  It exists so that the packing / schedule logic
(host logic) is testable without a GPU; arithmetic is float64 for clarity.

Not the oracle and not the product: a test helper.
"""
import math

import numpy as np

from zeroshape_amd import program as P

LANE = np.arange(64)
HI = LANE >> 5
COL = LANE & 31
ROWS = np.stack([P.ROW_TABLE[HI, r] for r in range(16)])   # [r][lane] -> D row index


def mfma(a_rec, b, acc):
    """acc[16,64] += A[32x2] @ B[2x32] in the 32x32x2 f32 layout."""
    A = a_rec.reshape(2, 32)          # [k][i]
    B = b.reshape(2, 32)              # [k][j]
    D = A.T @ B                       # [i][j]
    acc += D[ROWS, COL[None, :]]
    return acc


def partner(v):
    """value held by lane l ^ 32"""
    return np.concatenate([v[32:], v[:32]])


class Stream(object):
    def __init__(self, recs_flat):
        n_groups = recs_flat.size // P.GROUP_FLOATS
        self.g = recs_flat.reshape(n_groups, 64, 4).astype(np.float64)
        self.pos = 0

    def next_group(self):
        grp = self.g[self.pos]
        self.pos += 1
        return [grp[:, j] for j in range(4)]


def gemm_tile_f32(stream, X, acc):
    """acc[16,64] += W_tile @ X  with X = list of KT activation tiles [16,64];
    consumes KT*4 groups (kt-major, then register)."""
    for kt in range(len(X)):
        for g in range(4):
            recs = stream.next_group()
            for j in range(4):
                mfma(recs[j], X[kt][4 * g + j], acc)
    return acc


def rp(params, off, tile=0):
    """row-param read: [16,64] register image of params[off + tile*32 ...]."""
    base = off + tile * 32
    return np.stack([params[base + HI * 16 + r] for r in range(16)]).astype(np.float64)


def rp4(params, off, tile):
    base = off + tile * 128
    v = np.stack([[params[base + (HI * 16 + r) * 4 + c] for r in range(16)] for c in range(4)])
    return v.astype(np.float64)      # [4][16][64]


def layer_norm(x, params, g_off, b_off, eps=1e-6):
    s = sum(t.sum(axis=0) for t in x)
    s = s + partner(s)
    mean = s / 256.0
    v = sum(((t - mean) ** 2).sum(axis=0) for t in x)
    v = v + partner(v)
    rstd = 1.0 / np.sqrt(v / 256.0 + eps)
    return [(x[kt] - mean) * rstd * rp(params, g_off, kt) + rp(params, b_off, kt) for kt in range(8)]


def gelu(x):
    erf = np.vectorize(math.erf)
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def softplus100(x):
    z = x * 100.0
    return np.where(z > 20.0, x, np.log1p(np.exp(np.minimum(z, 20.0))) / 100.0)


class SplitStream(object):
    """The split-fp16 program (zeroshape_amd.program.split_program) consumed K-block by K-block in
    the order csrc/sdf_decoder_split.hip consumes it."""
    split = True

    def __init__(self, words):
        n = P.REC_FLOATS // P.KB_WORDS
        h = np.ascontiguousarray(words[:n * P.KB_WORDS]).view(np.uint16).reshape(n, 2, 64, 8)
        self.kb = h.view(np.float16).astype(np.float64)                            # [kb][hi|lo][lane][e]
        self.pos = 0

    def next_kblock(self):
        a = self.kb[self.pos]
        self.pos += 1
        return a[0], a[1]


def _split_f16(x):
    _, hi = P.f16_round(x.astype(np.float32))
    _, lo = P.f16_round(x.astype(np.float32) - hi)
    return hi.astype(np.float64), lo.astype(np.float64)


def mfma16(a, b, acc):
    """acc[16,64] += A[32x16] @ B[16x32] in the 32x32x16 layout: lane l supplies
    A[i = l & 31][k = (l >> 5, e)] and B[k = (l >> 5, e)][j = l & 31], e = 0..7."""
    A = np.concatenate([a[:32], a[32:]], axis=1)        # [i][16]
    B = np.concatenate([b[:32], b[32:]], axis=1).T      # [16][j]
    D = A @ B
    acc += D[ROWS, COL[None, :]]
    return acc


def gemm_tile_split(stream, X, acc):
    """as gemm_tile, on the split stream: per input tile two K-blocks (registers 8j..8j+7 of the
    tile as the B operand), three MFMAs each."""
    for kt in range(len(X)):
        for j in range(2):
            ahi, alo = stream.next_kblock()
            bhi, blo = _split_f16(X[kt][8 * j:8 * j + 8].T)        # [lane][e]
            mfma16(alo, bhi, acc)
            mfma16(ahi, blo, acc)
            mfma16(ahi, bhi, acc)
    return acc


def decode_wave(recs_flat, params, xyz, split=False):
    """xyz [32,3] -> logits [32] for one wave tile, following the kernel schedule.
    split: recs_flat is the split program (uint32 words) and the schedule that of
    csrc/sdf_decoder_split.hip (software-pipelined MLP)."""
    global gemm_tile
    if split:
        st = SplitStream(recs_flat)
        gemm = gemm_tile_split
    else:
        st = Stream(recs_flat)
        gemm = gemm_tile_f32
    return _decode_wave(st, gemm, params, xyz, split)


def _decode_wave(st, gemm_tile, params, xyz, split):
    L = P.PARAMS
    px = np.concatenate([xyz[:, 0], xyz[:, 0]]).astype(np.float64)
    py = np.concatenate([xyz[:, 1], xyz[:, 1]]).astype(np.float64)
    pz = np.concatenate([xyz[:, 2], xyz[:, 2]]).astype(np.float64)

    def xyz_affine(off, tile, x_, y_, z_):
        w = rp4(params, off, tile)
        return w[3] + w[0] * x_ + w[1] * y_ + w[2] * z_

    x = [xyz_affine(L.PP, kt, px, py, pz) for kt in range(8)]
    scale = 32 ** -0.5
    for blk in range(P.BLOCKS):
        d = L.blk[blk]
        h = layer_norm(x, params, d["ln1_g"], d["ln1_b"])
        y = [x[nt] + rp(params, d["bproj"], nt) for nt in range(8)]
        for hd in range(P.HEADS):
            qkv = []
            for part in range(3):
                acc = rp(params, d["bqkv"], hd * 3 + part).copy()
                qkv.append(gemm_tile(st, h, acc))
            q, k, v = qkv
            s_self = (q * k).sum(axis=0)
            s_self = (s_self + partner(s_self)) * scale
            m_run = np.full(64, -np.inf)
            z_run = np.zeros(64)
            o = np.zeros((16, 64))
            for lt in range(P.LT):
                S = gemm_tile(st, [q], np.zeros((16, 64))) * scale
                lidx = 32 * lt + ROWS
                S = np.where(lidx < P.L, S, -np.inf)
                m_t = S.max(axis=0)
                m_t = np.maximum(m_t, partner(m_t))
                m_new = np.maximum(m_run, m_t)
                alpha = np.exp(m_run - m_new)
                Pm = np.exp(S - m_new)
                z_run = z_run * alpha + Pm.sum(axis=0)
                o = o * alpha
                o = gemm_tile(st, [Pm], o)
                m_run = m_new
            m_new = np.maximum(m_run, s_self)
            alpha = np.exp(m_run - m_new)
            p_self = np.exp(s_self - m_new)
            z = (z_run + partner(z_run)) * alpha + p_self
            o = (o * alpha + p_self * v) / z
            for nt in range(8):
                gemm_tile(st, [o], y[nt])
        h2 = layer_norm(y, params, d["ln2_g"], d["ln2_b"])
        y = [y[nt] + rp(params, d["b2"], nt) for nt in range(8)]
        if split:      # fc1(0), [fc1(t+1), fc2(t)] ..., fc2(31)
            hid = gemm_tile(st, h2, rp(params, d["b1"], 0).copy())
            for ht in range(P.HT):
                nxt = gemm_tile(st, h2, rp(params, d["b1"], ht + 1).copy()) if ht + 1 < P.HT else None
                hid = gelu(hid)
                for nt in range(8):
                    gemm_tile(st, [hid], y[nt])
                hid = nxt
        else:
            for ht in range(P.HT):
                hid = gemm_tile(st, h2, rp(params, d["b1"], ht).copy())
                hid = gelu(hid)
                for nt in range(8):
                    gemm_tile(st, [hid], y[nt])
        x = y
    feat = layer_norm(x, params, L.lnf_g, L.lnf_b)
    sq2 = math.sqrt(2.0)
    # impl_mlp layer 0 (output parked in the LDS slab on the device)
    cur = []
    for nt in range(8):
        acc = xyz_affine(L.impl[0], nt, px, py, pz)
        cur.append(softplus100(gemm_tile(st, feat, acc)))
    feat_s = [t / sq2 for t in feat]
    sx, sy, sz = px / sq2, py / sq2, pz / sq2
    # feat halves of the skip layers: fp32 kernel = workspace "Z" tiles computed up front; split
    # kernel = 16 more K-blocks per output tile inside the skip layer (feat / sqrt(2) stays packed)
    Z = {}
    if not split:
        for l in P.SKIP_IN:
            Z[l] = [gemm_tile(st, feat_s, np.zeros((16, 64))) for nt in range(8)]
    for l in range(1, P.MLP_LAYERS - 1):
        nxt = []
        if l in P.SKIP_IN:
            for nt in range(8):
                acc = xyz_affine(L.impl[l], nt, sx, sy, sz)
                gemm_tile(st, cur, acc)          # cur already holds x / sqrt(2)
                if split:
                    gemm_tile(st, feat_s, acc)
                    nxt.append(softplus100(acc))
                else:
                    nxt.append(softplus100(acc + Z[l][nt]))
        else:
            post = sq2 if (l + 1) in P.SKIP_IN else 1.0
            for nt in range(8):
                acc = rp(params, L.impl[l], nt).copy()
                nxt.append(softplus100(gemm_tile(st, cur, acc)) / post)
        cur = nxt
    out = sum((cur[kt] * rp(params, L.w8, kt)).sum(axis=0) for kt in range(8))
    out = out + partner(out) + params[L.b8]
    assert st.pos == (P.KB_TOTAL if split else P.G_TOTAL), st.pos
    return out[:32]


gemm_tile = gemm_tile_f32     # name used by tests/test_program_packing.py
