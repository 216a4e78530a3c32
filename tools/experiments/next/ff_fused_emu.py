"""Lane-level replay of ff_fused.hip's fragment algebra for one wave (32 pixels) with the kernel's index arithmetic for the packed W1 / b1 / W2
layouts, the hidden-block B fragments and the output blocks, against the dense computation (exact GELU).  Checks indexing only.
    python tools/experiments/next/ff_fused_emu.py"""
import math
import numpy as np

rng = np.random.default_rng(2)
C, HID, DLD, L = 320, 1280, 36, 64
NB, KS = HID // 32, C // 16


def mfma(A, B, Dacc):
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        Am[l31, hh * 8:hh * 8 + 8] = A[lane]; Bm[hh * 8:hh * 8 + 8, l31] = B[lane]
    Cm = Am @ Bm
    out = Dacc.copy()
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        for g in range(4):
            out[lane, 4 * g:4 * g + 4] += Cm[8 * g + 4 * hh:8 * g + 4 * hh + 4, l31]
    return out


gelu = np.vectorize(lambda v: 0.5 * v * (1.0 + math.erf(v * 0.7071067811865476)))
xn = rng.standard_normal((32, C))                     # the wave's (already normalised) panel
W1 = rng.standard_normal((2 * HID, C)) / np.sqrt(C); b1 = 0.2 * rng.standard_normal(2 * HID)
W2 = rng.standard_normal((C, HID)) / np.sqrt(HID); bo = 0.3 * rng.standard_normal(C)
W1p = np.zeros_like(W1); b1p = np.zeros_like(b1)
for j in range(HID):
    blk, i = j // 32, j % 32
    W1p[blk * 64 + i], W1p[blk * 64 + 32 + i] = W1[j], W1[HID + j]
    b1p[blk * 64 + i], b1p[blk * 64 + 32 + i] = b1[j], b1[HID + j]
W2p = np.zeros((NB, C, DLD))
for hb in range(NB):
    W2p[hb, :, :32] = W2[:, hb * 32:(hb + 1) * 32]
xf = []
for ks in range(KS):
    B = np.zeros((L, 8))
    for lane in range(L):
        B[lane] = xn[lane & 31, ks * 16 + (lane >> 5) * 8: ks * 16 + (lane >> 5) * 8 + 8]
    xf.append(B)
yacc = [np.zeros((L, 16)) for _ in range(10)]
for hb in range(NB):
    ug = [np.zeros((L, 16)) for _ in range(2)]
    for kind in range(2):
        for k16 in range(KS):
            A = np.zeros((L, 8))
            for lane in range(L):
                A[lane] = W1p[(2 * hb + kind) * 32 + (lane & 31), k16 * 16 + (lane >> 5) * 8: k16 * 16 + (lane >> 5) * 8 + 8]
            ug[kind] = mfma(A, xf[k16], ug[kind])
    hf = [np.zeros((L, 8)) for _ in range(2)]
    for gp in range(2):
        for i in range(8):
            for lane in range(L):
                hh = lane >> 5
                r = 4 * (2 * gp + (i >> 2)) + (i & 3)
                row = 8 * (2 * gp + (i >> 2)) + 4 * hh + (i & 3)
                hf[gp][lane, i] = (ug[0][lane, r] + b1p[(2 * hb) * 32 + row]) * gelu(ug[1][lane, r] + b1p[(2 * hb) * 32 + 32 + row])
    for ob in range(10):
        for gp in range(2):
            A = np.zeros((L, 8))
            for lane in range(L):
                l31, hh = lane & 31, lane >> 5
                o = 16 * gp + 4 * hh
                A[lane, :4] = W2p[hb, 32 * ob + l31, o:o + 4]; A[lane, 4:] = W2p[hb, 32 * ob + l31, o + 8:o + 12]
            yacc[ob] = mfma(A, hf[gp], yacc[ob])
out = np.zeros((32, C))
for ob in range(10):
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        for g in range(4):
            for e in range(4):
                out[l31, ob * 32 + 8 * g + 4 * hh + e] = yacc[ob][lane, 4 * g + e] + bo[ob * 32 + 8 * g + 4 * hh + e]
u = xn @ W1[:HID].T + b1[:HID]; gg = xn @ W1[HID:].T + b1[HID:]
ref = (u * gelu(gg)) @ W2.T + bo
err = np.abs(out - ref).max()
print("max |emulated wave - dense reference| = %.3e" % err)
assert err < 1e-9
