# Wave-level emulation of v_mfma_f32_32x32x16_f16 operand layouts to check the "accumulator -> B operand of the next product" chain
# used by the fused to_q + cross-attention kernel: Q^T = Wq X^T ; S^T = K Q ; P = softmax ; O^T = V^T P.
import numpy as np
rng = np.random.default_rng(0)
L = 64  # lanes

def mfma(A, B, D):
    """A: [64 lanes][8] (lane (row=l31, hh): A[row][k=hh*8+i]); B: [64][8] (lane (col=l31, hh): B[k=hh*8+i][col]);
    D: [64][16] (lane (col=l31, hh): D[row=8g+4hh+e][col] at reg 4g+e).  Returns D + A*B."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        for i in range(8):
            Am[l31, hh * 8 + i] = A[lane, i]
            Bm[hh * 8 + i, l31] = B[lane, i]
    C = Am @ Bm
    out = D.copy()
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        for g in range(4):
            for e in range(4):
                out[lane, 4 * g + e] += C[8 * g + 4 * hh + e, l31]
    return out

def a_frag_natural(M, row0, k0):   # rows row0..row0+31 of M, k0..k0+15 natural order
    A = np.zeros((L, 8))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        A[lane] = M[row0 + l31, k0 + hh * 8: k0 + hh * 8 + 8]
    return A

def a_frag_chain(M, row0, c0, gp):  # two 4-element pieces at columns c0 + 16gp + 4hh (+8)
    A = np.zeros((L, 8))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        o = c0 + 16 * gp + 4 * hh
        A[lane, :4] = M[row0 + l31, o:o + 4]
        A[lane, 4:] = M[row0 + l31, o + 8:o + 12]
    return A

def b_from_acc(D, gp):              # lane's 8 values for k-step gp of a 32-row accumulator block
    B = np.zeros((L, 8))
    for i in range(8):
        B[:, i] = D[:, 4 * (2 * gp + (i >> 2)) + (i & 3)]
    return B

C, d, NK, NKP = 320, 64, 77, 96
X = rng.standard_normal((32, C))           # one wave's 32 pixels (already layer-normalised)
Wq = rng.standard_normal((C, C)) / np.sqrt(C)
K = np.zeros((NKP, d)); K[:NK] = rng.standard_normal((NK, d))
V = np.zeros((NKP, d)); V[:NK] = rng.standard_normal((NK, d))
Vt = V.T.copy()                             # [d][keys]
h = 2                                       # head under test
# X panel as B fragments (natural k order)
xf = []
for ks in range(C // 16):
    B = np.zeros((L, 8))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        B[lane] = X[l31, ks * 16 + hh * 8: ks * 16 + hh * 8 + 8]
    xf.append(B)
# 1. Q^T blocks (2 x 32 d rows)
qacc = [np.zeros((L, 16)) for _ in range(2)]
for b in range(2):
    for ks in range(C // 16):
        qacc[b] = mfma(a_frag_natural(Wq, h * 64 + b * 32, ks * 16), xf[ks], qacc[b])
scale = 0.125
# 2. S^T = K Q : 3 key blocks x 4 k-steps (b, gp)
sacc = [np.zeros((L, 16)) for _ in range(3)]
for kb in range(3):
    for b in range(2):
        for gp in range(2):
            sacc[kb] = mfma(a_frag_chain(K, kb * 32, 32 * b, gp), b_from_acc(qacc[b] * scale, gp), sacc[kb])
# 3. softmax per pixel column: lane-local over 48 values + the other half
m = np.full(L, -1e30)
for kb in range(3):
    for r in range(16):
        g, e = r >> 2, r & 3
        for lane in range(L):
            key = 32 * kb + 8 * g + 4 * (lane >> 5) + e
            if key >= NK: sacc[kb][lane, r] = -1e30
    m = np.maximum(m, sacc[kb].max(axis=1))
m = np.maximum(m, np.concatenate([m[32:], m[:32]]))   # shfl_xor 32
p = [np.exp(s - m[:, None]) for s in sacc]
l = sum(x.sum(axis=1) for x in p); l = l + np.concatenate([l[32:], l[:32]])
# 4. O^T = V^T P : 2 d blocks x 6 k-steps (kb, gp)
oacc = [np.zeros((L, 16)) for _ in range(2)]
for db in range(2):
    for kb in range(3):
        for gp in range(2):
            oacc[db] = mfma(a_frag_chain(Vt, db * 32, 32 * kb, gp), b_from_acc(p[kb], gp), oacc[db])
O = np.zeros((32, d))
for db in range(2):
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        for g in range(4):
            for e in range(4):
                O[l31, db * 32 + 8 * g + 4 * hh + e] = oacc[db][lane, 4 * g + e] / l[lane]
# reference
Q = X @ Wq[h * 64:(h + 1) * 64].T
S = (Q * scale) @ K[:NK].T
P = np.exp(S - S.max(axis=1, keepdims=True)); P /= P.sum(axis=1, keepdims=True)
ref = P @ V[:NK]
print("max |O - ref| =", np.abs(O - ref).max())
