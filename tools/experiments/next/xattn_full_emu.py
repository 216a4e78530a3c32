"""Functional emulation of ONE workgroup of xattn_full.hip on the CPU: LDS as an array of halves, every LDS-DMA piece, every fragment read
address, every MFMA (lane-level operand layouts of v_mfma_f32_32x32x16_f16), the transpose patches and the output rows are computed with
the kernel's own index arithmetic and compared with the dense computation.  What it cannot show: timing, the counted waits, bank conflicts.
    python tools/experiments/next/xattn_full_emu.py"""
import numpy as np

rng = np.random.default_rng(1)
C, HEADS, D, NKP, KLD, VLD, OLD, NK = 320, 5, 64, 96, 68, 100, 68, 77
KS = C // 16
STAGE = 32 * C * 2
OSTAGE = 160 * OLD * 2
SLOT = 6 * 4096
KVBUF = 16384
NST = 4            # ring depth (the K / V^T tiles are single-buffered)
PLD = 40
T = 256           # pixels per sample in this toy run (a workgroup's 128 pixels lie in one sample)
WG = 1            # the workgroup emulated (pixels 128..255)
n_samples = 2
P = n_samples * T
L = 64


def mfma(A, B, Dacc):
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        Am[l31, hh * 8:hh * 8 + 8] = A[lane]
        Bm[hh * 8:hh * 8 + 8, l31] = B[lane]
    Cm = Am @ Bm
    out = Dacc.copy()
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        for g in range(4):
            out[lane, 4 * g:4 * g + 4] += Cm[8 * g + 4 * hh:8 * g + 4 * hh + 4, l31]
    return out


def b_from_acc(Dacc, gp, scale=1.0):
    B = np.zeros((L, 8))
    for i in range(8):
        B[:, i] = Dacc[:, 4 * (2 * gp + (i >> 2)) + (i & 3)] * scale
    return B


# ---- tensors (values kept in float64; the emulation checks indexing, not rounding)
x = rng.standard_normal((P, C))
g_ln, b_ln = 1 + 0.2 * rng.standard_normal(C), 0.2 * rng.standard_normal(C)
Wq = rng.standard_normal((C, C)) / np.sqrt(C)
Wo = rng.standard_normal((C, C)) / np.sqrt(C)
bo = 0.3 * rng.standard_normal(C)
Kc = rng.standard_normal((n_samples, NK, C)); Vc = rng.standard_normal((n_samples, NK, C))
Kp = np.zeros((n_samples, HEADS, NKP, KLD)); Vt = np.zeros((n_samples, HEADS, D, VLD)); Wop = np.zeros((HEADS, C, OLD))
for h in range(HEADS):
    Kp[:, h, :NK, :D] = Kc[:, :, h * D:(h + 1) * D]
    Vt[:, h, :, :NK] = np.transpose(Vc[:, :, h * D:(h + 1) * D], (0, 2, 1))
    Wop[h, :, :D] = Wo[:, h * D:(h + 1) * D]
Wq_f, Kp_f, Vt_f, Wop_f = Wq.reshape(-1), Kp.reshape(-1), Vt.reshape(-1), Wop.reshape(-1)
scale, eps = 0.125, 1e-5

# ---- LDS (halves)
lds = np.zeros((NST * SLOT + 2 * KVBUF + 4 * 32 * PLD * 2) // 2)
RING, KB, VB = 0, NST * SLOT // 2, (NST * SLOT + KVBUF) // 2
PATCH0 = (NST * SLOT + 2 * KVBUF) // 2
n = (WG * 128) // T


def dma(dst_half, src, src_half_off, limit_bytes):
    """one 16-byte piece: zero beyond the descriptor's extent"""
    if src_half_off * 2 + 16 <= limit_bytes:
        lds[dst_half:dst_half + 8] = src[src_half_off:src_half_off + 8]
    else:
        lds[dst_half:dst_half + 8] = 0.0


def issue_stage(t, slot):
    h, kind = t >> 2, t & 3
    for tid in range(256):
        wave, lane = tid >> 6, tid & 63
        dst_b = RING * 2 + slot * SLOT + wave * 1024 + lane * 16
        if kind < 2:
            for j in range(5):
                i = j * 256 + tid
                row, cph = i // (C // 8), i % (C // 8)
                c = cph ^ ((row >> 1) & 7)
                wrel = (row * C + c * 8) * 2
                dma((dst_b + j * 4096) // 2, Wq_f, (wrel + (2 * h + kind) * STAGE) // 2, C * C * 2)
        else:
            base = (h * C + (kind - 2) * 160) * OLD
            for j in range(6):
                dma((dst_b + j * 4096) // 2, Wop_f[base:], ((j * 256 + tid) * 16) // 2, OSTAGE)


def issue_kv(h):
    gidx = n * HEADS + h
    for tid in range(256):
        wave, lane = tid >> 6, tid & 63
        for r in range(4):
            off = ((r * 256 + tid) * 16) // 2
            dma((KB * 2 + wave * 1024 + lane * 16 + r * 4096) // 2, Kp_f[gidx * NKP * KLD:], off, NKP * KLD * 2)
            dma((VB * 2 + wave * 1024 + lane * 16 + r * 4096) // 2, Vt_f[gidx * D * VLD:], off, D * VLD * 2)


def a_pieces(base_half, row_stride, row0, col):
    A = np.zeros((L, 8))
    for lane in range(L):
        l31, hh = lane & 31, lane >> 5
        o = base_half + (row0 + l31) * row_stride + col + 4 * hh
        A[lane, :4] = lds[o:o + 4]; A[lane, 4:] = lds[o + 8:o + 12]
    return A


out = np.zeros((P, C))
state = []
for wave in range(4):
    p0 = WG * 128 + wave * 32
    xf = []
    xn = x[p0:p0 + 32]
    mean = xn.mean(axis=1, keepdims=True); var = ((xn - mean) ** 2).mean(axis=1, keepdims=True)
    xn = (xn - mean) / np.sqrt(var + eps) * g_ln + b_ln
    for ks in range(KS):
        B = np.zeros((L, 8))
        for lane in range(L):
            B[lane] = xn[lane & 31, ks * 16 + (lane >> 5) * 8: ks * 16 + (lane >> 5) * 8 + 8]
        xf.append(B)
    state.append(dict(p0=p0, xf=xf, yacc=[np.zeros((L, 16)) for _ in range(10)]))

issue_kv(0)
for i in range(NST - 1):
    issue_stage(i, i)
rd_slot, wr_slot = 0, NST - 1
for h in range(HEADS):
    for st in state:
        st["qacc"] = [np.zeros((L, 16)) for _ in range(2)]
    for kind in range(4):
        s = 4 * h + kind
        # (barrier) DMAs land the moment they are issued here -- the earliest they can: an overwrite of something still needed shows up
        if kind == 2 and h + 1 < HEADS:
            issue_kv(h + 1)
        if s + NST - 1 < 4 * HEADS:
            issue_stage(s + NST - 1, wr_slot)
        wr_slot = 0 if wr_slot + 1 == NST else wr_slot + 1
        sW = (RING * 2 + rd_slot * SLOT) // 2
        rd_slot = 0 if rd_slot + 1 == NST else rd_slot + 1
        for st in state:
            if kind < 2:
                for k16 in range(KS):
                    A = np.zeros((L, 8))
                    for lane in range(L):
                        l31, hh = lane & 31, lane >> 5
                        tsw = hh ^ ((l31 >> 1) & 7)
                        aoff = l31 * (C * 2) + ((tsw ^ (2 * (k16 & 3))) << 4)
                        byte = aoff + (((2 * k16) & ~7) << 4)
                        A[lane] = lds[sW + byte // 2: sW + byte // 2 + 8]
                    st["qacc"][kind] = mfma(A, st["xf"][k16], st["qacc"][kind])
            else:
                for jb in range(5):
                    for ks2 in range(4):
                        A = a_pieces(sW, OLD, 32 * jb, 32 * (ks2 >> 1) + 16 * (ks2 & 1))
                        st["yacc"][5 * (kind - 2) + jb] = mfma(A, st["of"][ks2], st["yacc"][5 * (kind - 2) + jb])
            if kind == 1:
                kt = (KB * 2) // 2; vt = (VB * 2) // 2
                qf = [b_from_acc(st["qacc"][b], gp, scale) for b in range(2) for gp in range(2)]
                sc = [np.zeros((L, 16)) for _ in range(3)]
                for kb in range(3):
                    for ks in range(4):
                        sc[kb] = mfma(a_pieces(kt, KLD, 32 * kb, 32 * (ks >> 1) + 16 * (ks & 1)), qf[ks], sc[kb])
                mx = np.full(L, -3e38)
                for kb in range(3):
                    for r in range(16):
                        for lane in range(L):
                            key = 32 * kb + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)
                            if key >= NK:
                                sc[kb][lane, r] = -3e38
                    mx = np.maximum(mx, sc[kb].max(axis=1))
                mx = np.maximum(mx, np.concatenate([mx[32:], mx[:32]]))
                p = [np.exp(v - mx[:, None]) for v in sc]
                l = sum(v.sum(axis=1) for v in p); l = l + np.concatenate([l[32:], l[:32]])
                pf = [b_from_acc(p[kb], gp) for kb in range(3) for gp in range(2)]
                oa = [np.zeros((L, 16)) for _ in range(2)]
                for db in range(2):
                    for ks in range(6):
                        oa[db] = mfma(a_pieces(vt, VLD, 32 * db, 32 * (ks >> 1) + 16 * (ks & 1)), pf[ks], oa[db])
                st["of"] = [b_from_acc(oa[db] / l[:, None], gp) for db in range(2) for gp in range(2)]

# ---- epilogue
for wave, st in enumerate(state):
    patch = PATCH0 + wave * 32 * PLD
    for ob in range(10):
        for lane in range(L):
            l31, hh = lane & 31, lane >> 5
            for g4 in range(4):
                for e in range(4):
                    lds[patch + l31 * PLD + 8 * g4 + 4 * hh + e] = st["yacc"][ob][lane, 4 * g4 + e] + bo[ob * 32 + 8 * g4 + 4 * hh + e]
        for lane in range(L):
            rb_row, rb_chunk = lane >> 2, lane & 3
            for r in range(2):
                row = rb_row + 16 * r
                v = lds[patch + row * PLD + rb_chunk * 8: patch + row * PLD + rb_chunk * 8 + 8]
                pr = st["p0"] + row
                out[pr, ob * 32 + rb_chunk * 8: ob * 32 + rb_chunk * 8 + 8] = v + x[pr, ob * 32 + rb_chunk * 8: ob * 32 + rb_chunk * 8 + 8]

# ---- dense reference for the workgroup's 128 pixels
ref = np.zeros((128, C))
for i in range(128):
    pidx = WG * 128 + i
    xr = x[pidx]; m = xr.mean(); v = ((xr - m) ** 2).mean()
    xn = (xr - m) / np.sqrt(v + eps) * g_ln + b_ln
    q = Wq @ xn
    o = np.zeros(C)
    for h in range(HEADS):
        sc = (Kc[n, :, h * D:(h + 1) * D] @ q[h * D:(h + 1) * D]) * scale
        pr = np.exp(sc - sc.max()); pr /= pr.sum()
        o[h * D:(h + 1) * D] = pr @ Vc[n, :, h * D:(h + 1) * D]
    ref[i] = Wo @ o + bo + xr
err = np.abs(out[WG * 128:WG * 128 + 128] - ref).max()
print("max |emulated kernel - dense reference| = %.3e" % err)
assert err < 1e-9
