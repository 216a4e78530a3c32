"""Shared helpers for the parity tests (HIP path called through the C ABI vs the CPU oracle / plain torch fp32)."""
import ctypes
import math

import torch

from ladi_vton_amd import _lib
from ladi_vton_amd._lib import IGemmDesc, ptr, stream_ptr

ACT = dict(none=0, silu=1, gelu=2, geglu=3, relu=4)


def dev():
    return torch.device("cuda", 0)


def nhwc16(x):
    """[N,C,H,W] fp32 cpu -> NHWC fp16 on the GPU, channels zero-padded to a multiple of 64"""
    n, c, h, w = x.shape
    cp = (c + 63) // 64 * 64
    out = torch.zeros((n, h, w, cp), dtype=torch.float16)
    out[..., :c] = x.permute(0, 2, 3, 1).half()
    return out.to(dev())


def pack_conv_weight(w):
    """[cout,cin,k,k] -> [cout, k*k, cin_pad] fp16 (tap-major, channel-minor; the igemm layout)"""
    co, ci, k, _ = w.shape
    cp = (ci + 63) // 64 * 64
    out = torch.zeros((co, k * k, cp), dtype=torch.float16)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, k * k, ci).half()
    return out.reshape(co, k * k * cp).contiguous().to(dev())


def igemm(x, w_packed, cout, ksize=3, stride=1, pad=None, ups=0, x2=None, bias=None, rowadd=None, act="none", res0=None, res1=None,
          mask=None, out_f32=False, cfg=0, out_ld=None, ln=None, gn=None):
    """x, x2: NHWC fp16 gpu tensors [N,H,W,C]; returns NHWC output [N,Ho,Wo,ldo]"""
    lib = _lib.load()
    N, H, W, C0 = x.shape
    C1 = x2.shape[3] if x2 is not None else 0
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hl, Wl) if stride == 1 else (Hl // 2, Wl // 2)
    qout = cout // 2 if act == "geglu" else cout
    ldo = out_ld or qout
    out = torch.zeros((N, Ho, Wo, ldo), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
    d = IGemmDesc()
    d.src0, d.C0, d.ld0 = x.data_ptr(), C0, C0
    if x2 is not None:
        d.src1, d.C1, d.ld1 = x2.data_ptr(), C1, C1
    d.Hs, d.Ws, d.Ho, d.Wo, d.P = H, W, Ho, Wo, N * Ho * Wo
    d.ksize, d.stride, d.pad, d.ups = ksize, stride, (ksize // 2 if pad is None else pad), ups
    d.W, d.Q, d.K, d.ldw = w_packed.data_ptr(), cout, ksize * ksize * (C0 + C1), 0
    keep = []
    if bias is not None:
        b = bias.half().to(x.device); keep.append(b); d.bias = b.data_ptr()
    if rowadd is not None:
        r = rowadd.float().to(x.device); keep.append(r); d.rowadd = r.data_ptr()
    d.act, d.out_scale = ACT[act], 1.0
    if res0 is not None:
        d.res0, d.ldr0 = res0.data_ptr(), res0.shape[3]
    if res1 is not None:
        d.res1, d.ldr1 = res1.data_ptr(), res1.shape[3]
    if mask is not None:
        d.mask = mask.data_ptr()
    if ln is not None:       # (gamma, beta, eps[, with_scratch]): LayerNorm of the pixel operand (fused, or via the scratch)
        gm, bt = ln[0].half().to(x.device), ln[1].half().to(x.device)
        keep += [gm, bt]
        d.ln_gamma, d.ln_beta, d.ln_eps = gm.data_ptr(), bt.data_ptr(), float(ln[2])
        if len(ln) > 3 and ln[3]:
            scr = torch.empty((x.shape[0] * x.shape[1] * x.shape[2], x.shape[3]), dtype=torch.float16, device=x.device)
            keep.append(scr)
            d.ln_scratch = scr.data_ptr()
    if gn is not None:       # (scale_shift [n][C0][2] fp32 gpu tensor, pixels per sample): GroupNorm affine of the pixel operand
        keep.append(gn[0])
        d.gn_ss, d.gn_hw = gn[0].data_ptr(), int(gn[1])
    d.out, d.ldo, d.out_f32 = out.data_ptr(), ldo, int(out_f32)
    rc = lib.ladi_op_igemm(ctypes.byref(d), 1, cfg, stream_ptr())
    assert rc == 0, "igemm rc=%d %s" % (rc, _lib.last_error())
    torch.cuda.synchronize()
    return out


def to_nchw(y, c=None):
    y = y.float().cpu().permute(0, 3, 1, 2)
    return y[:, :c] if c is not None else y


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def psnr(a, b, peak=None):
    a, b = a.double(), b.double()
    peak = float(b.abs().max()) if peak is None else peak
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def cpu_quota_threads():
    """usable host threads: min(affinity, cgroup cpu.max quota) — torch oversubscribes badly when it sees 256 logical CPUs
    but the container is capped (the GPU box: quota 16 of 256)"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


PARITY_NAME = "parity_r06.json"


def library_digest():
    """sha256 over the sources / headers / flags libladi_native.so is built from (ladi_vton_amd/build.py: the stamp the loader checks)"""
    from ladi_vton_amd import build
    return build._digest()


def parity_path():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", PARITY_NAME)


def record_parity(key, value, name=None):
    """Measured parity values of the GPU suite -> gpurun_out/parity_r06.json (the only directory that comes back from the GPU box), stamped
    with the library digest and the pytest session id: tests/conftest.py removes the file at the start of every GPU session, so the record
    is ONE run of ONE binary, never a stitch.  `python tools/commit_parity.py` copies it to profiles/r06_parity.json after
    checking the digest against the sources; tests/test_cpu.py checks that every key the documents cite is in the committed file."""
    import json
    import os
    path = parity_path() if name is None else os.path.join(os.path.dirname(parity_path()), name)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        blob = json.load(open(path)) if os.path.exists(path) else {}
        blob["_library_digest"] = library_digest()
        blob.setdefault("_session", os.environ.get("LADI_PYTEST_SESSION", ""))
        blob[key] = value
        json.dump(blob, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
