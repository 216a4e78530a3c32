"""Launch one implicit-GEMM problem with fixed tile configurations a few times (for rocprofv3 --pmc runs on the GPU box).
python tools/run_one.py --shape 64,48,320,320,3 --cfgs 22,32,39 [--n 16] [--iters 5]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_shapes import make_problem  # noqa: E402
from ladi_vton_amd import _lib  # noqa: E402
from ladi_vton_amd._lib import stream_ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="64,48,320,320,3")
ap.add_argument("--cfgs", default="22,32,39")
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--epi", default="")
a = ap.parse_args()
H, W, cin, cout, k = [int(v) for v in a.shape.split(",")]
lib = _lib.load()
lib.ladi_igemm_set_autotune(0)
d, keep = make_problem(a.n, H, W, cin, cout, k, a.epi)
for c in [int(v) for v in a.cfgs.split(",")]:
    for _ in range(a.iters):
        rc = lib.ladi_op_igemm(ctypes.byref(d), 1, c, stream_ptr())
        assert rc == 0, (c, rc)
torch.cuda.synchronize()
