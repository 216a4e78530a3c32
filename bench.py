#!/usr/bin/env python
"""bench.py — try-on images/s of the native LaDI-VTON hot path (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch: B try-on pairs in (device resident) -> [producers, configs 2/3 only: CLIP
vision encoder -> inversion adapter -> CLIP text encoder with the pseudo-word splice] -> VAE encodes + EMASC -> N scheduler steps of
the CFG UNet (hipGraph) -> VAE decode with EMASC skips -> uint8 images (all-gathered over ranks when N_gpus > 1).

  --config 1 (default)  BASELINE.json configs[1]: batch 8 / GPU, 50 PNDM steps, 512x384            <- the configuration `metric` is quoted on
  --config 2            configs[2]: batch 32, 50 PNDM, 512x384, inversion adapter + vision / text encoders + EMASC inside the step
  --config 3            configs[3]: batch 256 over 8 GPUs = 32 / GPU (run with --gpus 8), producers on
  --config 4            configs[4]: 1024x768, 100 DDIM steps, batch 8 / GPU (64 over 8 GPUs)
Synthetic inputs with the datasets' shapes / value ranges and a deterministic random-init checkpoint of the released architecture
(no weights or datasets are reachable offline).  Global-batch row g depends only on g (per-row generators), rows are sharded
contiguously over ranks (ladi_vton_amd.parallel.run_sharded), so results do not depend on the world size.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N ...        (no launcher: bench.py re-executes itself under the torch.distributed.run line above, one rank per GPU)
`--gpus N` always means N ranks: fewer visible GPUs than N, or a launcher that started another WORLD_SIZE, is exit code 2 and no line.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import re
import statistics
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work (SURVEY.md §8d / BASELINE.md §2), 2*MAC FLOPs
UNET_FLOP_PER_SAMPLE_64x48 = 581.70e9
TRYON_FLOP_PER_IMAGE = {("ddim", 50, 512): 62.16e12, ("lms", 50, 512): 62.16e12, ("pndm", 50, 512): 63.32e12, ("ddim", 100, 1024): 644.9e12}
PEAK_F16_TFLOPS = 2500.0   # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
CONFIGS = {   # BASELINE.json `configs` index -> (batch per GPU, steps, scheduler, H, W, producers)
    1: dict(batch=8, steps=50, scheduler="pndm", H=512, W=384, producers=False,
            name="BASELINE configs[1]: VITON-HD-paired-like, batch 8 per GPU"),
    2: dict(batch=32, steps=50, scheduler="pndm", H=512, W=384, producers=True,
            name="BASELINE configs[2]: DressCode-upper_body-unpaired-like, batch 32, inversion adapter + CLIP vision / text encoders in the step"),
    3: dict(batch=32, steps=50, scheduler="pndm", H=512, W=384, producers=True,
            name="BASELINE configs[3]: VITON-HD-unpaired-like, batch 256 over 8 GPUs = 32 per GPU"),
    4: dict(batch=8, steps=100, scheduler="ddim", H=1024, W=768, producers=False,
            name="BASELINE configs[4]: DressCode-like 1024x768, batch 64 over 8 GPUs = 8 per GPU"),
}


def igemm_symbols(lib):
    """tile configuration id (csrc/igemm.hip kCfg) -> kernel symbol as rocprofv3 prints it; split-K variants launch the symbol of their
    base tile.  The table lives in the library (ladi_igemm_cfg_symbol_name), so a new tile shape needs no edit here."""
    n = lib.ladi_igemm_cfg_count()
    return n, {c: lib.ladi_igemm_cfg_symbol_name(c).decode() for c in range(1, n + 1)}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs[] index")
    p.add_argument("--batch", type=int, default=None, help="override: try-on pairs per GPU (weak scaling)")
    p.add_argument("--inference-steps", type=int, default=None)
    p.add_argument("--scheduler", default=None, choices=["pndm", "ddim", "lms"])
    p.add_argument("--height", type=int, default=None)
    p.add_argument("--width", type=int, default=None)
    p.add_argument("--size", default="full", choices=["full", "tiny"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-runs", type=int, default=3, help="end-to-end runs of config #0 on the host cores, median reported (BASELINE.md §3 protocol: 3)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--measure-traffic", action="store_true",
                   help="collect FETCH_SIZE / WRITE_SIZE of the dominant kernel NOW (two rocprofv3 --pmc child runs of --roofline-only) "
                        "instead of reading the committed, digest-checked profiles/")
    p.add_argument("--roofline-only", action="store_true",
                   help="only the dominant-kernel measurement: CFG UNet forwards at the bench batch (the command the rocprofv3 "
                        "summaries under profiles/ are taken from)")
    p.add_argument("--roofline-iters", type=int, default=4)
    p.add_argument("--no-tail", action="store_true", help="skip the D2H + PIL leg (with_d2h_pil_images_per_s)")
    p.add_argument("--stub-step", action="store_true",
                   help="rehearsal of the launch / sharding / collective / timing / reporting code WITHOUT the native pipeline: the step's compute is "
                        "a per-row CPU function, the collective backend is gloo (tests/test_cpu.py drives `bench.py --gpus 2 --stub-step`); the printed "
                        "line says data = 'stub' and can never be mistaken for a measurement")
    p.add_argument("--lanes", type=int, default=None,
                   help="sample-group lanes of the UNet forward inside the denoising loop (csrc/runtime.h UNetLanes); default: the library's "
                        "own choice (LADI_UNET_LANES, else 1: two concurrent half-batch forwards measured 3.5 % slower, DESIGN.md)")
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------------
# synthetic global batch: row g is a function of g only
# ---------------------------------------------------------------------------------------------------------------------------------
def make_rows(lo, hi, H, W, L, D, device, seed=1234):
    """rows [lo, hi) of the global batch (SURVEY.md §8d shapes and value ranges); the shared negative prompt is row-independent"""
    F = torch.nn.functional
    ys = torch.arange(H, dtype=torch.float32)[None, :, None]
    xs = torch.arange(W, dtype=torch.float32)[None, None, :]
    rows = {k: [] for k in ("image", "mask_image", "pose_map", "warped_cloth", "cloth", "prompt_embeds", "noise_cloth", "noise_latents",
                            "noise_masked", "word_ids")}
    for gidx in range(lo, hi):
        g = torch.Generator(device="cpu").manual_seed(seed * 1000003 + gidx)

        def smooth():
            low = torch.rand((1, 3, H // 8, W // 8), generator=g) * 2 - 1
            return F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(-1, 1)[0]

        rows["image"].append(smooth()); rows["warped_cloth"].append(smooth()); rows["cloth"].append(smooth())
        m = torch.zeros(1, H, W); m[:, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
        rows["mask_image"].append(m)
        cy = torch.rand((18, 1, 1), generator=g) * H
        cx = torch.rand((18, 1, 1), generator=g) * W
        pose = torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 81.0)
        pose[7::9] = 0.0
        rows["pose_map"].append(pose)
        rows["prompt_embeds"].append(torch.randn((L, D), generator=g))
        for k in ("noise_cloth", "noise_latents", "noise_masked"):
            rows[k].append(torch.randn((4, H // 8, W // 8), generator=g))
        # "a photo of a model wearing <category garment> $ x16" (src/inference.py:289): BOS, 8-10 words, 16 pseudo-word slots, EOS, padding
        ids = torch.zeros(77, dtype=torch.int32)
        nw = 8 + gidx % 3
        ids[0] = 49406
        ids[1:1 + nw] = torch.randint(300, 40000, (nw,), generator=g).int()
        ids[1 + nw:1 + nw + 16] = 259
        ids[1 + nw + 16] = 49407
        rows["word_ids"].append(ids)
    out = {}
    for k, v in rows.items():
        t = torch.stack(v)
        if k.startswith("noise"):
            out[k] = t.to(device)
        elif k == "word_ids":
            out[k] = t
        else:
            out[k] = t.to(device=device, dtype=torch.float16)
    gneg = torch.Generator(device="cpu").manual_seed(seed + 4)
    out["negative_prompt_embeds"] = torch.randn((1, L, D), generator=gneg).expand(hi - lo, L, D).contiguous().to(device=device, dtype=torch.float16)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# roofline.traffic: HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes
# ---------------------------------------------------------------------------------------------------------------------------------
def lib_digest():
    try:
        return open(os.path.join(ROOT, "ladi_vton_amd", "csrc", "_obj", "stamp")).read().strip()
    except OSError:
        return None


def _pmc_avg_kib(db_path, symbol):
    import sqlite3
    c = sqlite3.connect(db_path)
    rows = c.execute("select name, count(*), avg(counter_value) from pmc_events group by name").fetchall()
    for n, cnt, avg in rows:
        if symbol in re.sub(r"\(anonymous namespace\)::", "", n):
            return float(avg), int(cnt)
    return None, 0


def traffic_measured(symbol, extra_args):
    """two separate rocprofv3 --pmc child runs of `bench.py --roofline-only` (FETCH_SIZE and WRITE_SIZE cannot share a pass:
    MI355X_MICROARCH.md §rocprofv3 PMC slots); FETCH_SIZE doubled (gfx950 reports 64 B per 128-B request, §HBM)"""
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ladi_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--roofline-only",
               "--no-cpu-baseline"] + extra_args
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        if r.returncode != 0 or not dbs:
            return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
        avg, cnt = _pmc_avg_kib(dbs[0], symbol)
        if avg is None:
            return None, "kernel %s not in the %s pass" % (symbol, ctr)
        vals[ctr] = (avg * 1024.0, cnt)
        subprocess.run(["rm", "-rf", d])
    f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return {"bytes_per_launch": round(2.0 * f + w), "fetch_bytes_x2": round(2.0 * f), "write_bytes": round(w),
            "launches": vals["FETCH_SIZE"][1], "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH doubled)"}, None


PROFILE_ROUND = "r06"      # the committed rocprofv3 summaries this line may cite (profiles/<round>_*), digest-checked


def _pmc_file(prefix, tag):
    """{symbol line -> (avg KiB per dispatch, avg us under the PMC pass)} of profiles/<round>_<prefix>pmc_<tag>_size.txt, or (None, why)"""
    path = os.path.join(ROOT, "profiles", "%s_%spmc_%s_size.txt" % (PROFILE_ROUND, prefix, tag))
    if not os.path.exists(path):
        return None, "no committed PMC pass (%s)" % os.path.relpath(path, ROOT)
    lines = open(path).read().splitlines()
    m = re.match(r"# lib_digest=(\w+)", lines[0]) if lines else None
    if not m or m.group(1) != lib_digest():
        return None, "committed PMC passes are stale (taken with another library build); re-run tools/make_profiles.sh"
    out = {}
    for line in lines[2:]:
        f = line.split()
        if len(f) >= 6:
            out[line] = (float(f[-2]), float(f[-1]))
    return out, None


def _stats_file(prefix):
    """{symbol line -> (calls, avg us)} of profiles/<round>_<prefix>kernel_stats.txt (rocprofv3 --kernel-trace --stats of the same command)"""
    path = os.path.join(ROOT, "profiles", "%s_%skernel_stats.txt" % (PROFILE_ROUND, prefix))
    if not os.path.exists(path):
        return None
    out = {}
    for line in open(path).read().splitlines()[1:]:
        f = line.split()
        if len(f) >= 7 and not line.startswith("TOTAL"):
            try:
                out[line] = (int(f[-6]), float(f[-4]))
            except ValueError:
                pass
    return out


HBM_KERNELS = ("gn_norm_kernel", "gn_reduce_rows_kernel", "gn_apply_kernel", "gn_finalize_kernel", "gn_partial_kernel", "splitk_reduce_kernel", "layernorm_kernel", "sched_step_kernel",
               "image_post_kernel", "assemble_static_kernel")


def hbm_kernels():
    """north_star "achieved HBM GB/s": for the HBM-bound kernels of the path (normalisation, split-K reduce, scheduler step) and the
    convolutions of the VAE stages, bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE of the committed, digest-checked PMC passes (FETCH
    doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md) divided by the kernel's average duration in the committed
    rocprofv3 kernel trace of the same command.  profiles/README.md shows the recomputation."""
    res = {}
    for prefix, leg, pick in (("", "unet_forward", lambda n: any(k in n for k in HBM_KERNELS)),
                              ("vae_", "vae_stages", lambda n: True)):
        fe, why = _pmc_file(prefix, "fetch")
        wr, why2 = _pmc_file(prefix, "write")
        st = _stats_file(prefix if prefix else "unet_forward_")
        if fe is None or wr is None or st is None:
            res[leg] = {"note": why or why2 or "no committed kernel trace"}
            continue

        def key(line):
            return re.split(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+", line)[0].strip()
        fk = {key(l): v for l, v in fe.items()}
        wk = {key(l): v for l, v in wr.items()}
        rows = {}
        for line, (calls, us) in st.items():
            name = re.split(r"\s{2,}", line.strip())[0]
            if not pick(name):
                continue
            k78 = name[:78]
            if k78 in fk and k78 in wk and us > 0:
                b = 2.0 * fk[k78][0] * 1024.0 + wk[k78][0] * 1024.0
                # mangled names of the anonymous-namespace kernels: keep the identifier (lower-case snake case), drop template arguments and signature
                short = re.match(r"[a-z_0-9]+", re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)).group(0) if name.startswith("_ZN") else name.replace("void ", "")
                rows[short[:60]] = {"GBps": round(b / us / 1e3, 1), "frac_of_8TBps": round(b / us / 1e3 / 8000.0, 3), "avg_us": us,
                                    "MB_per_launch": round(b / 1e6, 2), "launches": calls}
        res[leg] = dict(sorted(rows.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:10])
    res["source"] = "profiles/%s_{,vae_}pmc_{fetch,write}_size.txt + %s_{unet_forward,vae}_kernel_stats.txt (same library digest)" % (PROFILE_ROUND, PROFILE_ROUND)
    return res


def traffic_committed(symbol):
    """profiles/<round>_pmc_{fetch,write}_size.txt, valid only for the library build they were taken with (first line: # lib_digest=...)"""
    vals = {}
    for tag in ("fetch", "write"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s_size.txt" % (PROFILE_ROUND, tag))
        if not os.path.exists(path):
            return None, "no committed PMC pass (profiles/%s_pmc_%s_size.txt)" % (PROFILE_ROUND, tag)
        lines = open(path).read().splitlines()
        m = re.match(r"# lib_digest=(\w+)", lines[0]) if lines else None
        if not m or m.group(1) != lib_digest():
            return None, "committed PMC passes are stale (taken with another library build); re-run with --measure-traffic"
        # the tile table names a kernel exactly as rocprofv3 prints it -- every template argument, defaulted ones included (round 6; the prefix
        # match of round 5 could hand a ring-halo symbol the 2-D blocked form's bytes when both were in one trace: ADVICE r05)
        for line in lines[1:]:
            if symbol in line:
                vals[tag] = float(line.split()[-2]) * 1024.0   # avg KiB per dispatch -> bytes
                break
    if len(vals) != 2:
        return None, "kernel %s not in the committed PMC passes" % symbol
    return {"bytes_per_launch": round(2.0 * vals["fetch"] + vals["write"]), "fetch_bytes_x2": round(2.0 * vals["fetch"]),
            "write_bytes": round(vals["write"]), "source": "profiles/%s_pmc_{fetch,write}_size.txt (same library digest; avg over all launches "
            "of this kernel in `bench.py --roofline-only`)" % PROFILE_ROUND}, None


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the fp32 oracle (a port) on BASELINE configs[0], end to end
# ---------------------------------------------------------------------------------------------------------------------------------
def vendor_gemm_calibration(dev, achieved_tflops, nmk=8192, iters=30):
    """one large fp16 GEMM through hipBLASLt (torch.matmul), timed with torch events on torch's current stream: the sustained dense-fp16 rate of
    this box (profiles/r06_library_path.json: 1 379 TFLOP/s = 0.55 of the nominal peak).  Reported, never required."""
    try:
        g = torch.Generator(device=dev).manual_seed(7)
        a_ = (torch.randn((nmk, nmk), device=dev, generator=g) * 0.05).half()
        b_ = (torch.randn((nmk, nmk), device=dev, generator=g) * 0.05).half()
        for _ in range(10):
            torch.matmul(a_, b_)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.matmul(a_, b_)
        e1.record()
        torch.cuda.synchronize()
        tf = 2.0 * nmk ** 3 * iters / e0.elapsed_time(e1) / 1e9
        return {"tflops": round(tf, 1), "shape": "%dx%dx%d fp16, torch.matmul (hipBLASLt)" % (nmk, nmk, nmk),
                "dominant_kernel_over_vendor_gemm": round(achieved_tflops / tf, 4), "what": "comparison point only; the product path uses no vendor library"}
    except Exception as e:
        return {"error": repr(e)[:200]}


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # respect the container's cgroup CPU quota (the GPU box shows 256 logical CPUs but grants 16)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    return cores


def cpu_baseline(sds, cfgs, runs, target_evals, scheduler, size="full"):
    """BASELINE configs[0] (B = 1, 20 scheduler steps, 512x384, CFG 7.5, EMASC on) END TO END through the fp32 CPU oracle (a port: the
    reference itself is not importable here) on this box's host cores; median over `runs`.  images/s for the metric's
    step count = 1 / (t_run + (target_evals - evals_20) * mean UNet evaluation time of the same run)."""
    from oracle import models as M      # test infrastructure: only used as the reported CPU baseline
    from oracle import pipeline as P
    cores = host_cores()
    torch.set_num_threads(cores)
    inp = P.synthetic_inputs(1, 512, 384, L=77, D=cfgs["unet"]["cross_attention_dim"])
    totals, evals_t = [], []
    n_evals = 0
    with torch.no_grad():
        for _ in range(max(1, runs)):
            ts = []

            def unet_fn(x, t, e):
                t0 = time.time()
                y = M.unet_forward(sds["unet"], cfgs["unet"], x, t, e)
                ts.append(time.time() - t0)
                return y

            t0 = time.time()
            P.tryon_pipeline(sds["unet"], cfgs["unet"], sds["vae"], cfgs["vae"], sds["emasc"], inp, num_inference_steps=20, guidance_scale=7.5,
                             scheduler=scheduler, unet_fn=unet_fn)
            totals.append(time.time() - t0)
            evals_t.append(sum(ts) / len(ts))
            n_evals = len(ts)
    t_run, t_eval = statistics.median(totals), statistics.median(evals_t)
    per_image = t_run + (target_evals - n_evals) * t_eval
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "fp32 torch-CPU oracle, " + size + "-size model, BASELINE configs[0] end to end (B=1, 20 %s steps = %d CFG UNet evaluations, 512x384, EMASC on): "
                      "%d run(s), median %.1f s (min %.1f s), %.2f s per CFG evaluation; images/s at the metric's %d evaluations = 1 / (t_run + %d x t_eval)"
                      % (scheduler.upper(), n_evals, len(totals), t_run, min(totals), t_eval, target_evals, target_evals - n_evals),
            "config0_seconds_per_image": round(t_run, 2), "config0_images_per_s": round(1.0 / t_run, 5)}


def make_step(run_local, local, lo, per_rank, global_B):
    """The step of the bench as a function (also driven, with a stub run_local, by the 2-process gloo rehearsal in tests/test_cpu.py): the
    row materialiser hands run_sharded this rank's resident rows of the GLOBAL batch [lo, lo + per_rank), run_sharded runs the local
    shard and performs the path's only collective (all-gather of the uint8 images; RCCL on the GPU box)."""
    from ladi_vton_amd.parallel import run_sharded

    def rows(lo_, hi_):
        assert (lo_, hi_) == (lo, lo + per_rank), ((lo_, hi_), (lo, lo + per_rank))
        return local

    def one_step():
        return run_sharded(run_local, rows, batch=global_B)
    return one_step


def d2h_pil_tail(images_u8):
    """what the reference does with the result (tryon_pipe.py:357-360 numpy_to_pil, inference.py:314-324 save as JPEG quality 95): device ->
    host copy, PIL image per sample, encode.  Encoded into memory (no disk in the measurement)."""
    import io
    from PIL import Image
    host = images_u8.cpu().numpy()
    n = 0
    for im in host:
        buf = io.BytesIO()
        Image.fromarray(im).save(buf, format="JPEG", quality=95)
        n += buf.tell()
    return n


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def launch_ranks(a, argv):
    """`--gpus N` is a promise about the number of ranks, not a label.  Called with --gpus N > 1 and no launcher around
    it (no WORLD_SIZE in the environment), this process becomes the launcher: it replaces itself by `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` -- one rank per GPU -- after checking that the box has
    N GPUs (exit code 2 and a message otherwise: one rank printing `n_gpus: 1` under a `--gpus 8` command line must not happen)."""
    if not a.stub_step:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this box; one process per GPU is the only mode -- refusing to run "
                             "fewer ranks than the command line promises\n" % (a.gpus, have))
            sys.exit(2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush(); sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def stub_run_local(inp):
    """--stub-step: a per-row CPU function of the row's tensors in place of producers + fused try-on (same stand-in as the gloo rehearsal
    in tests/test_cpu.py) -- [b, H, W, 3] floats in [0, 1]"""
    x = inp["image"].float() * 0.25 + inp["cloth"].float() * 0.25 + 0.5 + inp["noise_latents"].mean(dim=(1, 2, 3)).view(-1, 1, 1, 1) * 0.01
    return x.permute(0, 2, 3, 1).clamp(0, 1).float().contiguous()


def main():
    a = parse()
    cfg = CONFIGS[a.config]
    if a.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a, sys.argv[1:])          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; the line's n_gpus would contradict the command line\n"
                         % (a.gpus, world))
        sys.exit(2)
    stub = a.stub_step
    if stub:
        dev = torch.device("cpu")
        a.no_roofline = a.no_cpu_baseline = a.no_tail = True
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU: the native path has no CPU fallback"
        torch.cuda.set_device(local_rank)          # bind the rank to its GPU before RCCL is initialised
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if stub else "nccl", rank=rank, world_size=world)
    B = a.batch or cfg["batch"]
    H, W = a.height or cfg["H"], a.width or cfg["W"]
    steps_inf = a.inference_steps or cfg["steps"]
    scheduler = a.scheduler or cfg["scheduler"]
    producers = cfg["producers"]

    from ladi_vton_amd import configs as C      # shapes / configuration tables only: imports without a GPU
    ucfg, vcfg = (C.UNET_FULL, C.VAE_FULL) if a.size == "full" else (C.UNET_TINY, C.VAE_TINY)
    ecfg = C.emasc_for_vae(vcfg)
    cfgs = dict(unet=ucfg, vae=vcfg, emasc=ecfg)
    Ltok, D = 77, ucfg["cross_attention_dim"]
    global_B = B * world
    lo = rank * B
    want_cpu = (rank == 0 and world == 1 and not a.no_cpu_baseline and not a.roofline_only)
    t_build = time.time()
    pipe = unet = sds = None
    if stub:
        run_local = stub_run_local
    else:
        import ladi_vton_amd as L
        from ladi_vton_amd import _lib
        if want_cpu:   # the CPU baseline needs the fp32 checkpoint on the host; otherwise stream it tensor by tensor
            sds = dict(unet=C.synth_state_dict(C.unet_shapes(ucfg), "unet."), vae=C.synth_state_dict(C.vae_shapes(vcfg), "vae."),
                       emasc=C.synth_state_dict(C.emasc_shapes(ecfg), "emasc."))
            unet, vae, emasc = L.NativeUNet(ucfg, sds["unet"]), L.NativeVAE(vcfg, sds["vae"]), L.NativeEMASC(ecfg, sds["emasc"])
        else:
            unet = L.NativeUNet(ucfg, C.synth_items(C.unet_shapes(ucfg), "unet."))
            vae = L.NativeVAE(vcfg, C.synth_items(C.vae_shapes(vcfg), "vae."))
            emasc = L.NativeEMASC(ecfg, C.synth_items(C.emasc_shapes(ecfg), "emasc."))
        vision = adapter = text = None
        if producers and not a.roofline_only:
            assert a.size == "full", "the producers are only wired for the released sizes"
            vision = L.NativeCLIPVisionEncoder(C.VISION_FULL, C.synth_items(C.vision_shapes(C.VISION_FULL), "vision."))
            adapter = L.NativeInversionAdapter(C.ADAPTER_FULL, C.synth_items(C.adapter_shapes(C.ADAPTER_FULL), "adapter."))
            text = L.NativeCLIPTextEncoder(C.TEXT_FULL, C.synth_items(C.text_shapes(C.TEXT_FULL), "text."))
        sch = {"ddim": L.DDIMScheduler, "pndm": L.PNDMScheduler, "lms": L.LMSDiscreteScheduler}[scheduler]()
        pipe = L.StableDiffusionTryOnePipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=sch, emasc=emasc,
                                               emasc_int_layers=[1, 2, 3, 4, 5])
        if a.lanes is not None:
            pipe.lanes = a.lanes

        def run_local(inp):
            pe = inp["prompt_embeds"]
            if producers:   # src/inference.py:267-295: in-shop cloth -> CLIP ViT-H/14 -> inversion adapter -> pseudo-word splice -> CLIP text encoder
                feats = vision(L.clip_preprocess(inp["cloth"])).last_hidden_state     # resize + clamp + CLIP normalisation: one kernel
                words = adapter(feats).reshape(feats.shape[0], 16, -1)
                pe = L.encode_text_word_embedding(text, inp["word_ids"], words, 16).last_hidden_state
            return pipe._run_fused(inp["image"], inp["mask_image"], inp["pose_map"], inp["warped_cloth"], pe, inp["negative_prompt_embeds"],
                                   inp["noise_cloth"], inp["noise_latents"], inp["noise_masked"], H, W, steps_inf, 7.5, 1.0, False, not a.no_graph,
                                   return_device=True, out_uint8=True)     # uint8 straight from the decode epilogue (numpy_to_pil rounding)
    t_build = time.time() - t_build
    local = make_rows(lo, lo + B, H, W, Ltok, D, dev)      # this rank's rows of the global batch, resident in HBM before the timed region

    one_step = make_step(run_local, local, lo, B, global_B)   # contiguous row shards + the path's only collective (RCCL all-gather of uint8 images)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def fence():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    if a.roofline_only:
        # context for the stand-alone UNet forwards; one untimed forward first (per-shape tile measurement happens there)
        e = torch.cat([local["negative_prompt_embeds"], local["prompt_embeds"]]).contiguous()
        unet.set_context(e)
        unet.time_forward(2 * B, H // 8, W // 8, 1)
        out = torch.zeros((global_B, H, W, 3), dtype=torch.uint8, device=dev)
        a.warmup, a.steps = 0, 0
    for _ in range(a.warmup):
        out = one_step()
    fence()
    t0 = time.time()
    for _ in range(a.steps):
        out = one_step()
    fence()
    dt = max(time.time() - t0, 1e-9)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    assert out.shape[0] == global_B and out.dtype == torch.uint8
    images_per_s = global_B * a.steps / dt if a.steps else 0.0
    # who took part in the collective (outside the timed region): the RCCL version and the number of DISTINCT ranks an all-gather of the rank ids
    # returned on rank 0 -- so that a SCALE_rNN.json line can be checked for "RCCL saw N ranks" 
    rccl = {"version": None, "ranks_seen": 1, "backend": None}
    if dist.is_available() and dist.is_initialized():
        ids = torch.empty((world,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int32, device=dev))
        rccl["ranks_seen"] = int(ids.unique().numel())
        rccl["backend"] = dist.get_backend()
        try:
            rccl["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:       # noqa: BLE001  (reporting only)
            rccl["version"] = "unavailable: %r" % (e,)
    # the same step WITH the reference's output tail (D2H + PIL + JPEG encode of every image, on rank 0's host cores),
    # timed separately so that `value` stays the device-resident number
    tail = None
    if a.steps and not a.no_tail and not a.roofline_only:
        nt = min(a.steps, 3)
        fence()
        t1 = time.time()
        nbytes = 0
        for _ in range(nt):
            out = one_step()
            if rank == 0:
                nbytes = d2h_pil_tail(out)
        fence()
        dt_tail = max(time.time() - t1, 1e-9)
        tail = {"images_per_s": round(global_B * nt / dt_tail, 4), "ms_per_step": round(dt_tail / nt * 1000.0, 2), "steps": nt,
                "what": "step + D2H of the uint8 batch + PIL.Image + JPEG(quality 95) encode of every image on rank 0 (inference.py:314-324)",
                "jpeg_bytes_per_batch": nbytes}
    evals = steps_inf + (1 if scheduler == "pndm" else 0)
    lib = None if stub else _lib.load()
    stage = (ctypes.c_float * 3)()
    stage_ms = [float(stage[i]) for i in range(3)] if (pipe is not None and pipe._tryon and lib.ladi_tryon_stage_ms(pipe._tryon, stage) == 0) else None
    flop_img = TRYON_FLOP_PER_IMAGE.get((scheduler, steps_inf, H)) if a.size == "full" and W * 4 == H * 3 else None

    roofline = None
    if rank == 0 and not a.no_roofline:
        # dominant kernel = the kernel SYMBOL with the largest summed launch time inside CFG UNet forwards at the bench batch (UNet = 94 % of the
        # path's FLOPs); per-launch HIP events on the launch stream around the main kernel of every implicit-GEMM launch.
        n = 2 * B
        h, w = H // 8, W // 8
        whole_ms = unet.time_forward(n, h, w, a.roofline_iters)
        # the forward the way the denoising loop runs it (sample-group lanes on parallel streams inside one hipGraph), with the shader clock
        # read next to it by a one-wave probe on a side stream (DVFS: every "fraction of peak" here is against the NOMINAL 2.4 GHz)
        lanes_ms, lanes_n, clock = None, None, None
        try:
            probe = torch.zeros(2, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream()
            lanes_n = int(a.lanes) if a.lanes else int(os.environ.get("LADI_UNET_LANES", "1"))
            while lanes_n > 1 and n % lanes_n:
                lanes_n -= 1
            unet.time_forward_lanes(n, h, w, 1, lanes_n)                      # builds / tunes
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                lib.ladi_clock_probe(ctypes.c_ulonglong(int(whole_ms * 1e5 * 6)), ctypes.c_void_p(probe.data_ptr()), ctypes.c_void_p(side.cuda_stream))
            lanes_ms = unet.time_forward_lanes(n, h, w, max(a.roofline_iters, 8), lanes_n)
            torch.cuda.synchronize()
            cyc, ticks = [int(v) for v in probe.cpu()]
            clock = {"under_unet_forward_mhz": round(cyc / max(ticks, 1) * 100.0, 1), "nominal_mhz": 2400.0,
                     "how": "one-wave probe on a side stream (s_memtime / s_memrealtime over %.1f ms) while the forwards run" % (ticks / 1e5)}
        except Exception as e:   # reported, never required
            clock = {"error": repr(e)}
        lib.ladi_profile_igemm_enable(1)
        unet.time_forward(n, h, w, a.roofline_iters)   # 1 warm-up + roofline_iters timed forwards, all recorded
        lib.ladi_profile_igemm_enable(0)
        NCFG, SYMBOL = igemm_symbols(lib)
        # per-launch records grouped by the EXACT kernel symbol (as rocprofv3 prints it; the X-stationary kernel has one per <KH, PB, MODE, LN>)
        sbuf = ctypes.create_string_buffer(1 << 16)
        lib.ladi_profile_igemm_symbols(sbuf, len(sbuf))
        sym = {}
        for line in sbuf.value.decode().splitlines():
            name, ms_, fl_, cnt_ = line.split("\t")
            sym[name] = dict(ms=float(ms_), flop=float(fl_), launches=int(cnt_), cfgs=[])
        prof = (ctypes.c_double * (3 * (NCFG + 1)))()
        lib.ladi_profile_igemm_collect(prof, 3 * (NCFG + 1))
        for c_ in range(1, NCFG + 1):
            if prof[c_ * 3 + 2] > 0:
                for name in sym:
                    if name == SYMBOL.get(c_) or (SYMBOL.get(c_) == "linear_xs_kernel" and name.startswith("linear_xs_kernel")):
                        sym[name]["cfgs"].append(c_)
        if sym:
            dom = max(sym, key=lambda k: sym[k]["ms"])
            d = sym[dom]
            ach = d["flop"] / d["ms"] / 1e9
            extra = ["--config", str(a.config), "--batch", str(B), "--height", str(H), "--width", str(W), "--size", a.size]
            default_workload = (a.config == 1 and B == CONFIGS[1]["batch"] and (H, W) == (CONFIGS[1]["H"], CONFIGS[1]["W"]) and a.size == "full")
            if a.measure_traffic and not a.roofline_only:
                traffic, note = traffic_measured(dom, extra)
            elif default_workload:
                traffic, note = traffic_committed(dom)
            else:      # the committed PMC passes were taken on the default workload: another launch population, not comparable
                traffic, note = None, "committed PMC passes cover the default workload (config 1) only; use --measure-traffic"
            fwd_flop = UNET_FLOP_PER_SAMPLE_64x48 * n * (h * w / 3072.0) if a.size == "full" and (h, w) == (64, 48) else None
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_TFLOPS, 4),
                        "traffic": traffic, "traffic_note": note, "kernel": dom, "tile_cfgs": d["cfgs"],
                        "avg_launch_ms": round(d["ms"] / d["launches"], 5), "flop_per_launch": d["flop"] / d["launches"],
                        "launches_profiled": d["launches"], "share_of_igemm_time": round(d["ms"] / prof[0], 4) if prof[0] > 0 else None,
                        "igemm_all_tflops": round(prof[1] / prof[0] / 1e9, 2) if prof[0] > 0 else None,
                        "per_symbol": {k: {"tflops": round(v["flop"] / v["ms"] / 1e9, 1), "launches": v["launches"], "avg_ms": round(v["ms"] / v["launches"], 5),
                                           "total_ms": round(v["ms"], 3)} for k, v in sorted(sym.items(), key=lambda kv: -kv[1]["ms"])},
                        "unet_forward_ms": round(whole_ms, 3),
                        "unet_forward_tflops": round(fwd_flop / whole_ms / 1e9, 2) if fwd_flop else None,
                        "unet_forward_lanes": lanes_n, "unet_forward_lanes_ms": round(lanes_ms, 3) if lanes_ms else None,
                        "unet_forward_lanes_tflops": round(fwd_flop / lanes_ms / 1e9, 2) if (fwd_flop and lanes_ms) else None,
                        "clock": clock, "hbm_kernels": hbm_kernels() if default_workload else None,
                        "splitk_reduce": "charged to the symbol of the kernel that needed it (HIP events close after the reduce pass)",
                        "whole_path_frac": round(images_per_s / world * flop_img / 1e12 / PEAK_F16_TFLOPS, 4) if flop_img and a.steps else None}
            # calibration of `peak` on THIS box: what the vendor's own GEMM sustains on its best shape (a comparison point measured live -- torch.matmul is
            # hipBLASLt; nothing on the product path uses it).  `peak` stays the nominal 2.5 PFLOP/s at 2.4 GHz; the shader clock under this load is in `clock`.
            roofline["vendor_gemm"] = vendor_gemm_calibration(dev, ach)
    cpu = None
    if want_cpu:
        try:
            cpu = cpu_baseline(sds, cfgs, a.cpu_runs, evals, scheduler, a.size)
        except Exception as e:  # the baseline is reported, never required
            cpu = {"value": None, "unit": "images/s", "cores": host_cores(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        line = {
            "metric": "try-on images/sec @%dx%d, %d %s steps" % (H, W, steps_inf, scheduler.upper()), "value": round(images_per_s, 4), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1000.0, 2) if a.steps else None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic" if not stub else "stub (--stub-step: CPU stand-in compute over gloo; a rehearsal of launch / sharding / collective / timing, NOT a measurement)",
            "config": {"workload": "%s, %d %s steps (%d UNet evals, CFG 7.5), %dx%d, EMASC skips on, fp16 storage / fp32 accumulate, %s-size random-init "
                                   "checkpoint" % (cfg["name"], steps_inf, scheduler.upper(), evals, H, W, a.size),
                       "baseline_config_index": a.config, "batch_per_gpu": B, "global_batch": global_B, "producers_in_step": bool(producers),
                       "parallelism": "dp%d (contiguous row shards of the global batch + RCCL all-gather of uint8 images)" % world,
                       "hipgraph": not a.no_graph, "unet_lanes": (lib.ladi_tryon_lanes(pipe._tryon) if (pipe is not None and pipe._tryon) else None),
                       "rccl_ranks_seen": rccl["ranks_seen"], "rccl_version": rccl["version"], "collective_backend": rccl["backend"]},
            "stage_ms_rank0": stage_ms, "model_build_s": round(t_build, 1),
            "with_d2h_pil_images_per_s": tail["images_per_s"] if tail else None, "d2h_pil_tail": tail,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
