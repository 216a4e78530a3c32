"""Occupancy audit of the GEMM kernel family (CPU only: hipcc's register report + the LDS footprint each launcher requests).
For every tile configuration: VGPRs of its kernel symbol, waves per SIMD the registers allow (allocation granule 8, 512-entry file),
LDS bytes per workgroup, workgroups per CU the LDS admits (160 KB), and the waves per SIMD that results -- a configuration whose registers
allow more waves than its LDS admits is a candidate for a shallower ring (what `linear_xs_kernel<..., NST = 2>` did in round 4).
    python tools/occupancy_audit.py > profiles/r04_occupancy_audit.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ladi_vton_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def report(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src)], capture_output=True, text=True)
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"^void ", "", name).split("(")[0]
            out[name] = {}
        m = re.search(r"\s(VGPRs|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


def waves_by_regs(v):
    alloc = (v + 7) // 8 * 8
    return min(8, 512 // max(alloc, 8))


def main():
    rows = []
    # ring kernels: X(base, WQ, WP, TQ, TP, BK, NST, OCC, ILV) -> LDS = max(ring, epilogue patches)
    tiles = re.findall(r"X\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)", open(os.path.join(CSRC, "igemm_tiles.h")).read())
    regs = {}
    for g in "abcdefghi":
        regs.update(report("igemm_inst_%s.hip" % g))
    for t in tiles:
        base, WQ, WP, TQ, TP, BK, NST, OCC, ILV = map(int, t)
        sym = "igemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d>" % (WQ, WP, TQ, TP, BK, NST, OCC, ILV)
        BQ, BP = WQ * TQ * 32, WP * TP * 32
        ring = NST * (BQ + BP) * BK * 2
        epi = WQ * WP * 32 * (TQ * 32 + 4) * 2 + WQ * TQ * 32 * 4
        rows.append((sym, "cfg base %d: %dx%d BK%d ring %d" % (base, BQ, BP, BK, NST), WQ * WP, max(ring, epi), regs.get(sym, {})))
    for src, fam in (("igemm_halo.hip", "igemm_halo_kernel"), ("igemm8.hip", "igemm8_kernel"), ("igemm_lc.hip", "igemm_lc_kernel"), ("linear_xs.hip", "linear_xs_kernel"),
                     ("attention.hip", "flash_attn")):
        for sym, r in report(src).items():
            if not sym.startswith(fam):
                continue
            a = [int(x) for x in re.findall(r"-?\d+", sym.split("<", 1)[1])] if "<" in sym else []
            waves, lds, what = None, None, ""
            if fam == "igemm_halo_kernel":
                TQ, TP, NXB, NSTW, WPN, WMAX = a
                BQ, BP, RPP = 64 * TQ, 32 * WPN * TP, 16 * WPN
                RQ = (BQ + RPP - 1) // RPP
                XROWS = (BP + 2 * WMAX + 2 + RPP - 1) // RPP * RPP
                lds = (NSTW * RQ * RPP * 64 + NXB * XROWS * 64) * 2 + 128
                waves, what = 2 * WPN, "%dx%d, %d weight slots, %d halo buffers (rows <= %d)" % (BQ, BP, NSTW, NXB, WMAX)
            elif fam == "linear_xs_kernel":
                KH, PB, MODE, PRE, NST = a
                lds = NST * 20480 + 4 * (4096 if MODE == 1 else 2560) + 1280 + (KH * 2560 if PRE == 2 else 0)
                waves, what = 4, "K = %d, %d pixels per wave, mode %d, prologue %d, %d weight slots" % (320 * KH, 32 * PB, MODE, PRE, NST)
            elif fam == "igemm8_kernel":
                TQ, TP = a[0], a[1]
                lds = 2 * (64 * TQ + 128 * TP) * 64 * 2
                waves, what = 8, "%dx%d, 2 K-tile buffers" % (64 * TQ, 128 * TP)
            elif fam == "igemm_lc_kernel":
                WQ, WP, TQ, TP, NL, NST = a
                lds = NST * (WQ * TQ * 32 + WP * TP * 32) * 64 * 2
                waves, what = WQ * WP + NL, "%dx%d, %d-deep ring, %d loader waves" % (WQ * TQ * 32, WP * TP * 32, NST, NL)
            rows.append((sym, what, waves, lds, r))
    print("%-52s %-58s %5s %9s %7s | waves/SIMD: regs  LDS  -> gap" % ("kernel", "what", "VGPRs", "LDS B/WG", "WG/CU"))
    for sym, what, waves, lds, r in rows:
        v = r.get("VGPRs")
        if v is None:
            continue
        wr = waves_by_regs(v)
        if lds:
            wg = min(160 * 1024 // lds, 32 // waves if waves else 8)
            wl = wg * waves / 4.0
            flag = "  <-- registers allow %d, LDS admits %.1f" % (wr, wl) if wr >= wl + 1 else ""
            print("%-52s %-58s %5d %9d %7d | %16d %5.1f%s" % (sym[:52], what[:58], v, lds, wg, wr, wl, flag))
        else:
            print("%-52s %-58s %5d %9s %7s | %16d" % (sym[:52], what[:58], v, "-", "-", wr))


if __name__ == "__main__":
    main()
