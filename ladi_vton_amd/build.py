"""Build libladi_native.so (hand-written HIP for gfx950) in-tree with hipcc.

Usage: python -m ladi_vton_amd.build [--force]
The shared library lands next to this file so that it travels with the repo snapshot to the GPU box.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libladi_native.so")
SOURCES = ["igemm.hip"] + ["igemm_inst_%s.hip" % g for g in "abcdefghi"] + ["igemm8.hip", "igemm_lc.hip", "igemm_halo.hip"] + ["igemm_halo_inst_%s.hip" % g for g in "abcdefghij"] + ["linear_xs.hip", "xf_fused.hip", "norm.hip", "attention.hip", "elementwise.hip", "f32path.hip", "runtime_core.cpp", "runtime_f32.cpp", "runtime_unet.cpp",
           "runtime_vae.cpp", "runtime_text.cpp", "runtime_vision.cpp", "runtime_refine.cpp", "runtime_tps.cpp", "runtime_tryon.cpp", "capi.cpp"]
HEADERS = ["common.h", "kernels.h", "igemm_common.h", "igemm_kernel.h", "igemm_halo_kernel.h", "igemm_tiles.h", "runtime.h", os.path.join("..", "..", "include", "ladi_native.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile what changed and link.  Safe under `torchrun` (N ranks importing the package at once): the whole build runs under an
    exclusive flock on csrc/_obj/.lock, objects / library / stamps are written to temporary names and os.replace()d into place, so a
    concurrent rank either waits and then finds everything up to date, or dlopens a complete library -- never a half-written one."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest()

    def fresh():
        return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig

    if not force and fresh():
        return LIB
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():          # another process built it while this one waited for the lock
                return LIB
            return _build_locked(force, verbose, stamp, dig)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _write_atomic(path, text):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)


def _build_locked(force, verbose, stamp, dig):
    hipcc = _hipcc()

    def obj_hash(src, deps):
        h = hashlib.sha256()
        for f in [os.path.join(CSRC, src)] + sorted(deps):
            with open(f, "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(FLAGS).encode())
        return h.hexdigest()

    def read_deps(depfile):
        """headers of THIS repo a translation unit included (hipcc -MD depfile, make syntax); system / ROCm headers are not tracked"""
        try:
            txt = open(depfile).read()
        except OSError:
            return None
        root = os.path.dirname(HERE)
        deps = set()
        for tok in txt.split(":", 1)[-1].replace("\\\n", " ").split():
            ap = os.path.abspath(tok)
            if ap.startswith(root + os.sep) and ap.endswith(".h") and os.path.exists(ap):
                deps.add(ap)
        return deps

    def compile_one(src):
        # per-object stamp = sha256(source + the repo headers it actually includes (depfile of its last compile) + flags): a header edit
        # recompiles only the units that include it (round 6: every header used to be hashed into every unit -- ten minutes per edit)
        obj = os.path.join(OBJ, src + ".o")
        ostamp, odeps = obj + ".stamp", obj + ".d"
        deps = read_deps(odeps)
        if not force and deps is not None and os.path.exists(obj) and os.path.exists(ostamp):
            try:
                if open(ostamp).read() == obj_hash(src, deps):
                    return obj
            except OSError:
                pass
        tmp = "%s.tmp%d.o" % (obj, os.getpid())
        tmpd = "%s.tmp%d.d" % (obj, os.getpid())
        cmd = [hipcc] + FLAGS + ["-MD", "-MF", tmpd, "-c", os.path.join(CSRC, src), "-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        os.replace(tmp, obj)
        os.replace(tmpd, odeps)
        _write_atomic(ostamp, obj_hash(src, read_deps(odeps) or set()))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp_lib = "%s.tmp%d" % (LIB, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp_lib, LIB)
    _write_atomic(stamp, dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
