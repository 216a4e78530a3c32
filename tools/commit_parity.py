"""Copy the parity record of the last full `pytest -m gpu` run (gpurun_out/parity_r06.json, written by tests/util.py record_parity) to
profiles/r06_parity.json -- only if it was taken on the binary the current sources build (library digest) and holds every key the
documents cite (tests/test_cpu.py PARITY_KEYS_CITED).  python tools/commit_parity.py [--allow-stale-digest]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tests import util as U
    from tests.test_cpu import PARITY_KEYS_CITED
    src = os.path.join(ROOT, "gpurun_out", U.PARITY_NAME)
    blob = json.load(open(src))
    dig = U.library_digest()
    if blob.get("_library_digest") != dig and "--allow-stale-digest" not in sys.argv:
        sys.exit("parity record is from library %s..., the sources build %s...: re-run the GPU suite" % (str(blob.get("_library_digest"))[:8], dig[:8]))
    missing = [k for k in PARITY_KEYS_CITED if k not in blob]
    if missing:
        sys.exit("parity record lacks cited keys (partial run?): %s" % missing)
    shutil.copyfile(src, os.path.join(ROOT, "profiles", "r06_parity.json"))
    print("profiles/r06_parity.json <- %d keys, library %s" % (len(blob), dig[:8]))


if __name__ == "__main__":
    main()
