"""Identity of the kernel sources a measurement was taken with: SHA-256 over csrc/*.hip, csrc/*.h, include/bcx.h and
the Makefile (compiler flags).  profiles/*.json carry it; bench.py refuses measured constants (HBM traffic per
launch) whose stamp differs from the tree it runs from.  (The built .so is not hashed: a rebuild of the same
sources need not be byte-identical.)"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_digest():
    pkg = os.path.join(ROOT, "bayesian-coresets_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")))
    files += [os.path.join(ROOT, "include", "bcx.h"), os.path.join(pkg, "Makefile")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_digest())
