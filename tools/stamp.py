"""Identity of the kernel sources a measurement was taken with: SHA-256 over csrc/*.hip, csrc/*.h, include/bcx.h and
the Makefile (compiler flags).  profiles/*.json carry it; bench.py refuses measured constants (HBM traffic per
launch) whose stamp differs from the tree it runs from.  (The built .so is not hashed: a rebuild of the same
sources need not be byte-identical.)"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(files):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def source_digest():
    pkg = os.path.join(ROOT, "bayesian-coresets_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")))
    files += [os.path.join(ROOT, "include", "bcx.h"), os.path.join(pkg, "Makefile")]
    return _digest(files)


def kernel_digest(kind):
    """Identity of ONE kernel's sources: what a per-launch HBM traffic figure of that kernel depends on -- the kernel's own
    file, the headers it includes and the compiler flags -- so that an edit elsewhere in csrc/ does not void it.
    kind: "scan" (csrc/scan.hip) or "proj" (csrc/proj.hip)."""
    pkg = os.path.join(ROOT, "bayesian-coresets_amd")
    names = {"scan": ["scan.hip", "scan_core.h", "dev_util.h", "bcx_internal.h"],
             "proj": ["proj.hip", "proj_math.h", "moments_quad.h", "dev_util.h", "bcx_internal.h"]}[kind]
    return _digest([os.path.join(pkg, "csrc", n) for n in names] + [os.path.join(ROOT, "include", "bcx.h"), os.path.join(pkg, "Makefile")])


if __name__ == "__main__":
    import sys
    print(kernel_digest(sys.argv[1]) if len(sys.argv) > 1 else source_digest())
