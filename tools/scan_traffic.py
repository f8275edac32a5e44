"""(The file carries three stamps: `_stamp` = all kernel sources at the time of the pass, `_stamp_scan` / `_stamp_proj` = the sources
the scan / projection kernel is built from; bench.py checks the kernel's own.)
Builds profiles/scan_traffic.json entries from rocprofv3 --pmc FETCH_SIZE result databases (scan kernel; also the
projection kernel of the c5 line: key proj_colsum_linreg_n<N>_d<D>_s<S>).
usage: python tools/scan_traffic.py out.json key=path/to/results.db [key=db ...]
HBM bytes per scan launch = mean FETCH_SIZE (KB) over the scan_kernel dispatches x 1024 x 2 (gfx950 counts a 128-byte
request as 64 bytes for wide streaming reads: MI355X_MICROARCH.md, HBM section).  The file is stamped with the kernel
source digest (tools/stamp.py); bench.py ignores it when the stamp is not the tree's."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.stamp import source_digest, kernel_digest


def fetch_bytes(db, key):
    # the key ends in the storage type: only that instantiation of the scan kernel counts (a bench run may also hold
    # the fp64-rows scan of its exact-mode leg)
    if key.startswith("proj_colsum_linreg_"):
        inst = "proj_kernel<2, 1,"       # full-data COLSUM launches of the linear-regression family (bench.py --config c5)
    else:
        inst = {"float32": "scan_kernel<float", "float64": "scan_kernel<double", "float16": "scan_kernel<half_t"}[key.rsplit("_", 1)[1]]
    c = sqlite3.connect(db)
    # rocpd stores one row per (dispatch, counter instance): sum the instances of a dispatch first
    per = c.execute("select dispatch_id, sum(counter_value) from pmc_events where name like ? and counter_name = 'FETCH_SIZE' "
                    "group by dispatch_id", ("%" + inst + "%",)).fetchall()
    vals = sorted(v for _, v in per if v and v > 0)
    if not vals:
        raise SystemExit("no %s FETCH_SIZE rows in %s" % (inst, db))
    # launches issued after the state machine stopped read nothing: keep the ones within 2x of the median
    med = vals[len(vals) // 2]
    keep = [v for v in vals if v > 0.5 * med]
    return sum(keep) / len(keep) * 1024.0 * 2.0, len(keep)


def main():
    out = sys.argv[1]
    data = {"_comment": "HBM bytes per scan_kernel launch from rocprofv3 --pmc FETCH_SIZE (separate pass): mean KB x 1024 x 2 (gfx950 "
                        "half-count correction). Key = alg_nlocal_d_dtype as built by bench.py (proj_colsum_linreg_n_d_s: the projection kernel of the c5 line).", "_stamp": source_digest(), "_stamp_scan": kernel_digest("scan"), "_stamp_proj": kernel_digest("proj"), "_sources": {}}
    if os.path.exists(out):
        old = json.load(open(out))
        if old.get("_stamp") == data["_stamp"]:
            data = old
    for item in sys.argv[2:]:
        key, db = item.split("=", 1)
        b, n = fetch_bytes(db, key)
        data[key] = b
        data["_sources"][key] = "%s: %d dispatches" % (os.path.basename(os.path.dirname(db)), n)
        print(key, b, n)
    json.dump(data, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
