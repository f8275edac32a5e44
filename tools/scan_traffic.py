"""Builds profiles/scan_traffic.json entries from rocprofv3 --pmc FETCH_SIZE result databases.
usage: python tools/scan_traffic.py out.json key=path/to/results.db [key=db ...]
HBM bytes per scan launch = mean FETCH_SIZE (KB) over the scan_kernel dispatches x 1024 x 2 (gfx950 counts a 128-byte
request as 64 bytes for wide streaming reads: MI355X_MICROARCH.md, HBM section).  The file is stamped with the kernel
source digest (tools/stamp.py); bench.py ignores it when the stamp is not the tree's."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.stamp import source_digest


def fetch_bytes(db):
    c = sqlite3.connect(db)
    rows = c.execute("select avg(counter_value), count(*) from pmc_events where name like '%scan_kernel%' and counter_name = 'FETCH_SIZE' "
                     "and counter_value > 0").fetchall()
    # rocpd stores one row per (dispatch, counter instance); sum the instances of a dispatch first when there are several
    per = c.execute("select dispatch_id, sum(counter_value) from pmc_events where name like '%scan_kernel%' and counter_name = 'FETCH_SIZE' "
                    "group by dispatch_id").fetchall()
    vals = sorted(v for _, v in per if v and v > 0)
    if not vals:
        raise SystemExit("no scan_kernel FETCH_SIZE rows in " + db)
    # launches issued after the state machine stopped read nothing: keep the ones within 2x of the median
    med = vals[len(vals) // 2]
    keep = [v for v in vals if v > 0.5 * med]
    return sum(keep) / len(keep) * 1024.0 * 2.0, len(keep)


def main():
    out = sys.argv[1]
    data = {"_comment": "HBM bytes per scan_kernel launch from rocprofv3 --pmc FETCH_SIZE (separate pass): mean KB x 1024 x 2 (gfx950 "
                        "half-count correction). Key = alg_nlocal_d_dtype as built by bench.py.", "_stamp": source_digest(), "_sources": {}}
    if os.path.exists(out):
        old = json.load(open(out))
        if old.get("_stamp") == data["_stamp"]:
            data = old
    for item in sys.argv[2:]:
        key, db = item.split("=", 1)
        b, n = fetch_bytes(db)
        data[key] = b
        data["_sources"][key] = "%s: %d dispatches" % (os.path.basename(os.path.dirname(db)), n)
        print(key, b, n)
    json.dump(data, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
