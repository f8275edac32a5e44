"""Experiment result store: one CSV per argument set, named by a hash of the arguments, plus a manifest.

On-disk format of the reference's examples/common/results.py:8-59, so result folders written by either
harness can be mixed: `<md5 of the key-sorted JSON of the argument dict, minus 'func'>.csv` holds one row
per evaluated coreset size with every argument repeated as a column next to the result columns
(`Ms, csize, err, cput, ...`), and `manifest.csv` gets one `hash: {args}` line per saved run.
`load_matching` filters rows by the argument columns a file actually has.
"""
import hashlib
import json
import os

import pandas as pd

MANIFEST = "manifest.csv"


def _argdict(arguments):
    d = dict(vars(arguments)) if not isinstance(arguments, dict) else dict(arguments)
    d.pop("func", None)  # sub-command callbacks are not part of the experiment identity
    return d


def hash_namespace(arguments):
    return hashlib.md5(json.dumps(_argdict(arguments), sort_keys=True).encode("utf-8")).hexdigest()


class ResultStore:
    def __init__(self, folder="results/", manifest=MANIFEST):
        self.folder, self.manifest = folder, manifest

    def path(self, arguments):
        return os.path.join(self.folder, hash_namespace(arguments) + ".csv")

    def has(self, arguments):
        return os.path.exists(self.path(arguments))

    def put(self, arguments, **columns):
        args = _argdict(arguments)
        os.makedirs(self.folder, exist_ok=True)
        table = pd.DataFrame({**args, **columns})   # scalars broadcast against the result columns
        table.to_csv(self.path(arguments), index=False)
        with open(os.path.join(self.folder, self.manifest), "a") as f:
            f.write("%s: %s\n" % (hash_namespace(arguments), args))
        return table

    def files(self):
        if not os.path.isdir(self.folder):
            return []
        return sorted(os.path.join(self.folder, fn) for fn in os.listdir(self.folder)
                      if fn.endswith(".csv") and fn != self.manifest)

    def select(self, match):
        """Rows of every stored run whose argument columns equal `match` (keys a file lacks are ignored)."""
        parts = []
        for fn in self.files():
            t = pd.read_csv(fn)
            keep = pd.Series(True, index=t.index)
            for key, val in match.items():
                if key in t.columns and val is not None:
                    keep &= (t[key] == val)
            if keep.any():
                parts.append(t[keep])
        return pd.concat(parts, ignore_index=True) if parts else None


def check_exists(arguments, results_folder="results/"):
    return ResultStore(results_folder).has(arguments)


def save(arguments, results_folder="results/", log_file=MANIFEST, **kwargs):
    ResultStore(results_folder, log_file).put(arguments, **kwargs)


def load_matching(match_dict, results_folder="results/", log_file=MANIFEST):
    return ResultStore(results_folder, log_file).select(match_dict)
