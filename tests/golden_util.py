"""Loading of the golden fixtures (tests/golden): inputs + outputs of the reference's own
functions, written by tools/make_golden.py."""
import glob
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUTS = os.path.join(GOLD, "inputs")


def case_names():
    # scan cases have a .json (rows) and a .npz (count tables); other goldens (thresholds, cli_surface) are .json only
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLD, "*.npz")))


class Case:
    def __init__(self, name):
        self.name = name
        with open(os.path.join(GOLD, name + ".json")) as fh:
            self.doc = json.load(fh)
        self.arrays = np.load(os.path.join(GOLD, name + ".npz"))
        a = self.doc["args"]
        self.m, self.k = a["minWordSize"], a["maxWordSize"]
        self.w, self.i = a["windowlen"], a["increment"]
        self.mask_host, self.scaffolds_all, self.rip = a["maskHost"], a["scaffoldsAll"], a["RIP"]
        self.host = os.path.join(INPUTS, self.doc["host"])
        self.query = os.path.join(INPUTS, self.doc["query"]) if self.doc["query"] else None
        self.rows = self.doc["rows"]
        self.genome_meta = self.doc["genome_meta"]
        self.genome_counts = self.arrays["genome_counts"]
        self.window_counts = self.arrays["window_counts"]
        # IvomBuild's normalised distributions per row, dense over 4^k max-mers (stored for the small-K cases only)
        self.window_ivom = self.arrays["window_ivom"] if "window_ivom" in self.arrays.files else None
        self.genome_ivom = self.arrays["genome_ivom"] if "genome_ivom" in self.arrays.files else None

    @property
    def rip_on(self):
        return self.rip and self.m <= 2


def same_float(a, b):
    """bit-for-bit equality that treats NaN == NaN."""
    return (a != a and b != b) or a == b
