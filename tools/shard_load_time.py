#!/usr/bin/env python3
"""What one rank of an 8-rank job spends loading its tiles of the whole C5-shaped assembly (3.29 Gb FASTA, GPU box): parsing the
file (frisk_fasta_load_shard, 1/8 of the host's parser threads) against copying its tiles' bytes through the seek index
(frisk_fasta_load_shard_indexed).  Also times writing the index.  usage: shard_load_time.py [workdir]   (one JSON line per step;
the FASTA is the one tools/e2e_cli.py C5 writes)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frisk_amd import Engine  # noqa: E402
from frisk_amd.fasta import writeFastaIndex  # noqa: E402

work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/frisk_e2e"
fa = os.path.join(work, "C5.fa")
if not os.path.exists(fa):
    sys.exit("run tools/e2e_cli.py C5 %s first" % work)
idx = fa + ".frisk.fai"
t0 = time.time()
n = writeFastaIndex(fa, idx)
print(json.dumps({"step": "write index", "records": n, "s": round(time.time() - t0, 3), "index_bytes": os.path.getsize(idx)}), flush=True)
world = 8
with Engine(1, 8) as e:
    for rank in (0, 3, 7):
        out = {}
        for label, index in (("parse", None), ("indexed", idx), ("parse again", None), ("indexed again", idx)):
            t0 = time.time()
            names, cc = e.load_fasta_shard(fa, 5000, 1000, rank, world, index=index)
            out[label] = round(time.time() - t0, 3)
            assert (e.shard_index is not None) == (index is not None)
            words = [int(np.bitwise_xor.reduce(x)) for x in e.export_packed()]
            if "words" in out:
                assert out["words"] == words and out["cand"] == list(cc)
            out["words"], out["cand"] = words, list(cc)
        out.pop("words")
        print(json.dumps({"step": "rank %d of %d" % (rank, world), "resident_bases": int(e.padded_len), **out}), flush=True)
