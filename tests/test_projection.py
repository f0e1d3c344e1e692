"""Row f4: symmetric k-mer proportions for the projection, against a golden of the reference's own
computeKmers(sym=True) -> scrubMirrors -> flattenKmerMap(prop=True)."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLD

G = json.load(open(os.path.join(GOLD, "projection_counts.json")))


def test_mirror_folding_and_proportions_cpu():
    from frisk_amd.projection import feature_keys, proportions_from_forward
    from oracle import frisk_oracle_np as N
    kmin, kmax = G["pcaMin"], G["pcaMax"]
    assert feature_keys(kmin, kmax) == G["keys"]
    for w in G["windows"]:
        fwd, _ = N.forward_counts(N.Encoded(w["seq"]), kmin, kmax)
        vec = proportions_from_forward(fwd, kmin, kmax)
        assert vec.tolist() == w["vector"]            # integer counts + one correctly rounded division: bit-exact


@pytest.mark.gpu
def test_symmetric_counts_gpu():
    from frisk_amd.projection import symmetricCounts
    labels, counts = symmetricCounts([(w["label"], w["seq"]) for w in G["windows"]], G["pcaMin"], G["pcaMax"])
    assert labels[:, 0].tolist() == [w["label"] for w in G["windows"]]
    assert counts.shape == (len(G["windows"]), len(G["keys"]))
    for row, w in zip(counts, G["windows"]):
        assert row.tolist() == w["vector"]
    with pytest.raises(ValueError):
        symmetricCounts([("bad", "N" * 300 + "ACGT" * 50)], 1, 3)
