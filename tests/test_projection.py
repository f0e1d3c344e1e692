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


@pytest.mark.gpu
def test_symmetric_counts_of_a_region_longer_than_the_lds_counters():
    """An anomalous region of > 65 535 bases goes through the global-memory scan path; its vector must equal the one
    computed from the numpy oracle's forward counts."""
    import numpy as np
    from frisk_amd.projection import proportions_from_forward, symmetricCounts
    from oracle import frisk_oracle_np as N
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), size=n, p=[0.3, 0.2, 0.2, 0.3])) for n in (90000, 3000)]
    labels, counts = symmetricCounts([("big", seqs[0]), ("small", seqs[1])], 1, 4)
    for row, s in zip(counts, seqs):
        fwd, _ = N.forward_counts(N.Encoded(s), 1, 4)
        assert np.array_equal(row, proportions_from_forward(fwd, 1, 4))
