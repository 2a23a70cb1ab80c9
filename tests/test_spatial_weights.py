"""tangram_amd/spatial_weights.py against the UNMODIFIED reference function (tangram/spatial_weights.py:5-29): the fixtures in
tests/golden/spatial_weights.npz are its outputs (oracle/gen_spatial_golden.py), on a graph whose connectivity and distance
patterns coincide (squidpy's output) and on one where they do not (the reference pairs neighbours and weights positionally)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle.spatial_weights_oracle import spatial_weights_oracle
from tangram_amd.spatial_weights import spatial_weights, one_hot_encoding


class _Ad:
    def __init__(self, conn, dist):
        self.obsp = {"spatial_connectivities": conn, "spatial_distances": dist}


@pytest.fixture(scope="module")
def golden(golden_dir):
    import os
    return np.load(os.path.join(golden_dir, "spatial_weights.npz"))


@pytest.mark.parametrize("name", ["match", "mismatch"])
@pytest.mark.parametrize("standardized", [True, False])
@pytest.mark.parametrize("self_inclusion", [True, False])
def test_spatial_weights_equal_the_reference(golden, name, standardized, self_inclusion):
    conn, dist = golden[name + "_conn"], golden[name + "_dist"]
    want = golden[f"{name}_std{int(standardized)}_self{int(self_inclusion)}"]
    # the oracle restatement is pinned to the reference's output ...
    np.testing.assert_allclose(spatial_weights_oracle(conn, dist, standardized, self_inclusion), want, rtol=0, atol=1e-15)
    # ... and the product (CSR, float32) equals it to float32 rounding; the caller's matrices are left untouched
    cs, ds = sp.csr_matrix(conn), sp.csr_matrix(dist)
    d_before = ds.copy()
    got = spatial_weights(_Ad(cs, ds), standardized, self_inclusion)
    assert sp.issparse(got) and got.dtype == np.float32
    np.testing.assert_allclose(got.toarray(), want, rtol=1e-6, atol=1e-7)
    assert (ds != d_before).nnz == 0


def test_spatial_weights_missing_graph_raises():
    class A:
        obsp = {}
    with pytest.raises(ValueError, match="Missing spatial neighborhood parameters"):
        spatial_weights(A(), True, True)


def test_one_hot_encoding_order_of_first_appearance():
    E, cols = one_hot_encoding(["b", "a", "b", "c"])
    assert cols == ["b", "a", "c"]
    np.testing.assert_array_equal(E, np.array([[1, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]], np.float32))
