"""tangram_amd/host_rng.py: the mapper's initial logits come from NumPy's global legacy generator like the reference's
(`np.random.normal(0, 1, (n_cells, n_spots))`, tangram/mapping_optimizer.py:147-157), through the threaded C helper:
the values AND the generator state afterwards must be NumPy's, bit for bit."""
import numpy as np
import pytest

from tangram_amd import host_rng


@pytest.fixture(scope="module")
def helper():
    if host_rng.build() is None:
        pytest.skip("no C compiler for the host helper")
    old = host_rng.MIN_VALUES
    host_rng.MIN_VALUES = 1
    assert host_rng._load() is not None
    yield
    host_rng.MIN_VALUES = old


def _states_equal(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize("seed", [1, 42, 2 ** 31 + 7])
@pytest.mark.parametrize("shape", [(1,), (2,), (3, 3), (1, 65537), (7, 9852), (300, 4001), (5000, 997)])
@pytest.mark.parametrize("cached", [0, 1])
def test_values_and_generator_state_are_numpys(helper, seed, shape, cached):
    def prepare():
        np.random.seed(seed)
        for _ in range(cached):                      # an odd number of earlier draws leaves a cached second value of a pair
            np.random.normal()
    prepare()
    ref = np.random.normal(0, 1, shape).astype(np.float32)
    st_ref = np.random.get_state()
    prepare()
    got = host_rng.legacy_normal_f32(shape)
    assert got.shape == tuple(shape) and got.dtype == np.float32
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    assert _states_equal(st_ref, np.random.get_state())


def test_discarded_draws_advance_the_generator_alike(helper):
    np.random.seed(9)
    np.random.normal(0, 1, (123, 457))
    a = np.random.normal(0, 1, 7)
    np.random.seed(9)
    assert host_rng.legacy_normal_f32((123, 457), discard=True) is None
    b = np.random.normal(0, 1, 7)
    assert np.array_equal(a, b)


def test_mapper_initialisation_is_the_references(helper):
    """The same draw order as `Mapper` / `MapperConstrained` (one draw; a discarded draw, M, then F): oracle's restatement."""
    from oracle import tangram_oracle as orc
    C, V = 60, 2000
    np.random.seed(5)
    M = host_rng.legacy_normal_f32((C, V))
    assert np.array_equal(M, orc.reference_init_M(C, V, 5))
    np.random.seed(5)
    host_rng.legacy_normal_f32((C, V), discard=True)
    M2 = host_rng.legacy_normal_f32((C, V))
    F2 = np.random.normal(0, 1, C).astype(np.float32)
    Mo, Fo = orc.reference_init_MF_constrained(C, V, 5)
    assert np.array_equal(M2, Mo) and np.array_equal(F2, Fo)


def test_many_rounds_and_ragged_chunks(helper, tmp_path):
    """The chunk / round bookkeeping with tiny chunks (1 024 candidates) and two chunks per round: hundreds of rounds, the stream
    ending in the middle of a chunk -- same bits, same generator state."""
    import ctypes as ct
    import shutil
    import subprocess
    so = str(tmp_path / "rng_tiny.so")
    subprocess.run([shutil.which("gcc"), "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-DCHUNK_CAND=1024",
                    "-DROUND_CHUNKS=2", host_rng.SRC, "-o", so, "-lm"], check=True)
    lib = ct.CDLL(so)
    lib.tg_legacy_normal_f32.restype = ct.c_int
    lib.tg_legacy_normal_f32.argtypes = host_rng._load().tg_legacy_normal_f32.argtypes
    for seed, n, cached in ((3, 1, 0), (3, 2047, 1), (8, 100001, 0), (8, 262144, 1), (11, 777777, 0)):
        np.random.seed(seed)
        for _ in range(cached):
            np.random.normal()
        name, key, pos, hg, cg = np.random.get_state()
        ref = np.random.normal(0, 1, n).astype(np.float32)
        st_ref = np.random.get_state()
        key = key.copy()
        c_pos, c_has, c_g = ct.c_int(int(pos)), ct.c_int(int(hg)), ct.c_double(float(cg))
        out = np.empty(n, np.float32)
        assert lib.tg_legacy_normal_f32(key.ctypes.data, ct.byref(c_pos), ct.byref(c_has), ct.byref(c_g), out.ctypes.data, n, 5) == 0
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (seed, n, cached)
        assert np.array_equal(key, st_ref[1]) and (c_pos.value, c_has.value, c_g.value) == st_ref[2:], (seed, n, cached)
