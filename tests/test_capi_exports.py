"""The C-ABI library loads and exports every symbol declared in include/tangram_hip.h (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "tangram_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z_]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    from tangram_amd import _capi
    assert set(_capi.EXPORTS) == set(_declared())


def test_hip_library_exports_every_declared_symbol():
    from tangram_amd import _build
    path = _build.build()
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.tg_abi_version.restype = ctypes.c_int
    from tangram_amd import _capi
    assert lib.tg_abi_version() == _capi.TG_ABI_VERSION == 6


def test_argument_errors_are_reported_not_aborted():
    """tg_query_sizes validates its configuration on the host (no GPU involved)."""
    from tangram_amd import _build, _capi
    lib = _capi._declare(ctypes.CDLL(_build.build()))
    cfg = _capi.TgConfig()
    sizes = _capi.TgSizes()
    assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sizes)) == -1          # abi_version 0
    cfg.abi_version = _capi.TG_ABI_VERSION
    cfg.n_cells, cfg.n_genes, cfg.n_spots = 10, 5, 7
    cfg.lambda_g1 = 0.0
    assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sizes)) == -1          # lambda_g1 cannot be 0
    assert b"lambda_g1" in lib.tg_last_error()
    cfg.lambda_g1 = 1.0
    assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sizes)) == 0
    assert sizes.m_pitch == 64 and sizes.state_bytes > 0 and sizes.workspace_bytes > 0


def test_configuration_errors_of_round_three_fields():
    """bwd_tile and the shard geometry of runs with spatial terms are validated on the host (tg_query_sizes, no GPU involved)."""
    from tangram_amd import _build, _capi
    lib = _capi._declare(ctypes.CDLL(_build.build()))
    sizes = _capi.TgSizes()

    def cfg(**kw):
        c = _capi.TgConfig()
        c.abi_version = _capi.TG_ABI_VERSION
        c.n_cells, c.n_genes, c.n_spots, c.lambda_g1 = 5000, 40, 1000, 1.0
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    assert lib.tg_query_sizes(ctypes.byref(cfg()), ctypes.byref(sizes)) == 0
    assert lib.tg_query_sizes(ctypes.byref(cfg(bwd_tile=64)), ctypes.byref(sizes)) == -1 and b"bwd_tile" in lib.tg_last_error()
    for t in (128, 256):
        assert lib.tg_query_sizes(ctypes.byref(cfg(bwd_tile=t)), ctypes.byref(sizes)) == 0
    # spatial terms on a spot shard: blocks of ceil(V_total / ranks) spots, the offset names the block
    sp = dict(lambda_neighborhood_g1=0.5, nnz_w=7000, n_spots_total=1000, n_ranks=3)      # blocks of 334, 334, 332
    assert lib.tg_query_sizes(ctypes.byref(cfg(n_spots=334, spot_offset=334, **sp)), ctypes.byref(sizes)) == 0
    assert lib.tg_query_sizes(ctypes.byref(cfg(n_spots=332, spot_offset=668, **sp)), ctypes.byref(sizes)) == 0
    full = sizes.workspace_bytes
    assert lib.tg_query_sizes(ctypes.byref(cfg(n_spots=333, spot_offset=333, **sp)), ctypes.byref(sizes)) == -1      # balanced partition: refused
    assert b"ceil" in lib.tg_last_error()
    assert lib.tg_query_sizes(ctypes.byref(cfg(n_spots=334, spot_offset=334, lambda_neighborhood_g1=0.5, nnz_w=7000, n_spots_total=1000)),
                              ctypes.byref(sizes)) == -1                                                           # n_ranks missing
    # the gathered matrices are part of the workspace: a shard with spatial terms needs more than the same shard without
    assert lib.tg_query_sizes(ctypes.byref(cfg(n_spots=334, n_spots_total=1000, n_ranks=3)), ctypes.byref(sizes)) == 0
    assert sizes.workspace_bytes < full


def test_product_path_has_no_cpu_fallback():
    import numpy as np
    from tangram_amd import _capi
    from tangram_amd.mapping_optimizer import Mapper
    assert not _capi.is_emulated()
    with pytest.raises(RuntimeError):
        Mapper(np.ones((4, 3), np.float32), np.ones((5, 3), np.float32), device="cpu")


@pytest.mark.parametrize("n_major,n_minor", [(235, 79), (79, 235), (16, 1), (17, 3), (64, 64), (1563, 391), (118, 5), (40, 40), (3, 9), (118, 40)])
def test_xcd_tile_map_is_a_bijection(n_major, n_minor):
    """Every tile of the backward grid is visited exactly once by the XCD-banded supertile order."""
    from tangram_amd import _build
    lib = ctypes.CDLL(_build.build())
    mj, mn = ctypes.c_int(), ctypes.c_int()
    for mode in (0, 1):
        grid = lib.tg_debug_tilemap(mode, n_major, n_minor, -1, None, None)
        seen = set()
        for b in range(grid):
            if lib.tg_debug_tilemap(mode, n_major, n_minor, b, ctypes.byref(mj), ctypes.byref(mn)):
                assert 0 <= mj.value < n_major and 0 <= mn.value < n_minor
                assert (mj.value, mn.value) not in seen
                seen.add((mj.value, mn.value))
        assert len(seen) == n_major * n_minor
        if mode == 1:      # band property: a workgroup's XCD (b % 8) owns a contiguous band of the major axis; dense enumeration
            q, r = divmod(n_major, 8)
            assert grid == 8 * ((n_major + 7) // 8) * n_minor
            for b in range(0, grid, 97):
                x = b % 8
                row0, rows = x * q + min(x, r), q + (1 if x < r else 0)
                ok = lib.tg_debug_tilemap(1, n_major, n_minor, b, ctypes.byref(mj), ctypes.byref(mn))
                assert bool(ok) == ((b >> 3) < rows * n_minor)
                if ok:
                    assert row0 <= mj.value < row0 + rows
            # the first workgroups of an XCD fill ONE supertile: <= 8 rows x <= 8 columns, down the major axis first
            for x in (0, 7):
                rows = q + (1 if x < r else 0)
                n_first = min(rows, 8) * min(n_minor, 8)
                local = []
                for j in range(n_first):
                    assert lib.tg_debug_tilemap(1, n_major, n_minor, 8 * j + x, ctypes.byref(mj), ctypes.byref(mn))
                    local.append((mj.value, mn.value))
                if local:
                    assert len({a for a, _ in local}) == min(rows, 8) and len({c for _, c in local}) == min(n_minor, 8)
                    assert local[1][0] == local[0][0] + 1 or rows == 1


@pytest.mark.parametrize("nvt,nkt,nsplit", [(79, 8, 4), (1, 1, 1), (3, 2, 5), (10, 3, 1), (391, 16, 2)])
def test_forward_grid_map_is_a_bijection(nvt, nkt, nsplit):
    from tangram_amd import _build
    lib = ctypes.CDLL(_build.build())
    a, b_, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    grid = lib.tg_debug_fwd_map(nvt, nkt, nsplit, -1, None, None, None)
    seen = set()
    for b in range(grid):
        if lib.tg_debug_fwd_map(nvt, nkt, nsplit, b, ctypes.byref(a), ctypes.byref(b_), ctypes.byref(c)):
            key = (a.value, b_.value, c.value)
            assert key not in seen and a.value < nvt and b_.value < nkt and c.value < nsplit
            seen.add(key)
    assert len(seen) == nvt * nkt * nsplit


def _fwd_cover(lib, C, K, V, prec, v_total=0, ranks=0, tile=0, fwd_splits=0):
    from tangram_amd import _capi
    cfg = _capi.TgConfig()
    cfg.abi_version = _capi.TG_ABI_VERSION
    cfg.n_cells, cfg.n_genes, cfg.n_spots, cfg.n_spots_total, cfg.n_ranks = C, K, V, v_total, ranks
    cfg.lambda_g1, cfg.has_density, cfg.lambda_d, cfg.precision, cfg.tile_size, cfg.fwd_splits = 1.0, 1, 1.0, prec, tile, fwd_splits
    out = (ctypes.c_longlong * 6)()
    rc = lib.tg_debug_fwd_cover(ctypes.byref(cfg), out)
    assert rc == 0, lib.tg_last_error().decode()
    dec = (ctypes.c_int * 4)()
    assert lib.tg_debug_fwd_decomposition(ctypes.byref(cfg), dec) == 0
    return list(out), list(dec)


def test_forward_decomposition_covers_the_baseline_shapes():
    """Host replay of the forward launch (tg_debug_fwd_cover walks every workgroup through the kernel's own tg_fwd_walk): at the
    BASELINE shapes, which the CPU emulator cannot run, every (spot tile, gene tile, contraction step) is taken exactly once and
    the partial slots of every tile are the ones tg_ghat_reduce sums.  cfg2 in split-bf16 must be the stream-K decomposition
    (128 pieces x 2 wide gene tiles = one exactly full round of 256 CUs, every workgroup the same number of steps)."""
    from tangram_amd import _build
    lib = ctypes.CDLL(_build.build())
    lib.tg_last_error.restype = ctypes.c_char_p
    (grid, working, lo, hi, most_seg, slots), (units, nkt, nvt, nsteps) = _fwd_cover(lib, 30000, 1000, 10000, 2)
    assert (units, nkt, nvt) == (128, 2, 80) and units % nvt != 0 and (8 * nvt) % units == 0
    assert grid == working == 256 and hi - lo <= 1 and lo * units <= nvt * nsteps <= hi * units and most_seg == 2 and slots == 3
    # (a piece is 5/8 of a tile: a workgroup touches two tiles, a tile is summed from up to three partial slots)
    for shape in [(30000, 1000, 10000, 1), (30000, 1000, 10000, 0), (200000, 2000, 50000, 1), (200000, 2000, 50000, 2),
                  (30000, 1000, 1250, 2, 10000, 8), (200000, 2000, 6250, 1, 50000, 8), (26431, 249, 9852, 2), (4200, 1000, 1500, 2)]:
        (grid, working, lo, hi, most_seg, slots), (units, nkt, nvt, nsteps) = _fwd_cover(lib, *shape)
        assert working == units * nkt <= grid and lo >= 1 and slots >= 1
        if units % nvt:                                  # stream-K pieces: equal shares of the step space
            assert hi - lo <= 1
        if shape[:3] == (30000, 1000, 1250):             # round 6: a thin shard takes ONE exact round of stream-K pieces (measured: -7 % on its forward)
            assert units * nkt == 256 and units % nvt != 0


@pytest.mark.parametrize("seed", range(6))
def test_forward_decomposition_covers_random_shapes(seed):
    """The same replay over random shapes, tile sizes and FORCED piece counts (fwd_splits < 0: pieces that start and end anywhere,
    longer or shorter than a tile), incl. piece counts that do not divide anything."""
    from tangram_amd import _build
    lib = ctypes.CDLL(_build.build())
    lib.tg_last_error.restype = ctypes.c_char_p
    rng = np.random.default_rng(100 + seed)
    for _ in range(40):
        C, K, V = int(rng.integers(1, 5000)), int(rng.integers(1, 1400)), int(rng.integers(1, 6000))
        if C <= 32:
            C += 33                                      # (clusters-mode shapes do not launch this kernel)
        prec, tile = int(rng.integers(0, 3)), int(rng.choice([0, 128, 256]))
        forced = int(rng.integers(0, 3))
        fwd_splits = 0 if forced == 0 else (int(rng.integers(1, 9)) if forced == 1 else -int(rng.integers(1, 200)))
        (grid, working, lo, hi, most_seg, slots), (units, nkt, nvt, nsteps) = _fwd_cover(lib, C, K, V, prec, tile=tile, fwd_splits=fwd_splits)
        assert working == units * nkt <= grid and most_seg >= 1 and slots >= 1
        if fwd_splits < 0:
            assert units == min(-fwd_splits, nvt * nsteps)
        elif fwd_splits > 0:
            assert units == nvt * min(fwd_splits, nsteps)


def test_launch_geometry_of_the_baseline_shapes():
    """tg_make_layout on the host: cfg2 takes 256^2 tiles and, for split-bf16, the 128 x 512 forward tiles."""
    from tangram_amd import _build, _capi
    lib = ctypes.CDLL(_build.build())
    out = (ctypes.c_int * 8)()

    def geo(C, K, V, prec, v_total=0, ranks=0):
        cfg = _capi.TgConfig()
        cfg.abi_version = _capi.TG_ABI_VERSION
        cfg.n_cells, cfg.n_genes, cfg.n_spots, cfg.n_spots_total, cfg.n_ranks = C, K, V, v_total, ranks
        cfg.lambda_g1, cfg.has_density, cfg.lambda_d, cfg.precision = 1.0, 1, 1.0, prec
        assert lib.tg_debug_layout(ctypes.byref(cfg), out) == 0
        return list(out)

    T, nct, nvt, nkt, nsplit, wide, bands, _ = geo(30000, 1000, 10000, 2)
    assert (T, nct, nvt, nkt, wide, bands) == (256, 118, 40, 4, 1, 1)
    assert geo(30000, 1000, 10000, 1)[5] == 0            # plain bf16 keeps the 256^2 forward
    assert geo(30000, 1000, 10000, 0)[5] == 0            # and so does exact fp32
    assert geo(30000, 1000, 1250, 2, v_total=10000, ranks=8)[:3] == [256, 118, 5]
    assert geo(30000, 700, 10000, 2)[5] == 0             # 701 gene columns pad to 768: not a multiple of 512
    assert geo(2000, 100, 500, 2)[0] == 128


def test_tile_copies_stay_inside_their_32_bit_offsets():
    """The operand tiles are copied through one buffer descriptor per tile (TgKtileDma: 32-bit byte count and lane offsets; an
    out-of-range buffer load returns zeros instead of faulting).  tg_make_layout must pick a forward geometry whose S^T tile --
    rows x (4 bytes per cell) -- stays below 4 GiB, and refuse a pinned geometry that does not (round 4 advisor finding)."""
    from tangram_amd import _build, _capi
    lib = _capi._declare(ctypes.CDLL(_build.build()))
    out = (ctypes.c_int * 8)()

    def geo(C, K, V, prec, tile=0):
        cfg = _capi.TgConfig()
        cfg.abi_version = _capi.TG_ABI_VERSION
        cfg.n_cells, cfg.n_genes, cfg.n_spots, cfg.tile_size = C, K, V, tile
        cfg.lambda_g1, cfg.has_density, cfg.lambda_d, cfg.precision = 1.0, 1, 1.0, prec
        rc = lib.tg_debug_layout(ctypes.byref(cfg), out)
        return rc, list(out)

    def tile_bytes(C, prec, rows):
        cp = -(-C // 64) * 64
        return rows * (cp // (64 if prec == 1 else 32)) * 128

    rc, g = geo(2_000_000, 1000, 1000, 2)                 # just below the limit of the 128 x 512 forward tiles
    assert rc == 0 and g[0] == 256 and g[5] == 1 and tile_bytes(2_000_000, 2, 512) < 2**32
    rc, g = geo(2_200_000, 1000, 1000, 2)                 # 512 gene rows x 8.8 MB would wrap: 256^2 tiles instead
    assert rc == 0 and g[0] == 256 and g[5] == 0 and tile_bytes(2_200_000, 2, 512) >= 2**32 > tile_bytes(2_200_000, 2, 256)
    rc, g = geo(4_300_000, 1000, 1000, 2)                  # 256 rows wrap as well: 128^2
    assert rc == 0 and g[0] == 128 and tile_bytes(4_300_000, 2, 256) >= 2**32 > tile_bytes(4_300_000, 2, 128)
    rc, g = geo(4_300_000, 1000, 1000, 1)                  # plain bf16: 2 bytes per cell, 256^2 still fits
    assert rc == 0 and g[0] == 256
    rc, _ = geo(4_300_000, 1000, 1000, 2, tile=256)        # a pinned geometry that cannot address its tile is refused, with a message
    assert rc == -4 and b"32-bit" in lib.tg_last_error()      # TG_ERR_UNSUPPORTED
    rc, _ = geo(9_000_000, 1000, 1000, 2)                  # beyond every geometry
    assert rc == -4


def test_sizes_of_the_large_configurations_do_not_overflow():
    """tg_query_sizes on the host for BASELINE config 4 (200k x 2k x 50k, bf16: M + Adam moments = 120 GB on one GPU), its 1/8
    spot shard, and a problem far beyond any GPU: 64-bit byte counts, consistent with the documented layout."""
    from tangram_amd import _build, _capi
    lib = _capi._declare(ctypes.CDLL(_build.build()))

    def sizes(C, K, V, prec, v_total=0, ranks=0):
        cfg = _capi.TgConfig()
        cfg.abi_version = _capi.TG_ABI_VERSION
        cfg.n_cells, cfg.n_genes, cfg.n_spots, cfg.n_spots_total, cfg.n_ranks = C, K, V, v_total, ranks
        cfg.lambda_g1, cfg.has_density, cfg.lambda_d, cfg.precision = 1.0, 1, 1.0, prec
        sz = _capi.TgSizes()
        assert lib.tg_query_sizes(ctypes.byref(cfg), ctypes.byref(sz)) == 0, lib.tg_last_error()
        return sz

    s4 = sizes(200000, 2000, 50000, 1)
    pitch = s4.m_pitch
    assert pitch >= 50000 and pitch % 64 == 0
    assert s4.state_bytes >= 3 * 200000 * pitch * 4 and s4.state_bytes < 3 * 200000 * pitch * 4 + (1 << 26)     # M, m, v (+ nothing big)
    assert 200000 * pitch * 2 <= s4.workspace_bytes < 60 * (1 << 30)                 # bf16 X (20 GB) + operands + partials
    assert s4.state_bytes + s4.workspace_bytes < 288 * (1 << 30)                     # fits one MI355X
    s8 = sizes(200000, 2000, 6250, 1, v_total=50000, ranks=8)
    assert s8.state_bytes < s4.state_bytes // 7 and s8.state_bytes >= 3 * 200000 * 6250 * 4
    huge = sizes(2_000_000, 4000, 1_000_000, 2)
    assert huge.state_bytes >= 3 * 2_000_000 * 1_000_000 * 4                          # 24 TB: counted, not allocated
