"""Shared helpers of the parity tests (CPU-emulated and GPU).  The checker is the oracle / the golden
fixtures generated from the unmodified reference; the thing under test is always the C-ABI library."""
import os

import numpy as np

from oracle.gen_golden import CASES, build_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated fp32 tolerances (calibrated on the reference's own fp32-vs-fp64 / thread-count spread)
TOL = {
    #           per-epoch |d loss|   max|dP|   relFro(P^T S)
    "fp32":   dict(loss=1e-5, P=2e-4, ghat=1e-4),
    "bf16x3": dict(loss=1e-5, P=2e-4, ghat=1e-4),
    # bf16, max|dP| on the SMALL cases (tens to hundreds of spots: single probabilities of 0.1 - 1): calibrated in round 6 by running the 29
    # plain-bf16 cases of the GPU suite under shrinking bounds (scripts/gpu_r06f.sh, profiles/r06/run4_bf16_tolerance_scan): 28 pass at
    # 2e-2, 26 at 1e-2, 18 at 1e-3; the largest measured value is 3.5e-2 (golden constrained_entropy, 500 epochs), then 1.33e-2 (live
    # reference, spatial terms) and 1.26e-2 (golden cells_autocorr): the bound is 1.4 x the largest measurement, not a formality.  (At
    # full size the meaningful bound is relative: FULL_BOUNDS in tests/test_gpu_live_reference.py.)
    "bf16":   dict(loss=1e-3, P=float(os.environ.get("TG_TOL_BF16_P", 5e-2)), ghat=1e-2),      # (environment: the scan that calibrated the bound)
}


# The 500-epoch cases of the reference's own test grid are ill-conditioned: the reference's fp32 run drifts from its fp64 run (flat
# valley).  Beyond the well-conditioned prefix an implementation is held to this multiple of that drift, term by term and at the end
# point.  ONE bound for both kernel families (the clusters-mode kernels these 12-cluster cases run on by default, and the GEMM
# kernels pinned by `tile_size`), because the multiple an fp32 run ends at is amplified round-off of the transcendentals, not a
# property of a kernel family -- measured (scripts/exp_rounding_drift.py -> profiles/r04/exp_rounding/drift.json): the SAME kernel
# sources on the CPU emulator, with exp2 / exp / log rounded four different 1-ulp-accurate ways (host libm; towards zero; away from
# zero; hash-picked neighbour -- the last three also with the hardware's fp32 product x * log2(e) inside exp), end the seven cases at
#     largest multiple     libm    towards 0    away    hashed        MI355X (v_exp_f32 / v_log_f32)
#     clusters kernels     1.44      2.49       2.12     1.60          3.0   (round 3, profiles/r03)
#     GEMM kernels         2.22      1.72       2.74     2.00          2.2
# i.e. anywhere in 1.4 - 2.7 for either family from a 1-ulp change of three scalar functions; the hardware's 3.0 was one more draw
# from that distribution (round 4 therefore carried 4 = 1.5 x the largest emulated draw).
# Round 5 records what a GPU session measures (MEASURED_SPREAD below -> gpurun_out/own_spread_measured.json; committed copy:
# profiles/r05/final/own_spread_measured.json): with this round's kernels the largest multiples on MI355X are 1.47 (clusters-mode
# kernels, fp32 and bf16x3), 1.80 / 1.37 (GEMM kernels, bf16x3 / fp32) and 2.31 (GEMM kernels, plain bf16, against its own 100 x
# wider tolerance).  The kernels are bit-reproducible (no atomics), so these are properties of the build, not of a box: the bound
# goes back to 3 = 1.3 x the largest measured multiple (the round-4 advisor's request), one constant for both families.
OWN_SPREAD = 3.0
# every (case, term) whose whole-run check was decided by the drift bound leaves its measured multiple here; tests/conftest.py writes the
# list to gpurun_out/own_spread_measured.json at the end of a GPU session (round-4 advisor: record the multiples a round measures)
MEASURED_SPREAD = []


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def run_case(name, device, precision, epochs=None, pin_gemm=False):
    """Train the tangram_amd Mapper on golden case `name`; returns (P, history dict, Ghat, golden npz, epochs).
    pin_gemm: the engine is built with tile_size = 128, which keeps a problem of at most 32 cells on the GEMM kernels."""
    import functools
    import tangram_amd.mapping_optimizer as mo
    if pin_gemm:
        orig = mo.HipMapperEngine
        mo.HipMapperEngine = functools.partial(orig, tile_size=128)
        try:
            out = run_case(name, device, precision, epochs)
            out["pin_gemm"] = True
            return out
        finally:
            mo.HipMapperEngine = orig
    z = load_golden(name)
    args, n_epochs, mode = build_inputs(name)
    val_each = args.pop("val_each", None)
    if epochs is not None:
        n_epochs = min(n_epochs, epochs)
    if mode == "constrained":
        m = mo.MapperConstrained(device=device, gemm_precision=precision, M_init=z["f32_M0"], F_init=z["f32_F0"], **args)
        P, F, hist = m.train(num_epochs=n_epochs, learning_rate=0.1, print_each=None)
    else:
        m = mo.Mapper(device=device, gemm_precision=precision, M_init=z["f32_M0"], **args)
        P, hist = m.train(num_epochs=n_epochs, learning_rate=0.1, print_each=None, val_each=val_each)
        F = None
    Ghat = m.project_genes_device().detach().cpu().numpy()
    return dict(P=P, F=F, hist=hist, Ghat=Ghat, z=z, epochs=n_epochs, mode=mode, name=name)


def check_against_golden(res, precision, full_length, own_spread=None):
    """Compare with the reference's fp64 run (ground truth) within the stated fp32 tolerance."""
    OWN_SPREAD = own_spread if own_spread is not None else globals()["OWN_SPREAD"]
    tol = TOL[precision]
    z, n = res["z"], res["epochs"]
    keys = ["main_loss", "total_loss", "kl_reg", "vg_reg", "entropy_reg"]
    if res["mode"] == "constrained":
        keys += ["count_reg", "lambda_f_reg"]
    # (the spatial terms enter total_loss; the reference keeps no separate history for them, :378-392)
    for k in keys:
        ref = z["f64_hist_" + k][:n]
        got = np.array([float(x) for x in res["hist"][k]], dtype=np.float64)
        if np.isnan(ref).all():
            assert np.isnan(got).all(), f"{k}: expected NaN history like the reference"
            continue
        scale = max(1.0, float(np.abs(ref).max()))
        if res["mode"] == "constrained" and k == "total_loss":
            scale *= 20.0        # the reference stores str(tensor) here: 4 printed decimals (mapping_optimizer.py:630)
        # The 500-epoch grid cases (the reference's own test grid) leave the well-conditioned regime after ~170 epochs: the
        # REFERENCE's fp32 run then drifts up to 1.3e-4 from its fp64 run in main_loss / kl_reg (flat valley) while total_loss
        # stays within 6e-6 (tests/test_oracle_golden.py).  Each term is held to the flat tolerance for as long as the
        # reference's own fp32 arithmetic stays within a third of it; total_loss additionally over the whole run, bounded
        # by the flat tolerance or 5x the reference's own fp32-vs-fp64 spread on the case.
        own = np.abs(z["f32_hist_" + k][:n] - ref) if ("f32_hist_" + k) in z.files else np.zeros(n)
        if res["mode"] == "constrained" and k == "total_loss":
            own = np.zeros(n)
        over = np.nonzero(own > tol["loss"] * scale / 3.0)[0]
        well = int(over[0]) if len(over) else n
        if res["mode"] == "grid":
            well = min(well, 100 if precision != "bf16" else 50)   # round-off grows ~10x per 50 epochs on these cases; any two fp32
                                                                  # implementations part ways by ~150 (bf16 operands: earlier)
        assert well >= min(n, 50), f"{k}: fixture ill-conditioned from epoch {well}"
        err = float(np.abs(got[:well] - ref[:well]).max())
        assert err <= tol["loss"] * scale, f"{k}: max per-epoch |delta| {err:.3e} > {tol['loss'] * scale:.1e}"
        if well < n:
            # ... and over the WHOLE run EVERY term stays within OWN_SPREAD x the reference's own fp32-vs-fp64 drift on the case
            # (measured: 1.0 - 2.2 x; round 2 held only total_loss, to 5 x)
            spread = float(np.abs(z["f32_hist_" + k][:n] - ref).max())
            bound = max(tol["loss"] * scale, OWN_SPREAD * spread * (tol["loss"] / 1e-5))
            err = float(np.abs(got - ref).max())
            if spread > 0 and OWN_SPREAD * spread * (tol["loss"] / 1e-5) > tol["loss"] * scale:      # the drift bound is the binding one: keep the multiple
                MEASURED_SPREAD.append(dict(case=res.get("name"), mode=res["mode"], precision=precision, pinned_gemm=bool(res.get("pin_gemm")),
                                            term=k, multiple=err / (spread * (tol["loss"] / 1e-5)), bound=OWN_SPREAD))
            assert err <= bound, f"{k} (full run): max per-epoch |delta| {err:.3e} > {bound:.1e} (the reference's own fp32 drift: {spread:.1e})"
    if full_length:
        dP = float(np.abs(res["P"] - z["f64_P"]).max())
        boundP = tol["P"]
        if res["mode"] == "grid":       # end point of an ill-conditioned 500-epoch run: relative to the reference's own fp32 spread
            boundP = max(tol["P"], OWN_SPREAD * float(np.abs(z["f32_P"] - z["f64_P"]).max()) * (tol["P"] / 2e-4))
        assert dP <= boundP, f"max|dP| {dP:.3e} > {boundP:.1e}"
        rel = float(np.linalg.norm(res["Ghat"] - z["f64_Ghat"]) / np.linalg.norm(z["f64_Ghat"]))
        bound_g = tol["ghat"]
        if res["mode"] == "grid":
            bound_g = max(bound_g, OWN_SPREAD * float(np.linalg.norm(z["f32_Ghat"] - z["f64_Ghat"]) / np.linalg.norm(z["f64_Ghat"])) * (tol["ghat"] / 1e-4))
        assert rel <= bound_g, f"relFro(P^T S) {rel:.3e} > {bound_g:.1e}"
        if res["F"] is not None:
            dF = float(np.abs(res["F"] - z["f64_F_out"]).max())
            assert dF <= tol["P"], f"max|dF| {dF:.3e}"
        am = (res["P"].argmax(1) == z["f64_P"].argmax(1)).mean()
        # (grid cases: 12 cluster rows whose largest entries are ~0.04 and nearly tied; one flipped row is 8 %)
        need = (0.9 if precision != "bf16" else 0.75) if res["mode"] == "grid" else (0.98 if precision != "bf16" else 0.9)
        assert am >= need, f"argmax agreement {am:.3f}"
    if "f64_hist_val_gene_sim" in z.files:       # Mapper._val_loss_fn metrics (mapping_optimizer.py:311-356)
        for k in ("val_total_loss", "val_gene_sim", "val_sp_sparsity_weighted_sim", "val_entropy"):
            got = np.array(res["hist"][k], dtype=np.float64)
            ref = z["f64_hist_" + k][:len(got)]
            assert len(got) > 0 and float(np.abs(got - ref).max()) <= 10 * tol["loss"], (k, got, ref)
    np.testing.assert_allclose(res["P"].sum(axis=1), 1.0, atol=1e-5)
    assert (res["P"] >= 0).all()


def small_cluster_case(device, C, K, V, constrained, lambda_g2, seed, precision="bf16x3", n=3, tile_size=0):
    """One clusters-mode-sized problem (C <= 32: the library runs tg_sc_forward / tg_sc_backward instead of the GEMM kernels,
    asserted through tg_debug_layout) for n epochs against the fp64 oracle.  Tolerances: the fp32 row of TOL whatever `precision`
    says (that path computes in fp32 FMAs).  tile_size != 0 pins the GEMM path on the same problem."""
    import ctypes as ct
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    from oracle import tangram_oracle as orc
    rng = np.random.default_rng(seed)
    data = orc.make_synthetic(C, K, V, seed=seed)
    lam = dict(lambda_g1=1.0, lambda_d=float(rng.choice([0.0, 1.0])), lambda_g2=lambda_g2, lambda_r=float(rng.choice([0.0, 1e-3])))
    d = data["d"] if lam["lambda_d"] > 0 or constrained else None
    if constrained:
        lam["lambda_d"] = lam["lambda_d"] or 1.0
        lam.update(lambda_count=1.0, lambda_f_reg=1.0)
        tc = max(1.0, 0.5 * C)
        M0, F0 = orc.reference_init_MF_constrained(C, V, seed)
        o = orc.OracleMapperConstrained(data["S"], data["G"], d, M0=M0, F0=F0, target_count=tc, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(n, 0.1)
        e = HipMapperEngine(data["S"], data["G"], M0, d=d, F0=F0, mode="constrained", device=device, precision=precision, lambdas=lam,
                            target_count=tc, tile_size=tile_size)
    else:
        lam.update(lambda_l1=float(rng.choice([0.0, 1e-4])), lambda_l2=float(rng.choice([0.0, 1e-5])))
        ds = rng.dirichlet(np.ones(C)).astype(np.float32) if (d is not None and rng.integers(2)) else None
        M0 = orc.reference_init_M(C, V, seed)
        o = orc.OracleMapper(data["S"], data["G"], d=d, d_source=ds, M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(n, 0.1)
        e = HipMapperEngine(data["S"], data["G"], M0, d=d, d_source=ds, device=device, precision=precision, lambdas=lam, tile_size=tile_size)
    geo = (ct.c_int * 8)()
    assert e._lib.tg_debug_layout(ct.byref(e.cfg), geo) == 0
    assert geo[7] == int(tile_size == 0), "the small-C path is taken exactly when no tile size is pinned"
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    h = hist.cpu().numpy().astype(np.float64)
    tol = TOL["fp32"] if tile_size == 0 else TOL[precision]
    cols = [(_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss")]
    if lam["lambda_g2"] > 0:
        cols.append((_capi.H_VG, "vg_reg"))
    if lam["lambda_d"] > 0:
        cols.append((_capi.H_KL, "kl_reg"))
    if lam["lambda_r"] > 0:
        cols.append((_capi.H_ENTROPY, "entropy_reg"))
    for col, k in cols:
        ref = np.array([float(x) for x in ho[k]])
        err = np.abs(h[:, col] - ref).max()
        assert err <= 3 * tol["loss"] * max(1.0, np.abs(ref).max()), (C, K, V, constrained, k, err)
    if constrained:
        P, F = e.result(with_filter=True)
        assert np.abs(F.cpu().numpy() - Fo).max() <= 2e-5
    else:
        P = e.result()
    assert np.abs(P.cpu().numpy() - Po).max() <= tol["P"], (C, K, V)
    return e
