"""
Pin the oracle (oracle/) against outputs of the reference's own function bodies
executed in the build container (tests/golden/reference_golden.npz, made by
tests/golden/make_golden.py).  CPU only.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import scipy.sparse.linalg as spla


def test_blas_helpers(golden, oracle):
    # utilities/linear_algebra_funcs.py:16-44
    G = golden
    assert np.array_equal(oracle.dgemm(G["la_A"], G["la_B"]).shape, G["la_dgemm"].shape)
    np.testing.assert_allclose(oracle.dgemm(G["la_A"], G["la_B"]), G["la_dgemm"], rtol=1e-14)
    np.testing.assert_allclose(oracle.dgemm(G["la_A"], G["la_B"]), G["la_A"].T @ G["la_B"].T,
                               rtol=1e-14)
    assert oracle.norm2(G["la_q"]) == pytest.approx(float(G["la_norm2"]), rel=1e-15)
    assert oracle.scalprod(G["la_q"], G["la_q2"]) == pytest.approx(float(G["la_scalprod"]), rel=1e-14)


def test_generators(golden, oracle):
    # utilities/utilities_functions.py:99-107
    np.testing.assert_array_equal(oracle.angles_gen(0.3, 50), golden["gen_angles"])


@pytest.mark.parametrize("lam", [1, 2, 33])
def test_toeplitz_bitexact(golden, oracle, lam):
    # interfaces/linearoperators.py:582-595 -- same summation order => bit equal
    a, v, y = golden["toep_a%d" % lam], golden["toep_v"], golden["toep_y%d" % lam]
    np.testing.assert_array_equal(oracle.toeplitz_mult(a, v), y)
    np.testing.assert_array_equal(oracle.toeplitz_mult_numpy(a, v.copy()), y)


def test_toeplitz_band_longer_than_block(golden, oracle):
    np.testing.assert_array_equal(
        oracle.toeplitz_mult(golden["toep_a9"], golden["toep_vshort"]), golden["toep_yshort"])


def _weights(G, npix_key="bd_counts"):
    r = SimpleNamespace(counts=G["bd_counts"], cosine=G["bd_cos"], sine=G["bd_sin"],
                        cos2=G["bd_cos2"], sin2=G["bd_sin2"], sincos=G["bd_sincos"])
    r.new_npix = r.counts.shape[0]
    return r


@pytest.mark.parametrize("pol", [1, 2, 3])
def test_block_diagonal_lo_bitexact(golden, oracle, pol):
    # interfaces/linearoperators.py:728-746
    r = _weights(golden)
    np.testing.assert_array_equal(oracle.bd_mult(pol, r, golden["bd_x%d" % pol]),
                                  golden["bd_y%d" % pol])


def test_bd_preconditioner_pol1(golden, oracle):
    # interfaces/linearoperators.py:788-790
    r = _weights(golden)
    r.counts = golden["bdp1_counts"]
    y = oracle.bd_precond_mult(1, r, golden["bdp1_x"])
    np.testing.assert_array_equal(y, golden["bdp1_y"])
    assert y[2] == 0.0 and y[11] == 0.0            # unobserved pixels -> 0


def test_deflation(golden, oracle):
    # interfaces/linearoperators.py:1041-1056
    Z = golden["defl_Z"]
    np.testing.assert_array_equal(oracle.deflation_mult(Z, golden["defl_y"]), golden["defl_Zy"])
    np.testing.assert_allclose(oracle.deflation_rmult(Z, golden["defl_x"]), golden["defl_Ztx"],
                               rtol=1e-14)


def test_coarse(golden, oracle):
    # interfaces/linearoperators.py:969-1027
    Z, A, v = golden["defl_Z"], golden["coarse_A"], golden["coarse_v"]
    Az = A @ Z
    lu = oracle.Coarse(Z, Az, 4, apply='LU')
    np.testing.assert_allclose(lu.E, golden["coarse_E"], rtol=1e-14)
    np.testing.assert_allclose(lu.mult(v), golden["coarse_lu_x"], rtol=1e-12)
    eig = oracle.Coarse(Z, Az, 4, apply='eig')
    np.testing.assert_allclose(eig.invE, golden["coarse_invE"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(eig.mult(v), golden["coarse_eig_x"], rtol=1e-10)
    # degenerate E: one eigenvalue dropped (:997-999)
    Zd = Z.copy()
    Zd[:, 3] = Zd[:, 0]
    deg = oracle.Coarse(Zd, A @ Zd, 4, apply='eig')
    np.testing.assert_allclose(deg.invE, golden["coarse_deg_invE"], rtol=1e-8, atol=1e-12)
    assert np.linalg.matrix_rank(deg.invE, tol=1e-9) == 3


@pytest.mark.parametrize("pol", [1, 2, 3])
def test_repixelization(golden, oracle, pol):
    # utilities/process_ces.py:351-401 : old2new / compaction semantics -- the oracle's
    # repixelization() against what the reference's executed body produced
    G = golden
    mask = G["repix%d_mask" % pol]
    nold = G["repix%d_old2new" % pol].shape[0]
    keys = {1: ("counts",), 2: ("cos2", "sin2", "sincos"),
            3: ("counts", "cosine", "sine", "cos2", "sin2", "sincos")}[pol]
    old2new, new_npix, obspix, out = oracle.repixelization(
        mask, nold, np.arange(100, 100 + nold), {k: G["repix%d_in_%s" % (pol, k)] for k in keys})
    np.testing.assert_array_equal(old2new, G["repix%d_old2new" % pol])
    assert new_npix == int(G["repix%d_npix" % pol])
    np.testing.assert_array_equal(obspix, G["repix%d_obspix" % pol])
    for k in keys:
        np.testing.assert_array_equal(out[k], G["repix%d_out_%s" % (pol, k)])


def test_arnoldi_build_hess_build_Z(golden, oracle):
    # interfaces/deflationlib.py:17-184
    G = golden
    A, b = G["arn_A"], G["arn_b"]
    vs, hs, j = oracle.arnoldi(lambda x: A @ x, b, np.zeros_like(b), tol=1e-8, inner_m=30)
    assert j == int(G["arn_j"])
    np.testing.assert_allclose(np.asarray(vs), G["arn_V"], rtol=0, atol=1e-9)
    H = oracle.build_hess(hs, j)
    np.testing.assert_allclose(H, G["arn_H"], rtol=0, atol=1e-9)
    z, y = np.linalg.eigh(H)
    np.testing.assert_allclose(z, G["arn_ritz"], rtol=1e-8, atol=1e-10)
    Z, r = oracle.build_Z(G["arn_ritz"], np.linalg.eigh(G["arn_H"])[1], G["arn_V"].T.copy(), 1e-2)
    assert r == int(G["arn_r"])
    np.testing.assert_allclose(Z, G["arn_Z"], rtol=1e-12, atol=1e-14)
    assert int(G["arn_raises_at_inner_m"]) == 1
    with pytest.raises(RuntimeError):
        oracle.arnoldi(lambda x: A @ x, b, np.zeros_like(b), tol=1e-8, inner_m=3)
    assert int(G["arn_nonfinite_raises"]) == 1
    with pytest.raises(ValueError):
        oracle.arnoldi(lambda x: A @ x, b * np.nan, np.zeros_like(b))
    assert int(G["arn_zero_residual_returns_j0"]) == 1
    assert oracle.arnoldi(lambda x: A @ x, A @ np.ones(30), np.ones(30), tol=1e-5)[2] == 0


def test_cg_recurrence_equals_local_scipy(oracle):
    # scipy.sparse.linalg.cg is the reference's PCG driver
    # (tests/test_2level_preconditioner.py:52, src/test_BD_precond_onto_real_data.py:47)
    rng = np.random.default_rng(5)
    n = 60
    S = rng.standard_normal((n, n))
    A = S @ S.T + n * np.eye(n)
    Minv = np.diag(1.0 / np.diag(A))
    b = rng.standard_normal(n)
    for x0 in (None, np.ones(n)):
        its_s, its_o = [], []
        xs, info_s = spla.cg(A, b, x0=x0, rtol=1e-8, M=Minv, callback=lambda x: its_s.append(1))
        xo, info_o = oracle.cg(lambda v: A @ v, b, x0=x0, rtol=1e-8, M=lambda v: Minv @ v,
                               callback=lambda x: its_o.append(1))
        assert info_s == info_o == 0
        assert len(its_s) == len(its_o)
        np.testing.assert_array_equal(xs, xo)
    x, info = oracle.cg(lambda v: A @ v, b, rtol=1e-14, maxiter=3)
    assert info == 3


# ---------------------------------------------------------------- f1: FilterLO ----
def _filter_case(golden):
    ss = [[golden["filt_subscan0"], golden["filt_subscan1"]],
          [golden["filt_tstart0"], golden["filt_tstart1"]]]
    return ss, list(golden["filt_nsamples"]), list(golden["filt_nbolos"])


def test_legendre_tables(golden, oracle):
    np.testing.assert_array_equal(oracle.get_legendre_polynomials(3, 17), golden["leg_3_17"])
    np.testing.assert_array_equal(oracle.get_legendre_polynomials(1, 2), golden["leg_1_2"])


@pytest.mark.parametrize("order", [1, 2, 3])
def test_filter_poly_equals_reference_polyfilter(golden, oracle, order):
    ss, ns, nb = _filter_case(golden)
    y = oracle.filter_poly(golden["filt_d"], golden["filt_pix"], ss, ns, nb, order)
    np.testing.assert_array_equal(y, golden["filt_out%d" % order])


def test_filter_mean_properties(golden, oracle):
    """poly_order=0 is a weave loop in the reference (:141-162), so it is pinned by what
    the loop must do rather than by an executed vector."""
    ss, ns, nb = _filter_case(golden)
    d, pix = golden["filt_d"], golden["filt_pix"]
    y = oracle.filter_mean(d, pix, ss, ns, nb)
    starts, lens = oracle.filter_segments(*oracle.filter_normalise_args(ss, ns, nb))
    covered = np.zeros(d.size, dtype=bool)
    for a, n in zip(starts, lens):
        v = pix[a:a + n] != -1
        if not v.any():
            assert not y[a:a + n].any()                  # NaN mean -> chunk skipped (:163-164)
            continue
        covered[a:a + n] = True
        mean = d[a:a + n][v].sum() / v.sum()
        np.testing.assert_allclose(y[a:a + n], d[a:a + n] - mean, rtol=0, atol=1e-13)
        assert abs(y[a:a + n][v].sum()) < 1e-11         # offset of the unflagged samples is gone
    assert not y[~covered].any()                         # vec_out = d*0 outside the chunks (:130)
    np.testing.assert_allclose(oracle.filter_mean(y, pix, ss, ns, nb), y, atol=1e-13)  # idempotent
    # single-CES call convention (:269-273)
    one = oracle.filter_mean(d[:600], pix[:600], [ss[0][0], ss[1][0]], ns[0], nb[0])
    np.testing.assert_array_equal(one, y[:600])


def test_ground_filter_oracle():
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    g = rng.integers(-1, 12, size=500)
    g[g == 7] = 3                                        # an empty bin
    v = rng.standard_normal(500)
    y = O.ground_filter(g, v)
    for b in range(12):
        sel = g == b
        if sel.any():
            np.testing.assert_allclose(y[sel], v[sel] - v[sel].mean(), atol=1e-13)
    np.testing.assert_array_equal(y[g == -1], v[g == -1])
    np.testing.assert_allclose(O.ground_filter(g, y), y, atol=1e-13)


def test_all_cores_baseline_equals_serial_oracle(oracle):
    """bench.py's all-cores host baseline (OpenMP pointing loops + FFT Toeplitz per block)
    against the serial reference-order oracle."""
    rng = np.random.default_rng(0)
    pol, npix, nt, nb, lam = 3, 500, 40000, 4, 33
    d, pairs, phi, t, diag = oracle.system_setup(rng, nt, npix, nb)
    pairs[rng.random(nt) < 0.05] = -1
    c, s = np.cos(2 * phi), np.sin(2 * phi)
    kk = np.arange(lam)
    bands = [(1 + 0.1 * b) * np.exp(-kk / 8.0) for b in range(nb)]
    x = rng.standard_normal(pol * npix)
    ref = oracle.sparse_rmult(pol, npix, pairs, c, s, oracle.blocklo_mult(
        nt // nb, bands, True, oracle.sparse_mult(pol, pairs, c, s, x)))
    for threads in (1, 3):
        y = oracle.AllCoresMatvec(pol, npix, pairs, c, s, nt // nb, bands, threads)(x)
        assert np.linalg.norm(y - ref) / np.linalg.norm(ref) < 1e-13
    # a block convolved in pieces (with halos from inside the block only): the same operator, also when a
    # piece is shorter than the band and when the last piece of a block is ragged
    tod = rng.standard_normal(nt)
    ref_n = oracle.blocklo_mult(nt // nb, bands, True, tod)
    for piece in (1 << 20, 4096, 1000, 17):
        y = oracle.AllCoresMatvec(pol, npix, pairs, c, s, nt // nb, bands, 3, piece=piece).N(tod)
        assert np.linalg.norm(y - ref_n) / np.linalg.norm(ref_n) < 1e-13, piece


@pytest.mark.parametrize("pol", [1, 2, 3])
def test_full2cutskymap_and_reorganize_map(golden, oracle, pol):
    obs, full = golden["cut_obspix"], list(golden["cut_full"][:pol])
    res = oracle.full2cutskymap(full, pol, obs.size, obs)
    assert isinstance(res, list) == (pol == 1) == bool(golden["cut_pol1_returns_list"]) or pol > 1
    vec = res[0] if pol == 1 else res
    np.testing.assert_array_equal(vec, golden["cut_out%d" % pol])
    maps = oracle.reorganize_map(vec, obs, obs.size, 4, pol)      # needs healpy in the reference
    assert len(maps) == pol and all(m.size == 192 for m in maps)
    for k in range(pol):
        np.testing.assert_array_equal(maps[k][obs], full[k][obs])
        assert not np.delete(maps[k], obs).any()
