"""
oracle.py -- NumPy/SciPy + C (cm2_oracle.c) restatement of the COSMOMAP2 PCG hot
path, used ONLY as the checker.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``cosmomap2_amd/`` imports this module.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Pinning status per piece is in the header of cm2_oracle.c and
in DESIGN.md section "Oracle"; in short:

* executed-reference pins (tests/golden/make_golden.py runs the reference's own
  function bodies here and stores outputs): ToeplitzLO.mult, BlockDiagonalLO.mult,
  BlockDiagonalPreconditionerLO.mult (pol=1), ProcessTimeSamples.repixelization,
  DeflationLO.mult/rmult, CoarseLO (LU and eig), arnoldi/build_hess/build_Z,
  dgemm/norm2/scalprod, angles_gen/pairs_gen/noise_val/system_setup;
* invariant pins only (weave loops cannot run here): SparseLO mult/rmult,
  weight accumulation, flagging, M_BD pol=2/3;
* third-party, absent: krypy arnoldi/ritz (run_krypy_arnoldi,
  find_ritz_eigenvalues) -- restated from the published algorithm,
  PARITY UNPINNED for those two functions.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.linalg as sla

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcm2_oracle.so")
_SRC = os.path.join(_HERE, "cm2_oracle.c")
_lib = None

_D = ctypes.POINTER(ctypes.c_double)
_I32 = ctypes.POINTER(ctypes.c_int32)
_I64 = ctypes.POINTER(ctypes.c_int64)
_U8 = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile cm2_oracle.c with gcc (no FMA contraction, no fast-math)."""
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-o", _SO, _SRC, "-lm"]
    subprocess.check_call(cmd)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _d(a):
    return a.ctypes.data_as(_D)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------
# a2 / a3  SparseLO  (interfaces/linearoperators.py:356-526)
# --------------------------------------------------------------------------
def sparse_mult(pol, pix, cos, sin, x):
    pix = _i32(pix)
    nt = pix.shape[0]
    out = np.empty(nt)
    c = _f64(cos) if pol > 1 else np.zeros(1)
    s = _f64(sin) if pol > 1 else np.zeros(1)
    lib().orc_P_apply(ctypes.c_int(pol), ctypes.c_int64(nt), pix.ctypes.data_as(_I32),
                      _d(c), _d(s), _d(_f64(x)), _d(out))
    return out


def sparse_rmult(pol, npix, pix, cos, sin, v):
    pix = _i32(pix)
    nt = pix.shape[0]
    out = np.empty(npix * pol)
    c = _f64(cos) if pol > 1 else np.zeros(1)
    s = _f64(sin) if pol > 1 else np.zeros(1)
    lib().orc_Pt_apply(ctypes.c_int(pol), ctypes.c_int64(nt), ctypes.c_int64(npix),
                       pix.ctypes.data_as(_I32), _d(c), _d(s), _d(_f64(v)), _d(out))
    return out


def ptnp_diag(pol, npix, pix, cos, sin, w, x):
    """Reference-unfused P^T diag(w) P x (cpu_baseline leg)."""
    pix = _i32(pix)
    nt = pix.shape[0]
    tod = np.empty(nt)
    out = np.empty(npix * pol)
    c = _f64(cos) if pol > 1 else np.zeros(1)
    s = _f64(sin) if pol > 1 else np.zeros(1)
    lib().orc_PtNP_diag(ctypes.c_int(pol), ctypes.c_int64(nt), ctypes.c_int64(npix),
                        pix.ctypes.data_as(_I32), _d(c), _d(s), _d(_f64(w)),
                        _d(_f64(x)), _d(tod), _d(out))
    return out


# --------------------------------------------------------------------------
# a4  ToeplitzLO.mult  (interfaces/linearoperators.py:582-595)
# --------------------------------------------------------------------------
def toeplitz_mult(a, v):
    a = _f64(np.atleast_1d(a))
    v = _f64(v)
    y = np.empty_like(v)
    lib().orc_toeplitz_apply(ctypes.c_int64(a.shape[0]), _d(a),
                             ctypes.c_int64(v.shape[0]), _d(v), _d(y))
    return y


def toeplitz_mult_numpy(a, v):
    """Literal NumPy restatement of :587-595 (slow for long bands); used to
    cross-check the C loop's evaluation order."""
    y = a[0] * v
    for i in range(1, len(a)):
        temp = a[i] * v
        if i < len(v):
            y[:-i] += temp[i:]
            y[i:] += temp[:-i]
    return y


# --------------------------------------------------------------------------
# a5  BlockLO  (interfaces/linearoperators.py:655-690, interfaces/blkop.py:178-208)
# --------------------------------------------------------------------------
def block_sizes(blocksize, nblocks):
    """int blocksize -> nblocks equal blocks (the only form that works in the
    reference, SURVEY section 4 defect 1); a list is taken as per-block sizes
    (the documented intent of :638-639)."""
    if np.ndim(blocksize) == 0:
        return [int(blocksize)] * nblocks
    return [int(b) for b in blocksize]


def blocklo_diag(blocksize, t, offdiag=False):
    """BlockLO.diag (:677-683): per-sample weights for offdiag=False; for
    offdiag=True the reference exposes only covnoise[0] (:673, defect 3)."""
    if offdiag:
        return np.asarray(t[0], dtype=np.float64)
    sizes = block_sizes(blocksize, len(t))
    return np.concatenate([np.full(b, float(v)) for b, v in zip(sizes, t)])


def blocklo_mult(blocksize, t, offdiag, x):
    """blk_matvec (blkop.py:178-208): slice x per block, apply the block."""
    sizes = block_sizes(blocksize, len(t))
    y = np.empty(sum(sizes))
    o = 0
    for b, tb in zip(sizes, t):
        if offdiag:
            y[o:o + b] = toeplitz_mult(np.atleast_1d(tb), x[o:o + b])
        else:
            y[o:o + b] = float(tb) * x[o:o + b]          # lp.DiagonalOperator: diag*x
        o += b
    return y


# --------------------------------------------------------------------------
# a6 / a7  ProcessTimeSamples  (utilities/process_ces.py:58-89, 192-349, 403-555)
# --------------------------------------------------------------------------
class ProcessedSamples(object):
    pass


def repixelization(mask, npix, obspix, arrays):
    """new_repixelization / repixelization (utilities/process_ces.py:192-349, :351-401): keep the
    masked pixels in increasing order.  Returns ``(old2new, new_npix, obspix', arrays')`` with
    ``old2new[j]`` the rank of pixel j among the kept ones or -1, and every per-pixel array
    compacted to the kept pixels."""
    keep = np.zeros(npix, dtype=bool)
    keep[np.asarray(mask, dtype=np.int64)] = True
    old2new = np.full(npix, -1, dtype=np.int64)
    old2new[keep] = np.arange(int(keep.sum()))
    return (old2new, int(keep.sum()), np.asarray(obspix)[:npix][keep],
            {k: np.asarray(v)[keep] for k, v in arrays.items()})


def process_time_samples(pixs, npix, pol=1, phi=None, w=None, threshold_cond=1.e3,
                         obspix=None):
    """Restates __init__ -> initializeweights -> new_repixelization ->
    flagging_samples.  `pixs` is modified IN PLACE like the reference (:416).
    obspix default follows the documented intent np.arange(npix)
    (reference :67-68 uses nsamples, SURVEY defect 6)."""
    r = ProcessedSamples()
    pix = pixs                         # int32 ndarray, mutated in place
    assert pix.dtype == np.int32
    nt = pix.shape[0]
    if w is None:
        w = np.ones(nt)                                     # :65-66
    w = _f64(w)
    if obspix is None:
        obspix = np.arange(npix)
    z = lambda: np.zeros(npix)
    counts, cosine, sine, cos2, sin2, sincos = z(), z(), z(), z(), z(), z()
    if pol > 1:
        r.cos = np.cos(2. * np.asarray(phi))                # :493-494
        r.sin = np.sin(2. * np.asarray(phi))
    else:
        r.cos = np.zeros(1)
        r.sin = np.zeros(1)
    lib().orc_weights_accumulate(ctypes.c_int(pol), ctypes.c_int64(nt),
                                 pix.ctypes.data_as(_I32), _d(w), _d(r.cos), _d(r.sin),
                                 _d(counts), _d(cosine), _d(sine), _d(cos2), _d(sin2),
                                 _d(sincos))
    if pol == 1:
        mask = np.where(counts > 0)[0]                      # :491
    else:
        with np.errstate(all="ignore"):
            det = (cos2 * sin2) - (sincos * sincos)         # :544-550
            tr = cos2 + sin2
            sq = np.sqrt(tr * tr / 4. - det)
            lambda_max = tr / 2. + sq
            lambda_min = tr / 2. - sq
            cond_num = np.abs(lambda_max / lambda_min)
        mask = np.where(cond_num <= threshold_cond)[0]
        if pol == 3:
            mask2 = np.where(counts > 2)[0]                 # :554-555
            mask = np.intersect1d(mask2, mask)
    old2new, new_npix, r.obspix, compacted = repixelization(
        mask, npix, obspix, dict(counts=counts, cosine=cosine, sine=sine, cos2=cos2, sin2=sin2,
                                 sincos=sincos))
    r.mask = mask
    r.old2new = old2new
    r.new_npix = new_npix
    r.counts, r.cosine, r.sine = compacted["counts"], compacted["cosine"], compacted["sine"]
    r.cos2, r.sin2, r.sincos = compacted["cos2"], compacted["sin2"], compacted["sincos"]
    # flagging_samples (:411-418)
    lib().orc_flag_samples(ctypes.c_int64(nt), pix.ctypes.data_as(_I32),
                           old2new.ctypes.data_as(_I64))
    r.pixs = pix
    r.pol = pol
    r.nsamples = nt
    r.oldnpix = npix
    return r


# --------------------------------------------------------------------------
# a8  BlockDiagonalPreconditionerLO.mult  (interfaces/linearoperators.py:775-841)
# --------------------------------------------------------------------------
def bd_det_mask(pol, r):
    if pol == 1:
        return np.zeros(1), (r.counts > 0).astype(np.uint8)
    if pol == 3:                                            # :792-795
        determ = r.counts * (r.cos2 * r.sin2 - r.sincos * r.sincos) \
            - r.cosine * r.cosine * r.sin2 - r.sine * r.sine * r.cos2 \
            + 2. * r.cosine * r.sine * r.sincos
    else:                                                   # :820
        determ = (r.cos2 * r.sin2) - (r.sincos * r.sincos)
    return determ, (np.abs(determ) > 1e-5).astype(np.uint8)


def bd_precond_mult(pol, r, x):
    npix = r.new_npix
    det, mask = bd_det_mask(pol, r)
    y = np.empty(npix * pol)
    lib().orc_bdprecond_apply(ctypes.c_int(pol), ctypes.c_int64(npix), _d(_f64(r.counts)),
                              _d(_f64(r.cosine)), _d(_f64(r.sine)), _d(_f64(r.cos2)),
                              _d(_f64(r.sin2)), _d(_f64(r.sincos)), _d(_f64(det)),
                              mask.ctypes.data_as(_U8), _d(_f64(x)), _d(y))
    return y


# --------------------------------------------------------------------------
# a9  BlockDiagonalLO.mult  (interfaces/linearoperators.py:728-746)
# --------------------------------------------------------------------------
def bd_mult(pol, r, x):
    npix = r.new_npix
    y = np.empty(npix * pol)
    lib().orc_bd_apply(ctypes.c_int(pol), ctypes.c_int64(npix), _d(_f64(r.counts)),
                       _d(_f64(r.cosine)), _d(_f64(r.sine)), _d(_f64(r.cos2)),
                       _d(_f64(r.sin2)), _d(_f64(r.sincos)), _d(_f64(x)), _d(y))
    return y


# --------------------------------------------------------------------------
# a15  dgemm / norm2 / scalprod  (utilities/linear_algebra_funcs.py:16-44)
# --------------------------------------------------------------------------
def dgemm(A, B):
    """gemm(a=A.T, b=B, trans_b=True) = A^T B^T  (:27-29)."""
    return np.asarray(A).T.dot(np.asarray(B).T)


def norm2(q):
    return float(sla.get_blas_funcs('nrm2', dtype=np.float64)(np.asarray(q, dtype=np.float64)))


def scalprod(a, b):
    return float(sla.get_blas_funcs('dot', dtype=np.float64)(_f64(a), _f64(b)))


# --------------------------------------------------------------------------
# a10  DeflationLO  (interfaces/linearoperators.py:1041-1065)
# --------------------------------------------------------------------------
def deflation_mult(Z, x):
    Z = np.asarray(Z)
    y = np.zeros(Z.shape[0])
    for i in range(Z.shape[1]):                             # :1048-1049
        y += Z[:, i] * x[i]
    return y


def deflation_rmult(Z, x):
    Z = np.asarray(Z)
    return np.array([scalprod(Z[:, i], x) for i in range(Z.shape[1])])  # :1056


# --------------------------------------------------------------------------
# a11  CoarseLO  (interfaces/linearoperators.py:969-1027)
# --------------------------------------------------------------------------
class Coarse(object):
    def __init__(self, Z, Az, r, apply='LU'):
        M = dgemm(Z, np.asarray(Az).T)                      # :1019  E = Z^T (A Z)
        self.E = M.copy()
        self.apply = apply
        if apply == 'eig':                                  # :994-1015
            eigenvals, W = sla.eigh(M)
            lambda_max = max(eigenvals)
            diags = eigenvals * 0.
            nondegenerate = np.where(abs(eigenvals / lambda_max) > 1.e-6)[0]
            for i in nondegenerate:
                diags[i] = 1. / eigenvals[i]
            D = np.diag(diags)
            tmp = dgemm(D.T, W)
            self.invE = dgemm(W.T, tmp.T)
        elif apply == 'LU':                                 # :1025
            self.L, self.U = sla.lu(M, permute_l=True, check_finite=False)

    def mult(self, v):
        if self.apply == 'eig':
            return self.invE.dot(v)                         # :984
        y = sla.solve(self.L, v)                            # :975-976
        return sla.solve(self.U, y)


# --------------------------------------------------------------------------
# a12  two-level preconditioner  (src/test_M2_precond_onto_real_data.py:98-112,
#      tests/test_2level_preconditioner.py:45-48)
# --------------------------------------------------------------------------
def m2_apply(mbd, Z, AZ, coarse, r):
    """M2 r = Mbd (r - AZ y) + Z y,  y = E^-1 Z^T r."""
    y = coarse.mult(deflation_rmult(Z, r))
    return mbd(r - deflation_mult(AZ, y)) + deflation_mult(Z, y)


# --------------------------------------------------------------------------
# a13  arnoldi / build_hess / build_Z  (interfaces/deflationlib.py:17-184)
# --------------------------------------------------------------------------
def arnoldi(matvec, b, x0, tol=1e-5, inner_m=30, exhausted="raise"):
    """Modified Gram-Schmidt Arnoldi with the reference's stop rule
    abs(v_new[j]*h_{j+1,j}) <= tol (:101) and RuntimeError at inner_m (:111-112).
    exhausted="return" (bench.py's Ritz-space comparison only): hand back (vs, hs, inner_m) after
    inner_m steps instead of raising, so that a fixed number of steps can be compared."""
    if not np.isfinite(b).all():
        raise ValueError("RHS must contain only finite numbers")
    b_norm = norm2(b)
    if b_norm == 0:
        b_norm = 1
    r_outer = b - matvec(x0)
    r_norm = norm2(r_outer)
    if r_norm < tol * b_norm or r_norm < tol:
        return None, None, 0
    vs = [r_outer * (1.0 / r_norm)]
    hs = []
    for j in range(1, 1 + inner_m):
        v_new = matvec(vs[j - 1]).copy()
        hcur = []
        for v in vs:                                        # :94-97  (MGS)
            alpha = scalprod(v, v_new)
            hcur.append(alpha)
            v_new = v_new + (-alpha) * v
        hcur.append(norm2(v_new))
        v_new = v_new * (1.0 / hcur[-1])
        if abs(v_new[j] * hcur[-1]) <= tol:
            hs.append(hcur)
            return vs, hs, j
        vs.append(v_new)
        hs.append(hcur)
        if j == inner_m:
            if exhausted == "return":
                return vs, hs, j
            raise RuntimeError("Convergence not achieved within the Arnoldi algorithm")


def build_hess(h, m):                                       # :132-137
    hess = np.zeros((m, m))
    for q in range(m - 1):
        hess[:(q + 2), q] = h[q]
    hess[:m, m - 1] = h[-1][:m]
    return hess


def build_Z(z, y, w, eps):
    """:167-184.  The reference selects ROWS y[i] of eigh's eigenvector matrix
    (SURVEY defect 4) and needs w as an (npix x m) array whose COLUMNS are the
    basis vectors (dgemm(w.T, z) = w z^T)."""
    sel = [y[i] for i in range(len(z)) if abs(z[i]) <= eps]
    r = len(sel)
    if r == 0:
        raise RuntimeError("No Ritz eigenvalue are found smaller than fixed threshold %.1g " % eps)
    zz = np.asarray(sel)
    return dgemm(np.asarray(w).T, zz), r


# --------------------------------------------------------------------------
# a16  PCG -- scipy.sparse.linalg.cg recurrence (local scipy 1.15.3
#      _isolve/iterative.py::cg; call sites tests/test_2level_preconditioner.py:52,
#      src/test_BD_precond_onto_real_data.py:47)
# --------------------------------------------------------------------------
def cg(matvec, b, x0=None, rtol=1e-5, atol=0., maxiter=None, M=None, callback=None):
    b = _f64(b)
    n = b.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    bnrm2 = np.linalg.norm(b)
    atol = max(float(atol), float(rtol) * float(bnrm2))
    if bnrm2 == 0:
        return b.copy(), 0
    if maxiter is None:
        maxiter = n * 10
    psolve = (lambda v: v.copy()) if M is None else M
    r = b - matvec(x) if x.any() else b.copy()
    rho_prev, p = None, None
    for iteration in range(maxiter):
        if np.linalg.norm(r) < atol:
            return x, 0
        z = psolve(r)
        rho_cur = np.dot(r, z)
        if iteration > 0:
            beta = rho_cur / rho_prev
            p *= beta
            p += z
        else:
            p = np.empty_like(r)
            p[:] = z[:]
        q = matvec(p)
        alpha = rho_cur / np.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho_cur
        if callback:
            callback(x)
    return x, maxiter


# --------------------------------------------------------------------------
# synthetic inputs  (utilities/utilities_functions.py:99-122, 148-212), seeded
# --------------------------------------------------------------------------
def angles_gen(theta0, n, sample_freq=200., whwp_freq=2.5):
    return np.array([theta0 + 2 * np.pi * whwp_freq / sample_freq * i for i in range(n)])


def system_setup(rng, nt, npix, nb, bandsize=2):
    """Seeded version of system_setup (:190-212): d~U[0,1), pairs uniform,
    phi = HWP ramp from theta0~U(0,pi), t = nb arrays of `bandsize` uniforms."""
    d = rng.random(nt)
    pairs = rng.integers(0, npix, size=nt).astype(np.int32)
    phi = angles_gen(rng.uniform(0, np.pi), nt)
    t = [rng.random(bandsize) for _ in range(nb)]
    diag = [ti[0] for ti in t]
    return d, pairs, phi, t, diag


# --------------------------------------------------------------------------
# f1  FilterLO  (interfaces/linearoperators.py:94-322)
# --------------------------------------------------------------------------
def filter_normalise_args(subscan_nsample, samples_per_bolopair, bolos_per_ces):
    """__init__ :263-273: scalars (one CES) are wrapped into one-element lists."""
    nsamples, nbolos = samples_per_bolopair, bolos_per_ces
    subscans, tstart = subscan_nsample[0], subscan_nsample[1]
    if not isinstance(nsamples, list):
        nsamples, nbolos, subscans, tstart = [nsamples], [nbolos], [subscans], [tstart]
    return subscans, tstart, nsamples, nbolos


def filter_segments(subscans, tstart, nsamples, nbolos):
    """(start, length) of every chunk in the visiting order of the loops at
    :134-140 / :175-182: CES -> detector pair -> sub-scan."""
    starts, lens = [], []
    offset = 0
    for subsc, ts, ns, nb in zip(subscans, tstart, nsamples, nbolos):
        for bolo_iter in range(int(nb)):
            for i, j in zip(subsc, ts):
                starts.append(int(j) + int(ns) * bolo_iter + offset)
                lens.append(int(i))
        offset += int(nb) * int(ns)
    return np.asarray(starts, dtype=np.int64), np.asarray(lens, dtype=np.int64)


def get_legendre_polynomials(polyorder, size):
    """utilities/linear_algebra_funcs.py:47-59: columns L_k(linspace(-1,1,size))/||.||_2"""
    from scipy.special import legendre
    out = np.empty([size, polyorder + 1])
    x = np.linspace(-1, 1, size)
    for i in range(polyorder + 1):
        L = legendre(i)
        out[:, i] = L(x) / norm2(L(x))
    return out


def filter_mean(d, pixs, subscan_nsample, samples_per_bolopair, bolos_per_ces):
    """FilterLO.mult :129-168 (poly_order = 0) through orc_filter_mean."""
    starts, lens = filter_segments(*filter_normalise_args(
        subscan_nsample, samples_per_bolopair, bolos_per_ces))
    d = _f64(d)
    pix = _i32(pixs)
    out = np.zeros_like(d)
    lib().orc_filter_mean(ctypes.c_int64(len(starts)), starts.ctypes.data_as(_I64),
                          lens.ctypes.data_as(_I64), pix.ctypes.data_as(_I32), _d(d), _d(out))
    return out


def filter_poly(d, pixs, subscan_nsample, samples_per_bolopair, bolos_per_ces, poly_order):
    """poly_order > 0: polyfilter_multithreads -> globalprocsfilter :286-322
    (same arithmetic as polyfilter :170-204): per chunk, with `valid` the
    unflagged samples (:301-302):
      * fewer than poly_order+1 valid samples -> chunk left at 0 (:303-304);
      * some flagged: basis = Q of qr(legendres[valid]) (:307-308), projection
        removed on the valid samples only, flagged ones stay 0 (:310-315);
      * none flagged: basis = the normalised Legendre columns themselves
        (NOT re-orthogonalised), d - sum_k <L_k,d> L_k on the chunk (:317-321)."""
    subscans, tstart, nsamples, nbolos = filter_normalise_args(
        subscan_nsample, samples_per_bolopair, bolos_per_ces)
    starts, lens = filter_segments(subscans, tstart, nsamples, nbolos)
    d = _f64(d)
    valid = np.asarray(pixs) >= 0                                    # :256
    legendres = {}
    for n in lens:                                                   # :206-213
        if int(n) not in legendres:
            legendres[int(n)] = get_legendre_polynomials(poly_order, int(n))
    out = d * 0.
    for a, n in zip(starts, lens):
        b = a + n
        m = valid[a:b]
        size = int(np.count_nonzero(m))
        if size <= poly_order:
            continue
        basis = legendres[int(n)]
        if size != n:
            basis, _ = np.linalg.qr(basis[m])
            dd = d[a:b][m]
        else:
            dd = d[a:b]
        p = np.zeros(size)
        for k in range(poly_order + 1):
            fb = basis[:, k]
            p += scalprod(fb, dd) * fb
        if size != n:
            seg = out[a:b]
            seg[m] = dd - p
        else:
            out[a:b] = dd - p
    return out


# --------------------------------------------------------------------------
# f2  GroundFilterLO  (interfaces/linearoperators.py:24-61)
# --------------------------------------------------------------------------
def ground_filter(ground, v):
    """v - G (G^T G)^-1 G^T v with G the pol=1 pointing of the ground-bin ids
    (:52-60): nbins = max(ground)+1 (:50), bins nobody hit invert to 0 through
    BlockDiagonalPreconditionerLO's pol=1 branch (:788-790)."""
    ground = np.asarray(ground)
    nbins = int(ground.max()) + 1
    ok = ground >= 0
    hits = np.bincount(ground[ok], minlength=nbins).astype(np.float64)
    gtv = np.bincount(ground[ok], weights=np.asarray(v, dtype=np.float64)[ok], minlength=nbins)
    inv = np.zeros(nbins)
    nz = hits != 0
    inv[nz] = 1. / hits[nz]
    binned = inv * gtv
    back = np.where(ok, binned[np.where(ok, ground, 0)], 0.)
    return v - back


# --------------------------------------------------------------------------
# f3  map vector <-> full-sky HEALPix maps
# --------------------------------------------------------------------------
def reorganize_map(mapin, obspix, npix, nside, pol):
    """utilities/healpy_functions.py:47-102 without the FITS output: list of pol full-sky
    maps of 12*nside^2 pixels (hp.nside2npix), component k of observed pixel i at obspix[i]."""
    nfull = 12 * nside * nside
    mapin = np.asarray(mapin)
    out = []
    for k in range(pol):                                 # :80-100: i=mapin[::3], q=mapin[1::3] ...
        m = np.zeros(nfull)
        m[obspix] = mapin[k::pol]
        out.append(m)
    return out


def full2cutskymap(hp_map, pol, npix, observpix):
    """utilities/IOfiles.py:377-393."""
    x = np.zeros(pol * npix)
    obsmap = [np.asarray(m)[observpix] for m in hp_map]
    if pol == 1:
        return obsmap                                    # a list, as the reference returns it
    for i in range(npix):
        x[pol * i:pol * (i + 1)] = [obsmap[k][i] for k in range(pol)]
    return x


# --------------------------------------------------------------------------
# all-cores host baseline of P^T N^-1 P (SURVEY 8d "fair host baseline"): OpenMP pointing
# loops (cm2_oracle_omp.c) + FFT convolution per noise block on a thread pool.
# bench.py's cpu_baseline_all_cores leg only.
# --------------------------------------------------------------------------
_SO_OMP = os.path.join(_HERE, "_build", "libcm2_oracle_omp.so")
_SRC_OMP = os.path.join(_HERE, "cm2_oracle_omp.c")
_lib_omp = None


def build_omp(force=False):
    os.makedirs(os.path.dirname(_SO_OMP), exist_ok=True)
    if (not force and os.path.exists(_SO_OMP)
            and os.path.getmtime(_SO_OMP) >= os.path.getmtime(_SRC_OMP)):
        return _SO_OMP
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-o", _SO_OMP, _SRC_OMP])
    return _SO_OMP


def lib_omp():
    global _lib_omp
    if _lib_omp is None:
        build_omp()
        _lib_omp = ctypes.CDLL(_SO_OMP)
    return _lib_omp


class AllCoresMatvec(object):
    """(P^T N^-1 P) x on `threads` host threads for a pol-interleaved map: P and P^T through
    the OpenMP loops (interfaces/linearoperators.py:483-489, :509-516), N^-1 per noise block as
    interfaces/blkop.py:195-206 dispatches it: banded-Toeplitz blocks (zero boundary,
    linearoperators.py:582-595) as scipy.signal.fftconvolve 'same' with the symmetric 2*lambda-1
    kernel, one block per pool thread, or -- `bands` None and `diag` a per-sample weight vector --
    the diagonal blocks of BlockLO(offdiag=False) as one multiply.  `blocksize`: one int or the
    per-block sizes."""

    def __init__(self, pol, npix, pix, cos, sin, blocksize, bands, threads, diag=None, piece=1 << 20):
        """piece: a noise block longer than this is convolved in pieces of that many outputs, each read
        with a halo of lambda - 1 samples from INSIDE the block (nothing beyond the block's ends: the
        zero boundary of linearoperators.py:592-593) -- the same sums, and every host thread has work
        when a shard holds a few long detector blocks (C5: 8 blocks of 15 625 000 samples)."""
        from concurrent.futures import ThreadPoolExecutor
        self.piece = int(piece)
        self.pol, self.npix, self.threads = pol, int(npix), int(threads)
        self.pix, self.cos, self.sin = _i32(pix), _f64(cos), _f64(sin)
        self.nt = self.pix.size
        self.diag = None if diag is None else _f64(diag)
        if bands is not None:
            self.sizes = block_sizes(blocksize, len(bands))
            self.offs = np.concatenate([[0], np.cumsum(self.sizes)])
            self.kernels = [np.concatenate([np.asarray(b)[:0:-1], np.asarray(b)]) for b in bands]
        else:
            self.kernels = None
        # P^T: one private map per thread, summed afterwards -- at most ~4 GB of them (at nside 512 a map is
        # 75 MB: 256 private maps would cost far more to clear and to sum than the scatter itself takes)
        self.pt_threads = max(1, min(self.threads, int(4e9 // (8 * pol * max(self.npix, 1))) or 1))
        self.scratch = np.empty(self.pt_threads * pol * self.npix)
        self.pool = ThreadPoolExecutor(min(self.threads, 64))

    def P(self, x):
        tod = np.empty(self.nt)
        lib_omp().orc_omp_P_apply(self.pol, ctypes.c_int64(self.nt), self.pix.ctypes.data_as(_I32),
                                  _d(self.cos), _d(self.sin), _d(_f64(x)), _d(tod), self.threads)
        return tod

    def N(self, tod):
        if self.kernels is None:
            return self.diag * tod
        from scipy.signal import fftconvolve
        out_tod = np.empty(self.nt)

        def one(job):
            b, p0, p1, a, e = job
            halo = (len(self.kernels[b]) - 1) // 2                       # lambda - 1
            lo, hi = max(a, p0 - halo), min(e, p1 + halo)
            out_tod[p0:p1] = fftconvolve(tod[lo:hi], self.kernels[b], mode="same")[p0 - lo:p1 - lo]
        jobs = []
        for b in range(len(self.kernels)):
            a, e = int(self.offs[b]), min(int(self.offs[b + 1]), self.nt)
            for p0 in range(a, e, self.piece):
                jobs.append((b, p0, min(p0 + self.piece, e), a, e))
        list(self.pool.map(one, jobs))
        return out_tod

    def Pt(self, tod):
        out = np.empty(self.pol * self.npix)
        lib_omp().orc_omp_Pt_apply(self.pol, ctypes.c_int64(self.nt), ctypes.c_int64(self.npix),
                                   self.pix.ctypes.data_as(_I32), _d(self.cos), _d(self.sin),
                                   _d(_f64(tod)), _d(out), _d(self.scratch), self.pt_threads)
        return out

    def __call__(self, x):
        return self.Pt(self.N(self.P(x)))


# --------------------------------------------------------------------------
# the whole solve on the host at a BASELINE configuration's full size (tests/test_gpu_fullsize.py,
# bench.py's parity_full_size block): ProcessTimeSamples by the serial loops above, P^T N^-1 P by
# AllCoresMatvec (checked against the serial loops in tests/test_oracle_golden.py), M_BD by the
# serial per-pixel loop, PCG by `cg` above (scipy's recurrence).
# --------------------------------------------------------------------------
class HostProblem(object):
    """CPU restatement of `A x = b`, A = P^T N^-1 P, b = P^T N^-1 d, M = M_BD for host copies of
    the inputs: pix (int32, NOT yet flagged: flagged in place like the reference does,
    process_ces.py:416), phi, d, and either `bands` (one first row per noise block, offdiag=True)
    or `diag` (one weight per block, offdiag=False).  Call sites restated:
    src/test_BD_precond_onto_real_data.py:31-47 (ProcessTimeSamples -> SparseLO -> BlockLO ->
    BlockDiagonalPreconditionerLO -> cg)."""

    def __init__(self, pol, npix, pix, phi, blocksize, bands=None, diag=None, threads=None):
        if threads is None:
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count()
        self.pol, self.threads = pol, int(threads)
        pix = _i32(pix)
        nt = pix.shape[0]
        w = None
        if bands is None:
            w = blocklo_diag(blocksize, list(diag))            # BlockLO.diag feeds the weights (:677-683)
        self.ro = process_time_samples(pix, npix, pol=pol, phi=phi, w=w)
        self.n = self.ro.new_npix
        self.A = AllCoresMatvec(pol, self.n, pix, self.ro.cos, self.ro.sin, blocksize, bands, self.threads,
                                diag=w)
        self.nt = nt

    def M(self, v):
        return bd_precond_mult(self.pol, self.ro, v)

    def rhs(self, d):
        return self.A.Pt(self.A.N(_f64(d)))

    def solve(self, b, rtol=1e-6, maxiter=500, M=None):
        """(x, info, iterations counted as callback invocations -- the way the reference's scripts
        count them, src/test_BD_precond_onto_real_data.py:41-47).  M: another preconditioner
        (default M_BD), e.g. the one `two_level` returns."""
        its = []
        x, info = cg(self.A, b, rtol=rtol, maxiter=maxiter, M=self.M if M is None else M,
                     callback=lambda xk: its.append(1))
        return x, info, len(its)

    def two_level(self, Z, apply='eig'):
        """The reference's two-level build for a GIVEN deflation basis Z (n x r), all on the host
        (src/test_M2_precond_onto_real_data.py:96-112): Az[:, i] = A * Z[:, i] -- r applications of
        this problem's own A --, E = CoarseLO(Z, Az, r, apply) (interfaces/linearoperators.py:
        1018-1027, `Coarse` above), M2 = Mbd*R + Zd*E*Zd.T with R = I - AZd*E*Zd.T (`m2_apply`,
        with DeflationLO.mult / rmult of :1041-1056).  Returns (Az, Coarse, M2 as a callable)."""
        Z = np.asfortranarray(Z, dtype=np.float64)                 # contiguous columns, like the z list
        r = Z.shape[1]
        Az = np.empty_like(Z)
        for i in range(r):                                         # :98-101
            Az[:, i] = self.A(Z[:, i])
        co = Coarse(Z, Az, r, apply=apply)
        return Az, co, (lambda v: m2_apply(self.M, Z, Az, co, v))
