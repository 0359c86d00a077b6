#!/usr/bin/env python3
"""
Generate golden input/output vectors by executing the REFERENCE's own function
bodies in this container.  Run once here (needs /root/reference, which does not
exist on the GPU box); only the resulting data file
tests/golden/reference_golden.npz is committed -- no reference text is copied.

How: the reference is Python 2.  Each source file is read from /root/reference,
converted in memory by the standard library's lib2to3 (print statements, xrange,
...), and individual function / method definitions are compiled on their own
and called with plain data (a SimpleNamespace for `self`).  Only bodies that
need nothing absent from this image are run: no stand-ins are written for
weave, linop or krypy, so the weave loops and the krypy wrappers are NOT covered
here (see oracle/cm2_oracle.c header for their pinning status).

Usage:  python tests/golden/make_golden.py
"""
import ast
import io
import os
import random
import sys
import warnings
from contextlib import redirect_stdout
from types import SimpleNamespace

import numpy as np
import scipy.linalg
import scipy.special

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_golden.npz")


def converted_source(relpath):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        rt = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
        src = open(os.path.join(REF, relpath)).read().expandtabs(8) + "\n"
        return str(rt.refactor_string(src, relpath))


def extract(relpath, names, glob):
    """Compile the named top-level functions ('f') or methods ('Class.m') of a
    reference file into `glob` and return them."""
    tree = ast.parse(converted_source(relpath))
    found = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = node
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                key = "%s.%s" % (node.name, getattr(sub, "name", ""))
                if isinstance(sub, ast.FunctionDef) and key in names:
                    found[key] = sub
    out = {}
    for key, node in found.items():
        mod = ast.Module(body=[node], type_ignores=[])
        ns = dict(glob)
        exec(compile(mod, "%s:%s" % (relpath, key), "exec"), ns)
        out[key] = ns[node.name]
        # make top-level helpers visible to later-extracted functions
        if "." not in key:
            glob[node.name] = ns[node.name]
    missing = set(names) - set(out)
    assert not missing, missing
    return out


def quiet(f, *a, **k):
    with redirect_stdout(io.StringIO()):
        return f(*a, **k)


def main():
    G = {}
    rng = np.random.default_rng(20161202)

    # ---- utilities/linear_algebra_funcs.py:16-44 ------------------------------
    base = {"np": np, "get_blas_funcs": scipy.linalg.get_blas_funcs}
    la = extract("utilities/linear_algebra_funcs.py", ["dgemm", "norm2", "scalprod"], base)
    A = rng.standard_normal((7, 4))
    B = rng.standard_normal((5, 7))
    G["la_A"], G["la_B"] = A, B
    G["la_dgemm"] = la["dgemm"](A, B)                      # = A^T B^T
    q = rng.standard_normal(33)
    q2 = rng.standard_normal(33)
    G["la_q"], G["la_q2"] = q, q2
    G["la_norm2"] = np.float64(la["norm2"](q))
    G["la_scalprod"] = np.float64(la["scalprod"](q, q2))

    # ---- utilities/utilities_functions.py:99-122, 148-212 ----------------------
    ubase = {"np": np, "rd": random, "m": __import__("math"), "warnings": warnings,
             "get_blas_funcs": scipy.linalg.get_blas_funcs}
    uf = extract("utilities/utilities_functions.py",
                 ["angles_gen", "pairs_gen", "noise_val", "system_setup", "is_sorted",
                  "checking_output"], ubase)
    G["gen_angles"] = uf["angles_gen"](0.3, 50)
    np.random.seed(7)
    random.seed(7)
    d, pairs, phi, t, diag = uf["system_setup"](120, 17, 3)
    G["gen_seed"] = np.int64(7)
    G["gen_d"], G["gen_pairs"], G["gen_phi"] = d, np.asarray(pairs), phi
    G["gen_t"], G["gen_diag"] = np.asarray(t), np.asarray(diag)
    raised = 0
    try:
        uf["pairs_gen"](10, 2)
    except RuntimeError:
        raised = 1
    G["gen_pairs_small_raises"] = np.int64(raised)
    try:
        uf["checking_output"](3)
        G["checking_output_pos_raises"] = np.int64(0)
    except RuntimeError:
        G["checking_output_pos_raises"] = np.int64(1)
    G["checking_output_zero"] = np.int64(bool(uf["checking_output"](0)))

    # bash_colors class is needed (as data holder of escape codes) by later bodies
    tree = ast.parse(converted_source("utilities/utilities_functions.py"))
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "bash_colors":
            ns = {}
            exec(compile(ast.Module(body=[node], type_ignores=[]), "bash_colors", "exec"), ns)
            bash_colors = ns["bash_colors"]

    # ---- interfaces/linearoperators.py -----------------------------------------
    lbase = {"np": np, "dgemm": la["dgemm"], "norm2": la["norm2"], "scalprod": la["scalprod"],
             "solve": scipy.linalg.solve, "lu": scipy.linalg.lu, "eigh": scipy.linalg.eigh,
             "bash_colors": bash_colors}
    lo = extract("interfaces/linearoperators.py",
                 ["ToeplitzLO.mult", "BlockDiagonalLO.mult",
                  "BlockDiagonalPreconditionerLO.mult",
                  "DeflationLO.mult", "DeflationLO.rmult",
                  "CoarseLO.mult", "CoarseLO.mult_eig",
                  "CoarseLO.setting_inverse_w_eigenvalues"], lbase)

    # ToeplitzLO.mult :582-595
    v = rng.standard_normal(200)
    G["toep_v"] = v
    for lam in (1, 2, 33):
        a = rng.random(lam)
        G["toep_a%d" % lam] = a
        G["toep_y%d" % lam] = lo["ToeplitzLO.mult"](SimpleNamespace(array=a), v.copy())
    vs = rng.standard_normal(5)           # band longer than the block
    a9 = rng.random(9)
    G["toep_vshort"], G["toep_a9"] = vs, a9
    G["toep_yshort"] = lo["ToeplitzLO.mult"](SimpleNamespace(array=a9), vs.copy())

    # per-pixel weight arrays for the block operators
    npx = 23
    W = {k: rng.random(npx) + 0.1 for k in ("counts", "cos", "sin", "cos2", "sin2", "sincos")}
    W["counts"] = W["counts"] * 40 + 3
    for k, val in W.items():
        G["bd_" + k] = val
    # BlockDiagonalLO.mult :728-746
    for pol in (1, 2, 3):
        x = rng.standard_normal(pol * npx)
        me = SimpleNamespace(pol=pol, pixels=np.arange(npx), counts=W["counts"], cos=W["cos"],
                             sin=W["sin"], cos2=W["cos2"], sin2=W["sin2"], sincos=W["sincos"])
        G["bd_x%d" % pol] = x
        G["bd_y%d" % pol] = lo["BlockDiagonalLO.mult"](me, x.copy())
    # BlockDiagonalPreconditionerLO.mult, pol=1 branch :788-790 (pol 2/3 need weave)
    cnt = W["counts"].copy()
    cnt[[2, 11]] = 0.0
    x = rng.standard_normal(npx)
    me = SimpleNamespace(pol=1, size=npx, counts=cnt)
    G["bdp1_counts"], G["bdp1_x"] = cnt, x
    G["bdp1_y"] = lo["BlockDiagonalPreconditionerLO.mult"](me, x.copy())

    # DeflationLO :1041-1065 (z = list of columns, as __init__ :1059-1062 builds it)
    Z = rng.standard_normal((31, 4))
    me = SimpleNamespace(z=[Z[:, j] for j in range(4)], nrows=31, ncols=4)
    yv = rng.standard_normal(4)
    xv = rng.standard_normal(31)
    G["defl_Z"], G["defl_y"], G["defl_x"] = Z, yv, xv
    G["defl_Zy"] = lo["DeflationLO.mult"](me, yv)
    G["defl_Ztx"] = lo["DeflationLO.rmult"](me, xv)

    # CoarseLO :969-1027.  __init__ needs linop, so its two arithmetic lines are
    # replayed here: M=dgemm(Z,Az.T) (:1019) and lu(M,permute_l=True,...) (:1025).
    n, r = 31, 4
    S = rng.standard_normal((n, n))
    Aspd = S.dot(S.T) + n * np.eye(n)
    Az = Aspd.dot(Z)
    M = la["dgemm"](Z, Az.T)
    vv = rng.standard_normal(r)
    G["coarse_A"], G["coarse_v"], G["coarse_E"] = Aspd, vv, M.copy()
    me = SimpleNamespace()
    me.L, me.U = scipy.linalg.lu(M.copy(), permute_l=True, overwrite_a=True, check_finite=False)
    G["coarse_lu_x"] = lo["CoarseLO.mult"](me, vv.copy())
    me = SimpleNamespace()
    quiet(lo["CoarseLO.setting_inverse_w_eigenvalues"], me, M.copy())
    G["coarse_invE"] = me.invE
    G["coarse_eig_x"] = lo["CoarseLO.mult_eig"](me, vv.copy())
    # a rank-deficient E: duplicate deflation vector -> one eigenvalue dropped (:997-999)
    Zd = Z.copy()
    Zd[:, 3] = Zd[:, 0]
    Md = la["dgemm"](Zd, Aspd.dot(Zd).T)
    me = SimpleNamespace()
    quiet(lo["CoarseLO.setting_inverse_w_eigenvalues"], me, Md.copy())
    G["coarse_deg_E"], G["coarse_deg_invE"] = Md, me.invE

    # ---- utilities/process_ces.py:351-401  (pure-Python repixelization) --------
    pbase = {"np": np}
    pc = extract("utilities/process_ces.py", ["ProcessTimeSamples.repixelization"], pbase)
    for pol in (1, 2, 3):
        nold = 19
        arrs = {k: rng.random(nold) for k in ("counts", "cosine", "sine", "cos2", "sin2", "sincos")}
        mask = np.sort(rng.choice(nold, size=11, replace=False))
        me = SimpleNamespace(pol=pol, oldnpix=nold, mask=mask, obspix=np.arange(100, 100 + nold),
                             bashc=bash_colors(), nsamples=77,
                             **{k: val.copy() for k, val in arrs.items()})
        quiet(pc["ProcessTimeSamples.repixelization"], me)
        G["repix%d_mask" % pol] = mask
        for k, val in arrs.items():
            G["repix%d_in_%s" % (pol, k)] = val
        G["repix%d_old2new" % pol] = me.old2new
        G["repix%d_npix" % pol] = np.int64(getattr(me, "__new_npix"))
        G["repix%d_obspix" % pol] = me.obspix
        keys = {1: ("counts",), 2: ("cos2", "sin2", "sincos"),
                3: ("counts", "cosine", "sine", "cos2", "sin2", "sincos")}[pol]
        for k in keys:
            G["repix%d_out_%s" % (pol, k)] = getattr(me, k)

    # ---- interfaces/deflationlib.py:17-184 -------------------------------------
    dbase = {"np": np, "get_blas_funcs": scipy.linalg.get_blas_funcs,
             "norm2": la["norm2"], "dgemm": la["dgemm"]}
    dl = extract("interfaces/deflationlib.py", ["arnoldi", "build_hess", "build_Z"], dbase)
    n = 30
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    evals = np.repeat([0.004, 0.3, 1.0, 2.5, 4.0, 9.0], 5)     # 6 distinct -> Krylov dim 6
    Aar = (Q * evals).dot(Q.T)
    Aar = 0.5 * (Aar + Aar.T)
    b = rng.standard_normal(n)
    x0 = np.zeros(n)
    op = SimpleNamespace(matvec=lambda x: Aar.dot(x))
    vs_, hs_, j = quiet(dl["arnoldi"], op, b, x0=x0, tol=1e-8, inner_m=n)
    G["arn_A"], G["arn_b"] = Aar, b
    G["arn_j"] = np.int64(j)
    G["arn_V"] = np.asarray(vs_)
    G["arn_hlast"] = np.asarray(hs_[-1])
    H = dl["build_hess"](hs_, j)
    G["arn_H"] = H
    zz, yy = np.linalg.eigh(H)
    G["arn_ritz"] = zz
    # build_Z does dgemm(w.T, z): it needs w as an (npix x m) ARRAY whose columns are
    # the basis vectors (a plain list, as arnoldi returns, has no .T -- SURVEY defect 4)
    Zb, rb = quiet(dl["build_Z"], zz, yy, np.asarray(vs_).T.copy(), 1e-2)
    G["arn_Z"], G["arn_r"] = np.asarray(Zb), np.int64(rb)
    try:
        quiet(dl["arnoldi"], op, b, x0=x0, tol=1e-8, inner_m=3)
        G["arn_raises_at_inner_m"] = np.int64(0)
    except RuntimeError:
        G["arn_raises_at_inner_m"] = np.int64(1)
    try:
        quiet(dl["arnoldi"], op, b * np.nan, x0=x0)
        G["arn_nonfinite_raises"] = np.int64(0)
    except ValueError:
        G["arn_nonfinite_raises"] = np.int64(1)
    out0 = quiet(dl["arnoldi"], op, Aar.dot(np.ones(n)), x0=np.ones(n), tol=1e-5)
    G["arn_zero_residual_returns_j0"] = np.int64(out0[2] == 0 and out0[0] is None)

    # ---- FilterLO, poly_order>0: interfaces/linearoperators.py:170-213 and
    #      utilities/linear_algebra_funcs.py:47-59.  polyfilter is the serial twin of
    #      the Pool worker globalprocsfilter (:286-322, needs Python 2's time.clock);
    #      the poly_order=0 path (:129-168) is a weave loop and is not run here.
    rngf = np.random.default_rng(20161203)
    la2 = extract("utilities/linear_algebra_funcs.py", ["get_legendre_polynomials"],
                  {"np": np, "legendre": scipy.special.legendre, "norm2": la["norm2"]})
    G["leg_3_17"] = la2["get_legendre_polynomials"](3, 17)
    G["leg_1_2"] = la2["get_legendre_polynomials"](1, 2)
    fbase = {"np": np, "scalprod": la["scalprod"],
             "get_legendre_polynomials": la2["get_legendre_polynomials"]}
    fl = extract("interfaces/linearoperators.py",
                 ["FilterLO.polyfilter", "FilterLO.compute_legendres"], fbase)
    subscans = [np.array([40, 55, 40, 50]), np.array([60, 30, 50, 3])]
    tstart = [np.array([3, 45, 102, 145]), np.array([0, 62, 95, 146])]
    nsamples, nbolos = [200, 150], [3, 2]
    ntf = 200 * 3 + 150 * 2
    pixf = rngf.integers(0, 50, size=ntf).astype(np.int64)
    pixf[rngf.random(ntf) < 0.12] = -1
    pixf[3:43] = -1                              # a fully flagged chunk
    pixf[200 + 45:200 + 100] = -1                # a chunk with 2 valid samples only
    pixf[200 + 50] = 7
    pixf[200 + 77] = 9
    pixf[400 + 102:400 + 142] = np.abs(pixf[400 + 102:400 + 142])   # chunks without any flag
    pixf[600:660] = np.abs(pixf[600:660])
    df = rngf.standard_normal(ntf) + 0.01 * np.arange(ntf)
    G["filt_subscan0"], G["filt_subscan1"] = subscans
    G["filt_tstart0"], G["filt_tstart1"] = tstart
    G["filt_nsamples"], G["filt_nbolos"] = np.array(nsamples), np.array(nbolos)
    G["filt_pix"], G["filt_d"] = pixf, df
    for order in (1, 2, 3):
        me = SimpleNamespace(subscans=subscans, tstart=tstart, nsamples=nsamples, nbolos=nbolos,
                             pixels=pixf, poly_order=order)
        fl["FilterLO.compute_legendres"](me)
        G["filt_out%d" % order] = fl["FilterLO.polyfilter"](me, df.copy())

    # ---- utilities/IOfiles.py:377-393 full2cutskymap (pure NumPy; reorganize_map needs healpy)
    io = extract("utilities/IOfiles.py", ["full2cutskymap"], {"np": np})
    rngm = np.random.default_rng(20161204)
    nfull = 12 * 4 * 4
    obs = np.sort(rngm.choice(nfull, size=37, replace=False))
    full = [rngm.standard_normal(nfull) for _ in range(3)]
    G["cut_obspix"] = obs
    G["cut_full"] = np.asarray(full)
    for pol in (1, 2, 3):
        res = io["full2cutskymap"](full[:pol], pol, obs.size, obs)
        G["cut_out%d" % pol] = np.asarray(res[0] if pol == 1 else res)
    G["cut_pol1_returns_list"] = np.int64(isinstance(io["full2cutskymap"](full[:1], 1, obs.size, obs), list))

    np.savez_compressed(OUT, **G)
    print("wrote %s (%d arrays, %d bytes)" % (OUT, len(G), os.path.getsize(OUT)))


if __name__ == "__main__":
    sys.exit(main())
