"""
CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares (no compute calls), the operator algebra that stands in for `linop`, the seeded
generators against the reference's streams, and the "no CPU fallback" contract.
"""
import os
import random
import re
import sys

import numpy as np
import pytest
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "cosmomap2.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cm2_[a-zA-Z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from cosmomap2_amd import _hip
    lib = _hip.load()                     # dlopen only; no GPU call
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), "libcosmomap2_hip.so lacks %s" % s
    # the ctypes table binds exactly the declared interface
    assert sorted(_hip.PROTOTYPES) == syms
    assert lib.cm2_abi_version() == 2
    assert lib.cm2_last_error() is not None
    assert lib.cm2_reduce_work_doubles() > 0 and lib.cm2_gemm_tn_work_doubles(32, 32) > 0


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cosmomap2_amd import _hip, cg
    from cosmomap2_amd.interfaces import (SparseLO, BlockLO, ToeplitzLO, DeflationLO, FilterLO,
                                          GroundFilterLO)
    from cosmomap2_amd.utilities import ProcessTimeSamples, norm2, reorganize_map
    for make in (lambda: SparseLO(10, 20, np.zeros(20, dtype=np.int32)),
                 lambda: BlockLO(10, [1.0, 2.0]),
                 lambda: ToeplitzLO(np.ones(2), 10),
                 lambda: DeflationLO(np.ones((10, 2))),
                 lambda: FilterLO(20, [np.array([5]), np.array([0])], 10, 2,
                                  np.zeros(20, dtype=np.int32)),
                 lambda: GroundFilterLO(np.zeros(20, dtype=np.int32)),
                 lambda: ProcessTimeSamples(np.zeros(20, dtype=np.int32), 10),
                 lambda: norm2(np.ones(4)),
                 lambda: reorganize_map(np.ones(4), np.arange(4), 4, 2, 1),
                 lambda: cg(np.eye(3), np.ones(3))):
        with pytest.raises(_hip.HipError):
            make()
    with pytest.raises(RuntimeError):                       # bad pol is checked first (:549)
        SparseLO(10, 20, np.zeros(20, dtype=np.int32), pol=7)


def test_linop_algebra():
    from cosmomap2_amd import linop as lp
    rng = np.random.default_rng(0)
    Am, Bm = rng.standard_normal((5, 4)), rng.standard_normal((4, 6))
    A = lp.LinearOperator(4, 5, lambda x: Am @ x, rmatvec=lambda y: Am.T @ y)
    B = lp.LinearOperator(6, 4, lambda x: Bm @ x, rmatvec=lambda y: Bm.T @ y)
    x, y = rng.standard_normal(6), rng.standard_normal(5)
    assert A.shape == (5, 4) and A.nargin == 4 and A.nargout == 5 and A.dtype == np.float64
    np.testing.assert_allclose((A * B) * x, Am @ Bm @ x)
    np.testing.assert_allclose(A * B * x, Am @ Bm @ x)
    np.testing.assert_allclose((A * B).T * y, Bm.T @ Am.T @ y)
    np.testing.assert_allclose((A * B).H * y, Bm.T @ Am.T @ y)
    np.testing.assert_allclose(A.T.T * x[:4], Am @ x[:4])
    assert A.T.T is A
    np.testing.assert_allclose((2.5 * A) * x[:4], 2.5 * Am @ x[:4])
    np.testing.assert_allclose((A * 2.5) * x[:4], 2.5 * Am @ x[:4])
    np.testing.assert_allclose((A / 2) * x[:4], 0.5 * Am @ x[:4])
    np.testing.assert_allclose((-A) * x[:4], -Am @ x[:4])
    np.testing.assert_allclose((A + A) * x[:4], 2 * Am @ x[:4])
    np.testing.assert_allclose((A - 3 * A).T * y, -2 * Am.T @ y)
    np.testing.assert_allclose(A.to_array(), Am)
    np.testing.assert_allclose(A.matmat(Bm), Am @ Bm)
    with pytest.raises(lp.ShapeError):
        A * B.T
    with pytest.raises(lp.ShapeError):
        A * np.ones(7)
    with pytest.raises(lp.ShapeError):
        A + B
    S = lp.LinearOperator(3, 3, lambda v: 2 * v, symmetric=True)
    assert S.T is S and S.symmetric
    I = lp.IdentityOperator(3)
    Dg = lp.DiagonalOperator(np.array([1., 2., 3.]))
    v = np.array([1., 1., 1.])
    np.testing.assert_allclose((I - S * Dg) * v, [-1., -3., -5.])
    np.testing.assert_allclose(lp.ZeroOperator(3, 2) * v, [0., 0.])
    noT = lp.LinearOperator(3, 3, lambda v: v)
    with pytest.raises(NotImplementedError):
        noT.T
    # accepted by scipy
    xs, info = spla.cg(S + Dg, v, rtol=1e-12)
    assert info == 0
    np.testing.assert_allclose(xs, [1 / 3., 1 / 4., 1 / 5.])
    w = spla.eigsh(spla.aslinearoperator(S + Dg), k=1, which='LA')[0]
    np.testing.assert_allclose(w, [5.0])
    assert S.nMatvec > 0
    assert not lp.supports_device(A) and lp.supports_device(I) and lp.supports_device(I * Dg)


def test_block_diagonal_container():
    from cosmomap2_amd import linop as lp
    from cosmomap2_amd.interfaces.blkop import BlockDiagonalLinearOperator
    A = lp.DiagonalOperator(np.array([1., 2.]))
    Bm = np.array([[1., 2., 0.], [0., 1., 0.], [3., 0., 1.]])
    B = lp.LinearOperator(3, 3, lambda x: Bm @ x, rmatvec=lambda y: Bm.T @ y)
    K = BlockDiagonalLinearOperator([A, B])
    assert K.shape == (5, 5) and not K.symmetric and len(K.blocks) == 2 and K[1] is B
    x = np.arange(1., 6.)
    np.testing.assert_allclose(K * x, np.concatenate([[1., 4.], Bm @ x[2:]]))
    np.testing.assert_allclose(K.T * x, np.concatenate([[1., 4.], Bm.T @ x[2:]]))
    with pytest.raises(lp.ShapeError):
        K * np.ones(4)
    with pytest.raises(ValueError):
        BlockDiagonalLinearOperator([1, 2])


def test_generators_and_helpers_match_reference(golden):
    from cosmomap2_amd.utilities import utilities_functions as U
    from cosmomap2_amd.interfaces.deflationlib import build_hess
    G = golden
    np.testing.assert_array_equal(U.angles_gen(0.3, 50), G["gen_angles"])
    np.random.seed(int(G["gen_seed"]))
    random.seed(int(G["gen_seed"]))
    d, pairs, phi, t, diag = U.system_setup(120, 17, 3)
    np.testing.assert_array_equal(d, G["gen_d"])
    np.testing.assert_array_equal(pairs, G["gen_pairs"])
    np.testing.assert_array_equal(phi, G["gen_phi"])
    np.testing.assert_array_equal(np.asarray(t), G["gen_t"])
    np.testing.assert_array_equal(np.asarray(diag), G["gen_diag"])
    with pytest.raises(RuntimeError):
        U.pairs_gen(10, 2)                                  # utilities_functions.py:117-118
    assert int(G["gen_pairs_small_raises"]) == 1
    assert U.checking_output(0) is True and int(G["checking_output_zero"]) == 1
    with pytest.raises(RuntimeError):
        U.checking_output(3)
    with pytest.raises(RuntimeError):
        U.checking_output(-1)
    assert U.is_sorted([1, 2, 2, 5]) and not U.is_sorted([2, 1])
    assert U.bash_colors().bold("x") == "\033[1mx\033[0m"
    np.testing.assert_array_equal(U.subscan_resize(np.arange(10), [[2, 3], [1, 6]]),
                                  [1, 2, 6, 7, 8])
    # build_hess is pure host code: pinned by the reference-executed Hessenberg matrix
    H = G["arn_H"]
    j = int(G["arn_j"])
    cols = [H[:q + 2, q].copy() for q in range(j - 1)] + [np.concatenate([H[:, j - 1], [0.0]])]
    np.testing.assert_array_equal(build_hess(cols, j), H)


def _stieltjes_basis(x, K):
    """NumPy statement of what k_filter_setup builds per flagged chunk: the orthonormal
    polynomials of the point set x by the three-term recurrence."""
    P, nrm, alpha, beta = [np.ones(x.size)], [float(x.size)], [], [0.0]
    for k in range(K - 1):
        alpha.append(np.dot(x * P[k], P[k]) / nrm[k])
        nxt = (x - alpha[k]) * P[k] - (beta[k] * P[k - 1] if k else 0.0)
        P.append(nxt)
        nrm.append(np.dot(nxt, nxt))
        beta.append(nrm[k + 1] / nrm[k])
    return np.array([p / np.sqrt(n) for p, n in zip(P, nrm)]).T


@pytest.mark.parametrize("order", [1, 2, 3])
def test_filter_plan_reproduces_reference_polyfilter(golden, order):
    """The set-up FilterLO hands to the library (sorted chunks, shared Legendre tables) and
    the kernel's formula (table columns for chunks without flags, recurrence-built
    orthonormal polynomials for the others), evaluated with NumPy, against the reference's
    polyfilter output."""
    from cosmomap2_amd.interfaces.linearoperators import filter_plan
    from cosmomap2_amd.utilities.linear_algebra_funcs import get_legendre_polynomials
    subs = [golden["filt_subscan0"], golden["filt_subscan1"]]
    ts = [golden["filt_tstart0"], golden["filt_tstart1"]]
    ns, nb = list(golden["filt_nsamples"]), list(golden["filt_nbolos"])
    d, valid = golden["filt_d"], golden["filt_pix"] >= 0
    K = order + 1
    leg = {int(n): get_legendre_polynomials(order, int(n)) for a in subs for n in a}
    st, ln, toff, table = filter_plan(subs, ts, ns, nb, order, leg)
    assert (np.diff(st) > 0).all() and st.size == 3 * 4 + 2 * 4
    out = np.zeros_like(d)
    for s in range(st.size):
        a, n = st[s], ln[s]
        m = valid[a:a + n]
        T = table[toff[s]:toff[s] + n * K].reshape(n, K)
        np.testing.assert_array_equal(T, leg[int(n)])
        if m.sum() <= order:
            continue
        if m.all():
            out[a:a + n] = d[a:a + n] - T @ (T.T @ d[a:a + n])
            continue
        j = np.nonzero(m)[0]
        Q = _stieltjes_basis((2.0 * j - (j[0] + j[-1])) / (j[-1] - j[0]), K)
        np.testing.assert_allclose(Q.T @ Q, np.eye(K), atol=1e-13)
        out[a:a + n][m] = d[a:a + n][m] - Q @ (Q.T @ d[a:a + n][m])
    np.testing.assert_allclose(out, golden["filt_out%d" % order], rtol=0, atol=1e-12)


def test_filter_plan_order0_and_ces_offsets():
    from cosmomap2_amd.interfaces.linearoperators import filter_plan
    st, ln, toff, table = filter_plan(
        [np.array([4, 3]), np.array([5])], [np.array([1, 6]), np.array([2])], [10, 8], [2, 3],
        0, None)
    # CES 0: pairs at 0 and 10; CES 1 starts at 20: pairs at 20, 28, 36
    assert list(st) == [1, 6, 11, 16, 22, 30, 38]
    assert list(ln) == [4, 3, 4, 3, 5, 5, 5]
    assert toff is None and table is None


def test_header_is_plain_c_and_links(tmp_path):
    """include/cosmomap2.h must be consumable by a C (not C++) host: compile a C99 program
    against it with gcc and link it to the library (no GPU call: cm2_abi_version only)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "cosmomap2.h"\n'
                   "int main(void) { return cm2_abi_version() == CM2_ABI_VERSION ? 0 : 1; }\n")
    libdir = os.path.join(ROOT, "cosmomap2_amd")
    exe = str(tmp_path / "abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                           "-I" + os.path.join(ROOT, "include"), str(src), "-L" + libdir,
                           "-lcosmomap2_hip", "-Wl,-rpath," + libdir, "-o", exe])
    assert subprocess.call([exe]) == 0


def test_bench_launcher_command_line(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE starts N ranks itself: the command it runs is
    the driver's torch.distributed.run line with bench.py's own arguments passed through, and the
    relay fails unless the job reports N ranks."""
    import json
    import subprocess
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "5", "--config", "c3"], 29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5", "--config", "c3"]

    class _Done(object):
        def __init__(self, line, rc=0):
            self.stdout, self.returncode = (line + "\n").encode(), rc

    seen = {}

    def fake_run(cmd, env=None, stdout=None):
        seen["cmd"], seen["env"] = cmd, env
        return seen["result"]
    monkeypatch.setattr(subprocess, "run", fake_run)
    good = json.dumps({"metric": "m", "n_gpus": 2, "distributed": {"world_size": 2}})
    seen["result"] = _Done(good)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a job that ran on fewer ranks than asked for, a failed job, a job without a line: non-zero
    seen["result"] = _Done(json.dumps({"metric": "m", "n_gpus": 1, "distributed": None}))
    assert bench.launch_ranks(2, ["--gpus", "2"]) != 0
    seen["result"] = _Done(good, rc=3)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 3
    seen["result"] = _Done("no json here")
    assert bench.launch_ranks(2, ["--gpus", "2"]) != 0


def test_kernel_resource_table_of_the_shipped_build():
    """cosmomap2_amd/build.py keeps hipcc's per-kernel resource remarks of the objects it links
    (csrc/build/*.resources.json) and refuses a build in which a kernel of the default path spills
    registers: the table of the shipped library must list the overlap-save, tile and deflation kernels,
    and none of them may have spilled VGPRs or scratch."""
    from cosmomap2_amd import kernel_resources as KR
    text = ("x.hip:1:1: remark: Function Name: _Z3fooPd [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:1:1: remark:     VGPRs: 250 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:1:1: remark:     ScratchSize [bytes/lane]: 36 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:1:1: remark:     Occupancy [waves/SIMD]: 2 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hip:1:1: remark:     VGPRs Spill: 8 [-Rpass-analysis=kernel-resource-usage]\n")
    parsed = KR.parse_remarks(text)
    assert parsed == {"_Z3fooPd": {"vgpr": 250, "scratch_bytes_per_lane": 36, "occupancy": 2, "vgpr_spill": 8}}
    assert KR.short("void (anonymous namespace)::k_os_real<32, 2, true>(int, double*)") == "k_os_real<32, 2, true>"
    bad = KR.offenders([dict(unit="u", kernel="k_os_real<32, 2, true>", vgpr_spill=8, scratch_bytes_per_lane=36),
                        dict(unit="u", kernel="k_helper", vgpr_spill=3)])
    assert [r["kernel"] for r in bad] == ["k_os_real<32, 2, true>"]
    # SGPRs spilled to VGPR lanes count for the overlap-save instantiations of the default dispatch only
    bad = KR.offenders([dict(unit="u", kernel="k_os_real<32, 2, true>", sgpr_spill=16),
                        dict(unit="u", kernel="k_os_real<32, 2, false>", sgpr_spill=56),
                        dict(unit="u", kernel="k_tile_rank", sgpr_spill=129)])
    assert [r["kernel"] for r in bad] == ["k_os_real<32, 2, true>"]
    rows = KR.load_all()
    names = {r["kernel"] for r in rows}
    for k in ("k_os_real<32, 0, false>", "k_os_real<32, 2, true>", "k_os_real<32, 3, false>", "k_P_tiles<3, true>",
              "k_Pt_tiles_fixed<3, true, 4>", "k_gemm_tn_mfma_pairs<1>"):
        assert k in names, k
    assert not KR.offenders(rows)
    assert not any(r.get("vgpr_spill", 0) or r.get("scratch_bytes_per_lane", 0) for r in KR.own_kernels(rows))
    assert not [r for r in rows if r["kernel"] == "k_os_real<32, 2, true>" and r.get("sgpr_spill", 0)]


def test_host_problem_equals_the_serial_oracle(oracle):
    """oracle.HostProblem (the full-size checker of tests/test_gpu_fullsize.py and bench.py: all-cores
    P / N^-1 / P^T, serial ProcessTimeSamples and M_BD) against the serial reference-order oracle on a
    small problem: same right-hand side and matvec to 1e-13, the same PCG iteration count."""
    rng = np.random.default_rng(5)
    nt, npix, nb, pol = 60000, 400, 4, 3
    d, pairs, phi, t, diag = oracle.system_setup(rng, nt, npix, nb)
    pairs[rng.random(nt) < 0.03] = -1
    kk = np.arange(20)
    bands = [(1.0 + 0.1 * b) * np.exp(-kk / 5.0) for b in range(nb)]
    H = oracle.HostProblem(pol, npix, pairs.copy(), phi, nt // nb, bands=bands, threads=3)
    po = pairs.copy()
    ro = oracle.process_time_samples(po, npix, pol=pol, phi=phi)
    assert H.n == ro.new_npix

    def A(v):
        return oracle.sparse_rmult(pol, ro.new_npix, po, ro.cos, ro.sin, oracle.blocklo_mult(
            nt // nb, bands, True, oracle.sparse_mult(pol, po, ro.cos, ro.sin, v)))
    b = oracle.sparse_rmult(pol, ro.new_npix, po, ro.cos, ro.sin, oracle.blocklo_mult(nt // nb, bands, True, d))
    bh = H.rhs(d)
    assert np.linalg.norm(bh - b) / np.linalg.norm(b) < 1e-13
    x = rng.standard_normal(pol * H.n)
    assert np.linalg.norm(H.A(x) - A(x)) / np.linalg.norm(A(x)) < 1e-13
    its = []
    xs, info = oracle.cg(A, b, rtol=1e-6, M=lambda v: oracle.bd_precond_mult(pol, ro, v),
                         callback=lambda v: its.append(1))
    xh, info_h, its_h = H.solve(bh, rtol=1e-6)
    assert info == 0 and info_h == 0 and its_h == len(its)
    assert np.linalg.norm(xh - xs) / np.linalg.norm(xs) < 1e-9


def test_host_problem_two_level_build_has_the_reference_invariants(oracle):
    """oracle.HostProblem.two_level (the host-side two-level build of tests/test_gpu_fullsize.py and bench.py's
    parity_full_size.two_level) on a small raster-like problem: the invariants the reference's script prints
    (src/test_M2_precond_onto_real_data.py:112-117: M2 A z_i = z_i, ||R A z_i|| <= 1e-10) hold, PCG with M2
    reaches the M_BD solution in no more iterations; and oracle.arnoldi(exhausted="return") hands back the
    same vectors and Hessenberg columns the raising form had built before it raised."""
    rng = np.random.default_rng(6)
    nt, npix, nb, pol = 40000, 300, 4, 1
    d, pairs, phi, t, diag = oracle.system_setup(rng, nt, npix, nb)
    pairs = ((np.arange(nt) // 7) % npix).astype(np.int32)            # a coherent scan: slow modes exist
    kk = np.arange(30)
    bands = [(1.0 + 0.1 * b) * np.exp(-kk / 9.0) for b in range(nb)]
    H = oracle.HostProblem(pol, npix, pairs.copy(), phi, nt // nb, bands=bands, threads=2)
    b = H.rhs(d)
    vs, hs, m = oracle.arnoldi(lambda v: H.M(H.A(v)), H.M(b), np.zeros(b.size), tol=0.0, inner_m=12,
                               exhausted="return")
    assert m == 12 and len(vs) == 13 and len(hs) == 12
    with pytest.raises(RuntimeError):
        oracle.arnoldi(lambda v: H.M(H.A(v)), H.M(b), np.zeros(b.size), tol=0.0, inner_m=12)
    V = np.column_stack(vs[:m])
    assert np.abs(V.T.dot(V) - np.eye(m)).max() < 1e-8
    Hm = oracle.build_hess(hs, m)
    th, U = np.linalg.eigh(0.5 * (Hm + Hm.T))
    r = 4
    Z = V.dot(U[:, np.argsort(th)[:r]])
    Az, co, M2 = H.two_level(Z, apply='eig')
    for i in range(r):
        assert np.linalg.norm(Az[:, i] - H.A(Z[:, i])) == 0.0
        assert np.allclose(M2(Az[:, i]), Z[:, i])                       # M2 A z_i = z_i
        y = co.mult(oracle.deflation_rmult(Z, Az[:, i]))
        assert oracle.norm2(Az[:, i] - oracle.deflation_mult(Az, y)) <= 1e-10 * oracle.norm2(Az[:, i])   # R A z_i = 0
    x1, info1, its1 = H.solve(b, rtol=1e-8)
    x2, info2, its2 = H.solve(b, rtol=1e-8, M=M2)
    assert info1 == 0 and info2 == 0 and its2 <= its1
    assert np.linalg.norm(x2 - x1) / np.linalg.norm(x1) < 1e-6


def test_library_call_is_retried_once_after_an_out_of_memory_failure(monkeypatch):
    """_hip.call: a call that ran out of device memory is tried once more after torch's cached blocks went
    back to the driver (the library has released its own before reporting); any other failure, and a
    second out-of-memory failure, raise HipError with the library's message."""
    from cosmomap2_amd import _hip

    class Fake(object):
        def __init__(self, results, message):
            self.results, self.message, self.calls = list(results), message, 0

        def cm2_last_error(self):
            return self.message

        def cm2_something(self, *args):
            self.calls += 1
            return self.results.pop(0)

    freed = []
    monkeypatch.setattr(_hip, "_free_torch_cache", lambda: freed.append(1) or True)
    monkeypatch.setattr(_hip, "RESTARTABLE", frozenset(["cm2_something"]))
    oom = _hip.ERR_OUT_OF_MEMORY
    fake = Fake([oom, 0], b"d_temp.alloc(n) failed: out of memory (cm2_x.hip:1)")
    monkeypatch.setattr(_hip, "load", lambda: fake)
    _hip.call("cm2_something", 1, 2)
    assert fake.calls == 2 and freed == [1]
    fake = Fake([oom, oom], b"d_temp.alloc(n) failed: out of memory (cm2_x.hip:1)")
    monkeypatch.setattr(_hip, "load", lambda: fake)
    with pytest.raises(_hip.HipError, match="out of memory"):
        _hip.call("cm2_something")
    assert fake.calls == 2
    # the status code decides, not the text of the message
    fake = Fake([1, 0], b"hipLaunchKernel failed: out of memory")
    monkeypatch.setattr(_hip, "load", lambda: fake)
    with pytest.raises(_hip.HipError):
        _hip.call("cm2_something")
    assert fake.calls == 1
    fake = Fake([2, 0], b"cm2_tiles_create: bad pol=7")
    monkeypatch.setattr(_hip, "load", lambda: fake)
    with pytest.raises(_hip.HipError, match="bad pol"):
        _hip.call("cm2_something")
    assert fake.calls == 1
    # an entry point that updates in place is never run twice
    monkeypatch.setattr(_hip, "RESTARTABLE", frozenset())
    fake = Fake([oom, 0], b"x failed: out of memory")
    monkeypatch.setattr(_hip, "load", lambda: fake)
    with pytest.raises(_hip.HipError):
        _hip.call("cm2_something")
    assert fake.calls == 1
    monkeypatch.setattr(_hip, "RESTARTABLE", frozenset(["cm2_something"]))
    # without a GPU there is nothing to free: no second attempt
    monkeypatch.setattr(_hip, "_free_torch_cache", lambda: False)
    fake = Fake([oom, 0], b"x failed: out of memory")
    monkeypatch.setattr(_hip, "load", lambda: fake)
    with pytest.raises(_hip.HipError):
        _hip.call("cm2_something")
    assert fake.calls == 1


def test_restartable_entry_points_exist_and_exclude_the_in_place_updates():
    from cosmomap2_amd import _hip
    assert _hip.RESTARTABLE <= set(_hip.PROTOTYPES)
    for name in ("cm2_axpy", "cm2_scal", "cm2_Z_axpy", "cm2_panel_gemm", "cm2_pcg_update_p", "cm2_pcg_update_xr",
                 "cm2_flag_samples", "cm2_pcg", "cm2_pcg_sharded", "cm2_arnoldi", "cm2_compact_f64"):
        assert name not in _hip.RESTARTABLE
