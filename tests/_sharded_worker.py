"""Worker of tests/test_gpu_sharded.py: one rank of a 2-rank TOD-sharded matvec / PCG on ONE
GPU (gloo carries the collectives; the driver's multi-GPU runs use RCCL, one rank per GPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cosmomap2_amd
    from cosmomap2_amd.interfaces import SparseLO, BlockLO, BlockDiagonalPreconditionerLO
    from cosmomap2_amd.interfaces import linearoperators as L
    from cosmomap2_amd.utilities import ProcessTimeSamples
    from cosmomap2_amd.sharding import ShardedLO, make_sync

    pol, npix, nb, bs, lam = 3, 30000, 24, 100000, 600
    nt = nb * bs
    rng = np.random.default_rng(5)                      # every rank draws the whole problem
    pix = rng.integers(0, npix, nt).astype(np.int32)
    pix[rng.random(nt) < 0.03] = -1
    phi = 0.4 + 0.0785 * np.arange(nt)
    d = rng.standard_normal(nt)
    kk = np.arange(lam)
    bands = [(1.0 + 0.05 * b) * np.where(kk == 0, 1.0, 0.2 * np.exp(-kk / 150.0)) for b in range(nb)]
    x = torch.from_numpy(rng.standard_normal(pol * npix)).cuda()

    def build(lo_b, hi_b, allreduce):
        sl = slice(lo_b * bs, hi_b * bs)
        p = pix[sl].copy()
        ces = ProcessTimeSamples(p, npix, pol=pol, phi=phi[sl], allreduce=allreduce)
        n = ces.get_new_pixel[0]
        P = SparseLO(n, p.size, p, pol=pol, angle_processed=ces)
        N = BlockLO(bs, bands[lo_b:hi_b], offdiag=True, method=3)
        return ces, P, N, n

    per = nb // world
    ces, P, N, n = build(rank * per, (rank + 1) * per, lambda t: dist.all_reduce(t))
    assert n == npix and L._use_tiles(P)
    A = ShardedLO(P.T * N * P)
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "4"
    y4 = (A * x).clone()
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "1"
    y1 = (A * x).clone()
    err = float((y4 - y1).norm() / y1.norm())
    assert err < 1e-13, ("chunked vs single all-reduce", err)
    chk = y4.clone()
    dist.broadcast(chk, 0)
    assert torch.equal(chk, y4), "ranks disagree on the reduced map"
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "4"
    M = BlockDiagonalPreconditionerLO(ces, n, pol)
    b = P.T * (N * torch.from_numpy(d[rank * per * bs:(rank + 1) * per * bs]).cuda())
    dist.all_reduce(b)
    its = []
    xs, info = cosmomap2_amd.cg(A, b, M=M, rtol=1e-8, maxiter=200, callback=lambda v: its.append(1),
                                sync=make_sync())
    assert info == 0
    if rank == 0:                                       # the same problem on one rank
        ces1, P1, N1, n1 = build(0, nb, None)
        A1 = P1.T * N1 * P1
        y_full = A1 * x
        e2 = float((y4 - y_full).norm() / y_full.norm())
        assert e2 < 1e-12, ("sharded vs single-rank matvec", e2)
        M1 = BlockDiagonalPreconditionerLO(ces1, n1, pol)
        b1 = P1.T * (N1 * torch.from_numpy(d).cuda())
        its1 = []
        x1, info1 = cosmomap2_amd.cg(A1, b1, M=M1, rtol=1e-8, maxiter=200,
                                     callback=lambda v: its1.append(1))
        # (P^T sums in fixed order on every rank: the two-rank sum differs from the one-rank sum
        # only by where the partial sums are split, ~1e-16 relative)
        assert info1 == 0 and len(its1) == len(its), (len(its1), len(its))
        e3 = float((xs - x1).norm() / x1.norm())
        assert e3 < 1e-7, ("sharded vs single-rank PCG solution", e3)
        print("SHARDED-OK matvec %.1e / %.1e, PCG %d vs %d iterations, solution %.1e"
              % (err, e2, len(its), len(its1), e3), flush=True)
    # ---- row-sharded vectors (reduce-scatter variant), M_BD and two-level preconditioner:
    #      same iteration counts and, to rounding, the same solutions as the replicated variant
    from cosmomap2_amd.sharding import (RowShards, RowShardedNormalLO, row_sharded_bd,
                                        row_sharded_two_level)
    from cosmomap2_amd.interfaces import (CoarseLO, DeflationLO, TwoLevelPreconditionerLO,
                                          ritz_deflation_basis, apply_to_columns)
    sh = RowShards(n, pol)
    Ar = RowShardedNormalLO(P.T * N * P, sh)
    Mr = row_sharded_bd(ces, sh)
    b_loc = sh.local(b)
    y_rows = Ar * sh.local(x)
    e4 = float((sh.gather(y_rows) - y4).norm() / y4.norm())
    assert e4 < 1e-13, ("row-sharded vs replicated matvec", e4)
    its_r = []
    xr, info_r = cosmomap2_amd.cg(Ar, b_loc, M=Mr, rtol=1e-8, maxiter=200,
                                  callback=lambda v: its_r.append(1), dot_reduce=sh.allreduce_)
    assert info_r == 0 and len(its_r) == len(its), (len(its_r), len(its))
    e5 = float((sh.gather(xr) - xs).norm() / xs.norm())
    assert e5 < 1e-12, ("row-sharded vs replicated PCG solution", e5)
    # two-level: Z from the replicated operator (identical on both ranks), then its rows
    r = 8
    Z, theta = ritz_deflation_basis(A, M, b, r, 24)
    AZ = apply_to_columns(A, Z)
    M2 = TwoLevelPreconditionerLO(M, DeflationLO(Z), DeflationLO(AZ), CoarseLO(Z, AZ, r, apply='eig'))
    its2 = []
    x2, info2 = cosmomap2_amd.cg(A, b, M=M2, rtol=1e-8, maxiter=200, callback=lambda v: its2.append(1),
                                 sync=make_sync())

    def rows_of(mat):
        out = torch.zeros(sh.rows, mat.shape[1], dtype=torch.float64, device=mat.device)
        out[:sh.hi - sh.lo] = mat[sh.lo:sh.hi]
        return out
    M2r = row_sharded_two_level(Mr, rows_of(Z), rows_of(AZ), sh, apply='eig')
    rr0 = sh.local(x)
    e6 = float((sh.gather(M2r * rr0) - M2 * x).norm() / (M2 * x).norm())
    assert e6 < 1e-12, ("row-sharded vs replicated M2", e6)
    its2r = []
    x2r, info2r = cosmomap2_amd.cg(Ar, b_loc, M=M2r, rtol=1e-8, maxiter=200,
                                   callback=lambda v: its2r.append(1), dot_reduce=sh.allreduce_)
    assert info2 == 0 and info2r == 0 and len(its2r) == len(its2), (len(its2r), len(its2))
    e7 = float((sh.gather(x2r) - x2).norm() / x2.norm())
    assert e7 < 1e-12, ("row-sharded vs replicated two-level solution", e7)
    # ---- the deflation space built on row-sharded vectors (local rows of the Arnoldi basis,
    #      all-reduced coefficients): same Ritz values, same space, same two-level solve
    Zl, thl, AZl = ritz_deflation_basis(Ar, Mr, b_loc, r, 24, with_AZ=True, shards=sh)
    assert Zl.shape == (sh.rows, r) and AZl.shape == (sh.rows, r)
    e8 = float(np.abs(np.asarray(thl) - np.asarray(theta)).max() / np.abs(theta).max())
    assert e8 < 1e-10, ("Ritz values of the row-sharded build", e8)
    Zg = torch.zeros(sh.rows * world, r, dtype=torch.float64, device=Zl.device)
    dist.all_gather_into_tensor(Zg, Zl.contiguous())
    Zg = Zg[:pol * n]
    # same space: the replicated Ritz vectors are reproduced by projecting onto the sharded ones
    G = torch.linalg.lstsq(Zg, Z).solution
    e9 = float((Zg @ G - Z).norm() / Z.norm())
    assert e9 < 1e-8, ("span of the row-sharded Ritz vectors", e9)
    AZl_ref = rows_of(apply_to_columns(A, Zg))
    e10 = float((AZl - AZl_ref).norm() / AZl_ref.norm())
    assert e10 < 1e-9, ("A Z from the Arnoldi relation on local rows", e10)
    M2s = row_sharded_two_level(Mr, Zl, AZl, sh, apply='eig')
    its2s = []
    x2s, info2s = cosmomap2_amd.cg(Ar, b_loc, M=M2s, rtol=1e-8, maxiter=200,
                                   callback=lambda v: its2s.append(1), dot_reduce=sh.allreduce_)
    assert info2s == 0 and len(its2s) == len(its2), (len(its2s), len(its2))
    e11 = float((sh.gather(x2s) - x2).norm() / x2.norm())
    assert e11 < 1e-9, ("two-level solve with the row-sharded deflation space", e11)
    if rank == 0:
        print("ROWSHARDED-OK matvec %.1e, PCG %d its (%.1e), M2 %.1e, two-level %d its (%.1e); "
              "row-sharded build: Ritz values %.1e, span %.1e, AZ %.1e, two-level %d its (%.1e)"
              % (e4, len(its_r), e5, e6, len(its2r), e7, e8, e9, e10, len(its2s), e11), flush=True)
    # ---- persistent exchange buffers (round 5): the matvec of either layout makes NO allocation of its
    #      own between applications (what torch's counters still see inside the loop is the gloo
    #      transport's staging of device tensors, measured on bare collectives of the same shapes
    #      right here; over RCCL it is zero: tests/_nccl_worker.py asserts that), writes to the same
    #      addresses every time, and gives the bits of the allocating form; a whole PCG on the persistent
    #      operators gives the same count and map
    def allocations():
        st = torch.cuda.memory_stats()
        return np.array([st["allocation.all.allocated"], st["segment.all.allocated"]])

    def counted(fn, reps=5):
        torch.cuda.synchronize()
        a0 = allocations()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return allocations() - a0

    def transport_rows():
        g, tsum = sh.exchange_buffers(x.device)
        dist.all_gather_into_tensor(g, x_rows)
        dist.reduce_scatter_tensor(o_rows, tsum)

    def transport_replicated(chunks):
        step = -(-scratch.numel() // chunks)
        works = [dist.all_reduce(scratch[a:a + step], async_op=True) for a in range(0, scratch.numel(), step)] \
            if chunks > 1 else [dist.all_reduce(scratch, async_op=True)]
        for wk in works:
            wk.wait()
    x_rows, o_rows, scratch = sh.local(x), torch.empty(sh.rows, dtype=torch.float64, device=x.device), x.clone()
    for name, fresh, keep, vec, kw in (
            ("rows", Ar, RowShardedNormalLO(P.T * N * P, sh, persistent_output=True), x_rows,
             {"dot_reduce": sh.allreduce_}),
            ("replicated", A, ShardedLO(P.T * N * P, persistent_output=True), x, {"sync": make_sync()})):
        for chunks in (("1", "4") if name == "replicated" else ("1",)):
            os.environ["CM2_ALLREDUCE_CHUNKS"] = chunks
            y_f = (fresh * vec).clone()
            y_k = keep * vec                                   # first application: buffers are made here
            ptr_k = y_k.data_ptr()
            assert torch.equal(y_k, y_f), (name, chunks, "persistent buffers change the result")
            bare = counted(transport_rows if name == "rows" else (lambda: transport_replicated(int(chunks))))
            ours = counted(lambda: keep * vec)
            assert np.array_equal(ours, bare), (name, chunks, "allocations beyond the transport's", ours, bare)
            with_alloc = counted(lambda: fresh * vec)
            assert with_alloc[0] > bare[0], (name, "the allocating form should show in the counter", with_alloc)
            y_k = keep * vec
            assert y_k.data_ptr() == ptr_k and torch.equal(y_k, y_f), (name, chunks)
        os.environ["CM2_ALLREDUCE_CHUNKS"] = "4"
        its_k = []
        xk, info_k = cosmomap2_amd.cg(keep, b_loc if name == "rows" else b, M=Mr if name == "rows" else M,
                                      rtol=1e-8, maxiter=200, callback=lambda v: its_k.append(1), **kw)
        ref_x = xr if name == "rows" else xs
        assert info_k == 0 and len(its_k) == len(its), (name, len(its_k), len(its))
        assert torch.equal(xk, ref_x), (name, "PCG on the persistent operator differs")
    if rank == 0:
        print("PERSISTENT-OK both layouts: no allocation beyond the transport's in 5 matvecs, same bits, "
              "same PCG", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
