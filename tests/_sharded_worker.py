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
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
