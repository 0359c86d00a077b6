"""Worker of tests/test_gpu_sharded.py::test_one_rank_rccl_group: ONE rank with the "nccl"
(= RCCL) backend on the test box's one GPU.  With CM2_FORCE_COLLECTIVES / CM2_ALLREDUCE_CHUNKS
set, every collective of cosmomap2_amd/sharding.py goes through RCCL although the group has one
rank: the chunked asynchronous all-reduce on slices of the map, the device-side max-reduce of
||r||^2 folded into the PCG's deferred read, all_gather_into_tensor / reduce_scatter_tensor of
the row-sharded layout and the 8-byte all-reduces of its dot products.  A one-rank collective is
the identity, so every result must equal the plain single-process run bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")     # (the test passes a free port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    import cosmomap2_amd
    from cosmomap2_amd.interfaces import SparseLO, BlockLO, BlockDiagonalPreconditionerLO
    from cosmomap2_amd.interfaces import linearoperators as L
    from cosmomap2_amd.utilities import ProcessTimeSamples
    from cosmomap2_amd.sharding import (ShardedLO, make_sync, RowShards, RowShardedNormalLO,
                                        row_sharded_bd, allreduce_sum_)

    pol, npix, nb, bs, lam = 3, 30000, 12, 100000, 600
    nt = nb * bs
    rng = np.random.default_rng(5)
    pix = rng.integers(0, npix, nt).astype(np.int32)
    pix[rng.random(nt) < 0.03] = -1
    phi = 0.4 + 0.0785 * np.arange(nt)
    d = torch.from_numpy(rng.standard_normal(nt)).cuda()
    kk = np.arange(lam)
    bands = [(1.0 + 0.05 * b) * np.where(kk == 0, 1.0, 0.2 * np.exp(-kk / 150.0)) for b in range(nb)]
    x = torch.from_numpy(rng.standard_normal(pol * npix)).cuda()

    # plain single-process reference (no collectives at all)
    ces = ProcessTimeSamples(pix.copy(), npix, pol=pol, phi=phi)
    n = ces.get_new_pixel[0]
    P = SparseLO(n, nt, ces.pixs if hasattr(ces, "pixs") else pix, pol=pol, angle_processed=ces)
    N = BlockLO(bs, bands, offdiag=True, method=3)
    A_local = P.T * N * P
    assert L._use_tiles(P)
    M = BlockDiagonalPreconditionerLO(ces, n, pol)
    y_ref = (A_local * x).clone()
    b = P.T * (N * d)
    its_ref = []
    x_ref, info = cosmomap2_amd.cg(A_local, b, M=M, rtol=1e-8, maxiter=200,
                                   callback=lambda v: its_ref.append(1))
    assert info == 0

    # ---- replicated layout through RCCL: chunked async all-reduce on map slices + device max
    os.environ["CM2_FORCE_COLLECTIVES"] = "1"
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "4"
    ces_r = ProcessTimeSamples(pix.copy(), npix, pol=pol, phi=phi, allreduce=lambda t: dist.all_reduce(t))
    assert ces_r.get_new_pixel[0] == n
    A = ShardedLO(A_local)
    y = A * x
    assert A.collectives_issued == 4, A.collectives_issued
    assert torch.equal(y, y_ref), "chunked RCCL all-reduce changed the map"
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "1"
    y1 = A * x
    assert A.collectives_issued == 5 and torch.equal(y1, y_ref)
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "4"
    sync = make_sync()
    assert sync is not None and sync.reduce_ is not None      # RCCL: device-side form
    assert sync(3.5) == 3.5
    its = []
    xs, info = cosmomap2_amd.cg(A, b, M=M, rtol=1e-8, maxiter=200, callback=lambda v: its.append(1),
                                sync=sync)
    assert info == 0 and len(its) == len(its_ref), (len(its), len(its_ref))
    assert torch.equal(xs, x_ref), "PCG over RCCL differs from the single-process solve"

    # ---- row-sharded layout through RCCL: all-gather, reduce-scatter, scalar all-reduces
    sh = RowShards(n, pol)
    assert (sh.rank, sh.world) == (0, 1) and sh.rows == pol * n
    x_loc = sh.local(x)
    assert torch.equal(sh.gather(x_loc), x)
    assert torch.equal(sh.reduce_scatter(y_ref), y_ref)
    t1 = torch.tensor([2.5], dtype=torch.float64, device="cuda")
    assert float(allreduce_sum_(t1)) == 2.5
    Ar = RowShardedNormalLO(A_local, sh)
    assert torch.equal(Ar * x_loc, y_ref)
    Mr = row_sharded_bd(ces, sh)
    its_r = []
    xr, info_r = cosmomap2_amd.cg(Ar, sh.local(b), M=Mr, rtol=1e-8, maxiter=200,
                                  callback=lambda v: its_r.append(1), dot_reduce=sh.allreduce_)
    assert info_r == 0 and len(its_r) == len(its_ref), (len(its_r), len(its_ref))
    assert torch.equal(sh.gather(xr), x_ref)
    # ---- persistent exchange buffers (round 5), through RCCL: not one torch allocation inside a loop
    #      of matvecs of either layout, the same output address every time, the bits of y_ref
    def allocations():
        st = torch.cuda.memory_stats()
        return (st["allocation.all.allocated"], st["segment.all.allocated"])
    for name, keep, vec in (("rows", RowShardedNormalLO(A_local, sh, persistent_output=True), x_loc),
                            ("replicated", ShardedLO(A_local, persistent_output=True), x)):
        for chunks in ("1", "4"):
            os.environ["CM2_ALLREDUCE_CHUNKS"] = chunks
            y_k = keep * vec
            ptr_k = y_k.data_ptr()
            assert torch.equal(y_k, y_ref), (name, chunks)
            torch.cuda.synchronize()
            a0 = allocations()
            for _ in range(5):
                y_k = keep * vec
            torch.cuda.synchronize()
            assert allocations() == a0, (name, chunks, "torch allocations inside the matvec loop", a0, allocations())
            assert y_k.data_ptr() == ptr_k and torch.equal(y_k, y_ref), (name, chunks)
    os.environ["CM2_ALLREDUCE_CHUNKS"] = "4"
    print("RCCL-1RANK-OK backend %s, %d collectives on map slices, PCG %d iterations "
          "(replicated) / %d (row-sharded), bit-identical to the single-process solve"
          % (dist.get_backend(), A.collectives_issued, len(its), len(its_r)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
