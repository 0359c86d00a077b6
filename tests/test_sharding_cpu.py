"""
CPU tests of the multi-GPU layer (cosmomap2_amd/sharding.py) with world_size 2 over gloo.
The per-rank compute is done by the ORACLE (tests may use it); what is under test is the
host logic that ships: block-aligned TOD partition, the map all-reduce operator, the
replicated-vector PCG over it and the stop-test synchronisation.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_blocks_partition():
    from cosmomap2_amd.sharding import shard_blocks
    sizes = [500, 400, 124, 500, 400, 124, 77]
    for world in (1, 2, 3, 4, 8, 16):
        cover = []
        for r in range(world):
            b0, b1, s0, s1 = shard_blocks(sizes, world, r)
            assert 0 <= b0 <= b1 <= len(sizes)
            assert s0 == sum(sizes[:b0]) and s1 == sum(sizes[:b1])     # cuts at block boundaries
            cover.append((b0, b1))
        assert cover[0][0] == 0 and cover[-1][1] == len(sizes)
        for (a0, a1), (c0, c1) in zip(cover[:-1], cover[1:]):
            assert a1 == c0                                            # contiguous, disjoint
    # equal blocks split evenly: C5 = 64 detector blocks over 8 GPUs -> 8 each
    for r in range(8):
        b0, b1, s0, s1 = shard_blocks([15625000] * 64, 8, r)
        assert b1 - b0 == 8 and s1 - s0 == 8 * 15625000
    with pytest.raises(ValueError):
        shard_blocks(sizes, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from cosmomap2_amd import linop as lp
        from cosmomap2_amd.sharding import ShardedLO, shard_blocks, allreduce_sum_, make_sync, world as W
        assert W() == (rank, world)
        pol, npix, nb = 3, 48, 6
        sizes = [700, 300, 500, 800, 200, 500]
        nt = sum(sizes)
        rng = np.random.default_rng(123)                    # same global problem on every rank
        d, pairs, phi, t, diag = orc.system_setup(rng, nt, npix, nb)
        bands = [np.array([1.0 + 0.5 * ti[0], 0.3 * ti[1]]) for ti in t]
        c, s = np.cos(2 * phi), np.sin(2 * phi)

        def normal(pix, cc, ss, bl, szs, x):
            return orc.sparse_rmult(pol, npix, pix, cc, ss, orc.blocklo_mult(
                szs, bl, True, orc.sparse_mult(pol, pix, cc, ss, x)))

        # ---- this rank's shard: whole noise blocks only
        b0, b1, s0, s1 = shard_blocks(sizes, world, rank)
        lp_pix, lc, ls = pairs[s0:s1], c[s0:s1], s[s0:s1]
        A_local = lp.LinearOperator(pol * npix, pol * npix, lambda x: normal(
            lp_pix, lc, ls, bands[b0:b1], sizes[b0:b1], x), symmetric=True)
        A = ShardedLO(A_local)
        x = rng.standard_normal(pol * npix)
        y = A * x
        y_ref = normal(pairs, c, s, bands, sizes, x)
        assert np.allclose(y, y_ref, rtol=1e-12, atol=1e-12)

        # ---- per-pixel weights: sum of shard sums == global sums (setup all-reduce)
        z = lambda: np.zeros(npix)
        loc = [z() for _ in range(6)]
        orc.lib()
        import ctypes
        D_ = ctypes.POINTER(ctypes.c_double)
        w = np.ones(s1 - s0)
        orc.lib().orc_weights_accumulate(
            ctypes.c_int(pol), ctypes.c_int64(s1 - s0),
            np.ascontiguousarray(lp_pix).ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            w.ctypes.data_as(D_), np.ascontiguousarray(lc).ctypes.data_as(D_),
            np.ascontiguousarray(ls).ctypes.data_as(D_), *[a.ctypes.data_as(D_) for a in loc])
        for a in loc:
            allreduce_sum_(a)
        pg = pairs.copy()
        glob = orc.process_time_samples(pg, npix, pol=pol, phi=phi)
        assert glob.new_npix == npix
        assert np.allclose(loc[0], glob.counts) and np.allclose(loc[5], glob.sincos)

        # ---- replicated-vector PCG over the sharded operator
        b_loc = orc.sparse_rmult(pol, npix, lp_pix, lc, ls, orc.blocklo_mult(
            sizes[b0:b1], bands[b0:b1], True, d[s0:s1]))
        b = allreduce_sum_(b_loc.copy())
        M = lambda v: orc.bd_precond_mult(pol, glob, v)
        its = []
        xs, info = orc.cg(lambda v: A * v, b, M=M, rtol=1e-8, callback=lambda xk: its.append(1))
        its1 = []
        x1, info1 = orc.cg(lambda v: normal(pairs, c, s, bands, sizes, v),
                           normal_rhs(orc, pol, npix, pairs, c, s, bands, sizes, d), M=M,
                           rtol=1e-8, callback=lambda xk: its1.append(1))
        assert info == 0 and info1 == 0 and len(its) == len(its1)
        assert np.linalg.norm(xs - x1) <= 1e-9 * np.linalg.norm(x1)
        # every rank must hold the same bits (replicated vectors never diverge)
        chk = torch.from_numpy(xs.copy())
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
        sync = make_sync()
        assert sync(float(rank + 1)) == float(world)        # max over ranks
        assert sync.reduce_ is None                         # gloo: no device-side form

        # ---- the number of overlapped all-reduces per matvec is a COLLECTIVE choice: uneven
        #      shards on either side of a threshold (rank 0: 14e6 samples -> 1 chunk by itself,
        #      rank 1: 15e6 -> 2) must still issue the same collectives on every rank
        class _P(object):
            nrows = 14_000_000 + 1_000_000 * rank

        class _Chunked(lp.LinearOperator):
            """stand-in for the tile-order chain: reduces its output in `chunks` equal pieces"""
            P = _P()

            def __init__(self):
                super(_Chunked, self).__init__(8, 8, lambda v: v * 1.0, symmetric=True)

            def _compiled(self):
                return [self]

            def reduced_matvec(self, v, reducer, chunks):
                out = torch.from_numpy(np.array(v, dtype=np.float64))
                step = -(-out.numel() // chunks)
                works = [reducer(out[a:a + step]) for a in range(0, out.numel(), step)]
                for wk in works:
                    wk.wait()
                return out.numpy()

        os.environ.pop("CM2_ALLREDUCE_CHUNKS", None)
        Ac = ShardedLO(_Chunked())
        assert Ac.allreduce_chunks() is None                # not decided before the first matvec
        yc = Ac * np.arange(8.0)
        assert np.array_equal(yc, world * np.arange(8.0))
        assert Ac.allreduce_chunks() == 2                   # from the MAX over ranks, on every rank
        cnt = torch.tensor([Ac.collectives_issued], dtype=torch.int64)
        lo_c, hi_c = cnt.clone(), cnt.clone()
        dist.all_reduce(lo_c, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_c, op=dist.ReduceOp.MAX)
        assert int(lo_c) == int(hi_c) == 2, (int(lo_c), int(hi_c))

        # ---- row-sharded vectors (reduce-scatter variant): every rank owns n / N rows
        from cosmomap2_amd.sharding import RowShards, RowShardedNormalLO
        sh = RowShards(npix, pol)
        assert sh.rows == -(-npix // world) * pol and sh.lo == rank * sh.rows
        Ar = RowShardedNormalLO(A_local, sh)
        x_loc = sh.local(x)
        assert np.array_equal(sh.gather(x_loc), x)                       # all-gather restores x
        y_loc = Ar * x_loc
        assert np.allclose(y_loc[:sh.hi - sh.lo], y_ref[sh.lo:sh.hi], rtol=1e-12, atol=1e-12)
        assert np.all(y_loc[sh.hi - sh.lo:] == 0.0)
        # PCG in the scipy recurrence on the local rows: dots = local sums + all-reduce
        def rdot(u, v):
            return float(sh.allreduce_(np.array([np.dot(u, v)]))[0])
        Mfull = lambda v: orc.bd_precond_mult(pol, glob, v)
        def M_loc(v_loc):               # M_BD is per pixel: apply it on a padded whole vector
            full = np.zeros(pol * npix)
            full[sh.lo:sh.hi] = v_loc[:sh.hi - sh.lo]
            return sh.local(Mfull(full))
        b_rows = sh.reduce_scatter(b_loc)
        assert np.allclose(sh.gather(b_rows), b, rtol=1e-13, atol=1e-13)
        xr, rr = np.zeros(sh.rows), b_rows.copy()
        atol = 1e-8 * np.sqrt(rdot(b_rows, b_rows))
        nit, rho_prev, p = 0, None, None
        while np.sqrt(rdot(rr, rr)) >= atol and nit < 500:
            z = M_loc(rr)
            rho = rdot(rr, z)
            p = z.copy() if p is None else z + (rho / rho_prev) * p
            q = Ar * p
            alpha = rho / rdot(p, q)
            xr += alpha * p
            rr -= alpha * q
            rho_prev = rho
            nit += 1
        assert nit == len(its)                                           # same count as replicated
        assert np.linalg.norm(sh.gather(xr) - xs) <= 1e-10 * np.linalg.norm(xs)
        # ---- row-sharded deflation build: the shipped Arnoldi recurrence (_arnoldi_M: block
        #      Gram-Schmidt in the M inner product, Hessenberg matrix, stop on an exhausted space)
        #      with its vector operations supplied in NumPy, each rank holding its rows only and
        #      every inner product all-reduced -- against the same recurrence on whole vectors
        from cosmomap2_amd.interfaces.deflationlib import _arnoldi_M

        class _NpBasis(object):
            def __init__(self, n, allreduce):
                self.n, self.allreduce, self.vecs = n, allreduce, []

            def append(self, v):
                self.vecs.append(np.array(v))

            def dots(self, w):
                out = np.array([np.dot(v, w) for v in self.vecs] + [0.0] * 3)   # padded like a panel
                if self.allreduce is not None:
                    self.allreduce(out)
                return out

            def subtract(self, w, coeff):
                for v, c in zip(self.vecs, coeff):
                    w -= c * v

        class _NpOps(object):
            def __init__(self, allreduce=None):
                self.allreduce, self.reduced = allreduce, 0

            def basis(self, n):
                return _NpBasis(n, self._count if self.allreduce else None)

            def _count(self, t):
                self.reduced += 1
                return self.allreduce(t)

            def apply(self, op, v):
                return np.array(op(v))

            def scaled(self, a, v):
                return a * v

            def clone(self, v):
                return v.copy()

            def dot_dev(self, x, y):
                out = np.array([np.dot(x, y)])
                if self.allreduce is not None:
                    self._count(out)
                return out

            def dot(self, x, y):
                return float(self.dot_dev(x, y)[0])

            def column(self, hk, k, ss):
                return np.concatenate([hk[:k + 1], ss])

        steps = 10
        full = _arnoldi_M(lambda v: normal(pairs, c, s, bands, sizes, v), b, Mfull, steps, ops=_NpOps())
        ops_sh = _NpOps(sh.allreduce_)
        part = _arnoldi_M(lambda v: np.asarray(Ar * v), b_rows, M_loc, steps, ops=ops_sh)
        assert part[3] == full[3] == steps
        assert np.allclose(part[2], full[2], rtol=1e-10, atol=1e-12 * np.abs(full[2]).max())
        # per step: two coefficient vectors and one scalar; two scalars for the start vector's norm
        assert ops_sh.reduced == 3 * steps + 1, ops_sh.reduced
        for j in (0, 3, steps):                             # basis vectors: this rank's rows
            assert np.allclose(part[0].vecs[j][:sh.hi - sh.lo], full[0].vecs[j][sh.lo:sh.hi],
                               rtol=1e-9, atol=1e-11)
            assert np.all(part[0].vecs[j][sh.hi - sh.lo:] == 0.0)
        hs = torch.from_numpy(part[2].copy())               # every rank holds the same H, bit for bit
        lo_h, hi_h = hs.clone(), hs.clone()
        dist.all_reduce(lo_h, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_h, op=dist.ReduceOp.MAX)
        assert torch.equal(lo_h, hi_h)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("%d" % len(its))
    finally:
        dist.destroy_process_group()


def normal_rhs(orc, pol, npix, pairs, c, s, bands, sizes, d):
    return orc.sparse_rmult(pol, npix, pairs, c, s, orc.blocklo_mult(sizes, bands, True, d))


def test_sharded_operator_and_pcg_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
    assert (tmp_path / "ok0").read_text() == (tmp_path / "ok1").read_text()
