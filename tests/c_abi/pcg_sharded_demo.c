/*
 * A sharded solve from a C99 host: one process per rank, the TOD cut at noise-block
 * boundaries (interfaces/blkop.py:195-206: the blocks of N^-1 are independent, so the cut is exact), the
 * map-domain exchange and the scalar reductions done by the HOST's collective and handed to the
 * library's PCG driver as callbacks (cm2_pcg_sharded).  Both layouts of the map vectors:
 *
 *   layout 0 (replicated): A x = allreduce( P_k^T N_k^-1 P_k x ), whole vectors on every rank;
 *   layout 1 (rows):       rank k owns the pixel-aligned rows [row0, row1): all-gather p, local
 *                          P_k^T N_k^-1 P_k, reduce-scatter; M_BD on the rank's pixels only.
 *
 * The collective: with -DUSE_RCCL the real one (rccl.h; one rank, because the test box has one GPU --
 * what is exercised is the call path), otherwise a stand-in for the test box, where the ranks are
 * processes that share the one GPU: a file-backed exchange area (mmap) with a counting barrier, sums
 * taken in rank order so that every rank holds the same bits.  A production host passes ncclAllReduce
 * (INTEGRATION.md).
 *
 * Every rank also solves the WHOLE problem by itself with cm2_pcg and checks: same iteration count,
 * solution equal to 1e-10.  Prints "C-SHARDED-OK".
 *
 *   pcg_sharded_demo <rank> <world> <layout> <exchange file>
 */
#define _POSIX_C_SOURCE 200809L
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#ifdef USE_RCCL
#include <rccl/rccl.h>
#endif

#include "cosmomap2.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_CM2(x) do { if ((x) != 0) { \
    fprintf(stderr, "cm2 error at %s:%d: %s\n", __FILE__, __LINE__, cm2_last_error()); return 3; } } while (0)

static double *dmalloc(size_t n)
{
    void *p = NULL;
    return hipMalloc(&p, sizeof(double) * (n ? n : 1)) == hipSuccess ? (double *)p : NULL;
}

/* ------------------------------------------------------------------ the host's collective ---- */
typedef struct {
    int rank, world;
    int64_t n;                    /* longest vector exchanged */
    double *stage;                /* private host buffer, n doubles */
#ifdef USE_RCCL
    ncclComm_t comm;
#else
    int64_t *hdr;                 /* [0] arrivals, [1] phase */
    double *slots;                /* world x n doubles, shared */
#endif
} comm_t;

#ifndef USE_RCCL
static int barrier(comm_t *c)
{
    const int64_t phase = __atomic_load_n(&c->hdr[1], __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&c->hdr[0], 1, __ATOMIC_ACQ_REL) == c->world) {
        __atomic_store_n(&c->hdr[0], 0, __ATOMIC_RELAXED);
        __atomic_store_n(&c->hdr[1], phase + 1, __ATOMIC_RELEASE);
        return 0;
    }
    const time_t t0 = time(NULL);
    while (__atomic_load_n(&c->hdr[1], __ATOMIC_ACQUIRE) == phase) {
        struct timespec ts = {0, 20000};
        nanosleep(&ts, NULL);
        if (time(NULL) - t0 > 120) { fprintf(stderr, "rank %d: the other rank never arrived\n", c->rank); return 1; }
    }
    return 0;
}
#endif

/* in place over all ranks: d[0..count) = sum (op 0) or max (op 1); rows [lo, hi) only are brought
 * back when hi > lo (a reduce-scatter), everything otherwise */
static int all_reduce(comm_t *c, double *d, int64_t count, int op, int64_t lo, int64_t hi, double *d_out,
                      void *stream)
{
    if (hi <= lo) { lo = 0; hi = count; }
#ifdef USE_RCCL
    if (lo == 0 && hi == count)
        return ncclAllReduce(d, d_out, (size_t)count, ncclDouble, op == CM2_REDUCE_MAX ? ncclMax : ncclSum,
                             c->comm, (hipStream_t)stream) != ncclSuccess;
    /* equal pieces per rank: with one rank the piece is everything */
    return ncclReduceScatter(d, d_out, (size_t)(hi - lo), ncclDouble, ncclSum, c->comm, (hipStream_t)stream)
           != ncclSuccess;
#else
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (hipMemcpy(c->stage, d, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    memcpy(c->slots + (size_t)c->rank * c->n, c->stage, sizeof(double) * count);
    if (barrier(c)) return 1;
    for (int64_t i = lo; i < hi; ++i) {
        double s = c->slots[i];
        for (int k = 1; k < c->world; ++k) {
            const double v = c->slots[(size_t)k * c->n + i];
            s = op == CM2_REDUCE_MAX ? (v > s ? v : s) : s + v;
        }
        c->stage[i] = s;
    }
    if (barrier(c)) return 1;                      /* nobody overwrites a slot another rank still reads */
    return hipMemcpy(d_out, c->stage + lo, sizeof(double) * (hi - lo), hipMemcpyHostToDevice) != hipSuccess;
#endif
}

/* d_full[0..n) = the ranks' row ranges one after the other; mine is d_rows = rows [lo, hi) */
static int all_gather(comm_t *c, const double *d_rows, int64_t lo, int64_t hi, double *d_full, int64_t n,
                      void *stream)
{
#ifdef USE_RCCL
    (void)lo; (void)n;
    return ncclAllGather(d_rows, d_full, (size_t)(hi - lo), ncclDouble, c->comm, (hipStream_t)stream) != ncclSuccess;
#else
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (hipMemcpy(c->stage, d_rows, sizeof(double) * (hi - lo), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    memcpy(c->slots + lo, c->stage, sizeof(double) * (hi - lo));        /* slot 0 is the gathered vector */
    if (barrier(c)) return 1;
    memcpy(c->stage, c->slots, sizeof(double) * n);
    if (barrier(c)) return 1;
    return hipMemcpy(d_full, c->stage, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess;
#endif
}

static int cb_reduce(void *ctx, double *d_vals, int64_t count, int op, void *stream)
{
    return all_reduce((comm_t *)ctx, d_vals, count, op, 0, 0, d_vals, stream);
}

/* ------------------------------------------------------------------ operators ---------------- */
typedef struct {
    cm2_tiles *T;
    cm2_noise *N;
    double *tb1, *tb2;
    int64_t n;                    /* whole map length */
    /* sharding */
    comm_t *c;
    int layout;
    int64_t row0, row1;
    double *full_in, *full_out;
} normal_op;

static int cb_A_whole(void *ctx, const double *d_in, double *d_out, void *stream)
{
    const normal_op *A = (const normal_op *)ctx;
    return cm2_PtNP_tiles_apply(A->T, A->N, d_in, d_out, A->tb1, A->tb2, stream);
}

static int cb_A_sharded(void *ctx, const double *d_in, double *d_out, void *stream)
{
    const normal_op *A = (const normal_op *)ctx;
    if (A->layout == CM2_LAYOUT_REPLICATED) {
        if (cm2_PtNP_tiles_apply(A->T, A->N, d_in, A->full_out, A->tb1, A->tb2, stream)) return 1;
        return all_reduce(A->c, A->full_out, A->n, CM2_REDUCE_SUM, 0, 0, d_out, stream);
    }
    if (all_gather(A->c, d_in, A->row0, A->row1, A->full_in, A->n, stream)) return 1;
    if (cm2_PtNP_tiles_apply(A->T, A->N, A->full_in, A->full_out, A->tb1, A->tb2, stream)) return 1;
    return all_reduce(A->c, A->full_out, A->n, CM2_REDUCE_SUM, A->row0, A->row1, d_out, stream);
}

typedef struct {
    int pol;
    int64_t npix;                 /* pixels this preconditioner acts on */
    const double *w[6], *det;
    const uint8_t *mask;
} bd_ctx;

static int cb_M(void *ctx, const double *d_in, double *d_out, void *stream)
{
    const bd_ctx *m = (const bd_ctx *)ctx;
    return cm2_bdprecond_apply(m->pol, m->npix, m->w[0], m->w[1], m->w[2], m->w[3], m->w[4], m->w[5],
                               m->det, m->mask, d_in, d_out, stream);
}

/* pointing plan, noise model and right-hand side b = P^T N^-1 d of the blocks [b0, b1) */
static int build(normal_op *A, int pol, int64_t npix, int64_t bs, int b0, int b1, int lambda,
                 const int32_t *pix, const double *phi, const double *d, const double *bands,
                 double *w[6], double *d_b)
{
    const int64_t nt = (int64_t)(b1 - b0) * bs, off = (int64_t)b0 * bs;
    int32_t *d_pix = NULL;
    CHECK_HIP(hipMalloc((void **)&d_pix, sizeof(int32_t) * nt));
    double *d_phi = dmalloc(nt), *d_c = dmalloc(nt), *d_s = dmalloc(nt), *d_d = dmalloc(nt);
    CHECK_HIP(hipMemcpy(d_pix, pix + off, sizeof(int32_t) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_phi, phi + off, sizeof(double) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_d, d + off, sizeof(double) * nt, hipMemcpyHostToDevice));
    CHECK_CM2(cm2_cos_sin_2phi(nt, d_phi, d_c, d_s, NULL));
    CHECK_CM2(cm2_weights_accumulate(pol, nt, npix, d_pix, NULL, d_c, d_s, w[0], w[1], w[2], w[3], w[4], w[5], NULL));
    CHECK_CM2(cm2_tiles_create(&A->T, d_pix, d_c, d_s, nt, npix, pol, 512, 4096, NULL));
    int64_t info[12], sizes[16];
    CHECK_CM2(cm2_tiles_info(A->T, info));
    for (int b = 0; b < b1 - b0; ++b) sizes[b] = bs;
    CHECK_CM2(cm2_noise_create_toeplitz(&A->N, bands + (size_t)b0 * lambda, lambda, sizes, b1 - b0,
                                        CM2_TOEPLITZ_FUSED, NULL));
    A->tb1 = dmalloc(info[1]);
    A->tb2 = dmalloc(info[1]);
    A->n = pol * npix;
    CHECK_CM2(cm2_tod_time_to_tiles(A->T, d_d, A->tb1, NULL));
    CHECK_CM2(cm2_noise_apply_tiles(A->N, A->T, A->tb1, A->tb2, NULL));
    CHECK_CM2(cm2_Pt_tiles_apply(A->T, A->tb2, d_b, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(d_phi);
    (void)hipFree(d_d);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s rank world layout exchange-file\n", argv[0]); return 9; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]), layout = atoi(argv[3]);
    const int pol = 3, nside = 16, nblocks = 4, lambda = 64;
    const int64_t npix = 12 * nside * nside, bs = 100000, nt = nblocks * bs, n = pol * npix;
    if (world < 1 || nblocks % world || rank < 0 || rank >= world || (layout != 0 && layout != 1)) return 9;

    comm_t c;
    memset(&c, 0, sizeof c);
    c.rank = rank;
    c.world = world;
    c.n = n;
    c.stage = (double *)malloc(sizeof(double) * n);
#ifdef USE_RCCL
    if (world != 1) { fprintf(stderr, "the RCCL build of this demo runs one rank (one GPU on the box)\n"); return 9; }
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&c.comm, 1, id, 0) != ncclSuccess) {
        fprintf(stderr, "RCCL communicator could not be created\n");
        return 2;
    }
#else
    {
        const size_t bytes = 64 + sizeof(double) * (size_t)world * n;
        const int fd = open(argv[4], O_RDWR);
        if (fd < 0) { perror("exchange file"); return 9; }
        void *m = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 9; }
        c.hdr = (int64_t *)m;
        c.slots = (double *)((char *)m + 64);
    }
#endif

    /* the whole problem, the same on every rank */
    int32_t *pix = (int32_t *)malloc(sizeof(int32_t) * nt);
    double *phi = (double *)malloc(sizeof(double) * nt), *d = (double *)malloc(sizeof(double) * nt);
    uint64_t st = 88172645463325252ull;
    for (int64_t i = 0; i < nt; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        pix[i] = (int32_t)(st % (uint64_t)npix);
        phi[i] = 0.4 + (2.0 * 3.14159265358979323846 * 2.5 / 200.0) * (double)i;
        d[i] = (double)((st >> 20) % 1000003) / 1000003.0;
    }
    double *bands = (double *)malloc(sizeof(double) * nblocks * lambda);
    for (int b = 0; b < nblocks; ++b)
        for (int k = 0; k < lambda; ++k)
            bands[b * lambda + k] = (1.0 + 0.05 * b) * (k == 0 ? 1.0 : -0.2 * exp(-k / 7.0));

    /* ---- reference: this process alone on all blocks ----------------------------------------- */
    normal_op W;
    memset(&W, 0, sizeof W);
    double *ww[6], *wl[6];
    for (int k = 0; k < 6; ++k) { ww[k] = dmalloc(npix); wl[k] = dmalloc(npix); }
    double *b_whole = dmalloc(n), *x_whole = dmalloc(n), *det = dmalloc(npix);
    uint8_t *mask = NULL;
    CHECK_HIP(hipMalloc((void **)&mask, npix));
    if (build(&W, pol, npix, bs, 0, nblocks, lambda, pix, phi, d, bands, ww, b_whole)) return 3;
    CHECK_CM2(cm2_bd_det_mask(pol, npix, ww[0], ww[1], ww[2], ww[3], ww[4], ww[5], det, mask, NULL));
    bd_ctx Mw = {pol, npix, {ww[0], ww[1], ww[2], ww[3], ww[4], ww[5]}, det, mask};
    int64_t it_whole = 0;
    int info_whole = -1;
    CHECK_CM2(cm2_pcg(n, cb_A_whole, &W, cb_M, &Mw, b_whole, x_whole, 1, 1e-6, 0.0, 500, NULL, NULL,
                      &it_whole, &info_whole, NULL));

    /* ---- the rank's shard --------------------------------------------------------------------- */
    const int b0 = rank * (nblocks / world), b1 = b0 + nblocks / world;
    const int64_t p0 = npix * rank / world, p1 = npix * (rank + 1) / world;     /* my pixels (rows layout) */
    normal_op A;
    memset(&A, 0, sizeof A);
    double *b_loc = dmalloc(n), *b_sum = dmalloc(n);
    if (build(&A, pol, npix, bs, b0, b1, lambda, pix, phi, d, bands, wl, b_loc)) return 3;
    A.c = &c;
    A.layout = layout;
    A.row0 = pol * p0;
    A.row1 = pol * p1;
    A.full_in = dmalloc(n);
    A.full_out = dmalloc(n);
    /* per-pixel weight sums and the right-hand side: summed over the shards once */
    for (int k = 0; k < 6; ++k)
        if (all_reduce(&c, wl[k], npix, CM2_REDUCE_SUM, 0, 0, wl[k], NULL)) return 4;
    if (all_reduce(&c, b_loc, n, CM2_REDUCE_SUM, 0, 0, b_sum, NULL)) return 4;
    double *det2 = dmalloc(npix);
    uint8_t *mask2 = NULL;
    CHECK_HIP(hipMalloc((void **)&mask2, npix));
    CHECK_CM2(cm2_bd_det_mask(pol, npix, wl[0], wl[1], wl[2], wl[3], wl[4], wl[5], det2, mask2, NULL));

    const int rows = layout == CM2_LAYOUT_ROWS;
    const int64_t n_loc = rows ? A.row1 - A.row0 : n, q0 = rows ? p0 : 0, q1 = rows ? p1 : npix;
    bd_ctx Ms = {pol, q1 - q0, {wl[0] + q0, wl[1] + q0, wl[2] + q0, wl[3] + q0, wl[4] + q0, wl[5] + q0},
                 det2 + q0, mask2 + q0};
    double *x_sh = dmalloc(n_loc);
    int64_t it_sh = 0;
    int info_sh = -1;
    CHECK_CM2(cm2_pcg_sharded(n_loc, cb_A_sharded, &A, cb_M, &Ms, b_sum + (rows ? A.row0 : 0), x_sh, 1, 1e-6, 0.0,
                              rows ? -1 : 500, NULL, NULL, layout, cb_reduce, &c, &it_sh, &info_sh, NULL));

    /* ---- compare with the un-sharded solve on my rows ----------------------------------------- */
    double *xs = (double *)malloc(sizeof(double) * n_loc), *xw = (double *)malloc(sizeof(double) * n);
    CHECK_HIP(hipMemcpy(xs, x_sh, sizeof(double) * n_loc, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(xw, x_whole, sizeof(double) * n, hipMemcpyDeviceToHost));
    double num = 0.0, den = 0.0;
    for (int64_t i = 0; i < n_loc; ++i) {
        const double ref = xw[(rows ? A.row0 : 0) + i];
        num += (xs[i] - ref) * (xs[i] - ref);
        den += ref * ref;
    }
    const double err = sqrt(num / den);
    if (!(info_whole == 0 && info_sh == 0 && it_sh == it_whole && it_whole > 3 && err < 1e-10)) {
        fprintf(stderr, "rank %d layout %d: sharded %lld iterations (info %d), whole %lld (info %d), "
                "|dx|/|x| on my rows %.3e\n", rank, layout, (long long)it_sh, info_sh, (long long)it_whole,
                info_whole, err);
        return 1;
    }
    CHECK_CM2(cm2_noise_destroy(A.N));
    CHECK_CM2(cm2_tiles_destroy(A.T));
    CHECK_CM2(cm2_noise_destroy(W.N));
    CHECK_CM2(cm2_tiles_destroy(W.T));
#ifdef USE_RCCL
    ncclCommDestroy(c.comm);
#endif
    printf("C-SHARDED-OK rank %d of %d, %s layout%s: %lld iterations (whole problem: %lld), rows %lld..%lld "
           "within %.1e of the un-sharded solution\n", rank, world, rows ? "rows" : "replicated",
#ifdef USE_RCCL
           ", collectives through RCCL",
#else
           "",
#endif
           (long long)it_sh, (long long)it_whole, (long long)(rows ? A.row0 : 0),
           (long long)(rows ? A.row1 : n), err);
    return 0;
}
