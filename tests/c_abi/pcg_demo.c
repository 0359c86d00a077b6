/*
 * The whole hot path from a C99 host, no Python: weights and per-pixel blocks
 * (utilities/process_ces.py:480-555), tile-order P / N^-1 / P^T
 * (interfaces/linearoperators.py:463-526, :582-595), the block-diagonal preconditioner (:775-841)
 * and the PCG recurrence of scipy.sparse.linalg.cg with alpha and beta kept in HBM -- every
 * array-sized operation is one call of include/cosmomap2.h -- written out here and, second,
 * through the library's own driver cm2_pcg with the operator and M_BD as C callbacks (same
 * iteration count, bit-identical solution).  Solves P^T N^-1 P x = P^T N^-1 d to
 * rtol 1e-6 and checks the true residual.  Prints "C-PCG-OK".
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

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

typedef struct {
    cm2_tiles *T;
    cm2_noise *N;
    double *tb1, *tb2;
} normal_op;

/* y = P^T N^-1 P x on the tile order */
static int apply_A(const normal_op *A, const double *d_x, double *d_y)
{
    if (cm2_P_tiles_apply(A->T, d_x, A->tb1, NULL)) return 1;
    if (cm2_noise_apply_tiles(A->N, A->T, A->tb1, A->tb2, NULL)) return 1;
    return cm2_Pt_tiles_apply(A->T, A->tb2, d_y, NULL);
}

static int cb_A(void *ctx, const double *d_in, double *d_out, void *stream)
{
    (void)stream;                                    /* this demo runs on the default stream */
    return apply_A((const normal_op *)ctx, d_in, d_out);
}

typedef struct {
    int pol;
    int64_t npix;
    const double *w[6], *det;
    const uint8_t *mask;
} bd_ctx;

static int cb_M(void *ctx, const double *d_in, double *d_out, void *stream)
{
    const bd_ctx *m = (const bd_ctx *)ctx;
    return cm2_bdprecond_apply(m->pol, m->npix, m->w[0], m->w[1], m->w[2], m->w[3], m->w[4], m->w[5],
                               m->det, m->mask, d_in, d_out, stream);
}

int main(void)
{
    const int pol = 3, nside = 16, nblocks = 4, lambda = 64;
    const int64_t npix = 12 * nside * nside, bs = 300000, nt = nblocks * bs, n = pol * npix;
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
    int64_t sizes[4];
    for (int b = 0; b < nblocks; ++b) {
        sizes[b] = bs;
        for (int k = 0; k < lambda; ++k)
            bands[b * lambda + k] = (1.0 + 0.05 * b) * (k == 0 ? 1.0 : -0.2 * exp(-k / 7.0));
    }

    int32_t *d_pix = NULL;
    CHECK_HIP(hipMalloc((void **)&d_pix, sizeof(int32_t) * nt));
    double *d_phi = dmalloc(nt), *d_c = dmalloc(nt), *d_s = dmalloc(nt), *d_d = dmalloc(nt);
    CHECK_HIP(hipMemcpy(d_pix, pix, sizeof(int32_t) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_phi, phi, sizeof(double) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_d, d, sizeof(double) * nt, hipMemcpyHostToDevice));
    CHECK_CM2(cm2_cos_sin_2phi(nt, d_phi, d_c, d_s, NULL));

    /* ProcessTimeSamples: the six per-pixel sums, then det and mask of the 3x3 blocks */
    double *w[6];
    for (int k = 0; k < 6; ++k) w[k] = dmalloc(npix);
    double *d_det = dmalloc(npix);
    uint8_t *d_mask = NULL;
    CHECK_HIP(hipMalloc((void **)&d_mask, npix));
    CHECK_CM2(cm2_weights_accumulate(pol, nt, npix, d_pix, NULL, d_c, d_s, w[0], w[1], w[2], w[3], w[4], w[5], NULL));
    CHECK_CM2(cm2_bd_det_mask(pol, npix, w[0], w[1], w[2], w[3], w[4], w[5], d_det, d_mask, NULL));

    normal_op A;
    CHECK_CM2(cm2_tiles_create(&A.T, d_pix, d_c, d_s, nt, npix, pol, 512, 4096, NULL));
    CHECK_CM2(cm2_tiles_prepare_pt(A.T, NULL));      /* fixed-order P^T lists now, not in the first apply */
    int64_t info[10];
    CHECK_CM2(cm2_tiles_info(A.T, info));
    const int64_t nvalid = info[1];
    CHECK_CM2(cm2_noise_create_toeplitz(&A.N, bands, lambda, sizes, nblocks, CM2_TOEPLITZ_FUSED, NULL));
    A.tb1 = dmalloc(nvalid);
    A.tb2 = dmalloc(nvalid);

    double *b = dmalloc(n), *x = dmalloc(n), *r = dmalloc(n), *z = dmalloc(n), *p = dmalloc(n), *q = dmalloc(n);
    double *work = dmalloc((size_t)cm2_reduce_work_doubles());
    double *sc = dmalloc(8);                         /* rho, rho_prev, pq, rr, bb */
    /* b = P^T N^-1 d */
    CHECK_CM2(cm2_tod_time_to_tiles(A.T, d_d, A.tb1, NULL));
    CHECK_CM2(cm2_noise_apply_tiles(A.N, A.T, A.tb1, A.tb2, NULL));
    CHECK_CM2(cm2_Pt_tiles_apply(A.T, A.tb2, b, NULL));

    /* scipy.sparse.linalg.cg: x0 = 0, r = b, atol = rtol ||b|| */
    CHECK_HIP(hipMemset(x, 0, sizeof(double) * n));
    CHECK_HIP(hipMemcpy(r, b, sizeof(double) * n, hipMemcpyDeviceToDevice));
    double h[2];
    CHECK_CM2(cm2_dot(n, b, b, sc + 4, work, NULL));
    CHECK_HIP(hipMemcpy(h, sc + 4, sizeof(double), hipMemcpyDeviceToHost));
    const double atol = 1e-6 * sqrt(h[0]);
    double rr = h[0];
    int it = 0;
    for (; it < 500; ++it) {
        if (sqrt(rr) < atol) break;
        CHECK_CM2(cm2_bdprecond_apply(pol, npix, w[0], w[1], w[2], w[3], w[4], w[5], d_det, d_mask, r, z, NULL));
        CHECK_HIP(hipMemcpyAsync(sc + 1, sc, sizeof(double), hipMemcpyDeviceToDevice, NULL));     /* rho_prev */
        CHECK_CM2(cm2_dot(n, r, z, sc, work, NULL));                                               /* rho */
        if (it == 0) CHECK_HIP(hipMemcpyAsync(p, z, sizeof(double) * n, hipMemcpyDeviceToDevice, NULL));
        else CHECK_CM2(cm2_pcg_update_p(n, sc, sc + 1, z, p, NULL));
        if (apply_A(&A, p, q)) { fprintf(stderr, "%s\n", cm2_last_error()); return 3; }
        CHECK_CM2(cm2_dot(n, p, q, sc + 2, work, NULL));                                           /* pq */
        CHECK_CM2(cm2_pcg_update_xr(n, sc, sc + 2, p, q, x, r, sc + 3, work, NULL));
        CHECK_HIP(hipMemcpy(&rr, sc + 3, sizeof(double), hipMemcpyDeviceToHost));
    }
    /* true residual */
    if (apply_A(&A, x, q)) return 3;
    CHECK_CM2(cm2_axpy(n, -1.0, b, q, NULL));                /* q = A x - b */
    CHECK_CM2(cm2_dot(n, q, q, sc + 3, work, NULL));
    CHECK_HIP(hipMemcpy(h, sc + 3, sizeof(double), hipMemcpyDeviceToHost));
    const double rel = sqrt(h[0]) / (atol / 1e-6);
    if (!(it > 0 && it < 500 && rel < 2e-6)) {
        fprintf(stderr, "PCG failed: %d iterations, true relative residual %.3e\n", it, rel);
        return 1;
    }
    /* the same solve by the library's driver, operator and preconditioner as C callbacks */
    bd_ctx Mc = {pol, npix, {w[0], w[1], w[2], w[3], w[4], w[5]}, d_det, d_mask};
    double *x2 = dmalloc(n);
    int64_t iters2 = 0;
    int info2 = -1;
    CHECK_CM2(cm2_pcg(n, cb_A, &A, cb_M, &Mc, b, x2, 1, 1e-6, 0.0, 500, NULL, NULL, &iters2, &info2, NULL));
    CHECK_CM2(cm2_axpy(n, -1.0, x, x2, NULL));               /* x2 - x */
    CHECK_CM2(cm2_dot(n, x2, x2, sc + 3, work, NULL));
    CHECK_HIP(hipMemcpy(h, sc + 3, sizeof(double), hipMemcpyDeviceToHost));
    if (info2 != 0 || iters2 != it || h[0] != 0.0) {
        fprintf(stderr, "cm2_pcg: info %d, %lld iterations (loop above: %d), |dx|^2 = %.3e\n", info2,
                (long long)iters2, it, h[0]);
        return 1;
    }
    CHECK_CM2(cm2_noise_destroy(A.N));
    CHECK_CM2(cm2_tiles_destroy(A.T));
    printf("C-PCG-OK %d iterations, true relative residual %.2e (%lld samples, %lld pixels, lambda %d)\n",
           it, rel, (long long)nt, (long long)npix, lambda);
    return 0;
}
