/*
 * A host that is not Python: plain C99 + the HIP runtime C API + include/cosmomap2.h.
 * Builds a small IQU pointing on the GPU, applies P and P^T through the C ABI and compares with
 * the reference's serial loops (interfaces/linearoperators.py:463-497 and :498-526) on the host;
 * then the block-diagonal preconditioner build (utilities/process_ces.py:480-555) + apply.
 * Exit code 0 and "C-ABI-OK" on success.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi/matvec_demo.c \
 *       -Lcosmomap2_amd -lcosmomap2_hip -L/opt/rocm/lib -lamdhip64 -lm -o demo
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

int main(void)
{
    const int64_t nt = 50000, npix = 700;
    const int pol = 3;
    int32_t *pix = (int32_t *)malloc(sizeof(int32_t) * nt);
    double *c = (double *)malloc(sizeof(double) * nt), *s = (double *)malloc(sizeof(double) * nt);
    double *x = (double *)malloc(sizeof(double) * pol * npix);
    double *tod = (double *)calloc(nt, sizeof(double)), *map = (double *)calloc(pol * npix, sizeof(double));
    double *tod_g = (double *)malloc(sizeof(double) * nt), *map_g = (double *)malloc(sizeof(double) * pol * npix);
    uint64_t st = 88172645463325252ull;                      /* xorshift: no libc rand() differences */
    for (int64_t i = 0; i < nt; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        pix[i] = (st % 23 == 0) ? -1 : (int32_t)(st % (uint64_t)npix);
        const double phi = 0.3 + 0.0785 * (double)i;
        c[i] = cos(2.0 * phi);
        s[i] = sin(2.0 * phi);
    }
    for (int64_t j = 0; j < pol * npix; ++j) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        x[j] = (double)(st % 2000001) / 1.0e6 - 1.0;
    }
    /* the reference's loops */
    for (int64_t i = 0; i < nt; ++i) {
        if (pix[i] < 0) continue;
        const double *m = x + 3 * (int64_t)pix[i];
        tod[i] = m[0] + m[1] * c[i] + m[2] * s[i];
    }
    for (int64_t i = 0; i < nt; ++i) {
        if (pix[i] < 0) continue;
        double *m = map + 3 * (int64_t)pix[i];
        m[0] += tod[i];
        m[1] += tod[i] * c[i];
        m[2] += tod[i] * s[i];
    }

    int32_t *d_pix; double *d_c, *d_s, *d_x, *d_tod, *d_map;
    CHECK_HIP(hipMalloc((void **)&d_pix, sizeof(int32_t) * nt));
    CHECK_HIP(hipMalloc((void **)&d_c, sizeof(double) * nt));
    CHECK_HIP(hipMalloc((void **)&d_s, sizeof(double) * nt));
    CHECK_HIP(hipMalloc((void **)&d_x, sizeof(double) * pol * npix));
    CHECK_HIP(hipMalloc((void **)&d_tod, sizeof(double) * nt));
    CHECK_HIP(hipMalloc((void **)&d_map, sizeof(double) * pol * npix));
    CHECK_HIP(hipMemcpy(d_pix, pix, sizeof(int32_t) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_c, c, sizeof(double) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_s, s, sizeof(double) * nt, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_x, x, sizeof(double) * pol * npix, hipMemcpyHostToDevice));

    cm2_pointing *P = NULL;
    CHECK_CM2(cm2_pointing_create(&P, d_pix, d_c, d_s, nt, npix, pol, NULL));
    CHECK_CM2(cm2_P_apply(P, d_x, d_tod, NULL));
    CHECK_CM2(cm2_Pt_apply(P, d_tod, d_map, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(tod_g, d_tod, sizeof(double) * nt, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(map_g, d_map, sizeof(double) * pol * npix, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int64_t i = 0; i < nt; ++i) bad += (tod_g[i] != tod[i]);
    for (int64_t j = 0; j < pol * npix; ++j) bad += (map_g[j] != map[j]);   /* same order of additions */
    if (bad) {
        fprintf(stderr, "%lld values differ from the serial loops\n", (long long)bad);
        return 1;
    }
    /* a wrong polarisation key must fail with the reference's message class, not crash */
    cm2_pointing *Q = NULL;
    if (cm2_pointing_create(&Q, d_pix, d_c, d_s, nt, npix, 4, NULL) == 0 || Q != NULL) return 4;
    CHECK_CM2(cm2_pointing_destroy(P));
    hipFree(d_pix); hipFree(d_c); hipFree(d_s); hipFree(d_x); hipFree(d_tod); hipFree(d_map);
    free(pix); free(c); free(s); free(x); free(tod); free(map); free(tod_g); free(map_g);
    printf("C-ABI-OK abi=%d P and P^T bit-identical to the serial loops (%lld samples, %lld pixels)\n",
           cm2_abi_version(), (long long)nt, (long long)npix);
    return 0;
}
