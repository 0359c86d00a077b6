/*
 * cm2_oracle_omp.c -- all-cores CPU baseline of the P^T N^-1 P matvec (TEST / BENCH
 * INFRASTRUCTURE ONLY, like cm2_oracle.c: nothing under cosmomap2_amd/ uses it).
 *
 * SURVEY 8(d) asks for two host numbers: the reference-faithful serial loops
 * (cm2_oracle.c, one thread, direct band sum) and a fair all-cores figure.  This file is
 * the second one's pointing half: the same loops as interfaces/linearoperators.py:483-489
 * (P) and :509-516 (P^T), split over OpenMP threads; P^T accumulates into one private map
 * per thread and the maps are summed afterwards (no atomics).  The Toeplitz half of that
 * baseline is an FFT convolution per noise block in oracle/oracle.py (scipy.fft, one block
 * per thread).  Built with: gcc -O3 -fopenmp -fPIC -shared.
 */
#include <omp.h>
#include <stdint.h>
#include <string.h>

int orc_omp_max_threads(void) { return omp_get_max_threads(); }

void orc_omp_P_apply(int pol, int64_t nt, const int32_t *pix, const double *c,
                     const double *s, const double *x, double *out, int nthreads)
{
    int64_t i;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (i = 0; i < nt; ++i) {
        const int64_t p = pix[i];
        if (p < 0) { out[i] = 0.0; continue; }
        if (pol == 1) out[i] = x[p];
        else if (pol == 2) out[i] = x[2 * p] * c[i] + x[2 * p + 1] * s[i];
        else out[i] = x[3 * p] + x[3 * p + 1] * c[i] + x[3 * p + 2] * s[i];
    }
}

/* scratch: nthreads * pol * npix doubles */
void orc_omp_Pt_apply(int pol, int64_t nt, int64_t npix, const int32_t *pix, const double *c,
                      const double *s, const double *v, double *out, double *scratch,
                      int nthreads)
{
    const int64_t n = pol * npix;
#pragma omp parallel num_threads(nthreads)
    {
        const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
        double *mine = scratch + (int64_t)tid * n;
        const int64_t lo = nt * tid / nth, hi = nt * (tid + 1) / nth;
        int64_t i, k;
        int t;
        memset(mine, 0, (size_t)n * sizeof(double));
        for (i = lo; i < hi; ++i) {
            const int64_t p = pix[i];
            if (p < 0) continue;
            if (pol == 1) mine[p] += v[i];
            else if (pol == 2) { mine[2 * p] += v[i] * c[i]; mine[2 * p + 1] += v[i] * s[i]; }
            else { mine[3 * p] += v[i]; mine[3 * p + 1] += v[i] * c[i]; mine[3 * p + 2] += v[i] * s[i]; }
        }
#pragma omp barrier
#pragma omp for schedule(static)
        for (k = 0; k < n; ++k) {
            double acc = 0.0;
            for (t = 0; t < nth; ++t) acc += scratch[(int64_t)t * n + k];
            out[k] = acc;
        }
    }
}
