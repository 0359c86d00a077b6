// cm2_pixel.hip -- per-pixel kernels: weight accumulation, pixel mask, compaction,
// sample flagging (ProcessTimeSamples, utilities/process_ces.py:58-555) and the 1x1 /
// 2x2 / 3x3 Stokes block operators (interfaces/linearoperators.py:700-859).
//
// All are HBM-bound; the per-pixel operators move 56-80 B in + 8*pol B out per pixel.
#include "cm2_pixindex.h"

#include <climits>
#include <cstdlib>
#include <cstring>
#include "cm2_blocks.h"

#include <hipcub/hipcub.hpp>

using namespace cm2;

// ------------------------------------------------------------------ a6 weights ---
// One thread per pixel walks that pixel's samples in time order, so every sum is
// formed in exactly the order of the reference's serial loop
// (process_ces.py:480-487 / :505-514 / :527-539).  Products keep the reference's
// association: w*c*c = (w*c)*c, w*s*c = (w*s)*c.
// (cos, sin) pairs side by side: the per-pixel walk below gathers them at random sample indices, and
// one 16-byte read touches one sector where two 8-byte reads touched two (k_weights 5.3 -> 2.9 ms at C4)
__global__ __launch_bounds__(256) void k_cos_sin_pairs(int64_t nt, const double *__restrict__ c,
                                                        const double *__restrict__ s,
                                                        double2 *__restrict__ cs)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride)
        cs[i] = make_double2(c[i], s[i]);
}

template <int POL>
__global__ __launch_bounds__(256) void k_weights(
    int64_t npix, const int64_t *__restrict__ ptr, const uint32_t *__restrict__ sorted_t,
    const double *__restrict__ w, const double2 *__restrict__ cpair,
    double *__restrict__ counts, double *__restrict__ cosine, double *__restrict__ sine,
    double *__restrict__ cos2, double *__restrict__ sin2, double *__restrict__ sincos,
    int64_t hot_min, int32_t *__restrict__ hot_pix, unsigned int *__restrict__ hot_n,
    unsigned long long *__restrict__ hot_longest)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const int64_t b = ptr[p], e = ptr[p + 1];
        if (e - b >= hot_min) {                          // left to k_weights_hot (at most nt / hot_min pixels)
            hot_pix[atomicAdd(hot_n, 1u)] = (int32_t)p;
            atomicMax(hot_longest, (unsigned long long)(e - b));
            continue;
        }
        double n = 0.0, sc = 0.0, ss = 0.0, c2 = 0.0, s2 = 0.0, cs = 0.0;
        for (int64_t k = b; k < e; ++k) {
            const uint32_t t = sorted_t[k];
            const double wt = w ? w[t] : 1.0;
            if (POL == 1) {
                n += wt;
            } else {
                const double2 q = cpair[t];
                const double ct = q.x, st = q.y;
                if (POL == 3) {
                    n += wt;
                    sc += wt * ct;
                    ss += wt * st;
                }
                c2 += wt * ct * ct;
                s2 += wt * st * st;
                cs += wt * st * ct;
            }
        }
        if (POL != 2) counts[p] = n;
        if (POL == 3) {
            cosine[p] = sc;
            sine[p] = ss;
        }
        if (POL >= 2) {
            cos2[p] = c2;
            sin2[p] = s2;
            sincos[p] = cs;
        }
    }
}

// Hot pixels.  One thread walking a pixel's samples is a chain of dependent gathers: 0.37 us a
// sample, 1.9 s for a pixel that holds 5 % of 1e8 samples (a stare at a source).  A pixel with at
// least kWeightsHotMin samples is therefore summed like the hot tiles of the fixed-order P^T
// (cm2_tiles_fixed.hip): its sample list is cut into ranges of kWeightsChunk entries, one workgroup per
// range, thread t adding the terms at positions t, t + 256, ... in that order, the 256 thread sums
// combined by a fixed halving tree; k_weights_hot_combine adds the range sums in time order.  Every
// boundary and order follows from the pixel's sample count alone: reproducible bit for bit,
// independent of the rest of the hit map; a regrouping of the serial sum (~1e-16 relative per level;
// the hit count itself, a sum of ones, is exact).  CM2_WEIGHTS_ORDER=exact keeps the serial walk.
constexpr int64_t kWeightsHotMin = 8192;
constexpr int kWeightsChunk = 4096;

template <int POL>
__global__ __launch_bounds__(256) void k_weights_hot(
    const int32_t *__restrict__ hot_pix, int64_t max_chunks, const int64_t *__restrict__ ptr,
    const uint32_t *__restrict__ sorted_t, const double *__restrict__ w, const double2 *__restrict__ cpair,
    double *__restrict__ partial)
{
    __shared__ double red[6][256];
    const int64_t h = blockIdx.y, ch = blockIdx.x;
    const int64_t p = hot_pix[h];
    const int64_t b = ptr[p] + ch * kWeightsChunk, end = ptr[p + 1];
    if (b >= end) return;
    const int64_t e = b + kWeightsChunk < end ? b + kWeightsChunk : end;
    const int t = threadIdx.x;
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // n, sum c, sum s, c2, s2, cs
    for (int64_t k = b + t; k < e; k += 256) {
        const uint32_t ts = sorted_t[k];
        const double wt = w ? w[ts] : 1.0;
        if (POL == 1) {
            a[0] += wt;
        } else {
            const double2 q = cpair[ts];
            const double ct = q.x, st = q.y;
            if (POL == 3) {
                a[0] += wt;
                a[1] += wt * ct;
                a[2] += wt * st;
            }
            a[3] += wt * ct * ct;
            a[4] += wt * st * st;
            a[5] += wt * st * ct;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) red[q][t] = a[q];
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (t < half)
#pragma unroll
            for (int q = 0; q < 6; ++q) red[q][t] += red[q][t + half];
        __syncthreads();
    }
    if (t < 6) partial[(h * max_chunks + ch) * 6 + t] = red[t][0];
}

template <int POL>
__global__ __launch_bounds__(64) void k_weights_hot_combine(
    int64_t nhot, const int32_t *__restrict__ hot_pix, int64_t max_chunks,
    const int64_t *__restrict__ ptr, const double *__restrict__ partial,
    double *__restrict__ counts, double *__restrict__ cosine, double *__restrict__ sine,
    double *__restrict__ cos2, double *__restrict__ sin2, double *__restrict__ sincos)
{
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nhot) return;
    const int64_t p = hot_pix[h];
    const int64_t nch = (ptr[p + 1] - ptr[p] + kWeightsChunk - 1) / kWeightsChunk;
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t ch = 0; ch < nch; ++ch)
#pragma unroll
        for (int q = 0; q < 6; ++q) a[q] += partial[(h * max_chunks + ch) * 6 + q];
    if (POL != 2) counts[p] = a[0];
    if (POL == 3) {
        cosine[p] = a[1];
        sine[p] = a[2];
    }
    if (POL >= 2) {
        cos2[p] = a[3];
        sin2[p] = a[4];
        sincos[p] = a[5];
    }
}

extern "C" int cm2_weights_accumulate(int pol, int64_t nt, int64_t npix, const int32_t *d_pix,
                                      const double *d_w, const double *d_cos,
                                      const double *d_sin, double *d_counts, double *d_cosine,
                                      double *d_sine, double *d_cos2, double *d_sin2,
                                      double *d_sincos, void *stream_)
{
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_weights_accumulate: bad pol=%d", pol);
    CM2_CHECK(pol == 1 || (d_cos && d_sin), "cm2_weights_accumulate: cos/sin required");
    CM2_CHECK(pol == 2 || d_counts, "cm2_weights_accumulate: d_counts is NULL");
    CM2_CHECK(pol == 1 || (d_cos2 && d_sin2 && d_sincos), "cm2_weights_accumulate: NULL output");
    CM2_CHECK(pol != 3 || (d_cosine && d_sine), "cm2_weights_accumulate: NULL output");
    hipStream_t stream = as_stream(stream_);
    PixIndex ix;
    struct IxGuard { PixIndex *ix; ~IxGuard() { ix->release(); } } guard{&ix};
    if (int rc = build_pixindex(ix, d_pix, nt, npix, stream)) return rc;
    const int g = grid_for(npix);
    DevTemp<double2> d_cs;
    if (pol > 1) {
        CM2_HIP(d_cs.alloc(nt));
        k_cos_sin_pairs<<<grid_for(nt), kBlock, 0, stream>>>(nt, d_cos, d_sin, d_cs);
        CM2_LAUNCH_OK();
    }
    // pixels with >= hot_min samples are listed by k_weights and summed by k_weights_hot
    int64_t hot_min = kWeightsHotMin;
    if (const char *e = getenv("CM2_WEIGHTS_ORDER"))
        if (strcmp(e, "exact") == 0) hot_min = INT64_MAX;
    if (cm2::exact_order_setting() >= 0) hot_min = cm2::exact_order_setting() ? INT64_MAX : kWeightsHotMin;
    DevTemp<int32_t> d_hot_pix;
    DevTemp<unsigned int> d_hot_n;
    DevTemp<unsigned long long> d_hot_longest;
    CM2_HIP(d_hot_pix.alloc((size_t)(nt / kWeightsHotMin + 1)));
    CM2_HIP(d_hot_n.alloc(1));
    CM2_HIP(d_hot_longest.alloc(1));
    CM2_HIP(hipMemsetAsync(d_hot_n.p, 0, sizeof(unsigned int), stream));
    CM2_HIP(hipMemsetAsync(d_hot_longest.p, 0, sizeof(unsigned long long), stream));
#define CM2_W(POL)                                                                          \
    k_weights<POL><<<g, kBlock, 0, stream>>>(npix, ix.d_ptr, ix.d_sorted_t, d_w, d_cs.p,    \
                                             d_counts, d_cosine, d_sine, d_cos2,            \
                                             d_sin2, d_sincos, hot_min, d_hot_pix, d_hot_n, \
                                             d_hot_longest)
    if (pol == 1) CM2_W(1); else if (pol == 2) CM2_W(2); else CM2_W(3);
#undef CM2_W
    CM2_LAUNCH_OK();
    unsigned int nhot = 0;
    unsigned long long longest = 0;
    CM2_HIP(cm2::download(&nhot, d_hot_n.p, sizeof(nhot), stream));
    CM2_HIP(cm2::download(&longest, d_hot_longest.p, sizeof(longest), stream));
    CM2_HIP(hipStreamSynchronize(stream));
    if (nhot > 0) {
        const int64_t max_chunks = ((int64_t)longest + kWeightsChunk - 1) / kWeightsChunk;
        CM2_CHECK(max_chunks <= 0x7FFFFFFF, "cm2_weights_accumulate: a pixel with %llu samples", longest);
        // (grid.y <= 65535: the hot pixels are taken in batches)
        const unsigned int batch_max = 65535u;
        DevTemp<double> d_partial;
        CM2_HIP(d_partial.alloc((size_t)(nhot < batch_max ? nhot : batch_max) * (size_t)max_chunks * 6));
#define CM2_WH(POL)                                                                              \
    for (unsigned int h0 = 0; h0 < nhot; h0 += batch_max) {                                      \
        const unsigned int nb = nhot - h0 < batch_max ? nhot - h0 : batch_max;                   \
        k_weights_hot<POL><<<dim3((unsigned)max_chunks, nb), 256, 0, stream>>>(                  \
            d_hot_pix.p + h0, max_chunks, ix.d_ptr, ix.d_sorted_t, d_w, d_cs.p, d_partial);       \
        k_weights_hot_combine<POL><<<(nb + 63) / 64, 64, 0, stream>>>(                           \
            nb, d_hot_pix.p + h0, max_chunks, ix.d_ptr, d_partial, d_counts, d_cosine, d_sine,   \
            d_cos2, d_sin2, d_sincos);                                                           \
    }
        if (pol == 1) CM2_WH(1) else if (pol == 2) CM2_WH(2) else CM2_WH(3)
#undef CM2_WH
        CM2_LAUNCH_OK();
        CM2_HIP(hipStreamSynchronize(stream));
    }
    return 0;
}

// --------------------------------------------------------------- a6 pixel mask ---
// process_ces.py:491 (pol=1) and :544-555 (pol>=2).  NaN condition numbers
// (unobserved pixels: 0/0) compare false, exactly as np.where(cond_num<=thr) does.
__global__ __launch_bounds__(256) void k_pixel_mask(int pol, int64_t npix,
                                                     const double *__restrict__ counts,
                                                     const double *__restrict__ cos2,
                                                     const double *__restrict__ sin2,
                                                     const double *__restrict__ sincos,
                                                     double threshold, uint8_t *__restrict__ keep)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        bool k;
        if (pol == 1) {
            k = counts[p] > 0.0;
        } else {
            const double det = (cos2[p] * sin2[p]) - (sincos[p] * sincos[p]);
            const double tr = cos2[p] + sin2[p];
            const double sq = sqrt(tr * tr / 4. - det);
            const double lmax = tr / 2. + sq;
            const double lmin = tr / 2. - sq;
            const double cond = fabs(lmax / lmin);
            k = cond <= threshold;
            if (pol == 3) k = k && (counts[p] > 2.0);
        }
        keep[p] = k ? 1 : 0;
    }
}

extern "C" int cm2_pixel_mask(int pol, int64_t npix, const double *d_counts,
                              const double *d_cos2, const double *d_sin2,
                              const double *d_sincos, double threshold, uint8_t *d_keep,
                              void *stream_)
{
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_pixel_mask: bad pol=%d", pol);
    CM2_CHECK(d_keep != nullptr, "cm2_pixel_mask: d_keep is NULL");
    k_pixel_mask<<<grid_for(npix), kBlock, 0, as_stream(stream_)>>>(pol, npix, d_counts, d_cos2,
                                                                    d_sin2, d_sincos, threshold,
                                                                    d_keep);
    CM2_LAUNCH_OK();
    return 0;
}

// -------------------------------------------------------------- a7 compaction ---
__global__ __launch_bounds__(256) void k_keep_to_i32(int64_t n, const uint8_t *__restrict__ keep,
                                                      int32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = keep[i] ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_rank_to_old2new(int64_t n,
                                                          const uint8_t *__restrict__ keep,
                                                          int32_t *__restrict__ old2new)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (!keep[i]) old2new[i] = -1;
}

extern "C" int cm2_pixel_compact(int64_t npix, const uint8_t *d_keep, int32_t *d_old2new,
                                 int64_t *h_new_npix, void *stream_)
{
    CM2_CHECK(d_keep && d_old2new && h_new_npix, "cm2_pixel_compact: NULL argument");
    hipStream_t stream = as_stream(stream_);
    DevTemp<int32_t> flags;
    DevTemp<char> d_temp;
    size_t tb = 0;
    CM2_HIP(flags.alloc(npix + 1));
    k_keep_to_i32<<<grid_for(npix), kBlock, 0, stream>>>(npix, d_keep, flags);
    CM2_LAUNCH_OK();
    // exclusive prefix sum of the keep flags = rank among kept pixels
    CM2_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, flags.p, d_old2new, npix, stream));
    CM2_HIP(d_temp.alloc(tb + 16));
    CM2_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp.p, tb, flags.p, d_old2new, npix, stream));
    int32_t last_rank = 0;
    uint8_t last_keep = 0;
    CM2_HIP(cm2::download(&last_rank, d_old2new + (npix - 1), sizeof(int32_t), stream));
    CM2_HIP(cm2::download(&last_keep, d_keep + (npix - 1), sizeof(uint8_t), stream));
    k_rank_to_old2new<<<grid_for(npix), kBlock, 0, stream>>>(npix, d_keep, d_old2new);
    CM2_LAUNCH_OK();
    CM2_HIP(hipStreamSynchronize(stream));
    *h_new_npix = (int64_t)last_rank + (last_keep ? 1 : 0);
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_compact(int64_t n, const int32_t *__restrict__ old2new,
                                                  const T *__restrict__ in, T *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t j = old2new[i];
        if (j >= 0) out[j] = in[i];
    }
}

extern "C" int cm2_compact_f64(int64_t npix, const int32_t *d_old2new, const double *d_in,
                               double *d_out, void *stream_)
{
    CM2_CHECK(d_old2new && d_in && d_out, "cm2_compact_f64: NULL argument");
    k_compact<double><<<grid_for(npix), kBlock, 0, as_stream(stream_)>>>(npix, d_old2new, d_in,
                                                                         d_out);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_compact_i64(int64_t npix, const int32_t *d_old2new, const int64_t *d_in,
                               int64_t *d_out, void *stream_)
{
    CM2_CHECK(d_old2new && d_in && d_out, "cm2_compact_i64: NULL argument");
    k_compact<int64_t><<<grid_for(npix), kBlock, 0, as_stream(stream_)>>>(npix, d_old2new, d_in,
                                                                          d_out);
    CM2_LAUNCH_OK();
    return 0;
}

// process_ces.py:411-418
__global__ __launch_bounds__(256) void k_flag(int64_t nt, int32_t *__restrict__ pix,
                                               const int32_t *__restrict__ old2new)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const int32_t p = pix[i];
        if (p == -1) continue;
        pix[i] = old2new[p];
    }
}

extern "C" int cm2_flag_samples(int64_t nt, int32_t *d_pix, const int32_t *d_old2new,
                                void *stream_)
{
    CM2_CHECK(d_pix && d_old2new, "cm2_flag_samples: NULL argument");
    if (nt == 0) return 0;
    k_flag<<<grid_for(nt), kBlock, 0, as_stream(stream_)>>>(nt, d_pix, d_old2new);
    CM2_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------- a8 det / mask / M_BD ---
// NumPy lines linearoperators.py:792-795 (pol 3) and :820-821 (pol 2), left to right.
__global__ __launch_bounds__(256) void k_bd_det_mask(
    int pol, int64_t npix, const double *__restrict__ counts, const double *__restrict__ c,
    const double *__restrict__ s, const double *__restrict__ c2, const double *__restrict__ s2,
    const double *__restrict__ cs, double *__restrict__ det, uint8_t *__restrict__ mask)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npix; j += stride) {
        if (pol == 1) {
            det[j] = counts[j];
            mask[j] = counts[j] > 0.0 ? 1 : 0;
        } else {
            double d;
            if (pol == 3) {
                d = counts[j] * (c2[j] * s2[j] - cs[j] * cs[j]) - c[j] * c[j] * s2[j]
                    - s[j] * s[j] * c2[j] + 2. * c[j] * s[j] * cs[j];
            } else {
                d = (c2[j] * s2[j]) - (cs[j] * cs[j]);
            }
            det[j] = d;
            mask[j] = fabs(d) > 1e-5 ? 1 : 0;
        }
    }
}

extern "C" int cm2_bd_det_mask(int pol, int64_t npix, const double *d_counts,
                               const double *d_cosine, const double *d_sine,
                               const double *d_cos2, const double *d_sin2,
                               const double *d_sincos, double *d_det, uint8_t *d_mask,
                               void *stream_)
{
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_bd_det_mask: bad pol=%d", pol);
    CM2_CHECK(d_det && d_mask, "cm2_bd_det_mask: NULL output");
    k_bd_det_mask<<<grid_for(npix), kBlock, 0, as_stream(stream_)>>>(
        pol, npix, d_counts, d_cosine, d_sine, d_cos2, d_sin2, d_sincos, d_det, d_mask);
    CM2_LAUNCH_OK();
    return 0;
}

template <int POL>
__global__ __launch_bounds__(256) void k_bdprecond(
    int64_t npix, const double *__restrict__ hits, const double *__restrict__ c,
    const double *__restrict__ s, const double *__restrict__ c2, const double *__restrict__ s2,
    const double *__restrict__ cs, const double *__restrict__ det,
    const uint8_t *__restrict__ mask, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npix; j += stride) {
        double xin[3], yo[3];
#pragma unroll
        for (int k = 0; k < POL; ++k) xin[k] = x[POL * j + k];
        const bool m = mask[j] != 0;
        if (POL == 1)
            bd_inverse_block<1>(hits[j], 0, 0, 0, 0, 0, 0, m, xin, yo);
        else if (POL == 2)
            bd_inverse_block<2>(0, 0, 0, c2[j], s2[j], cs[j], det[j], m, xin, yo);
        else
            bd_inverse_block<3>(hits[j], c[j], s[j], c2[j], s2[j], cs[j], det[j], m, xin, yo);
#pragma unroll
        for (int k = 0; k < POL; ++k) y[POL * j + k] = yo[k];
    }
}

extern "C" int cm2_bdprecond_apply(int pol, int64_t npix, const double *d_counts,
                                   const double *d_cosine, const double *d_sine,
                                   const double *d_cos2, const double *d_sin2,
                                   const double *d_sincos, const double *d_det,
                                   const uint8_t *d_mask, const double *d_x, double *d_y,
                                   void *stream_)
{
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_bdprecond_apply: bad pol=%d", pol);
    CM2_CHECK(d_mask && d_x && d_y, "cm2_bdprecond_apply: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const int g = grid_for(npix);
#define CM2_BDP(POL)                                                                      \
    k_bdprecond<POL><<<g, kBlock, 0, stream>>>(npix, d_counts, d_cosine, d_sine, d_cos2,  \
                                               d_sin2, d_sincos, d_det, d_mask, d_x, d_y)
    if (pol == 1) CM2_BDP(1); else if (pol == 2) CM2_BDP(2); else CM2_BDP(3);
#undef CM2_BDP
    CM2_LAUNCH_OK();
    return 0;
}

// ----------------------------------------------------------- a9 forward block ---
template <int POL>
__global__ __launch_bounds__(256) void k_bd_apply(
    int64_t npix, const double *__restrict__ hits, const double *__restrict__ c,
    const double *__restrict__ s, const double *__restrict__ c2, const double *__restrict__ s2,
    const double *__restrict__ cs, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        if (POL == 1) {
            y[p] = x[p] * hits[p];                                       // :735
        } else if (POL == 2) {
            const double x0 = x[2 * p], x1 = x[2 * p + 1];               // :743-745
            y[2 * p] = c2[p] * x0 + cs[p] * x1;
            y[2 * p + 1] = cs[p] * x0 + s2[p] * x1;
        } else {
            const double x0 = x[3 * p], x1 = x[3 * p + 1], x2 = x[3 * p + 2];   // :737-741
            y[3 * p] = hits[p] * x0 + c[p] * x1 + s[p] * x2;
            y[3 * p + 1] = c[p] * x0 + c2[p] * x1 + cs[p] * x2;
            y[3 * p + 2] = s[p] * x0 + cs[p] * x1 + s2[p] * x2;
        }
    }
}

extern "C" int cm2_bd_apply(int pol, int64_t npix, const double *d_counts,
                            const double *d_cosine, const double *d_sine, const double *d_cos2,
                            const double *d_sin2, const double *d_sincos, const double *d_x,
                            double *d_y, void *stream_)
{
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_bd_apply: bad pol=%d", pol);
    CM2_CHECK(d_x && d_y, "cm2_bd_apply: NULL argument");
    hipStream_t stream = as_stream(stream_);
    const int g = grid_for(npix);
#define CM2_BD(POL)                                                                       \
    k_bd_apply<POL><<<g, kBlock, 0, stream>>>(npix, d_counts, d_cosine, d_sine, d_cos2,   \
                                              d_sin2, d_sincos, d_x, d_y)
    if (pol == 1) CM2_BD(1); else if (pol == 2) CM2_BD(2); else CM2_BD(3);
#undef CM2_BD
    CM2_LAUNCH_OK();
    return 0;
}

// ---- f3: cut-sky map vector <-> full-sky HEALPix maps ------------------------------------
// reorganize_map (utilities/healpy_functions.py:47-102): map component k of observed pixel i,
// mapin[pol*i + k], goes to full[k][obspix[i]]; every other full-sky pixel is 0.
// full2cutskymap (utilities/IOfiles.py:377-393) is the way back.
namespace cm2 {

__global__ __launch_bounds__(256) void k_cut_to_full(int pol, int64_t npix,
                                                      const int64_t *__restrict__ obspix,
                                                      const double *__restrict__ map,
                                                      int64_t nfull, double *__restrict__ full)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const int64_t p = obspix[i];
        for (int k = 0; k < pol; ++k) full[(int64_t)k * nfull + p] = map[pol * i + k];
    }
}

__global__ __launch_bounds__(256) void k_full_to_cut(int pol, int64_t npix,
                                                      const int64_t *__restrict__ obspix,
                                                      const double *__restrict__ full, int64_t nfull,
                                                      double *__restrict__ map)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const int64_t p = obspix[i];
        for (int k = 0; k < pol; ++k) map[pol * i + k] = full[(int64_t)k * nfull + p];
    }
}

__global__ __launch_bounds__(256) void k_obspix_range(int64_t npix, const int64_t *__restrict__ obspix,
                                                       int64_t nfull, unsigned int *__restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride)
        if (obspix[i] < 0 || obspix[i] >= nfull) atomicAdd(bad, 1u);
}

static int check_obspix(int64_t npix, const int64_t *d_obspix, int64_t nfull, hipStream_t st)
{
    DevTemp<unsigned int> d_bad;
    CM2_HIP(d_bad.alloc(1));
    CM2_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), st));
    k_obspix_range<<<grid_for(npix), kBlock, 0, st>>>(npix, d_obspix, nfull, d_bad);
    CM2_LAUNCH_OK();
    unsigned int h_bad = 0;
    CM2_HIP(cm2::download(&h_bad, d_bad, sizeof(h_bad), st));
    CM2_HIP(hipStreamSynchronize(st));
    CM2_CHECK(h_bad == 0, "%u observed-pixel ids lie outside the full-sky map of %lld pixels", h_bad,
              (long long)nfull);
    return 0;
}

}  // namespace cm2

extern "C" int cm2_cutsky_to_fullsky(int pol, int64_t npix, const int64_t *d_obspix,
                                     const double *d_map, int64_t nfull, double *d_full,
                                     void *stream)
{
    CM2_CHECK(pol >= 1 && pol <= 3, "cm2_cutsky_to_fullsky: pol=%d", pol);
    CM2_CHECK(npix >= 0 && nfull >= 0, "cm2_cutsky_to_fullsky: negative size");
    hipStream_t st = cm2::as_stream(stream);
    if (nfull) {
        CM2_CHECK(d_full, "cm2_cutsky_to_fullsky: null output");
        CM2_HIP(hipMemsetAsync(d_full, 0, sizeof(double) * (size_t)pol * nfull, st));
    }
    if (npix == 0) return 0;
    CM2_CHECK(d_obspix && d_map, "cm2_cutsky_to_fullsky: null input");
    if (int rc = cm2::check_obspix(npix, d_obspix, nfull, st)) return rc;
    cm2::k_cut_to_full<<<cm2::grid_for(npix), cm2::kBlock, 0, st>>>(pol, npix, d_obspix, d_map, nfull,
                                                                     d_full);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_fullsky_to_cutsky(int pol, int64_t npix, const int64_t *d_obspix,
                                     const double *d_full, int64_t nfull, double *d_map,
                                     void *stream)
{
    CM2_CHECK(pol >= 1 && pol <= 3, "cm2_fullsky_to_cutsky: pol=%d", pol);
    CM2_CHECK(npix >= 0 && nfull >= 0, "cm2_fullsky_to_cutsky: negative size");
    if (npix == 0) return 0;
    CM2_CHECK(d_obspix && d_full && d_map, "cm2_fullsky_to_cutsky: null argument");
    hipStream_t st = cm2::as_stream(stream);
    if (int rc = cm2::check_obspix(npix, d_obspix, nfull, st)) return rc;
    cm2::k_full_to_cut<<<cm2::grid_for(npix), cm2::kBlock, 0, st>>>(pol, npix, d_obspix, d_full, nfull,
                                                                     d_map);
    CM2_LAUNCH_OK();
    return 0;
}
