// cm2_pointing.hip -- the pointing matrix P (gather), P^T (scatter as a fixed-order
// per-pixel reduction) and the fused P^T diag(w) P, for gfx950.
//
// Reference loops replaced: SparseLO.mult/mult_qu/mult_iqu and
// rmult/rmult_qu/rmult_iqu, interfaces/linearoperators.py:356-526.
//
// Data layout in HBM
//   time order   pix[nt] i32, cos[nt], sin[nt] f64           (caller's buffers, borrowed)
//   pixel-major  sliced ELL, slice = 64 pixels = one wavefront, lane = pixel:
//                  element (slice s, row j, lane l) at slice_ptr[s] + 64*j + l
//                row j of a lane is the j-th sample (time order) of that lane's pixel.
//                Pixels are ordered by descending hit count, so the lanes of a slice
//                have (nearly) equal lengths and padding is small.
//                sell_t (u32 sample id), sell_cos, sell_sin, sell_w (f64).
//   Every load of the pixel-major arrays is a full 256/512-byte wave access.
//   Each lane owns its pixel's three accumulators in registers: no atomics, no LDS,
//   and the additions happen in the reference's order (time order per pixel).
//
// Roofline: all kernels here are HBM-bound.  Algorithmic bytes per sample (pol=3):
//   P        pix 4 + cos 8 + sin 8 + out 8          = 28
//   P^T      t 4 + cos 8 + sin 8 + v 8 (gathered)   = 28
//   P^T W P  cos 8 + sin 8 + w 8                    = 24 streamed (28 in SURVEY 8d terms)
//   plus 48 B per pixel for reading and writing the map.
#include "cm2_pixindex.h"

#include <hipcub/hipcub.hpp>
#include <vector>

using namespace cm2;

struct cm2_pointing {
    int64_t nt = 0, npix = 0;
    int pol = 0;
    const int32_t *d_pix = nullptr;   // borrowed
    const double *d_cos = nullptr;    // borrowed
    const double *d_sin = nullptr;    // borrowed
    int64_t nvalid = 0, nslots = 0, nslices = 0, sell_len = 0;
    int32_t *d_sell_pix = nullptr;    // [nslots] pixel of the slot, -1 for padding slots
    int32_t *d_sell_cnt = nullptr;    // [nslots] hits of that pixel
    int64_t *d_slice_ptr = nullptr;   // [nslices+1]
    uint32_t *d_sell_t = nullptr;     // [sell_len]
    double *d_sell_cos = nullptr, *d_sell_sin = nullptr, *d_sell_w = nullptr;
    bool has_w = false;
    bool sell_built = false;          // the pixel-major plan is built by its first user
};

// ---------------------------------------------------------------- plan build ---
__global__ __launch_bounds__(256) void k_counts(const int64_t *__restrict__ ptr, int64_t npix,
                                                 int32_t *__restrict__ cnt,
                                                 int32_t *__restrict__ ids)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        cnt[p] = (int32_t)(ptr[p + 1] - ptr[p]);
        ids[p] = (int32_t)p;
    }
}

__global__ __launch_bounds__(256) void k_pad_slots(int64_t npix, int64_t nslots,
                                                    int32_t *__restrict__ sell_pix,
                                                    int32_t *__restrict__ sell_cnt)
{
    const int64_t i = npix + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        sell_pix[i] = -1;
        sell_cnt[i] = 0;
    }
}

// slice length in elements = 64 * (hits of the slice's first = fullest lane)
__global__ __launch_bounds__(256) void k_slice_len(const int32_t *__restrict__ sell_cnt,
                                                    int64_t nslices, int64_t *__restrict__ len)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslices) len[s] = (int64_t)sell_cnt[s * 64] * 64;
    if (s == nslices) len[s] = 0;
}

// one thread per slot writes its lane of the slice (coalesced across the wave)
template <int POL>
__global__ __launch_bounds__(256) void k_fill_sell(
    int64_t nslots, const int32_t *__restrict__ sell_pix, const int32_t *__restrict__ sell_cnt,
    const int64_t *__restrict__ slice_ptr, const int64_t *__restrict__ ptr,
    const uint32_t *__restrict__ sorted_t, const double *__restrict__ c,
    const double *__restrict__ s, uint32_t *__restrict__ sell_t, double *__restrict__ sell_cos,
    double *__restrict__ sell_sin)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const int64_t sl = slot >> 6;
    const int64_t base = slice_ptr[sl] + (slot & 63);
    const int width = (int)((slice_ptr[sl + 1] - slice_ptr[sl]) >> 6);
    const int32_t p = sell_pix[slot];
    const int cnt = sell_cnt[slot];
    const int64_t first = (p >= 0) ? ptr[p] : 0;
    for (int j = 0; j < width; ++j) {
        const int64_t idx = base + (int64_t)j * 64;
        if (j < cnt) {
            const uint32_t t = sorted_t[first + j];
            sell_t[idx] = t;
            if (POL > 1) {
                sell_cos[idx] = c[t];
                sell_sin[idx] = s[t];
            }
        } else {
            sell_t[idx] = kInvalidSample;
            if (POL > 1) {
                sell_cos[idx] = 0.0;
                sell_sin[idx] = 0.0;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gather_w(int64_t n, const uint32_t *__restrict__ sell_t,
                                                   const double *__restrict__ w,
                                                   double *__restrict__ sell_w)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t t = sell_t[i];
        sell_w[i] = (t == kInvalidSample) ? 0.0 : (w ? w[t] : 1.0);
    }
}

// -------------------------------------------------------------------- P x ------
// linearoperators.py:371-374 / :426-429 / :485-488.  x(i) starts at 0 and gets
// "+= value", so the stored result is 0.0 + value.
template <int POL>
__global__ __launch_bounds__(256) void k_P_time(int64_t nt, const int32_t *__restrict__ pix,
                                                 const double *__restrict__ c,
                                                 const double *__restrict__ s,
                                                 const double *__restrict__ x,
                                                 double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const int32_t p = pix[i];
        double r = 0.0;
        if (p >= 0) {
            if (POL == 1) {
                r += x[p];
            } else if (POL == 2) {
                const double *xp = x + 2 * (int64_t)p;
                r += xp[0] * c[i] + xp[1] * s[i];
            } else {
                const double *xp = x + 3 * (int64_t)p;
                r += xp[0] + xp[1] * c[i] + xp[2] * s[i];
            }
        }
        out[i] = r;
    }
}

// ------------------------------------------------------------------ P^T v ------
// linearoperators.py:396-400 / :449-453 / :511-516, one lane per pixel, samples of
// the pixel visited in time order => identical rounding to the serial loop.
template <int POL>
__global__ __launch_bounds__(256) void k_Pt_sell(
    int64_t nslots, const int32_t *__restrict__ sell_pix, const int32_t *__restrict__ sell_cnt,
    const int64_t *__restrict__ slice_ptr, const uint32_t *__restrict__ sell_t,
    const double *__restrict__ sell_cos, const double *__restrict__ sell_sin,
    const double *__restrict__ v, double *__restrict__ out)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const int32_t p = sell_pix[slot];
    const int cnt = sell_cnt[slot];
    const int64_t base = slice_ptr[slot >> 6] + (slot & 63);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int j = 0;
    for (; j + 4 <= cnt; j += 4) {          // 4 rows in flight per lane
        uint32_t t[4];
        double vv[4], cc[4], ss[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t idx = base + (int64_t)(j + u) * 64;
            t[u] = sell_t[idx];
            if (POL > 1) {
                cc[u] = sell_cos[idx];
                ss[u] = sell_sin[idx];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = v[t[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (POL == 1) {
                a0 += vv[u];
            } else if (POL == 2) {
                a0 += vv[u] * cc[u];
                a1 += vv[u] * ss[u];
            } else {
                a0 += vv[u];
                a1 += vv[u] * cc[u];
                a2 += vv[u] * ss[u];
            }
        }
    }
    for (; j < cnt; ++j) {
        const int64_t idx = base + (int64_t)j * 64;
        const double val = v[sell_t[idx]];
        if (POL == 1) {
            a0 += val;
        } else if (POL == 2) {
            a0 += val * sell_cos[idx];
            a1 += val * sell_sin[idx];
        } else {
            a0 += val;
            a1 += val * sell_cos[idx];
            a2 += val * sell_sin[idx];
        }
    }
    if (p >= 0) {
        double *o = out + (int64_t)POL * p;
        o[0] = a0;
        if (POL >= 2) o[1] = a1;
        if (POL == 3) o[2] = a2;
    }
}

// -------------------------------------------------------- fused P^T diag(w) P ---
// d = P x for the sample, v = w*d, accumulate v*(1,c,s): the three reference stages
// (linearoperators.py:485-488, linop DiagonalOperator, :511-516) on one sample
// without ever writing the TOD.  8 rows per lane are kept in flight.
template <int POL>
__global__ __launch_bounds__(256) void k_PtNP_sell(
    int64_t nslots, const int32_t *__restrict__ sell_pix, const int32_t *__restrict__ sell_cnt,
    const int64_t *__restrict__ slice_ptr, const double *__restrict__ sell_cos,
    const double *__restrict__ sell_sin, const double *__restrict__ sell_w,
    const double *__restrict__ x, double *__restrict__ out)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const int32_t p = sell_pix[slot];
    const int cnt = sell_cnt[slot];
    const int64_t base = slice_ptr[slot >> 6] + (slot & 63);
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (p >= 0) {
        const double *xp = x + (int64_t)POL * p;
        x0 = xp[0];
        if (POL >= 2) x1 = xp[1];
        if (POL == 3) x2 = xp[2];
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    constexpr int U = 8;
    int j = 0;
    for (; j + U <= cnt; j += U) {
        double ww[U], cc[U], ss[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t idx = base + (int64_t)(j + u) * 64;
            ww[u] = sell_w[idx];
            if (POL > 1) {
                cc[u] = sell_cos[idx];
                ss[u] = sell_sin[idx];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (POL == 1) {
                const double d = 0.0 + x0;
                a0 += ww[u] * d;
            } else if (POL == 2) {
                const double d = 0.0 + (x0 * cc[u] + x1 * ss[u]);
                const double vv = ww[u] * d;
                a0 += vv * cc[u];
                a1 += vv * ss[u];
            } else {
                const double d = 0.0 + (x0 + x1 * cc[u] + x2 * ss[u]);
                const double vv = ww[u] * d;
                a0 += vv;
                a1 += vv * cc[u];
                a2 += vv * ss[u];
            }
        }
    }
    for (; j < cnt; ++j) {
        const int64_t idx = base + (int64_t)j * 64;
        const double w = sell_w[idx];
        if (POL == 1) {
            const double d = 0.0 + x0;
            a0 += w * d;
        } else if (POL == 2) {
            const double c = sell_cos[idx], s = sell_sin[idx];
            const double d = 0.0 + (x0 * c + x1 * s);
            const double vv = w * d;
            a0 += vv * c;
            a1 += vv * s;
        } else {
            const double c = sell_cos[idx], s = sell_sin[idx];
            const double d = 0.0 + (x0 + x1 * c + x2 * s);
            const double vv = w * d;
            a0 += vv;
            a1 += vv * c;
            a2 += vv * s;
        }
    }
    if (p >= 0) {
        double *o = out + (int64_t)POL * p;
        o[0] = a0;
        if (POL >= 2) o[1] = a1;
        if (POL == 3) o[2] = a2;
    }
}

// ------------------------------------------------------------------ C ABI ------
static void free_plan(cm2_pointing *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_sell_pix, p->d_sell_cnt, p->d_slice_ptr, p->d_sell_t,
                    p->d_sell_cos, p->d_sell_sin, p->d_sell_w};
    for (void *q : ptrs)
        if (q) (void)cm2::dev_free(q);
    delete p;
}

// one pass over the pixel stream: samples outside [-1, npix) and unflagged samples
__global__ __launch_bounds__(256) void k_check_pix(const int32_t *__restrict__ pix, int64_t nt,
                                                    int64_t npix, unsigned long long *__restrict__ cnt)
{
    unsigned long long bad = 0, valid = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const int32_t p = pix[i];
        bad += (p < -1 || p >= npix) ? 1 : 0;
        valid += (p >= 0 && p < npix) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        bad += __shfl_down(bad, o);
        valid += __shfl_down(valid, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd(cnt, bad);
        if (valid) atomicAdd(cnt + 1, valid);
    }
}

extern "C" int cm2_pointing_create(cm2_pointing **out, const int32_t *d_pix,
                                   const double *d_cos, const double *d_sin, int64_t nt,
                                   int64_t npix, int pol, void *stream_)
{
    CM2_CHECK(out != nullptr, "cm2_pointing_create: out is NULL");
    *out = nullptr;
    // same message class as the RuntimeError at interfaces/linearoperators.py:549-550
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3,
              "No valid polarization key set! pol=%d (possible values 1 (I), 2 (QU), 3 (IQU))", pol);
    CM2_CHECK(d_pix != nullptr || nt == 0, "cm2_pointing_create: d_pix is NULL");
    CM2_CHECK(pol == 1 || (d_cos && d_sin), "cm2_pointing_create: cos/sin required for pol=%d", pol);
    CM2_CHECK(nt >= 0 && nt < (int64_t)0xFFFFFFFF, "nt=%lld out of range (must fit uint32)",
              (long long)nt);
    CM2_CHECK(npix > 0 && npix < (int64_t)0x7FFFFFFF, "npix=%lld out of range", (long long)npix);
    hipStream_t stream = as_stream(stream_);
    // The time-order P needs nothing but the three streams; the pixel-major (sliced-ELL) copy
    // behind the exact P^T and the fused P^T diag(w) P costs a sort and 12-20 B per sample and is
    // built by the first call that needs it (an operator that only ever runs on the tile order
    // never pays for it).  The pixel ids are checked here, so that a bad stream fails at
    // construction like the reference's index error would.
    unsigned long long h_cnt[2] = {0, 0};
    if (nt > 0) {
        DevTemp<unsigned long long> d_cnt;
        CM2_HIP(d_cnt.alloc(2));
        CM2_HIP(hipMemsetAsync(d_cnt.p, 0, sizeof(h_cnt), stream));
        k_check_pix<<<grid_for(nt), kBlock, 0, stream>>>(d_pix, nt, npix, d_cnt.p);
        CM2_LAUNCH_OK();
        CM2_HIP(cm2::read_back(h_cnt, d_cnt.p, sizeof(h_cnt), stream));
    }
    CM2_CHECK(h_cnt[0] == 0, "%llu samples have a pixel id outside [-1, npix=%lld)", h_cnt[0],
              (long long)npix);
    cm2_pointing *p = new cm2_pointing();
    p->nt = nt; p->npix = npix; p->pol = pol;
    p->d_pix = d_pix; p->d_cos = d_cos; p->d_sin = d_sin;
    p->nvalid = (int64_t)h_cnt[1];
    p->nslots = ((npix + 63) / 64) * 64;
    p->nslices = p->nslots / 64;
    *out = p;
    return 0;
}

static int ensure_sell(const cm2_pointing *cp, hipStream_t stream)
{
    if (cp->sell_built) return 0;
    cm2_pointing *p = const_cast<cm2_pointing *>(cp);
    const int64_t nt = p->nt, npix = p->npix;
    const int pol = p->pol;
    const double *d_cos = p->d_cos, *d_sin = p->d_sin;
    PixIndex ix;
    if (int rc = build_pixindex(ix, p->d_pix, nt, npix, stream)) {
        ix.release();
        return rc;
    }
    struct PlanGuard {             // frees the half-built arrays and the pixel index on early return
        cm2_pointing *plan;
        PixIndex *ix;
        ~PlanGuard()
        {
            if (plan) {
                void *ptrs[] = {plan->d_sell_pix, plan->d_sell_cnt, plan->d_slice_ptr, plan->d_sell_t,
                                plan->d_sell_cos, plan->d_sell_sin};
                for (void *q : ptrs)
                    if (q) (void)cm2::dev_free(q);
                plan->d_sell_pix = plan->d_sell_cnt = nullptr;
                plan->d_slice_ptr = nullptr;
                plan->d_sell_t = nullptr;
                plan->d_sell_cos = plan->d_sell_sin = nullptr;
            }
            ix->release();
        }
    } guard{p, &ix};
    DevTemp<int32_t> cnt_in, ids_in;
    DevTemp<int64_t> d_len;
    DevTemp<char> d_temp;
    size_t tb1 = 0, tb2 = 0;
    CM2_HIP(cnt_in.alloc(npix));
    CM2_HIP(ids_in.alloc(npix));
    CM2_HIP(cm2::dev_malloc(&p->d_sell_pix, sizeof(int32_t) * p->nslots));
    CM2_HIP(cm2::dev_malloc(&p->d_sell_cnt, sizeof(int32_t) * p->nslots));
    CM2_HIP(cm2::dev_malloc(&p->d_slice_ptr, sizeof(int64_t) * (p->nslices + 1)));
    CM2_HIP(d_len.alloc(p->nslices + 1));
    k_counts<<<grid_for(npix), kBlock, 0, stream>>>(ix.d_ptr, npix, cnt_in, ids_in);
    CM2_LAUNCH_OK();
    // pixels by descending hit count (stable): equal-length lanes share a slice
    CM2_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb1, cnt_in.p, p->d_sell_cnt,
                                                         ids_in.p, p->d_sell_pix, npix, 0, 32,
                                                         stream));
    CM2_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, d_len.p, p->d_slice_ptr,
                                             p->nslices + 1, stream));
    CM2_HIP(d_temp.alloc((tb1 > tb2 ? tb1 : tb2) + 16));
    CM2_HIP(hipcub::DeviceRadixSort::SortPairsDescending(d_temp.p, tb1, cnt_in.p, p->d_sell_cnt,
                                                         ids_in.p, p->d_sell_pix, npix, 0, 32,
                                                         stream));
    if (p->nslots > npix) {
        k_pad_slots<<<1, kBlock, 0, stream>>>(npix, p->nslots, p->d_sell_pix, p->d_sell_cnt);
        CM2_LAUNCH_OK();
    }
    k_slice_len<<<(int)((p->nslices + 1 + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
        p->d_sell_cnt, p->nslices, d_len);
    CM2_LAUNCH_OK();
    CM2_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp.p, tb2, d_len.p, p->d_slice_ptr,
                                             p->nslices + 1, stream));
    CM2_HIP(cm2::download(&p->sell_len, p->d_slice_ptr + p->nslices, sizeof(int64_t), stream));
    CM2_HIP(hipStreamSynchronize(stream));

    const int64_t L = p->sell_len > 0 ? p->sell_len : 1;
    CM2_HIP(cm2::dev_malloc(&p->d_sell_t, sizeof(uint32_t) * L));
    if (pol > 1) {
        CM2_HIP(cm2::dev_malloc(&p->d_sell_cos, sizeof(double) * L));
        CM2_HIP(cm2::dev_malloc(&p->d_sell_sin, sizeof(double) * L));
    }
    const int g = (int)((p->nslots + kBlock - 1) / kBlock);
#define CM2_FILL(POL)                                                                       \
    k_fill_sell<POL><<<g, kBlock, 0, stream>>>(p->nslots, p->d_sell_pix, p->d_sell_cnt,     \
                                               p->d_slice_ptr, ix.d_ptr, ix.d_sorted_t,     \
                                               d_cos, d_sin, p->d_sell_t, p->d_sell_cos,    \
                                               p->d_sell_sin)
    if (pol == 1) CM2_FILL(1); else if (pol == 2) CM2_FILL(2); else CM2_FILL(3);
#undef CM2_FILL
    CM2_LAUNCH_OK();
    CM2_HIP(hipStreamSynchronize(stream));
    guard.plan = nullptr;          // success (the index is still released)
    p->sell_built = true;
    return 0;
}

extern "C" int cm2_pointing_destroy(cm2_pointing *p)
{
    free_plan(p);
    return 0;
}

extern "C" int cm2_pointing_build_sell(cm2_pointing *p, void *stream_)
{
    CM2_CHECK(p, "cm2_pointing_build_sell: NULL plan");
    return ensure_sell(p, as_stream(stream_));
}

extern "C" int cm2_pointing_info(const cm2_pointing *p, int64_t *h_info)
{
    CM2_CHECK(p && h_info, "cm2_pointing_info: NULL argument");
    h_info[0] = p->nt; h_info[1] = p->npix; h_info[2] = p->pol;
    if (p->sell_built) {
        h_info[3] = p->nvalid; h_info[4] = p->sell_len; h_info[5] = p->nslices;
    } else {
        h_info[3] = h_info[4] = h_info[5] = -1;           // pixel-major copy not built (builds nothing)
    }
    return 0;
}

extern "C" int cm2_P_apply(const cm2_pointing *p, const double *d_x, double *d_out, void *stream_)
{
    CM2_CHECK(p && d_x && (d_out || p->nt == 0), "cm2_P_apply: NULL argument");
    if (p->nt == 0) return 0;
    hipStream_t stream = as_stream(stream_);
    const int g = grid_for(p->nt);
    if (p->pol == 1)
        k_P_time<1><<<g, kBlock, 0, stream>>>(p->nt, p->d_pix, p->d_cos, p->d_sin, d_x, d_out);
    else if (p->pol == 2)
        k_P_time<2><<<g, kBlock, 0, stream>>>(p->nt, p->d_pix, p->d_cos, p->d_sin, d_x, d_out);
    else
        k_P_time<3><<<g, kBlock, 0, stream>>>(p->nt, p->d_pix, p->d_cos, p->d_sin, d_x, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_Pt_apply(const cm2_pointing *p, const double *d_v, double *d_out, void *stream_)
{
    CM2_CHECK(p && d_out && (d_v || p->nt == 0), "cm2_Pt_apply: NULL argument");
    hipStream_t stream = as_stream(stream_);
    if (int rc = ensure_sell(p, stream)) return rc;
    const int g = (int)((p->nslots + kBlock - 1) / kBlock);
#define CM2_PT(POL)                                                                          \
    k_Pt_sell<POL><<<g, kBlock, 0, stream>>>(p->nslots, p->d_sell_pix, p->d_sell_cnt,        \
                                             p->d_slice_ptr, p->d_sell_t, p->d_sell_cos,     \
                                             p->d_sell_sin, d_v, d_out)
    if (p->pol == 1) CM2_PT(1); else if (p->pol == 2) CM2_PT(2); else CM2_PT(3);
#undef CM2_PT
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_pointing_set_weights(cm2_pointing *p, const double *d_w, void *stream_)
{
    CM2_CHECK(p != nullptr, "cm2_pointing_set_weights: NULL plan");
    hipStream_t stream = as_stream(stream_);
    if (int rc = ensure_sell(p, stream)) return rc;
    const int64_t L = p->sell_len > 0 ? p->sell_len : 1;
    if (!p->d_sell_w) CM2_HIP(cm2::dev_malloc(&p->d_sell_w, sizeof(double) * L));
    if (p->sell_len > 0) {
        k_gather_w<<<grid_for(p->sell_len), kBlock, 0, stream>>>(p->sell_len, p->d_sell_t, d_w,
                                                                 p->d_sell_w);
        CM2_LAUNCH_OK();
    }
    p->has_w = true;
    return 0;
}

extern "C" int cm2_PtNP_diag_apply(const cm2_pointing *p, const double *d_x, double *d_out,
                                   void *stream_)
{
    CM2_CHECK(p && d_x && d_out, "cm2_PtNP_diag_apply: NULL argument");
    CM2_CHECK(p->has_w, "cm2_PtNP_diag_apply: call cm2_pointing_set_weights first");
    hipStream_t stream = as_stream(stream_);
    const int g = (int)((p->nslots + kBlock - 1) / kBlock);
#define CM2_FUSED(POL)                                                                        \
    k_PtNP_sell<POL><<<g, kBlock, 0, stream>>>(p->nslots, p->d_sell_pix, p->d_sell_cnt,       \
                                               p->d_slice_ptr, p->d_sell_cos, p->d_sell_sin,  \
                                               p->d_sell_w, d_x, d_out)
    if (p->pol == 1) CM2_FUSED(1); else if (p->pol == 2) CM2_FUSED(2); else CM2_FUSED(3);
#undef CM2_FUSED
    CM2_LAUNCH_OK();
    return 0;
}
