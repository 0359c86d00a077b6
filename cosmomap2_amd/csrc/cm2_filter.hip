// cm2_filter.hip -- sub-scan filtering (FilterLO) and the ground-template subtraction
// (GroundFilterLO) of interfaces/linearoperators.py:24-322, see include/cosmomap2.h (f1, f2).
//
// One wavefront owns one chunk (CES x detector pair x sub-scan): a masked reduction over the
// chunk, a 64-lane butterfly so that every lane holds the same sums, then the write pass (the
// second read of d comes from L2).  The wave also zero-fills the gap in front of its chunk, so
// the output needs no separate memset: algorithmic traffic is 8 (d) + 4 (pix) + 8 (out) bytes
// per sample.  The Legendre tables are shared by all chunks of one length and stay in L2.
#include "cm2_common.h"

#include <hipcub/hipcub.hpp>

#include <vector>

namespace cm2 {

// butterfly sum: every lane ends with the same value, fixed order
__device__ __forceinline__ double wave_allsum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void zero_range(double *__restrict__ out, int64_t a, int64_t b, int lane)
{
    for (int64_t j = a + lane; j < b; j += kWave) out[j] = 0.0;
}

// A wave's view of its chunk.  RegChunk holds up to R samples per lane in registers (all loads
// in flight at once, the chunk is read from HBM once); MemChunk streams a longer chunk twice.
// Both visit a lane's samples in the same order, so the sums do not depend on the choice.
// MEAN selects the flag test: pix == -1 (inline C at :148) or pix < 0 (mask at :256).
template <int R, bool MEAN>
struct RegChunk {
    double r[R];
    uint32_t ok;
    int64_t n;
    int lane;
    __device__ __forceinline__ void load(const double *__restrict__ d, const int32_t *__restrict__ pix,
                                         int64_t a, int64_t n_, int lane_)
    {
        n = n_;
        lane = lane_;
        ok = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t j = lane + (int64_t)u * kWave;
            const bool in = j < n;
            r[u] = in ? d[a + j] : 0.0;
            const int32_t px = in ? pix[a + j] : -1;
            if (MEAN ? (px != -1) : (px >= 0)) ok |= 1u << u;
        }
    }
    template <class F>
    __device__ __forceinline__ void each(F f) const
    {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t j = lane + (int64_t)u * kWave;
            if (j < n) f(j, r[u], ((ok >> u) & 1u) != 0);
        }
    }
};

template <bool MEAN>
struct MemChunk {
    const double *__restrict__ d;
    const int32_t *__restrict__ pix;
    int64_t a, n;
    int lane;
    __device__ __forceinline__ void load(const double *__restrict__ d_, const int32_t *__restrict__ pix_,
                                         int64_t a_, int64_t n_, int lane_)
    {
        d = d_;
        pix = pix_;
        a = a_;
        n = n_;
        lane = lane_;
    }
    template <class F>
    __device__ __forceinline__ void each(F f) const
    {
        for (int64_t j = lane; j < n; j += kWave) {
            const int32_t px = pix[a + j];
            f(j, d[a + j], MEAN ? (px != -1) : (px >= 0));
        }
    }
};

constexpr int kRegSmall = 8, kRegLarge = 32;      // chunks up to 512 / 2048 samples stay in registers

template <class Chunk>
__device__ __forceinline__ void mean_body(const Chunk &ch, int64_t a, int64_t b, int lane,
                                          double *__restrict__ out)
{
    double sum = 0.0, cnt = 0.0;
    ch.each([&](int64_t, double dv, bool ok) {
        if (ok) {
            sum += dv;
            cnt += 1.0;
        }
    });
    sum = wave_allsum(sum);
    cnt = wave_allsum(cnt);
    const double mean = sum / cnt;
    if (isnan(mean) || isinf(mean)) {                 // :163-164
        zero_range(out, a, b, lane);
        return;
    }
    ch.each([&](int64_t j, double dv, bool) { out[a + j] = dv - mean; });   // :165
}

__global__ __launch_bounds__(256) void k_filter_mean(int64_t nseg, const int64_t *__restrict__ start,
                                                      const int64_t *__restrict__ len,
                                                      const int64_t *__restrict__ prev_end, int64_t nt,
                                                      const int32_t *__restrict__ pix,
                                                      const double *__restrict__ d,
                                                      double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const int64_t a = start[s], n = len[s], b = a + n;
    zero_range(out, prev_end[s], a, lane);
    if (s == nseg - 1) zero_range(out, b, nt, lane);
    if (n <= kWave * kRegSmall) {
        RegChunk<kRegSmall, true> ch;
        ch.load(d, pix, a, n, lane);
        mean_body(ch, a, b, lane, out);
    } else if (n <= kWave * kRegLarge) {
        RegChunk<kRegLarge, true> ch;
        ch.load(d, pix, a, n, lane);
        mean_body(ch, a, b, lane, out);
    } else {
        MemChunk<true> ch;
        ch.load(d, pix, a, n, lane);
        mean_body(ch, a, b, lane, out);
    }
}

// Discrete orthogonal polynomials on the unflagged samples of a chunk (Stieltjes three-term
// recurrence).  They span the same space as the Legendre columns restricted to those samples,
// so sum_k q_k q_k^T is the projector Q Q^T the reference gets from qr(legendres[unflagged])
// (:307-315), without forming a Gram matrix: the basis is orthonormal to rounding whatever the
// conditioning of the restricted Legendre block.  Per chunk: xc, xs (sample index -> [-1,1] on
// the support of the unflagged samples), alpha[K], beta[K], 1/||p_k||[K].
template <int K>
struct OrthoCoef {
    double xc, xs, alpha[K], beta[K], inorm[K];
};

template <int K, int UPTO>
__device__ __forceinline__ void ortho_eval(double x, const double *alpha, const double *beta,
                                           double (&p)[K])
{
    p[0] = 1.0;
    if (UPTO >= 1) p[1] = x - alpha[0];
#pragma unroll
    for (int k = 1; k < UPTO; ++k) p[k + 1] = (x - alpha[k]) * p[k] - beta[k] * p[k - 1];
}

template <int K, int LEVEL>
struct StieltjesStep {
    // computes ||p_LEVEL||^2 and alpha_LEVEL from the coefficients of the lower levels
    static __device__ __forceinline__ void run(int64_t a, int64_t n, int lane, double xc, double xs,
                                               const int32_t *__restrict__ pix, double *alpha,
                                               double *beta, double *nrm)
    {
        StieltjesStep<K, LEVEL - 1>::run(a, n, lane, xc, xs, pix, alpha, beta, nrm);
        double s0 = 0.0, s1 = 0.0;
        for (int64_t j = lane; j < n; j += kWave) {
            if (pix[a + j] >= 0) {
                const double x = ((double)j - xc) * xs;
                double p[K];
                ortho_eval<K, LEVEL>(x, alpha, beta, p);
                const double pp = p[LEVEL] * p[LEVEL];
                s0 += pp;
                s1 += x * pp;
            }
        }
        s0 = wave_allsum(s0);
        s1 = wave_allsum(s1);
        nrm[LEVEL] = s0;
        alpha[LEVEL] = s1 / s0;
        beta[LEVEL] = (LEVEL > 0) ? s0 / nrm[LEVEL > 0 ? LEVEL - 1 : 0] : 0.0;
    }
};
template <int K>
struct StieltjesStep<K, -1> {
    static __device__ __forceinline__ void run(int64_t, int64_t, int, double, double,
                                               const int32_t *, double *, double *, double *) {}
};

// set-up: classify every chunk (0 = too few unflagged samples, 1 = no flag, 2 = some flags)
// and build the recurrence coefficients of the kind-2 chunks
template <int K>
__global__ __launch_bounds__(256) void k_filter_setup(int64_t nseg, const int64_t *__restrict__ start,
                                                       const int64_t *__restrict__ len,
                                                       const int32_t *__restrict__ pix,
                                                       uint8_t *__restrict__ kind,
                                                       OrthoCoef<K> *__restrict__ coef)
{
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const int64_t a = start[s], n = len[s];
    int64_t jmin = n, jmax = -1, cnt = 0;
    for (int64_t j = lane; j < n; j += kWave) {
        if (pix[a + j] >= 0) {
            jmin = (j < jmin) ? j : jmin;
            jmax = j;
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t omin = __shfl_xor(jmin, off, 64), omax = __shfl_xor(jmax, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
        jmin = (omin < jmin) ? omin : jmin;
        jmax = (omax > jmax) ? omax : jmax;
    }
    const uint8_t kd = (cnt <= K - 1) ? 0 : (cnt == n ? 1 : 2);      // :303, :306
    if (lane == 0) kind[s] = kd;
    if (kd != 2) return;
    const double xc = 0.5 * (double)(jmin + jmax), xs = 2.0 / (double)(jmax - jmin);
    double alpha[K], beta[K], nrm[K];
    StieltjesStep<K, K - 1>::run(a, n, lane, xc, xs, pix, alpha, beta, nrm);
    if (lane == 0) {
        OrthoCoef<K> c;
        c.xc = xc;
        c.xs = xs;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c.alpha[k] = alpha[k];
            c.beta[k] = beta[k];
            c.inorm[k] = 1.0 / sqrt(nrm[k]);
        }
        coef[s] = c;
    }
}

template <int K, class Chunk>
__device__ __forceinline__ void poly_body(const Chunk &ch, int kd, int64_t a,
                                          const double *__restrict__ T, const OrthoCoef<K> &cf,
                                          double *__restrict__ out)
{
    double c[K];
#pragma unroll
    for (int k = 0; k < K; ++k) c[k] = 0.0;
    if (kd == 1) {
        // no flag in the chunk: the normalised Legendre columns as they are (:317-321)
        ch.each([&](int64_t j, double dv, bool) {
#pragma unroll
            for (int k = 0; k < K; ++k) c[k] += T[j * K + k] * dv;
        });
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = wave_allsum(c[k]);
        ch.each([&](int64_t j, double dv, bool) {
            double p = 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k) p += c[k] * T[j * K + k];
            out[a + j] = dv - p;
        });
        return;
    }
    // some flags: orthonormal polynomials of the unflagged samples (:307-315)
    ch.each([&](int64_t j, double dv, bool ok) {
        if (ok) {
            double p[K];
            ortho_eval<K, K - 1>(((double)j - cf.xc) * cf.xs, cf.alpha, cf.beta, p);
#pragma unroll
            for (int k = 0; k < K; ++k) c[k] += (p[k] * cf.inorm[k]) * dv;
        }
    });
#pragma unroll
    for (int k = 0; k < K; ++k) c[k] = wave_allsum(c[k]);
    ch.each([&](int64_t j, double dv, bool ok) {
        double o = 0.0;
        if (ok) {
            double p[K];
            ortho_eval<K, K - 1>(((double)j - cf.xc) * cf.xs, cf.alpha, cf.beta, p);
            double proj = 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k) proj += c[k] * (p[k] * cf.inorm[k]);
            o = dv - proj;
        }
        out[a + j] = o;
    });
}

template <int K>
__global__ __launch_bounds__(256) void k_filter_poly(int64_t nseg, const int64_t *__restrict__ start,
                                                      const int64_t *__restrict__ len,
                                                      const int64_t *__restrict__ prev_end, int64_t nt,
                                                      const uint8_t *__restrict__ kind,
                                                      const int64_t *__restrict__ toff,
                                                      const double *__restrict__ table,
                                                      const OrthoCoef<K> *__restrict__ coef,
                                                      const int32_t *__restrict__ pix,
                                                      const double *__restrict__ d,
                                                      double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const int64_t a = start[s], n = len[s], b = a + n;
    zero_range(out, prev_end[s], a, lane);
    if (s == nseg - 1) zero_range(out, b, nt, lane);
    const int kd = kind[s];
    if (kd == 0) {                                    // :303-304
        zero_range(out, a, b, lane);
        return;
    }
    const double *__restrict__ T = table + toff[s];
    OrthoCoef<K> cf;
    if (kd == 2) cf = coef[s];
    if (n <= kWave * kRegSmall) {
        RegChunk<kRegSmall, false> ch;
        ch.load(d, pix, a, n, lane);
        poly_body<K>(ch, kd, a, T, cf, out);
    } else if (n <= kWave * kRegLarge) {
        RegChunk<kRegLarge, false> ch;
        ch.load(d, pix, a, n, lane);
        poly_body<K>(ch, kd, a, T, cf, out);
    } else {
        MemChunk<false> ch;
        ch.load(d, pix, a, n, lane);
        poly_body<K>(ch, kd, a, T, cf, out);
    }
}

// ---- the same filters on the tile-bucketed TOD order of cm2_tiles.hip ------------------
// A window is a stretch of <= 8192 consecutive time samples holding whole chunks.  One
// workgroup gathers its window through an address-sorted list (k = position in the tile
// order, q = offset in the window; the window's samples of one pixel tile are one contiguous
// run), filters the chunks in LDS -- one wavefront per chunk, same arithmetic as above -- and
// scatters the window back through the list entries it kept in registers: 6 (list) + 8 + 8
// bytes per sample.  Flagged samples have no slot in the tile order: they are simply absent.
constexpr int kWinLen = 8192, kWinT = 256, kWinPer = kWinLen / kWinT;

struct FilterWin {
    int64_t t0;            // time of window offset 0
    int32_t span, s0, s1;  // samples in the window, chunks [s0, s1)
    int32_t pad;
};

template <bool MEAN>
struct LdsChunk {
    const double *d;
    const uint8_t *ok;
    int64_t a, n;
    int lane;
    template <class F>
    __device__ __forceinline__ void each(F f) const
    {
        for (int64_t j = lane; j < n; j += kWave) f(j, d[a + j], ok[a + j] != 0);
    }
};

template <int K>          // K = 0: mean removal, K >= 2: Legendre order K - 1
__global__ __launch_bounds__(kWinT, 2) void k_filter_windows(
    const FilterWin *__restrict__ wins, int nwin, const int64_t *__restrict__ start,
    const int64_t *__restrict__ len, const uint8_t *__restrict__ kind,
    const int64_t *__restrict__ toff, const double *__restrict__ table, const void *__restrict__ coef_,
    const uint32_t *__restrict__ lst_k, const uint16_t *__restrict__ lst_q,
    const double *__restrict__ in, double *__restrict__ out)
{
    extern __shared__ double win_lds[];
    double *data = win_lds;
    uint8_t *ok = reinterpret_cast<uint8_t *>(win_lds + kWinLen);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per_xcd = (nwin + 7) / 8;
    const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wid >= nwin) return;
    const FilterWin w = wins[wid];
    for (int i = t; i < kWinLen / 8; i += kWinT) reinterpret_cast<uint64_t *>(ok)[i] = 0;
    uint32_t kk[kWinPer];
    uint16_t qq[kWinPer];
    const int64_t base = (int64_t)wid * kWinLen;
#pragma unroll
    for (int u = 0; u < kWinPer; ++u) {
        kk[u] = lst_k[base + t + u * kWinT];
        qq[u] = lst_q[base + t + u * kWinT];
    }
    double vv[kWinPer];
#pragma unroll
    for (int u = 0; u < kWinPer; ++u) vv[u] = (kk[u] != kInvalidSample) ? in[kk[u]] : 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kWinPer; ++u)
        if (kk[u] != kInvalidSample) {
            data[qq[u]] = vv[u];
            ok[qq[u]] = 1;
        }
    __syncthreads();
    for (int s = w.s0 + wave; s < w.s1; s += kWinT / 64) {
        const int64_t a = start[s] - w.t0, n = len[s];
        const int64_t prev = (s == w.s0) ? 0 : start[s - 1] + len[s - 1] - w.t0;
        zero_range(data, prev, a, lane);                       // gap in front of the chunk
        if (s == w.s1 - 1) zero_range(data, a + n, w.span, lane);
        if constexpr (K == 0) {
            LdsChunk<true> ch{data, ok, a, n, lane};
            mean_body(ch, a, a + n, lane, data);
        } else {
            const int kd = kind[s];
            if (kd == 0) {
                zero_range(data, a, a + n, lane);
            } else {
                const OrthoCoef<K> *coef = static_cast<const OrthoCoef<K> *>(coef_);
                OrthoCoef<K> cf;
                if (kd == 2) cf = coef[s];
                LdsChunk<false> ch{data, ok, a, n, lane};
                poly_body<K>(ch, kd, a, table + toff[s], cf, data);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kWinPer; ++u)
        if (kk[u] != kInvalidSample) out[kk[u]] = data[qq[u]];
}

// keys of the window lists: (window << 32) | position in the tile order, value = offset in the
// window; slots past the window's span and flagged samples get an invalid address
__global__ __launch_bounds__(256) void k_win_keys(const FilterWin *__restrict__ wins, int64_t nwin,
                                                   const uint32_t *__restrict__ idx,
                                                   uint64_t *__restrict__ keys,
                                                   uint16_t *__restrict__ vals)
{
    const int64_t total = nwin * kWinLen;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const int64_t wi = g / kWinLen;
        const int q = (int)(g - wi * kWinLen);
        const FilterWin w = wins[wi];
        const uint32_t k = q < w.span ? idx[w.t0 + q] : kInvalidSample;
        keys[g] = ((uint64_t)wi << 32) | (uint64_t)k;
        vals[g] = (uint16_t)q;
    }
}

__global__ __launch_bounds__(256) void k_win_unpack(int64_t total, const uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ lst_k)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride)
        lst_k[g] = (uint32_t)(keys[g] & 0xFFFFFFFFull);
}

__global__ __launch_bounds__(256) void k_ground_subtract(int64_t nt, const int32_t *__restrict__ bin,
                                                          const double *__restrict__ binned,
                                                          const double *__restrict__ v,
                                                          double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const int32_t g = bin[i];
        out[i] = (g < 0) ? v[i] : v[i] - binned[g];
    }
}

// G^T v for a few thousand ground bins: every workgroup bins a contiguous stretch of the time
// stream into LDS (ds_add_f64), then adds its non-empty bins to HBM.  The pixel-major
// fixed-order P^T of cm2_pointing.hip has one lane per pixel, which leaves most of the chip
// idle when ~2000 bins hold ~50000 samples each.
constexpr int kGroundLdsBins = 8192;

__global__ __launch_bounds__(256) void k_ground_bin(int64_t nt, int nbins,
                                                     const int32_t *__restrict__ bin,
                                                     const double *__restrict__ v,
                                                     double *__restrict__ sums)
{
    extern __shared__ double acc[];
    for (int i = threadIdx.x; i < nbins; i += 256) acc[i] = 0.0;
    __syncthreads();
    const int64_t per = (nt + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = per * blockIdx.x, b1 = (b0 + per < nt) ? b0 + per : nt;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) {
        const int32_t g = bin[i];
        if ((uint32_t)g < (uint32_t)nbins) atomicAdd(&acc[g], v[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 256) {
        const double a = acc[i];
        if (a != 0.0) atomicAdd(&sums[i], a);
    }
}

}  // namespace cm2

using namespace cm2;

struct cm2_filter {
    int64_t nt = 0, nseg = 0, covered = 0;
    int order = 0;
    const int32_t *d_pix = nullptr;      // borrowed
    int64_t *d_start = nullptr, *d_len = nullptr, *d_prev_end = nullptr, *d_toff = nullptr;
    uint8_t *d_kind = nullptr;
    double *d_table = nullptr;
    void *d_coef = nullptr;              // OrthoCoef<order+1>[nseg]
    int64_t nkind[3] = {0, 0, 0};
    std::vector<int64_t> h_start, h_len; // host copies of the chunk table (window planning)
    // tile-order plan, built for one tile index at a time
    uint64_t win_plan = 0;            // id of the tile plan the window lists were built for
    bool win_ok = false, win_memset = false;
    int64_t nwin = 0;
    FilterWin *d_wins = nullptr;
    uint32_t *d_win_k = nullptr;
    uint16_t *d_win_q = nullptr;
};

extern "C" void cm2_filter_destroy(cm2_filter *f)
{
    if (!f) return;
    void *bufs[] = {f->d_start, f->d_len, f->d_prev_end, f->d_toff, f->d_kind, f->d_table, f->d_coef,
                    f->d_wins, f->d_win_k, f->d_win_q};
    for (void *p : bufs)
        if (p) (void)cm2::dev_free(p);
    delete f;
}

namespace {

template <typename T>
int upload(T **dst, const T *src, size_t count, hipStream_t st)
{
    CM2_HIP(cm2::dev_malloc(dst, sizeof(T) * (count ? count : 1)));
    if (count) CM2_HIP(cm2::upload(*dst, src, sizeof(T) * count, st));
    return 0;
}

template <int K>
int setup_poly(cm2_filter *f, hipStream_t st)
{
    CM2_HIP(cm2::dev_malloc(&f->d_coef, sizeof(OrthoCoef<K>) * (size_t)(f->nseg ? f->nseg : 1)));
    CM2_HIP(cm2::dev_malloc(&f->d_kind, (size_t)(f->nseg ? f->nseg : 1)));
    if (f->nseg == 0) return 0;
    const int64_t blocks = (f->nseg + 3) / 4;
    k_filter_setup<K><<<dim3((unsigned)blocks), 256, 0, st>>>(
        f->nseg, f->d_start, f->d_len, f->d_pix, f->d_kind, static_cast<OrthoCoef<K> *>(f->d_coef));
    CM2_LAUNCH_OK();
    return 0;
}

template <int K>
void launch_poly(const cm2_filter *f, const double *d_in, double *d_out, hipStream_t st)
{
    const int64_t blocks = (f->nseg + 3) / 4;
    k_filter_poly<K><<<dim3((unsigned)blocks), 256, 0, st>>>(
        f->nseg, f->d_start, f->d_len, f->d_prev_end, f->nt, f->d_kind, f->d_toff, f->d_table,
        static_cast<const OrthoCoef<K> *>(f->d_coef), f->d_pix, d_in, d_out);
}

int filter_fill(cm2_filter *f, int64_t nt, int64_t nseg, const int64_t *h_start,
                const int64_t *h_len, const int32_t *d_pix, int order,
                const int64_t *h_table_off, const double *h_table, int64_t table_len,
                hipStream_t st)
{
    CM2_CHECK(nt >= 0 && nseg >= 0, "cm2_filter_create: negative size");
    CM2_CHECK(order >= 0 && order <= 7, "cm2_filter_create: poly order %d outside [0,7]", order);
    CM2_CHECK(nseg == 0 || (h_start && h_len), "cm2_filter_create: null chunk arrays");
    CM2_CHECK(nt == 0 || d_pix, "cm2_filter_create: null pixel array");
    CM2_CHECK((nseg + 3) / 4 < (int64_t)0x7FFFFFFF, "cm2_filter_create: too many chunks");
    const int K = order + 1;
    if (order > 0)
        CM2_CHECK(nseg == 0 || (h_table_off && h_table),
                  "cm2_filter_create: order %d needs the Legendre tables", order);
    std::vector<int64_t> prev_end((size_t)nseg);
    int64_t end = 0, covered = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        CM2_CHECK(h_len[s] >= 0, "cm2_filter_create: chunk %lld has negative length", (long long)s);
        CM2_CHECK(h_start[s] >= end,
                  "cm2_filter_create: chunk %lld starts at %lld, inside or before the previous "
                  "chunk (ends at %lld): chunks must be ascending and disjoint",
                  (long long)s, (long long)h_start[s], (long long)end);
        CM2_CHECK(h_start[s] + h_len[s] <= nt, "cm2_filter_create: chunk %lld ends past nt=%lld",
                  (long long)s, (long long)nt);
        if (order > 0)
            CM2_CHECK(h_table_off[s] >= 0 && h_table_off[s] + h_len[s] * K <= table_len,
                      "cm2_filter_create: chunk %lld: table block outside the table", (long long)s);
        prev_end[(size_t)s] = end;
        end = h_start[s] + h_len[s];
        covered += h_len[s];
    }
    f->nt = nt;
    f->nseg = nseg;
    f->order = order;
    f->covered = covered;
    f->d_pix = d_pix;
    f->h_start.assign(h_start, h_start + nseg);
    f->h_len.assign(h_len, h_len + nseg);
    if (upload(&f->d_start, h_start, (size_t)nseg, st)) return 1;
    if (upload(&f->d_len, h_len, (size_t)nseg, st)) return 1;
    if (upload(&f->d_prev_end, prev_end.data(), (size_t)nseg, st)) return 1;
    if (order > 0) {
        if (upload(&f->d_toff, h_table_off, (size_t)nseg, st)) return 1;
        if (upload(&f->d_table, h_table, (size_t)table_len, st)) return 1;
        int rc = 0;
        switch (K) {
            case 2: rc = setup_poly<2>(f, st); break;
            case 3: rc = setup_poly<3>(f, st); break;
            case 4: rc = setup_poly<4>(f, st); break;
            case 5: rc = setup_poly<5>(f, st); break;
            case 6: rc = setup_poly<6>(f, st); break;
            case 7: rc = setup_poly<7>(f, st); break;
            default: rc = setup_poly<8>(f, st); break;
        }
        if (rc) return rc;
        std::vector<uint8_t> h_kind((size_t)nseg);
        if (nseg)
            CM2_HIP(cm2::download(h_kind.data(), f->d_kind, (size_t)nseg, st));
        CM2_HIP(hipStreamSynchronize(st));
        for (uint8_t k : h_kind) f->nkind[k < 3 ? k : 0]++;
    }
    CM2_HIP(hipStreamSynchronize(st));       // the host arrays may go away after return
    return 0;
}

}  // namespace

extern "C" int cm2_filter_create(cm2_filter **out, int64_t nt, int64_t nseg,
                                 const int64_t *h_start, const int64_t *h_len,
                                 const int32_t *d_pix, int order, const int64_t *h_table_off,
                                 const double *h_table, int64_t table_len, void *stream)
{
    CM2_CHECK(out, "cm2_filter_create: null output handle");
    *out = nullptr;
    cm2_filter *f = new cm2_filter();
    const int rc = filter_fill(f, nt, nseg, h_start, h_len, d_pix, order, h_table_off, h_table,
                               table_len, as_stream(stream));
    if (rc) {
        cm2_filter_destroy(f);
        return rc;
    }
    *out = f;
    return 0;
}

extern "C" int cm2_filter_info(const cm2_filter *f, int64_t *info)
{
    CM2_CHECK(f && info, "cm2_filter_info: null argument");
    info[0] = f->nt;
    info[1] = f->nseg;
    info[2] = f->order;
    info[3] = f->covered;
    info[4] = f->nkind[0];
    info[5] = f->nkind[1];
    info[6] = f->nkind[2];
    return 0;
}

extern "C" int cm2_filter_apply(const cm2_filter *f, const double *d_in, double *d_out, void *stream)
{
    CM2_CHECK(f, "cm2_filter_apply: null handle");
    if (f->nt == 0) return 0;
    CM2_CHECK(d_in && d_out, "cm2_filter_apply: null vector");
    CM2_CHECK(d_in != d_out, "cm2_filter_apply: output must not alias the input");
    hipStream_t st = as_stream(stream);
    if (f->nseg == 0) {
        CM2_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * f->nt, st));
        return 0;
    }
    const int64_t blocks = (f->nseg + 3) / 4;
    if (f->order == 0) {
        k_filter_mean<<<dim3((unsigned)blocks), 256, 0, st>>>(f->nseg, f->d_start, f->d_len,
                                                              f->d_prev_end, f->nt, f->d_pix, d_in,
                                                              d_out);
    } else {
        switch (f->order + 1) {
            case 2: launch_poly<2>(f, d_in, d_out, st); break;
            case 3: launch_poly<3>(f, d_in, d_out, st); break;
            case 4: launch_poly<4>(f, d_in, d_out, st); break;
            case 5: launch_poly<5>(f, d_in, d_out, st); break;
            case 6: launch_poly<6>(f, d_in, d_out, st); break;
            case 7: launch_poly<7>(f, d_in, d_out, st); break;
            default: launch_poly<8>(f, d_in, d_out, st); break;
        }
    }
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" const uint32_t *cm2_tiles_index(const cm2_tiles *t);     // cm2_tiles.hip
extern "C" int64_t cm2_tiles_nt(const cm2_tiles *t);
extern "C" int64_t cm2_tiles_nvalid(const cm2_tiles *t);
extern "C" uint64_t cm2_tiles_plan_id(const cm2_tiles *t);

namespace {

int filter_windows_build(cm2_filter *f, const uint32_t *d_idx, uint64_t plan_id, hipStream_t st)
{
    void **old[] = {(void **)&f->d_wins, (void **)&f->d_win_k, (void **)&f->d_win_q};
    for (void **q : old) {
        if (*q) (void)cm2::dev_free(*q);
        *q = nullptr;
    }
    f->win_plan = 0;                                  // set once the lists are complete
    f->win_ok = false;
    f->win_memset = false;
    f->nwin = 0;
    std::vector<FilterWin> wins;
    const int64_t nseg = f->nseg;
    const auto &S = f->h_start;
    const auto &L = f->h_len;
    if (nseg == 0 || S[0] > 0) f->win_memset = true;
    for (int64_t s0 = 0; s0 < nseg;) {
        const int64_t t0 = S[(size_t)s0];
        int64_t s1 = s0;
        while (s1 < nseg && S[(size_t)s1] + L[(size_t)s1] - t0 <= kWinLen && s1 - s0 < (1 << 20)) ++s1;
        if (s1 == s0) {                               // a chunk longer than a window: not tileable
            f->win_plan = plan_id;
            return 0;
        }
        int64_t t1 = S[(size_t)s1 - 1] + L[(size_t)s1 - 1];
        const int64_t next = s1 < nseg ? S[(size_t)s1] : f->nt;
        if (next - t0 <= kWinLen) t1 = next;          // the trailing gap rides along (zeroed in LDS)
        else f->win_memset = true;                    // samples between windows: zeroed by a memset
        FilterWin w;
        w.t0 = t0;
        w.span = (int32_t)(t1 - t0);
        w.s0 = (int32_t)s0;
        w.s1 = (int32_t)s1;
        w.pad = 0;
        wins.push_back(w);
        s0 = s1;
    }
    f->nwin = (int64_t)wins.size();
    f->win_ok = true;
    if (f->nwin == 0) {
        f->win_plan = plan_id;
        return 0;
    }
    CM2_CHECK(f->nseg < ((int64_t)1 << 31) && f->nwin < ((int64_t)1 << 31), "too many chunks");
    CM2_HIP(cm2::dev_malloc(&f->d_wins, sizeof(FilterWin) * wins.size()));
    CM2_HIP(cm2::upload(f->d_wins, wins.data(), sizeof(FilterWin) * wins.size(), st));
    const int64_t total = f->nwin * kWinLen;
    DevTemp<uint64_t> keys_in, keys_out;
    DevTemp<uint16_t> vals_in;
    DevTemp<char> d_temp;
    CM2_HIP(keys_in.alloc(total));
    CM2_HIP(keys_out.alloc(total));
    CM2_HIP(vals_in.alloc(total));
    CM2_HIP(cm2::dev_malloc(&f->d_win_k, sizeof(uint32_t) * total));
    CM2_HIP(cm2::dev_malloc(&f->d_win_q, sizeof(uint16_t) * total));
    k_win_keys<<<grid_for(total), kBlock, 0, st>>>(f->d_wins, f->nwin, d_idx, keys_in, vals_in);
    CM2_LAUNCH_OK();
    int end_bit = 33;
    while (((int64_t)1 << (end_bit - 32)) <= f->nwin && end_bit < 64) ++end_bit;
    size_t tb = 0;
    CM2_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in.p, keys_out.p, vals_in.p,
                                               f->d_win_q, total, 0, end_bit, st));
    CM2_HIP(d_temp.alloc(tb + 16));
    CM2_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp.p, tb, keys_in.p, keys_out.p, vals_in.p,
                                               f->d_win_q, total, 0, end_bit, st));
    k_win_unpack<<<grid_for(total), kBlock, 0, st>>>(total, keys_out, f->d_win_k);
    CM2_LAUNCH_OK();
    CM2_HIP(hipStreamSynchronize(st));
    f->win_plan = plan_id;
    return 0;
}

template <int K>
int launch_windows(const cm2_filter *f, const double *d_in, double *d_out, hipStream_t st)
{
    constexpr size_t lds = sizeof(double) * kWinLen + kWinLen;
    static size_t granted[64] = {0};
    CM2_HIP(ensure_dynamic_lds((const void *)k_filter_windows<K>, lds, granted));
    const int grid = (int)(((f->nwin + 7) / 8) * 8);
    k_filter_windows<K><<<grid, kWinT, lds, st>>>(f->d_wins, (int)f->nwin, f->d_start, f->d_len,
                                                  f->d_kind, f->d_toff, f->d_table, f->d_coef,
                                                  f->d_win_k, f->d_win_q, d_in, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

}  // namespace

extern "C" int cm2_filter_apply_tiles(cm2_filter *f, const cm2_tiles *tiles, const double *d_in_tb,
                                      double *d_out_tb, int *h_done, void *stream)
{
    CM2_CHECK(f && tiles && h_done, "cm2_filter_apply_tiles: null argument");
    *h_done = 0;
    CM2_CHECK(cm2_tiles_nt(tiles) == f->nt, "filter has %lld samples, tile plan %lld",
              (long long)f->nt, (long long)cm2_tiles_nt(tiles));
    hipStream_t st = as_stream(stream);
    const uint32_t *d_idx = cm2_tiles_index(tiles);
    // the lists belong to ONE tile plan: keyed on its id, not on a device address that a later
    // plan could be given again by the allocator
    const uint64_t plan_id = cm2_tiles_plan_id(tiles);
    if (f->win_plan != plan_id)
        if (int rc = filter_windows_build(f, d_idx, plan_id, st)) return rc;
    if (!f->win_ok) return 0;                         // caller falls back to the time order
    const int64_t nvalid = cm2_tiles_nvalid(tiles);
    CM2_CHECK(nvalid == 0 || (d_in_tb && d_out_tb && d_in_tb != d_out_tb),
              "cm2_filter_apply_tiles: null or aliased vectors");
    if (f->win_memset && nvalid) CM2_HIP(hipMemsetAsync(d_out_tb, 0, sizeof(double) * nvalid, st));
    *h_done = 1;
    if (f->nwin == 0) return 0;
    switch (f->order == 0 ? 0 : f->order + 1) {
        case 0: return launch_windows<0>(f, d_in_tb, d_out_tb, st);
        case 2: return launch_windows<2>(f, d_in_tb, d_out_tb, st);
        case 3: return launch_windows<3>(f, d_in_tb, d_out_tb, st);
        case 4: return launch_windows<4>(f, d_in_tb, d_out_tb, st);
        case 5: return launch_windows<5>(f, d_in_tb, d_out_tb, st);
        case 6: return launch_windows<6>(f, d_in_tb, d_out_tb, st);
        case 7: return launch_windows<7>(f, d_in_tb, d_out_tb, st);
        default: return launch_windows<8>(f, d_in_tb, d_out_tb, st);
    }
}

extern "C" int cm2_ground_subtract(int64_t nt, const int32_t *d_bin, const double *d_binned,
                                   const double *d_v, double *d_out, void *stream)
{
    CM2_CHECK(nt >= 0, "cm2_ground_subtract: negative size");
    if (nt == 0) return 0;
    CM2_CHECK(d_bin && d_binned && d_v && d_out, "cm2_ground_subtract: null array");
    k_ground_subtract<<<grid_for(nt), kBlock, 0, as_stream(stream)>>>(nt, d_bin, d_binned, d_v,
                                                                       d_out);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_ground_bin_sums(int64_t nt, int nbins, const int32_t *d_bin, const double *d_v,
                                   double *d_sums, void *stream)
{
    CM2_CHECK(nt >= 0 && nbins > 0, "cm2_ground_bin_sums: bad size");
    CM2_CHECK(nbins <= kGroundLdsBins, "cm2_ground_bin_sums: %d bins exceed the LDS histogram (%d); "
              "use cm2_Pt_apply", nbins, kGroundLdsBins);
    CM2_CHECK(d_sums && (nt == 0 || (d_bin && d_v)), "cm2_ground_bin_sums: null array");
    hipStream_t st = as_stream(stream);
    CM2_HIP(hipMemsetAsync(d_sums, 0, sizeof(double) * nbins, st));
    if (nt == 0) return 0;
    const int grid = grid_for(nt, 256 * 64, kNumCU * 4);
    k_ground_bin<<<grid, 256, sizeof(double) * nbins, st>>>(nt, nbins, d_bin, d_v, d_sums);
    CM2_LAUNCH_OK();
    return 0;
}
