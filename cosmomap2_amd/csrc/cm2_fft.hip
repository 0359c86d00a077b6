// cm2_fft.hip -- banded-Toeplitz N^-1 by overlap-save with a hand-written fp64 FFT: one kernel
// reads the TOD once (x 1.5 for the window overlap) and writes the filtered TOD once.
//
// Reference semantics: ToeplitzLO.mult, interfaces/linearoperators.py:582-595 (symmetric
// band, ZERO boundary at both ends of every block), dispatched per block as
// interfaces/blkop.py:195-206.
//
// Algorithm per workgroup (256 threads, two workgroups per CU):
//   * two overlap-save segments A, B of the same noise block (each: 4096 new samples + a halo of
//     2048 >= lambda-1 samples on both sides, zero outside the block) are packed as ONE complex
//     signal z = A + iB of length N = 8192;
//   * forward complex FFT of length N = 32*16*16 (decimation in frequency, the signal lives in
//     registers -- 32 complex points per thread -- and passes exchange it through LDS one plane
//     at a time, output in digit-reversed order);
//   * pointwise product with the band's spectrum H.  H is real and even (the band is
//     symmetric), so  ifft(H fft(A + iB)) = h*A + i h*B : no untangling is needed, and H is
//     stored pre-permuted in the digit-reversed order and with 1/N folded in;
//   * inverse FFT (the same passes backwards, conjugate twiddles);
//   * the 4096 results of A (real part) and of B (imaginary part) are written out.
//
// LDS: N + N/32 doubles (one plane, one pad double every 32) = 66 KB per workgroup.
// HBM traffic per output sample: 1.5 x 8 B read + 8 B write against the 16 B algorithmic figure
// (+ 15 B of address / position lists on the tile order); the rocFFT route (pack, 2 x [FFT +
// real pre/post kernel], spectrum multiply, unpack) moves ~7x that.
// This translation unit is compiled with FMA contraction ON (results are compared with
// the direct sum at 1e-12, not bit for bit).
//
// k_overlap_save_reg<LISTS> serves every band length the fused path supports (lambda <= 2049), on
// the time order (computed addresses) and on the tile-bucketed order of cm2_tiles.hip (three
// address-sorted lists per segment pair).  The LDS-resident kernels of round 1 (both planes in
// LDS, one workgroup per CU, transforms of 512 / 2048 / 8192 points) were slower in every case
//  and are gone.
#include "cm2_fft.h"

#include <hipcub/hipcub.hpp>

using namespace cm2;

namespace {

constexpr double kCos32[32] = {1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476, 0.5555702330196022, 0.3826834323650898, 0.19509032201612828, 0.0, -0.19509032201612828, -0.3826834323650898, -0.5555702330196022, -0.7071067811865476, -0.8314696123025452, -0.9238795325112867, -0.9807852804032304, -1.0, -0.9807852804032304, -0.9238795325112867, -0.8314696123025452, -0.7071067811865476, -0.5555702330196022, -0.3826834323650898, -0.19509032201612828, 0.0, 0.19509032201612828, 0.3826834323650898, 0.5555702330196022, 0.7071067811865476, 0.8314696123025452, 0.9238795325112867, 0.9807852804032304};
constexpr double kSin32[32] = {0.0, 0.19509032201612828, 0.3826834323650898, 0.5555702330196022, 0.7071067811865476, 0.8314696123025452, 0.9238795325112867, 0.9807852804032304, 1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476, 0.5555702330196022, 0.3826834323650898, 0.19509032201612828, 0.0, -0.19509032201612828, -0.3826834323650898, -0.5555702330196022, -0.7071067811865476, -0.8314696123025452, -0.9238795325112867, -0.9807852804032304, -1.0, -0.9807852804032304, -0.9238795325112867, -0.8314696123025452, -0.7071067811865476, -0.5555702330196022, -0.3826834323650898, -0.19509032201612828};

__host__ __device__ constexpr int ilog2(int r) { return r <= 1 ? 0 : 1 + ilog2(r >> 1); }

template <int R>
__host__ __device__ constexpr int brev(int m)
{
    int out = 0;
    for (int b = 0; b < ilog2(R); ++b) out |= ((m >> b) & 1) << (ilog2(R) - 1 - b);
    return out;
}

__device__ __forceinline__ int padi(int a) { return a + (a >> 5); }

struct PairDesc {          // one workgroup's work: two segments of one noise block
    int64_t a_start, a_len, b_start, b_len, lo, hi;
    int32_t blk, pad;
};

// ------------------------------------------------------------------------------------
// The signal lives in registers -- 256 threads x 32 complex points -- and LDS is only the
// exchange buffer between passes, ONE plane at a time (66 KB): two workgroups share a CU and
// one computes while the other waits on HBM.  (R1, R2, R3) = (32, 16, 16); Hperm is laid out
// for that factorisation.
//
// Layouts (position a of the N-point signal held by thread t in slot m):
//   P1: a = t + 256 m                      radix-32 pass over stride 256  (n = N)
//   P2: a = (16 (m>>4) + (t>>4)) 256 + (t&15) + 16 (m&15)
//                                          two radix-16 butterflies over stride 16  (n = 256)
//   P3: a = 32 t + m                       two radix-16 butterflies on contiguous points
constexpr int kRegT = 256, kRegN = 8192;

template <int L>
__device__ __forceinline__ int reg_pos(int t, int m)
{
    if (L == 1) return t + 256 * m;
    if (L == 2) return (16 * (m >> 4) + (t >> 4)) * 256 + (t & 15) + 16 * (m & 15);
    return 32 * t + m;
}

// padded LDS index of reg_pos<L>(t, m) split into a per-thread base and a compile-time offset
// (padi(a) = a + (a >> 5) is additive for these layouts), so that the 32 accesses of an
// exchange are one base register plus immediate offsets
template <int L>
__device__ __forceinline__ int reg_base(int t)
{
    if (L == 1) return t + (t >> 5);
    if (L == 2) return (t >> 4) * 264 + (t & 15);
    return 33 * t;
}
template <int L>
__host__ __device__ constexpr int reg_off(int m)
{
    return L == 1 ? 264 * m : (L == 2 ? 4224 * (m >> 4) + 16 * (m & 15) + ((m & 15) >> 1) : m);
}

// Where slot m of a 32-array sits after in-place butterflies: dft_sub leaves output m of a
// radix-R block at index brev<R>(m) of that block.  PERM = 0: natural, 32: one radix-32 block,
// 16: two radix-16 blocks.  The exchanges apply it as a compile-time index, so no value is
// ever moved between registers.
template <int PERM>
__host__ __device__ constexpr int reg_slot(int m)
{
    return PERM == 32 ? brev<32>(m) : (PERM == 16 ? 16 * (m >> 4) + brev<16>(m & 15) : m);
}

template <int FROM, int TO, int PERM>
__device__ __forceinline__ void reg_exchange(double (&a)[32], double *__restrict__ buf, int t)
{
    double *__restrict__ wp = buf + reg_base<FROM>(t);
    const double *__restrict__ rp = buf + reg_base<TO>(t);
#pragma unroll
    for (int m = 0; m < 32; ++m) wp[reg_off<FROM>(m)] = a[reg_slot<PERM>(m)];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 32; ++m) a[m] = rp[reg_off<TO>(m)];
    __syncthreads();
}

// dft_regs on the sub-block [OFF, OFF + R) of a 32-array, in place
template <int R, int OFF>
__device__ __forceinline__ void dft_sub(double (&re)[32], double (&im)[32])
{
#pragma unroll
    for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int i = 0; i < h; ++i) {
                const int a = OFF + blk + i, b = a + h;
                const int tw = i * (32 / (2 * h));
                const double ar = re[a], ai = im[a], br = re[b], bi = im[b];
                re[a] = ar + br;
                im[a] = ai + bi;
                const double dr = ar - br, di = ai - bi;
                if (tw == 0) {
                    re[b] = dr;
                    im[b] = di;
                } else if (tw == 8) {
                    re[b] = di;
                    im[b] = -dr;
                } else {
                    const double c = kCos32[tw], s = kSin32[tw];
                    re[b] = dr * c + di * s;
                    im[b] = di * c - dr * s;
                }
            }
        }
    }
}

// decimation-in-time counterpart of dft_sub: input m at index brev<R>(m), output natural.
// Used for the inverse half of the middle pass, so that the spectrum product happens in
// place on the bit-reversed output of dft_sub and nothing is permuted or copied.
template <int R, int OFF>
__device__ __forceinline__ void dit_sub(double (&re)[32], double (&im)[32])
{
#pragma unroll
    for (int h = 1; h <= R / 2; h <<= 1) {
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int i = 0; i < h; ++i) {
                const int a = OFF + blk + i, b = a + h;
                const int tw = i * (32 / (2 * h));          // exponent in units of 2 pi / 32
                double tr, ti;
                if (tw == 0) {
                    tr = re[b];
                    ti = im[b];
                } else if (tw == 8) {                        // times -i
                    tr = im[b];
                    ti = -re[b];
                } else {                                     // times (c - i s)
                    const double c = kCos32[tw], s = kSin32[tw];
                    tr = re[b] * c + im[b] * s;
                    ti = im[b] * c - re[b] * s;
                }
                const double ar = re[a], ai = im[a];
                re[a] = ar + tr;
                im[a] = ai + ti;
                re[b] = ar - tr;
                im[b] = ai - ti;
            }
        }
    }
}

// forward pass on a block: butterfly, then output m (at index brev(m)) times w1^m
template <int R, int OFF>
__device__ __forceinline__ void reg_fwd(double (&xr)[32], double (&xi)[32], double2 w1)
{
    dft_sub<R, OFF>(xr, xi);
    double cr = 1.0, ci = 0.0;
#pragma unroll
    for (int m = 1; m < R; ++m) {
        const double nr = cr * w1.x - ci * w1.y;
        ci = cr * w1.y + ci * w1.x;
        cr = nr;
        const int i = OFF + brev<R>(m);
        const double tr = xr[i] * cr - xi[i] * ci;
        xi[i] = xr[i] * ci + xi[i] * cr;
        xr[i] = tr;
    }
}

// inverse pass on a block: input m (natural index) times conj(w1^m), then the inverse
// butterfly (swap . forward . swap); output m ends at index brev(m)
template <int R, int OFF>
__device__ __forceinline__ void reg_inv(double (&xr)[32], double (&xi)[32], double2 w1)
{
    double cr = 1.0, ci = 0.0;
#pragma unroll
    for (int m = 1; m < R; ++m) {
        const double nr = cr * w1.x - ci * w1.y;
        ci = cr * w1.y + ci * w1.x;
        cr = nr;
        const int i = OFF + m;
        const double tr = xr[i] * cr + xi[i] * ci;
        xi[i] = xi[i] * cr - xr[i] * ci;
        xr[i] = tr;
    }
    dft_sub<R, OFF>(xi, xr);
}

// The pair's two segments are hop = 4096 = 16 * 256 apart (halo 2048 on both sides whatever
// lambda), so that in layout P1 a thread's 32 points of segment B are its points 16..47 of the
// 12288-sample union window U: 48 values per thread feed both planes, and the 2 x 4096 results
// form one contiguous 8192-sample window.  In tile order the window is reached through three
// address-sorted lists per pair (U[0, 8192), U[8192, 12288), results), each routed through the
// exchange buffer in chunks, so nothing but the 48 window values stays live in registers.
constexpr int kRegHop = 4096, kRegHalo = 2048;
#ifndef CM2_REG_CH
#define CM2_REG_CH 16
#endif
constexpr int kRegCh = CM2_REG_CH;
constexpr int kRegL1 = kRegN, kRegL2 = kRegHop, kRegLS = 2 * kRegHop;     // list lengths per pair
constexpr int kRegPer = kRegL1 + kRegL2 + kRegLS;                          // list entries per pair

// walk LEN (address, position) entries, kRegCh per thread at a time: gen(e, k, q) yields entry e
template <int LEN, class G, class F>
__device__ __forceinline__ void reg_walk(int t, G gen, F f)
{
#pragma unroll 1
    for (int c0 = 0; c0 < LEN / kRegT; c0 += kRegCh) {
        uint32_t kk[kRegCh];
        int qq[kRegCh];
#pragma unroll
        for (int u = 0; u < kRegCh; ++u) gen(t + (c0 + u) * kRegT, kk[u], qq[u]);
        f(kk, qq);
    }
}

// LISTS: the TODs are in the tile-bucketed order and reached through the three address-sorted
// lists of the pair; otherwise they are in time order and entry e of a walk is simply sample
// (window start + e) -- the same code path with computed instead of loaded addresses.
// On the tile order (LISTS) the walk is software-pipelined: both chunks of the first window list
// are requested up front, their gathers are issued back to back together with the second list,
// the first half of the result list is requested behind the last inverse pass; invalid entries
// are read from address 0 and zeroed by a select instead of a branch: 4 dependent memory round
// trips per pair (8 with one list chunk -> its gathers at a time).
template <bool LISTS>
__global__ __launch_bounds__(kRegT, 2) void k_overlap_save_reg(
    const PairDesc *__restrict__ pairs, int npairs, const double2 *__restrict__ W,
    const double *__restrict__ Hperm,
    const uint32_t *__restrict__ lst_k, const uint16_t *__restrict__ lst_q,
    const double *__restrict__ v, double *__restrict__ out)
{
    constexpr int N = kRegN;
    extern __shared__ double buf[];
    const int t = threadIdx.x;
    const int per_xcd = (npairs + 7) / 8;
    const int pair_id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (pair_id >= npairs) return;
    const PairDesc pd = pairs[pair_id];
    const double *hperm = Hperm + (int64_t)pd.blk * N;
    const bool has_b = pd.b_len > 0;
    double *__restrict__ b1 = buf + reg_base<1>(t);

    // ---- load the union window: U[m] = sample (a_start - HALO) + t + 256 m ----
    double U[48];
    if constexpr (LISTS) {
        constexpr int C = 16;
        static_assert(kRegL1 == 2 * C * kRegT && kRegL2 == C * kRegT, "chunking of the window lists");
        // the pair's three lists lie one behind the other: window part 1, part 2, results
        const uint32_t *__restrict__ lk1 = lst_k + (int64_t)pair_id * kRegPer + t;
        const uint16_t *__restrict__ lq1 = lst_q + (int64_t)pair_id * kRegPer + t;
        const uint32_t *__restrict__ lk2 = lk1 + kRegL1;
        const uint16_t *__restrict__ lq2 = lq1 + kRegL1;
        uint32_t ka[C], kb[C], kc[C];
        uint32_t qa[C], qb[C], qc[C];
        double vv[C], vw[C];
#pragma unroll
        for (int u = 0; u < C; ++u) { ka[u] = lk1[u * kRegT]; qa[u] = lq1[u * kRegT]; }
#pragma unroll
        for (int u = 0; u < C; ++u) { kb[u] = lk1[(C + u) * kRegT]; qb[u] = lq1[(C + u) * kRegT]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < C; ++u) vv[u] = v[ka[u] != kInvalidSample ? ka[u] : 0u];
#pragma unroll
        for (int u = 0; u < C; ++u) vw[u] = v[kb[u] != kInvalidSample ? kb[u] : 0u];
#pragma unroll
        for (int u = 0; u < C; ++u) { kc[u] = lk2[u * kRegT]; qc[u] = lq2[u * kRegT]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < C; ++u) buf[padi((int)qa[u])] = ka[u] != kInvalidSample ? vv[u] : 0.0;
#pragma unroll
        for (int u = 0; u < C; ++u) buf[padi((int)qb[u])] = kb[u] != kInvalidSample ? vw[u] : 0.0;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 32; ++m) U[m] = b1[reg_off<1>(m)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < C; ++u) vv[u] = v[kc[u] != kInvalidSample ? kc[u] : 0u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < C; ++u) buf[padi((int)qc[u])] = kc[u] != kInvalidSample ? vv[u] : 0.0;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) U[32 + m] = b1[reg_off<1>(m)];
        __syncthreads();
    } else {
        // time order: entry e of a walk is simply sample (window start + e)
        const int64_t w0 = pd.a_start - kRegHalo;         // time of window position 0
        reg_walk<kRegL1>(t, [&](int e, uint32_t &k, int &q) {
            const int64_t ts = w0 + e;
            k = (ts >= pd.lo && ts < pd.hi) ? (uint32_t)ts : kInvalidSample;
            q = e;
        }, [&](const uint32_t (&kk)[kRegCh], const int (&qq)[kRegCh]) {
            double vv[kRegCh];
#pragma unroll
            for (int u = 0; u < kRegCh; ++u) vv[u] = (kk[u] != kInvalidSample) ? v[kk[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < kRegCh; ++u) buf[padi(qq[u])] = vv[u];
        });
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 32; ++m) U[m] = b1[reg_off<1>(m)];
        __syncthreads();
        if (has_b) {
            reg_walk<kRegL2>(t, [&](int e, uint32_t &k, int &q) {
                const int64_t ts = w0 + kRegL1 + e;
                k = (ts >= pd.lo && ts < pd.hi) ? (uint32_t)ts : kInvalidSample;
                q = e;
            }, [&](const uint32_t (&kk)[kRegCh], const int (&qq)[kRegCh]) {
                double vv[kRegCh];
#pragma unroll
                for (int u = 0; u < kRegCh; ++u) vv[u] = (kk[u] != kInvalidSample) ? v[kk[u]] : 0.0;
#pragma unroll
                for (int u = 0; u < kRegCh; ++u) buf[padi(qq[u])] = vv[u];
            });
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; ++m) U[32 + m] = b1[reg_off<1>(m)];
            __syncthreads();
        }
    }
    double zr[32], zi[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        zr[m] = U[m];
        zi[m] = has_b ? U[m + 16] : 0.0;
    }

    const double2 w_a = W[t];                       // n = N:   w = exp(-2 pi i j / N),   j = t
    const double2 w_b = W[32 * (t & 15)];           // n = 256: w = exp(-2 pi i j / 256), j = t & 15

    // ---- forward: radix 32, radix 16, then (radix 16, spectrum, inverse radix 16) ----
    reg_fwd<32, 0>(zr, zi, w_a);
    reg_exchange<1, 2, 32>(zr, buf, t);
    reg_exchange<1, 2, 32>(zi, buf, t);
    reg_fwd<16, 0>(zr, zi, w_b);
    reg_fwd<16, 16>(zr, zi, w_b);
    reg_exchange<2, 3, 16>(zr, buf, t);
    reg_exchange<2, 3, 16>(zi, buf, t);
    // middle: radix 16 (output k at index brev(k)), spectrum product in place, inverse radix
    // 16 as a decimation-in-time network on the bit-reversed data (swap . forward . swap)
    dft_sub<16, 0>(zr, zi);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double hv = hperm[32 * t + k];
        zr[brev<16>(k)] *= hv;
        zi[brev<16>(k)] *= hv;
    }
    dit_sub<16, 0>(zi, zr);
    __builtin_amdgcn_sched_barrier(0);
    dft_sub<16, 16>(zr, zi);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double hv = hperm[32 * t + 16 + k];
        zr[16 + brev<16>(k)] *= hv;
        zi[16 + brev<16>(k)] *= hv;
    }
    dit_sub<16, 16>(zi, zr);
    // ---- inverse: radix 16, radix 32 ----
    // The twiddle powers are recomputed from an opaque copy of w: otherwise the compiler
    // keeps the 46 powers of the forward passes alive (spilled) across the whole transform.
    double2 w_bi = w_b, w_ai = w_a;
    asm volatile("" : "+v"(w_bi.x), "+v"(w_bi.y), "+v"(w_ai.x), "+v"(w_ai.y));
    reg_exchange<3, 2, 0>(zr, buf, t);
    reg_exchange<3, 2, 0>(zi, buf, t);
    reg_inv<16, 0>(zr, zi, w_bi);
    reg_inv<16, 16>(zr, zi, w_bi);
    reg_exchange<2, 1, 16>(zr, buf, t);
    reg_exchange<2, 1, 16>(zi, buf, t);
    // first half of the result list, requested behind the last inverse pass
    uint32_t ks0[16], qs0[16];
    if constexpr (LISTS) {
        const uint32_t *__restrict__ lks = lst_k + (int64_t)pair_id * kRegPer + kRegL1 + kRegL2 + t;
        const uint16_t *__restrict__ lqs = lst_q + (int64_t)pair_id * kRegPer + kRegL1 + kRegL2 + t;
#pragma unroll
        for (int u = 0; u < 16; ++u) { ks0[u] = lks[u * kRegT]; qs0[u] = lqs[u * kRegT]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    reg_inv<32, 0>(zr, zi, w_ai);                      // result slot m at index brev<32>(m)

    // ---- store: results of A are slots 8..23 of zr (window positions 2048..6143), of B the
    //      same slots of zi; together the result window R[0, 8192), R[j] at P1 slot j / 256 ----
    {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            b1[reg_off<1>(m)] = zr[brev<32>(m + 8)];
            b1[reg_off<1>(m + 16)] = zi[brev<32>(m + 8)];
        }
        if constexpr (LISTS) {
            constexpr int C = kRegLS / kRegT;
            const uint32_t *__restrict__ lks = lst_k + (int64_t)pair_id * kRegPer + kRegL1 + kRegL2 + t;
            const uint16_t *__restrict__ lqs = lst_q + (int64_t)pair_id * kRegPer + kRegL1 + kRegL2 + t;
            uint32_t ks[C], qs[C];
#pragma unroll
            for (int u = 0; u < 16; ++u) { ks[u] = ks0[u]; qs[u] = qs0[u]; }
#pragma unroll
            for (int u = 16; u < C; ++u) { ks[u] = lks[u * kRegT]; qs[u] = lqs[u * kRegT]; }
            __syncthreads();
            double rv[C];                     // all LDS reads first: one wait, not one per store
#pragma unroll
            for (int u = 0; u < C; ++u) rv[u] = buf[padi((int)qs[u])];
#pragma unroll
            for (int u = 0; u < C; ++u)
                if (ks[u] != kInvalidSample) out[ks[u]] = rv[u];
            return;
        }
        __syncthreads();
        reg_walk<kRegLS>(t, [&](int e, uint32_t &k, int &q) {
            // result j: segment A for j < 4096, segment B (b_start = a_start + 4096) after
            const bool ok = e < kRegHop ? e < pd.a_len : e - kRegHop < pd.b_len;
            k = ok ? (uint32_t)(pd.a_start + e) : kInvalidSample;
            q = e;
        }, [&](const uint32_t (&kk)[kRegCh], const int (&qq)[kRegCh]) {
#pragma unroll
            for (int u = 0; u < kRegCh; ++u)
                if (kk[u] != kInvalidSample) out[kk[u]] = buf[padi(qq[u])];
        });
    }
}

// entries of the three lists of pairs [p0, p0 + np), 20480 per pair, in the order they are stored:
//   [0, 8192)      U[q], q = e              -> value q
//   [8192, 12288)  U[q], q = e              -> value q - 8192
//   [12288, 20480) results R[j], j = e - 12288 (A: j < 4096, B: j - 4096) -> value j
// key = address in the tile order (0xFFFFFFFF: no sample); a segmented sort then orders every
// list by address.
__global__ __launch_bounds__(256) void k_reg_keys(const PairDesc *__restrict__ pairs, int64_t p0,
                                                   int64_t np, const uint32_t *__restrict__ idx,
                                                   uint32_t *__restrict__ keys,
                                                   uint16_t *__restrict__ vals)
{
    const int64_t total = np * kRegPer;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const int64_t p = g / kRegPer;
        const int e = (int)(g - p * kRegPer);
        const PairDesc pd = pairs[p0 + p];
        uint32_t k = kInvalidSample;
        int val;
        if (e < kRegL1 + kRegL2) {
            val = e < kRegL1 ? e : e - kRegL1;
            const int64_t ts = pd.a_start - kRegHalo + e;
            if (ts >= pd.lo && ts < pd.hi && (e < kRegL1 || pd.b_len > 0)) k = idx[ts];
        } else {
            val = e - (kRegL1 + kRegL2);
            const int ja = val, jb = val - kRegHop;
            if (ja < pd.a_len && ja < kRegHop) k = idx[pd.a_start + ja];
            else if (jb >= 0 && jb < pd.b_len) k = idx[pd.b_start + jb];
        }
        keys[g] = k;
        vals[g] = (uint16_t)val;
    }
}

// first (end = 0) / one-past-last (end = 1) entry of list s % 3 of pair s / 3, relative to the
// chunk being sorted (both offset iterators of the segmented sort must have one type)
struct ListOffset {
    int end;
    __host__ __device__ int operator()(int s) const
    {
        const int l = s % 3 + end;
        return (s / 3) * kRegPer + (l == 0 ? 0 : (l == 1 ? kRegL1 : (l == 2 ? kRegL1 + kRegL2 : kRegPer)));
    }
};

// W[t] = exp(-2 pi i t / N)
__global__ void k_twiddles(int N, double2 *__restrict__ W)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N) W[t] = make_double2(cospi(2.0 * t / N), -sinpi(2.0 * t / N));
}

// Hperm[b][a] = H_b(k(a)) / N with a = d1*(R2*R3) + d2*R3 + d3  <->  k = d1 + R1*d2 + R1*R2*d3
// and H_b(k) = a0 + 2 sum_{j>=1} a_j cos(2 pi j k / N)   (real, even: symmetric band)
__global__ __launch_bounds__(256) void k_spectrum_perm(int nb, int64_t lambda, int N, int R1,
                                                        int R2, int R3,
                                                        const double *__restrict__ bands,
                                                        double *__restrict__ Hperm)
{
    const int64_t total = (int64_t)nb * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / N;
        const int a = (int)(e - b * N);
        const int d1 = a / (R2 * R3), d2 = (a / R3) % R2, d3 = a % R3;
        const int64_t k = d1 + (int64_t)R1 * d2 + (int64_t)R1 * R2 * d3;
        const double *band = bands + b * lambda;
        double acc = 0.0;
        for (int64_t j = lambda - 1; j >= 1; --j) {
            const int64_t m = (j * k) % N;
            acc += band[j] * cospi(2.0 * (double)m / (double)N);
        }
        Hperm[e] = (band[0] + 2.0 * acc) / (double)N;
    }
}

}  // namespace

namespace cm2 {

struct FusedOS {
    int halo = 0;
    int64_t npairs = 0;
    PairDesc *d_pairs = nullptr;
    double *d_Hperm = nullptr;           // spectrum of every block in the (32, 16, 16) digit order
    double2 *d_W = nullptr;              // exp(-2 pi i t / 8192)
    // address-sorted lists of the tile-order path, built for one tile plan at a time
    uint64_t list_plan = 0;              // id of the tile plan the lists were built for
    uint32_t *d_lst_k = nullptr;         // [npairs][20480] addresses: window part 1 | part 2 | results
    uint16_t *d_lst_q = nullptr;         // positions in the window / result window
    // the one-real-window kernel (cm2_fft_real.hip), built on first use when CM2_OS_KERNEL selects it
    RealOS *real = nullptr;
    int real_pt = 0;                     // 0: segment-pair kernel of this file
    int real_lists = 0;                  // 0 chosen by tile count, 1 plain, 2 run-coded, 3 inverse
    double *d_bands = nullptr;           // borrowed from the noise operator (lives as long as it does)
    int64_t lambda = 0;
    std::vector<int64_t> off;
};

// CM2_OS_KERNEL = pair | real16 | real32 selects the overlap-save kernel of the tile-order and
// time-order applications; CM2_OS_LISTS = auto | rc | plain | inv the list format of the real-window
// kernel (auto: inverse lists from 768 pixel tiles up, where a half window's address runs get short).
static void os_choice(int *pt, int *lists)
{
    *pt = 32;               // fastest on MI355X (profiles/r03_os_variants.md)
    *lists = 0;
    if (const char *e = getenv("CM2_OS_KERNEL")) {
        if (!strcmp(e, "pair")) *pt = 0;
        else if (!strcmp(e, "real32")) *pt = 32;
        else if (!strcmp(e, "real16")) *pt = 16;
        else if (!strcmp(e, "wide32")) *pt = 64;         // 32-point windows on 512 threads x 16 points
    }
    if (const char *e = getenv("CM2_OS_LISTS"))
    {
        if (!strcmp(e, "plain")) *lists = 1;
        else if (!strcmp(e, "inv")) *lists = 3;
        else if (!strcmp(e, "rc")) *lists = 2;
    }
}

static int ensure_pair_state(FusedOS *f, hipStream_t stream);

static int ensure_real(FusedOS *f, hipStream_t stream)
{
    int pt, lists;
    os_choice(&pt, &lists);
    f->real_lists = lists;
    if (pt == f->real_pt && (pt == 0 || f->real)) return 0;
    if (f->real) real_os_destroy(f->real);
    f->real = nullptr;
    f->real_pt = pt;
    if (pt == 0) return 0;
    return real_os_create(&f->real, pt, f->d_bands, f->lambda, f->off, stream);
}

static void free_lists(FusedOS *f)
{
    if (f->d_lst_k) (void)cm2::dev_free(f->d_lst_k);
    if (f->d_lst_q) (void)cm2::dev_free(f->d_lst_q);
    f->d_lst_k = nullptr;
    f->d_lst_q = nullptr;
    f->list_plan = 0;
}

void fused_os_destroy(FusedOS *f)
{
    if (!f) return;
    free_lists(f);
    if (f->real) real_os_destroy(f->real);
    void *ptrs[] = {f->d_pairs, f->d_W, f->d_Hperm};
    for (void *q : ptrs)
        if (q) (void)cm2::dev_free(q);
    delete f;
}

bool fused_os_supported(int64_t lambda) { return lambda >= 1 && lambda - 1 <= kRegHalo; }

int64_t fused_os_length(const FusedOS *f) { return f ? kRegN : 0; }

template <bool LISTS>
static int launch_reg(const FusedOS *f, const double *d_v, double *d_out, hipStream_t stream)
{
    constexpr size_t lds = sizeof(double) * (size_t)(kRegN + kRegN / 32);
    static size_t granted[64] = {0};
    CM2_HIP(ensure_dynamic_lds((const void *)k_overlap_save_reg<LISTS>, lds, granted));
    if (f->npairs == 0) return 0;
    const int grid = (int)(((f->npairs + 7) / 8) * 8);       // whole rounds over the 8 XCDs
    k_overlap_save_reg<LISTS><<<grid, kRegT, lds, stream>>>(
        f->d_pairs, (int)f->npairs, f->d_W, f->d_Hperm, f->d_lst_k, f->d_lst_q, d_v, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

int fused_os_apply(const FusedOS *f_, const double *d_v, double *d_out, hipStream_t stream)
{
    FusedOS *f = const_cast<FusedOS *>(f_);
    if (int rc = ensure_real(f, stream)) return rc;
    if (f->real) return real_os_apply(f->real, d_v, d_out, stream);
    if (int rc = ensure_pair_state(f, stream)) return rc;
    return launch_reg<false>(f, d_v, d_out, stream);
}

double fused_os_tile_info(const FusedOS *f, int *kernel)
{
    if (kernel) {
        kernel[0] = f ? f->real_pt : 0;
        kernel[1] = f ? (f->real ? real_os_list_mode(f->real) : (f->d_lst_k ? 1 : 0)) : 0;
    }
    if (f && f->real) return real_os_tile_bytes_per_sample(f->real);
    return 35.0;           // 1.5 x 8 gathered + 8 written + 2.5 list entries of 6 bytes
}

// the three address-sorted lists of every pair for the tile plan whose index is d_idx
static int build_lists(FusedOS *f, const uint32_t *d_idx, uint64_t plan_id, hipStream_t stream)
{
    free_lists(f);
    if (f->npairs == 0) {
        f->list_plan = plan_id;
        return 0;
    }
    const int64_t total = f->npairs * kRegPer;
    struct Guard { FusedOS *f; ~Guard() { if (f) free_lists(f); } } guard{f};     // early returns
    CM2_HIP(cm2::dev_malloc(&f->d_lst_k, sizeof(uint32_t) * total));
    CM2_HIP(cm2::dev_malloc(&f->d_lst_q, sizeof(uint16_t) * total));
    // One segment per list, 32-bit keys: every segment (4096 or 8192 entries) is sorted inside one
    // workgroup -- 4 ms at 1e8 samples where a global sort on (pair, list, address) keys took 12.
    // hipCUB counts items in int: pairs go through in chunks of at most 2^30 entries.
    int64_t chunk_pairs = ((int64_t)1 << 30) / kRegPer;
    if (const char *e = getenv("CM2_OS_LIST_CHUNK_PAIRS"))      // test hook: force several chunks
        if (atoll(e) > 0 && atoll(e) < chunk_pairs) chunk_pairs = atoll(e);
    const int64_t cp_max = f->npairs < chunk_pairs ? f->npairs : chunk_pairs;
    DevTemp<uint32_t> keys_in;
    DevTemp<uint16_t> vals_in;
    DevTemp<char> d_temp;
    CM2_HIP(keys_in.alloc(cp_max * kRegPer));
    CM2_HIP(vals_in.alloc(cp_max * kRegPer));
    hipcub::CountingInputIterator<int> seg_id(0);
    using OffsetIt = hipcub::TransformInputIterator<int, ListOffset, hipcub::CountingInputIterator<int>>;
    OffsetIt seg_begin(seg_id, ListOffset{0}), seg_end(seg_id, ListOffset{1});
    size_t tb = 0;
    CM2_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(
        nullptr, tb, keys_in.p, f->d_lst_k, vals_in.p, f->d_lst_q, (int)(cp_max * kRegPer),
        (int)(3 * cp_max), seg_begin, seg_end, 0, 32, stream));
    CM2_HIP(d_temp.alloc(tb + 16));
    for (int64_t p0 = 0; p0 < f->npairs; p0 += chunk_pairs) {
        const int64_t np = f->npairs - p0 < chunk_pairs ? f->npairs - p0 : chunk_pairs;
        k_reg_keys<<<grid_for(np * kRegPer), kBlock, 0, stream>>>(f->d_pairs, p0, np, d_idx, keys_in, vals_in);
        CM2_LAUNCH_OK();
        size_t tbc = tb;
        CM2_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(
            d_temp.p, tbc, keys_in.p, f->d_lst_k + p0 * kRegPer, vals_in.p, f->d_lst_q + p0 * kRegPer,
            (int)(np * kRegPer), (int)(3 * np), seg_begin, seg_end, 0, 32, stream));
    }
    CM2_HIP(hipStreamSynchronize(stream));
    guard.f = nullptr;
    f->list_plan = plan_id;
    return 0;
}

int fused_os_apply_indexed(FusedOS *f, const uint32_t *d_idx, const int64_t *d_tile_off, uint64_t plan_id,
                           int64_t ntiles, int64_t nvalid, const double *d_v, double *d_out, hipStream_t stream)
{
    if (int rc = ensure_real(f, stream)) return rc;
    if (f->real)
        return real_os_apply_indexed(f->real, d_idx, d_tile_off, plan_id, ntiles, nvalid, f->real_lists, d_v, d_out,
                                     stream);
    if (int rc = ensure_pair_state(f, stream)) return rc;
    // the lists belong to ONE tile plan; keyed on its id (a device address may be handed out
    // again to a later plan of the same size)
    if (f->list_plan != plan_id || (f->npairs > 0 && !f->d_lst_k))
        if (int rc = build_lists(f, d_idx, plan_id, stream)) return rc;
    return launch_reg<true>(f, d_v, d_out, stream);
}

// segment pairs, permuted spectra and twiddles of the segment-pair kernel: built when that kernel
// is the one selected (CM2_OS_KERNEL=pair), not for every operator
static int ensure_pair_state(FusedOS *f, hipStream_t stream)
{
    if (f->d_pairs) return 0;
    const std::vector<int64_t> &off = f->off;
    const int64_t lambda = f->lambda;
    const int64_t nb = (int64_t)off.size() - 1;
    // segment pairs: fixed geometry (hop 4096, halo 2048 whatever lambda is)
    std::vector<PairDesc> pairs;
    for (int64_t b = 0; b < nb; ++b) {
        for (int64_t s0 = off[b]; s0 < off[b + 1]; s0 += 2 * kRegHop) {
            PairDesc pd;
            pd.lo = off[b]; pd.hi = off[b + 1]; pd.blk = (int32_t)b; pd.pad = 0;
            pd.a_start = s0;
            pd.a_len = (off[b + 1] - s0 < kRegHop) ? off[b + 1] - s0 : kRegHop;
            pd.b_start = s0 + kRegHop;
            pd.b_len = pd.b_start < off[b + 1]
                           ? ((off[b + 1] - pd.b_start < kRegHop) ? off[b + 1] - pd.b_start : kRegHop)
                           : 0;
            if (pd.b_len == 0) pd.b_start = s0;
            pairs.push_back(pd);
        }
    }
    f->npairs = (int64_t)pairs.size();
    CM2_HIP(cm2::dev_malloc(&f->d_pairs, sizeof(PairDesc) * (pairs.size() ? pairs.size() : 1)));
    if (!pairs.empty())
        CM2_HIP(hipMemcpy(f->d_pairs, pairs.data(), sizeof(PairDesc) * pairs.size(),
                          hipMemcpyHostToDevice));
    CM2_HIP(cm2::dev_malloc(&f->d_Hperm, sizeof(double) * (nb > 0 ? nb : 1) * kRegN));
    if (nb > 0) {
        k_spectrum_perm<<<grid_for(nb * kRegN), kBlock, 0, stream>>>((int)nb, lambda, kRegN, 32, 16, 16,
                                                                    f->d_bands, f->d_Hperm);
        CM2_LAUNCH_OK();
    }
    CM2_HIP(cm2::dev_malloc(&f->d_W, sizeof(double2) * kRegN));
    k_twiddles<<<(kRegN + 255) / 256, 256, 0, stream>>>(kRegN, f->d_W);
    CM2_LAUNCH_OK();
    CM2_HIP(hipStreamSynchronize(stream));
    return 0;
}

int fused_os_create(FusedOS **out, const double *d_bands, int64_t lambda,
                    const std::vector<int64_t> &off, hipStream_t stream)
{
    CM2_CHECK(out != nullptr, "fused_os_create: out is NULL");
    *out = nullptr;
    CM2_CHECK(fused_os_supported(lambda), "fused overlap-save supports lambda <= 2049, got %lld",
              (long long)lambda);
    FusedOS *f = new FusedOS();
    struct Guard { FusedOS *f; ~Guard() { if (f) fused_os_destroy(f); } } guard{f};   // early returns
    f->halo = (int)(lambda - 1);
    f->d_bands = const_cast<double *>(d_bands);
    f->lambda = lambda;
    f->off = off;
    // the selected kernel's plan (windows, spectra) is built now, so that a bad band fails here
    if (int rc = ensure_real(f, stream)) return rc;
    if (!f->real)
        if (int rc = ensure_pair_state(f, stream)) return rc;
    guard.f = nullptr;
    *out = f;
    return 0;
}

}  // namespace cm2
