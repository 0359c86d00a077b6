// cm2_tiles_fixed.hip -- P^T on the tile-bucketed order with every pixel's terms added IN TIME
// ORDER starting from 0, exactly like the reference's serial scatter loop
// (interfaces/linearoperators.py:394-400, :452-458, :509-516): no atomics, bitwise reproducible
// from run to run, and bit-identical to the serial loop when the plan keeps cos and sin
// (CM2_TILE_ANGLES=full; the default half-angle storage rebuilds them to ~2e-16).
//
// ONE workgroup owns a tile from its first sample to its last: the tile's accumulators stay in
// LDS for the whole bucket and are written out once with plain stores.  The bucket is walked
// slice by slice (S consecutive TB samples = S consecutive-in-time samples of this tile).  At
// plan time the samples of a slice are sorted by (pixel, time) and packed into GROUPS of four
// list entries such that a group holds whole runs (run = the slice's samples of one pixel):
//
//   entry  = pl word (pixel in tile, sign of cos) | offset in the slice << 16 | level << 28
//   group  = 4 entries + their 4 half angles (or cos, sin), padded with null entries
//   thread t of the workgroup owns group t of the slice: it reads its 16 + 32 bytes with
//   three 16-byte loads straight into registers -- two slices ahead of the one it reduces --,
//   picks the 4 TOD values out of the slice staged in LDS, and adds its entries to the pixels' LDS
//   accumulators one after the other (ds_add_f64: adds of one thread to one address happen in
//   program order).  A pixel is touched by one thread per pass and the slices follow each other
//   in time, so each sum is the reference's, term by term.
//
// Runs of 5..60 samples are cut into pieces of 4 in consecutive groups of ONE wave (the packer pads
// so that a run never straddles a multiple of 64 groups), piece p carrying level p: every wave
// makes one pass per level (the slice's highest level is known at plan time); the LDS executes a
// wave's instructions in issue order, so the pieces are added in order without a barrier between
// the passes (round 2 had one per level: dense tiles, 5-6 hits per pixel and slice, paid 2-3 of
// them per slice).  Longer runs (a pixel hit
// > 60 times inside one slice: hot pixels) are kept out of the groups and walked by one thread
// each from a separate list.
//
// HBM per sample: 8 B TOD + ~1.14 x 12 B of list (padding of the groups) = ~21.7 B, every load
// 16 B per lane and coalesced.  What the kernel had to get right to stream at HBM speed
// (measured, 1e8 samples, nside 256: 0.37 ms with the reduction switched off, 0.45 ms with it):
//   * no register spill in the slice loop: a scratch reload is a VMEM operation, and waiting for
//     it (vmcnt counts in order) drains every prefetched load issued before it;
//   * unconditional loads (clamped indices) and a scheduling barrier per slice, so that the
//     compiler waits with s_waitcnt vmcnt(N > 0) for the oldest slice only;
//   * LDS adds instead of a load / add / store chain per run (12 of 24 LDS operations less per
//     thread and slice, and no dependent round trip).
#include "cm2_tiles.h"

#include <algorithm>
#include <cstring>
#include <mutex>

#include <hipcub/hipcub.hpp>

using namespace cm2;

namespace {

constexpr int kFxT = 512;               // threads = groups per slice handled in one round
constexpr int kFxDepth = 2;             // slices fetched ahead of the one being reduced
constexpr uint32_t kFxNull = 0xFFFFFFFFu;
constexpr int kFxMaxLevel = 14;         // pieces of 4: runs up to 60 samples go into groups
// Hot pixels.  A run longer than `chunk_min` entries inside one slice (a pixel that takes a large
// share of a tile's samples: a stare at a source, a tile that is one pixel) is not walked term by
// term by one thread -- 1536 dependent additions per slice while 511 threads wait -- but cut into
// chunks of kFxChunk consecutive entries: one thread per chunk sums its terms in time order in
// registers, the chunk sums of a run are added in time order by one thread, and the result joins
// the pixel's accumulator.  The chunk boundaries and both orders are fixed by the plan, so the
// result is reproducible bit for bit and independent of the hit map; it differs from the serial
// sum by the rounding of a regrouped sum (~1e-16 relative per level).  cm2_tiles_set_pt_order(t, 2)
// / CM2_PT_ORDER=exact keep the pure time order for every run.
constexpr int kFxChunk = 32;
constexpr int kFxChunkMinDefault = 256;
constexpr int kFxMaxChunks = 128;       // per slice: S / kFxChunk + long runs <= 64 + 8

// ------------------------------------------------------------------- fused form -----
// The ranges of the hot tiles (a one-pixel tile of very many samples is reduced by ranges of kHotChunk samples)
// used to be two more launches behind the main one (k_Pt_hot: 306 workgroups for a 5e6-sample pixel, 22 us;
// k_hot_combine, 7 us).  They are work items at the END of the main launch's grid now, and the last range of a
// tile to finish adds the tile's range sums.  Who is last depends on the dispatch; WHAT is added and in which
// order does not (range after range, as k_hot_combine did): the same bits, whoever does it.  Measured at C4
// with 5 % of the samples on one pixel (profiles/r05_pt_fused.md): P^T 0.402-0.411 against 0.440-0.449 ms.
// The copies of SPLIT tiles stay with k_parts_combine: adding them in the main launch as well (the tile's last
// part to finish reads the other parts' copies) was built and measured -- P^T of the uneven hit map 0.438-0.446
// against 0.423-0.428 ms: one workgroup reading k copies at the tail of the launch loses to a short, wide
// kernel -- and removed.
// Hand-off between workgroups inside a launch (cdna_hip_programming.md, Guideline 16, counter form with
// write-through payload): the producer stores its three sums sc1 (no release fence: a release writes back the
// XCD's whole L2), drains them, ONE lane does a relaxed agent-scope fetch_add on the tile's counter; the
// workgroup that draws the last ticket: agent-scope acquire fence by one lane, drain, barrier, then plain loads
// of the others' sums.  The counters are zeroed by a memset in front of every launch (fx_launch_inst).
struct FxFused {
    unsigned int *count;            // [nhot] arrival counters
    const int64_t *hot_range;       // [ranges][2] first / one-past-last TB position
    const int *hot_range_tile;      // [ranges] index in hot_tiles
    const int64_t *hot_tiles;       // [nhot][3] first pixel, first range, ranges
    double *hot_partial;            // [ranges][3]
    const uint16_t *pl;             // TB-order streams of the plan (hot ranges read them directly)
    const double *a_tb, *b_tb;
};

constexpr int kHotChunk = 16384, kHotT = 1024;          // samples per range, (virtual) threads per range
// (1024 threads: 306 workgroups of 256 left the chip with one wave per SIMD, 45 us for 5e6 samples)

// a handed-off double: WRITE-THROUGH (sc1) store, so that the producer needs no release fence
__device__ __forceinline__ void fx_publish(double *p, double x)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, x),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool fx_last_arriver(unsigned int *counter, unsigned int expected, int *lds_flag, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its (sc1) stores
    __syncthreads();
    if (tid == 0) {
        const unsigned int old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old + 1u == expected) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// One range of a hot tile (kHotChunk consecutive samples of ONE pixel) by one workgroup of kFxT = 512
// threads doing the work of k_Pt_hot's 1024: thread t is the virtual threads t and t + 512, each adding the
// terms at positions vt, vt + 1024, ... of the range in that order; the 1024 sums are combined by the same
// halving tree.  Then the last range of the tile adds the tile's range sums in time order (k_hot_combine).
template <int POL, bool HALF>
__device__ __forceinline__ void fx_hot_item(const FxFused &z, int64_t c, const double *__restrict__ v_tb,
                                            double *__restrict__ out, double *sm, int tid)
{
    double *red = sm;                                        // [3][kHotT]
    const int64_t k0 = z.hot_range[2 * c], k1 = z.hot_range[2 * c + 1];
    double sv[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    constexpr int U = 4;
    for (int64_t kk = k0 + tid; kk < k1; kk += (int64_t)U * kHotT) {
        double v[2][U], a[2][U], b2[2][U];
        uint16_t w[2][U];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t k = kk + (int64_t)hh * kFxT + (int64_t)u * kHotT;
                const int64_t kc = k < k1 ? k : k1 - 1;
                v[hh][u] = v_tb[kc];
                a[hh][u] = POL > 1 ? z.a_tb[kc] : 0.0;
                b2[hh][u] = (POL > 1 && !HALF) ? z.b_tb[kc] : 0.0;
                w[hh][u] = (POL > 1 && HALF) ? z.pl[kc] : (uint16_t)0;
            }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (kk + (int64_t)hh * kFxT + (int64_t)u * kHotT >= k1) continue;
                sv[hh] += v[hh][u];
                if (POL > 1) {
                    double cc, ss;
                    if (HALF) {
                        const double h = a[hh][u], h2 = h * h, inv = 1.0 / (1.0 + h2);
                        cc = (1.0 - h2) * inv;
                        ss = (h + h) * inv;
                        if (w[hh][u] & 0x8000u) cc = -cc;
                    } else {
                        cc = a[hh][u];
                        ss = b2[hh][u];
                    }
                    s1[hh] += v[hh][u] * cc;
                    s2[hh] += v[hh][u] * ss;
                }
            }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        red[tid + hh * kFxT] = sv[hh];
        red[kHotT + tid + hh * kFxT] = s1[hh];
        red[2 * kHotT + tid + hh * kFxT] = s2[hh];
    }
    __syncthreads();
    for (int h = kHotT / 2; h >= 1; h >>= 1) {
        if (tid < h) {
            red[tid] += red[tid + h];
            red[kHotT + tid] += red[kHotT + tid + h];
            red[2 * kHotT + tid] += red[2 * kHotT + tid + h];
        }
        __syncthreads();
    }
    if (tid == 0) {
        fx_publish(z.hot_partial + 3 * c, red[0]);
        fx_publish(z.hot_partial + 3 * c + 1, red[kHotT]);
        fx_publish(z.hot_partial + 3 * c + 2, red[2 * kHotT]);
    }
    const int h = z.hot_range_tile[c];
    const int64_t p0 = z.hot_tiles[3 * h], c0 = z.hot_tiles[3 * h + 1], nc = z.hot_tiles[3 * h + 2];
    int *flag = reinterpret_cast<int *>(red + 3 * kHotT);
    if (!fx_last_arriver(z.count + h, (unsigned int)nc, flag, tid)) return;
    // the tile's range sums in time order (k_hot_combine): staged 256 ranges at a time, one thread adds
    double *st = red;
    double tv = 0.0, t1 = 0.0, t2 = 0.0;
    for (int64_t base = 0; base < nc; base += 256) {
        const int64_t n = nc - base < 256 ? nc - base : 256;
        for (int64_t i = tid; i < 3 * n; i += kFxT) st[i] = z.hot_partial[3 * (c0 + base) + i];
        __syncthreads();
        if (tid == 0)
            for (int64_t q = 0; q < n; ++q) {
                tv += st[3 * q];
                t1 += st[3 * q + 1];
                t2 += st[3 * q + 2];
            }
        __syncthreads();
    }
    if (tid != 0) return;
    if (POL == 1) {
        out[p0] = tv;
    } else if (POL == 2) {
        out[2 * p0] = t1;
        out[2 * p0 + 1] = t2;
    } else {
        out[3 * p0] = tv;
        out[3 * p0 + 1] = t1;
        out[3 * p0 + 2] = t2;
    }
}

// ------------------------------------------------------------------- kernel --------
template <int POL, bool HALF, int VPT>
__global__ __launch_bounds__(kFxT, 4) void k_Pt_tiles_fixed(
    int tp, const int64_t *__restrict__ tile_p0, int tile0, const uint2 *__restrict__ sk,
    const int64_t *__restrict__ slice0, const uint2 *__restrict__ meta,
    const uint4 *__restrict__ gent, const double2 *__restrict__ ga,
    const double2 *__restrict__ gb, const uint2 *__restrict__ trun,
    const uint32_t *__restrict__ tent, const double *__restrict__ ta,
    const double *__restrict__ tb, const double *__restrict__ v_tb, double *__restrict__ out,
    uint32_t chunk_min, const uint8_t *__restrict__ hot, const int4 *__restrict__ parts,
    int part0, double *__restrict__ scratch, const FxFused *__restrict__ fz, int nmain, int64_t hot_c0)
{
    constexpr int D = kFxDepth;
    constexpr bool ANG = POL > 1, TWO = POL > 1 && !HALF;
    constexpr uint32_t QM = HALF ? 0x7FFFu : 0xFFFFu;
    extern __shared__ double sm[];
    double *tile = sm;                                   // tp * POL accumulators
    // the slice's TOD values, TB order, in one of TWO buffers (slice j in buffer j & 1): a wave that
    // is ahead stages the next slice while another still gathers from this one, so a slice costs ONE
    // workgroup barrier (behind the staging; it also completes every wave's LDS adds of the slice
    // before, which keeps the slices' adds to one pixel in time order)
    double *vbuf0 = sm + (int64_t)tp * POL;              // VPT * kFxT values
    double *vbuf1 = vbuf0 + VPT * kFxT;
    double *part = vbuf1 + VPT * kFxT;                   // 3 x kFxMaxChunks chunk sums of hot runs
    const int tid = threadIdx.x;
    if (fz && (int)blockIdx.x >= nmain) {                // fused form: the ranges of the hot tiles come last
        fx_hot_item<POL, HALF>(*fz, hot_c0 + ((int)blockIdx.x - nmain), v_tb, out, sm, tid);
        return;
    }
    // the workgroup's work: a whole tile, or (plans with parts) some consecutive slices of one, summed
    // into a scratch copy of the tile that k_parts_combine adds to the other parts' copies
    int b = tile0 + (int)blockIdx.x, nsl = 0;
    int64_t s0 = 0;
    double *o_part = nullptr;
    if (parts) {
        const int4 pd = parts[part0 + (int)blockIdx.x];
        b = pd.x;
        nsl = pd.y;
        s0 = pd.z;
        if (pd.w >= 0) o_part = scratch + (int64_t)pd.w * ((int64_t)tp * POL);
    }
    if (hot && hot[b]) return;                           // reduced by k_Pt_hot (many workgroups)
    const int64_t p0 = tile_p0[b];
    const int64_t np = tile_p0[b + 1] - p0;
    const int nvals = (int)(np * POL);
    for (int i = tid; i < nvals; i += kFxT) tile[i] = 0.0;
    if (!parts) {
        s0 = slice0[b];
        nsl = (int)(slice0[b + 1] - s0);
    }

    // register ring: slice j lives in slot j % D from the moment slice j - D has been staged.
    // All loads are unconditional (indices clamped) and kept together per slice, so that the
    // compiler counts outstanding loads (s_waitcnt vmcnt(N)) instead of draining them.
    double pv[D][VPT];
    uint4 pe[D];
    double2 pa[D][2], pb[D][2];
    uint2 pm[D][2];                                      // meta of slice j and j + 1
    // {first group, first tail run} of the NEXT slice to fetch and of its successor: loaded one
    // fetch ahead, so that the group addresses never wait for them
    // ... and {first TB address, samples} of it: the slices of a tile are its segments (one per span
    // of the plan, cut where longer than the slice length), not consecutive addresses
    uint2 nx0 = meta[s0], nx1 = meta[s0 + (nsl > 0 ? 1 : 0)], nk = sk[s0];
    auto fetch = [&](int slot, int j) {
        const int64_t kb = (int64_t)nk.x;
        const int len = (int)nk.y;
        const uint2 m0 = nx0, m1 = nx1;
        pm[slot][0] = m0;
        pm[slot][1] = m1;
        {
            const int jn = j + 1 < nsl ? j + 1 : nsl - 1;
            nx0 = meta[s0 + jn];
            nx1 = meta[s0 + jn + 1];
            nk = sk[s0 + jn];
        }
        const uint32_t G = m1.x - m0.x;
        const int64_t g = (int64_t)m0.x + ((uint32_t)tid < G ? tid : 0);
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            const int i = tid + u * kFxT;
            pv[slot][u] = v_tb[kb + (i < len ? i : len - 1)];
        }
        pe[slot] = gent[g];
        if (ANG) {
            pa[slot][0] = ga[2 * g];
            pa[slot][1] = ga[2 * g + 1];
        }
        if (TWO) {
            pb[slot][0] = gb[2 * g];
            pb[slot][1] = gb[2 * g + 1];
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // terms of one list entry: (v, v cos, v sin)
    auto terms = [&](uint32_t w, double a, double bsin, double v, double &t1, double &t2) {
        if (POL == 1) return;
        double cc, ss;
        if (HALF) {
            const double h2 = a * a, inv = 1.0 / (1.0 + h2);
            cc = (1.0 - h2) * inv;
            ss = (a + a) * inv;
            if (w & 0x8000u) cc = -cc;
        } else {
            cc = a;
            ss = bsin;
        }
        t1 = v * cc;
        t2 = v * ss;
    };
    // tile[pixel] += term with the LDS adder (ds_add_f64, nothing returned): the same IEEE addition
    // the serial loop performs on its accumulator, and the adds one thread issues to one address
    // are performed in program order -- so a run is summed term after term without a register
    // round trip.  Within a pass no two threads touch the same pixel.
    auto tile_add = [&](int q, double v, double t1, double t2) {
        if (POL == 1) {
            atomicAdd(&tile[q], v);
        } else if (POL == 2) {
            atomicAdd(&tile[2 * q], t1);
            atomicAdd(&tile[2 * q + 1], t2);
        } else {
            atomicAdd(&tile[3 * q], v);
            atomicAdd(&tile[3 * q + 1], t1);
            atomicAdd(&tile[3 * q + 2], t2);
        }
    };
    // one group: its entries added to the tile accumulators in list (= time) order
    auto reduce_group = [&](const uint32_t (&w)[4], const double (&v)[4], const double (&t1)[4],
                            const double (&t2)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (w[m] != kFxNull) tile_add((int)(w[m] & QM), v[m], t1[m], t2[m]);
    };

    if (nsl > 0) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) fetch(dd, dd);
    }
    for (int jj = 0; jj < nsl; jj += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const int j = jj + dd;
            if (j >= nsl) break;
            double *vbuf = dd ? vbuf1 : vbuf0;               // (kFxDepth = 2: slot dd = j & 1)
            // ---- stage the slice's TOD values (into the buffer slice j - 2 used: every wave left
            //      that slice before it arrived at slice j - 1's barrier; j = 0: the barrier below
            //      also covers the zeroing of the tile) ----
#pragma unroll
            for (int u = 0; u < VPT; ++u) vbuf[tid + u * kFxT] = pv[dd][u];
            const uint2 m0 = pm[dd][0], m1 = pm[dd][1];
            uint32_t w[4] = {pe[dd].x, pe[dd].y, pe[dd].z, pe[dd].w};
            double a[4] = {0.0, 0.0, 0.0, 0.0}, bs[4] = {0.0, 0.0, 0.0, 0.0};
            if (ANG) {
                a[0] = pa[dd][0].x; a[1] = pa[dd][0].y; a[2] = pa[dd][1].x; a[3] = pa[dd][1].y;
            }
            if (TWO) {
                bs[0] = pb[dd][0].x; bs[1] = pb[dd][0].y; bs[2] = pb[dd][1].x; bs[3] = pb[dd][1].y;
            }
            __syncthreads();
            fetch(dd, j + D);
            const uint32_t G = m1.x - m0.x, ntail = (m1.y & 0x0FFFFFFFu) - (m0.y & 0x0FFFFFFFu);
            const int maxlevel = (int)(m0.y >> 28);      // highest level in this slice (plan time)
            for (uint32_t g0 = 0; g0 < G || g0 == 0; g0 += kFxT) {
                const bool mine = g0 + tid < G;
                if (g0 > 0) {                             // more groups than threads: direct loads
                    const int64_t g = (int64_t)m0.x + (mine ? g0 + tid : 0);
                    const uint4 e = gent[g];
                    w[0] = e.x; w[1] = e.y; w[2] = e.z; w[3] = e.w;
                    if (ANG) {
                        const double2 x0 = ga[2 * g], x1 = ga[2 * g + 1];
                        a[0] = x0.x; a[1] = x0.y; a[2] = x1.x; a[3] = x1.y;
                    }
                    if (TWO) {
                        const double2 x0 = gb[2 * g], x1 = gb[2 * g + 1];
                        bs[0] = x0.x; bs[1] = x0.y; bs[2] = x1.x; bs[3] = x1.y;
                    }
                }
                if (!mine) w[0] = w[1] = w[2] = w[3] = kFxNull;
                double v[4], t1[4], t2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t off = (w[m] >> 16) & 0xFFFu;
                    v[m] = vbuf[w[m] != kFxNull ? off : 0];
                    t1[m] = t2[m] = 0.0;
                    terms(w[m], a[m], bs[m], v[m], t1[m], t2[m]);
                }
                const int level = mine ? (int)((w[0] >> 28) & 15u) : 0;
                if (g0 == 0 && ntail > 0) {
                    const int64_t tr0 = (int64_t)(m0.y & 0x0FFFFFFFu);
                    // runs too long for the groups: one thread walks a whole run ...
                    for (uint32_t r = tid; r < ntail; r += kFxT) {
                        const uint2 r0 = trun[tr0 + r], r1 = trun[tr0 + r + 1];
                        if (r1.x - r0.x > chunk_min) continue;
                        const int q = (int)r0.y;
                        for (uint32_t e = r0.x; e < r1.x; ++e) {
                            const uint32_t we = tent[e];
                            const double ve = vbuf[(we >> 16) & 0xFFFu];
                            double u1 = 0.0, u2 = 0.0;
                            terms(we, ANG ? ta[e] : 0.0, TWO ? tb[e] : 0.0, ve, u1, u2);
                            tile_add(q, ve, u1, u2);
                        }
                    }
                    // ... unless it is a hot run: chunk sums by one thread per chunk (the walk over
                    // the runs is the same for every thread: trun is read uniformly), then the
                    // chunk sums of a run added in time order by the thread of its first chunk
                    int cbase = 0, my_first = -1, my_n = 0, my_q = 0;
                    for (uint32_t r = 0; r < ntail; ++r) {
                        const uint2 r0 = trun[tr0 + r], r1 = trun[tr0 + r + 1];
                        const uint32_t len = r1.x - r0.x;
                        if (len <= chunk_min) continue;
                        const int nch = (int)((len + kFxChunk - 1) / kFxChunk);
                        const int c = tid - cbase;
                        if (c >= 0 && c < nch && tid < kFxMaxChunks) {
                            const uint32_t e0 = r0.x + (uint32_t)c * kFxChunk;
                            const uint32_t e1 = e0 + kFxChunk < r1.x ? e0 + kFxChunk : r1.x;
                            double sv = 0.0, s1 = 0.0, s2 = 0.0;
                            for (uint32_t e = e0; e < e1; ++e) {
                                const uint32_t we = tent[e];
                                const double ve = vbuf[(we >> 16) & 0xFFFu];
                                double u1 = 0.0, u2 = 0.0;
                                terms(we, ANG ? ta[e] : 0.0, TWO ? tb[e] : 0.0, ve, u1, u2);
                                sv += ve;
                                s1 += u1;
                                s2 += u2;
                            }
                            part[tid] = sv;
                            part[kFxMaxChunks + tid] = s1;
                            part[2 * kFxMaxChunks + tid] = s2;
                            if (c == 0) {
                                my_first = tid;
                                my_n = nch;
                                my_q = (int)r0.y;
                            }
                        }
                        cbase += nch;
                    }
                    if (cbase > 0) {                      // (uniform: some run of this slice is hot)
                        __syncthreads();
                        if (my_first >= 0) {
                            double sv = part[my_first], s1 = part[kFxMaxChunks + my_first],
                                   s2 = part[2 * kFxMaxChunks + my_first];
                            for (int k = 1; k < my_n; ++k) {
                                sv += part[my_first + k];
                                s1 += part[kFxMaxChunks + my_first + k];
                                s2 += part[2 * kFxMaxChunks + my_first + k];
                            }
                            tile_add(my_q, sv, s1, s2);
                        }
                    }
                }
                // one pass per level: piece p of a long run is added after piece p - 1.  The pieces
                // of a run are groups of one wave (k_fx_pack), and the LDS executes a wave's
                // instructions in issue order, so the passes need no barrier between them; the
                // next slice's staging barrier separates this slice's adds from the next slice's.
                for (int p = 0; p <= maxlevel; ++p)
                    if (mine && level == p) reduce_group(w, v, t1, t2);
                if (g0 + kFxT < G) __syncthreads();          // (a further round of groups of this slice)
            }
        }
    }
    __syncthreads();
    double *o = o_part ? o_part : out + p0 * POL;
    for (int i = tid; i < nvals; i += kFxT) o[i] = tile[i];
}

// the copies of a split tile added in time order (part after part), written to the map
__global__ __launch_bounds__(256) void k_parts_combine(const int64_t *__restrict__ multi, int64_t m0,
                                                        int64_t stride, const double *__restrict__ scratch,
                                                        double *__restrict__ out)
{
    const int64_t *m = multi + 4 * (m0 + blockIdx.x);
    const int64_t o = m[0], nvals = m[1], slot = m[2], k = m[3];
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;       // one value per thread: the launch is short
    if (i >= nvals) return;
    double acc = scratch[slot * stride + i];
    for (int64_t j = 1; j < k; ++j) acc += scratch[(slot + j) * stride + i];
    out[o + i] = acc;
}

// ------------------------------------------------------------------- hot tiles ------
// A tile that is one pixel with very many samples: its bucket is cut into ranges of kHotChunk
// consecutive samples; one workgroup per range, thread t adding the terms at positions t, t + 1024,
// ... of the range in that order, the 1024 thread sums combined by a fixed halving tree; k_hot_combine
// then adds the range sums of a tile in time order and writes the pixel.  Every boundary and every
// order depends on the bucket's length only: reproducible bit for bit, independent of the rest of
// the hit map; a regrouping of the serial sum, ~1e-16 relative per level away from it.
constexpr int64_t kHotMin = kHotTileMin;                // samples that make a one-pixel tile hot
static_assert(kHotMin == 2 * kHotChunk, "hot tiles: at least two ranges");

template <int POL, bool HALF>
__global__ __launch_bounds__(kHotT) void k_Pt_hot(const int64_t *__restrict__ range, int64_t c0,
                                                   const uint16_t *__restrict__ pl,
                                                   const double *__restrict__ a_tb,
                                                   const double *__restrict__ b_tb,
                                                   const double *__restrict__ v_tb,
                                                   double *__restrict__ partial)
{
    __shared__ double red[3][kHotT];
    const int64_t c = c0 + blockIdx.x;
    const int64_t k0 = range[2 * c], k1 = range[2 * c + 1];
    const int t = threadIdx.x;
    double sv = 0.0, s1 = 0.0, s2 = 0.0;
    // (four positions' loads in flight at a time; the terms are added in the order of the positions)
    constexpr int U = 4;
    for (int64_t kk = k0 + t; kk < k1; kk += (int64_t)U * kHotT) {
        double v[U], a[U], b2[U];
        uint16_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = kk + (int64_t)u * kHotT;
            const int64_t kc = k < k1 ? k : k1 - 1;
            v[u] = v_tb[kc];
            a[u] = POL > 1 ? a_tb[kc] : 0.0;
            b2[u] = (POL > 1 && !HALF) ? b_tb[kc] : 0.0;
            w[u] = (POL > 1 && HALF) ? pl[kc] : (uint16_t)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kk + (int64_t)u * kHotT >= k1) break;
            sv += v[u];
            if (POL > 1) {
                double cc, ss;
                if (HALF) {
                    const double h = a[u], h2 = h * h, inv = 1.0 / (1.0 + h2);
                    cc = (1.0 - h2) * inv;
                    ss = (h + h) * inv;
                    if (w[u] & 0x8000u) cc = -cc;
                } else {
                    cc = a[u];
                    ss = b2[u];
                }
                s1 += v[u] * cc;
                s2 += v[u] * ss;
            }
        }
    }
    red[0][t] = sv;
    red[1][t] = s1;
    red[2][t] = s2;
    __syncthreads();
    for (int h = kHotT / 2; h >= 1; h >>= 1) {
        if (t < h) {
            red[0][t] += red[0][t + h];
            red[1][t] += red[1][t + h];
            red[2][t] += red[2][t + h];
        }
        __syncthreads();
    }
    if (t == 0) {
        partial[3 * c] = red[0][0];
        partial[3 * c + 1] = red[1][0];
        partial[3 * c + 2] = red[2][0];
    }
}

// One workgroup per hot tile: the range sums are staged in LDS 256 ranges at a time (coalesced loads: one
// thread walking them in HBM paid a memory round trip every few terms -- 305 ranges for 5e6 samples), and
// thread 0 adds them in time order.
template <int POL>
__global__ __launch_bounds__(256) void k_hot_combine(const int64_t *__restrict__ tiles, int64_t h0,
                                                      const double *__restrict__ partial,
                                                      double *__restrict__ out)
{
    __shared__ double st[3 * 256];
    const int64_t h = h0 + blockIdx.x;
    const int64_t p0 = tiles[3 * h], c0 = tiles[3 * h + 1], nc = tiles[3 * h + 2];
    double sv = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t base = 0; base < nc; base += 256) {
        const int64_t n = nc - base < 256 ? nc - base : 256;
        for (int64_t i = threadIdx.x; i < 3 * n; i += 256) st[i] = partial[3 * (c0 + base) + i];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int64_t c = 0; c < n; ++c) {               // range sums in time order
                sv += st[3 * c];
                s1 += st[3 * c + 1];
                s2 += st[3 * c + 2];
            }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    if (POL == 1) {
        out[p0] = sv;
    } else if (POL == 2) {
        out[2 * p0] = s1;
        out[2 * p0 + 1] = s2;
    } else {
        out[3 * p0] = sv;
        out[3 * p0 + 1] = s1;
        out[3 * p0 + 2] = s2;
    }
}

// ------------------------------------------------------------------- plan -----------
// keys of the per-slice sort: (global slice number << 16) | pixel in tile; value = list entry
__global__ __launch_bounds__(256) void k_fx_keys(int64_t nvalid, int64_t ntiles, int S, uint32_t qmask,
                                                  const int64_t *__restrict__ tile_off,
                                                  const int64_t *__restrict__ tile_slice0,
                                                  const uint16_t *__restrict__ pl,
                                                  uint64_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nvalid; k += stride) {
        int64_t lo = 0, hi = ntiles;                      // largest b with tile_off[b] <= k
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (tile_off[mid] <= k) lo = mid; else hi = mid;
        }
        const int64_t r = k - tile_off[lo];
        const uint32_t w = pl[k];
        keys[k] = ((uint64_t)(tile_slice0[lo] + r / S) << 16) | (uint64_t)(w & qmask);
        vals[k] = w | ((uint32_t)(r % S) << 16);
    }
}

// One thread per slice walks the slice's sorted entries and packs the runs into groups.
// WRITE = false: counts[4 s + {0, 1, 2, 3}] = groups, tail runs, tail entries, highest level.
// WRITE = true: the groups / tail lists are written at the offsets of the slice.
// NANG = angle arrays to carry along: 0 (pol = 1), 1 (half angle), 2 (cos and sin); a compile-time
// switch, because a run-time "if (ga)" does not keep the compiler from issuing the a_tb load.
template <bool WRITE, int NANG>
__global__ __launch_bounds__(64) void k_fx_pack(
    int64_t nslices, uint32_t qmask, const int64_t *__restrict__ slice_k0,
    const uint32_t *__restrict__ ent, const double *__restrict__ a_tb,
    const double *__restrict__ b_tb, uint32_t *__restrict__ counts,
    const uint2 *__restrict__ meta, const uint32_t *__restrict__ tent_off,
    uint32_t *__restrict__ gent, double *__restrict__ ga, double *__restrict__ gb,
    uint2 *__restrict__ trun, uint32_t *__restrict__ tent, double *__restrict__ ta,
    double *__restrict__ tb)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    const int64_t k0 = slice_k0[s];
    const int len = (int)(slice_k0[s + 1] - k0);
    int64_t g = WRITE ? (int64_t)meta[s].x : 0;
    uint32_t ntr = 0, nte = 0, ng = 0, maxlev = 0;
    int fill = 0;
    auto put = [&](int slot, uint32_t w, uint32_t level) {
        if (!WRITE) return;
        const int64_t at = 4 * g + slot;
        gent[at] = w | (level << 28);
        const int64_t src = k0 + (int64_t)((w >> 16) & 0xFFFu);
        if (NANG >= 1) ga[at] = a_tb[src];
        if (NANG == 2) gb[at] = b_tb[src];
    };
    auto close = [&]() {
        if (WRITE)
            for (int slot = fill; slot < 4; ++slot) {
                gent[4 * g + slot] = kFxNull;
                if (NANG >= 1) ga[4 * g + slot] = 0.0;
                if (NANG == 2) gb[4 * g + slot] = 0.0;
            }
        ++g;
        ++ng;
        fill = 0;
    };
    int i = 0;
    while (i < len) {
        const uint32_t q = ent[k0 + i] & qmask;
        int L = 1;
        while (i + L < len && (ent[k0 + i + L] & qmask) == q) ++L;
        if (L > 4 * (kFxMaxLevel + 1)) {
            if (WRITE) {
                const uint32_t e0 = tent_off[s] + nte;
                trun[(int64_t)(meta[s].y & 0x0FFFFFFFu) + ntr] = make_uint2(e0, q);
                for (int m = 0; m < L; ++m) {
                    const uint32_t w = ent[k0 + i + m];
                    tent[e0 + m] = w;
                    const int64_t src = k0 + (int64_t)((w >> 16) & 0xFFFu);
                    if (NANG >= 1) ta[e0 + m] = a_tb[src];
                    if (NANG == 2) tb[e0 + m] = b_tb[src];
                }
            }
            ++ntr;
            nte += (uint32_t)L;
        } else if (L <= 4) {
            if (fill + L > 4) close();
            for (int m = 0; m < L; ++m) put(fill + m, ent[k0 + i + m], 0);
            fill += L;
            if (fill == 4) close();
        } else {
            if (fill > 0) close();
            if ((uint32_t)((L - 1) / 4) > maxlev) maxlev = (uint32_t)((L - 1) / 4);
            // all pieces of a run inside ONE wave (64 consecutive groups of the slice): the kernel
            // orders the pieces by the program order of that wave's LDS adds, not by barriers
            while ((int)(ng % 64u) + (L + 3) / 4 > 64) close();
            for (int m = 0; m < L; ++m) {
                put(fill, ent[k0 + i + m], (uint32_t)(m / 4));
                if (++fill == 4) close();
            }
            if (fill > 0) close();
        }
        i += L;
    }
    if (fill > 0) close();
    if (!WRITE) {
        counts[4 * s] = ng;
        counts[4 * s + 1] = ntr;
        counts[4 * s + 2] = nte;
        counts[4 * s + 3] = maxlev;
    }
}

// ---- the same lists built by one workgroup per slice ------------------------------------------------
// k_fx_keys + a global radix sort + one THREAD per slice walking ~1500 sorted entries (k_fx_pack) cost
// 14 ms at C4.  k_fx_build does the whole slice in LDS: a bitonic sort of (pixel, position) keys, the
// runs from a flag scan, and a packing that needs no walk: runs are placed by CLASS with ranks from a
// scan --
//   runs of 5 .. 60 entries ("long") first, each in ceil(L / 4) consecutive groups with levels 0, 1, ..;
//     rows of 64 groups (= one wave of the P^T kernel): with R = 64 - (groups of the slice's longest run)
//     + 1, run i with u_i = groups of the long runs before it goes to row u_i / R at offset u_i - (u of
//     the row's first run) <= R - 1, so it ends inside the row;
//   then the runs of 4, the runs of 3 (slot 3 takes a single), the runs of 2 in pairs (an odd one out
//     takes two singles), the remaining singles four to a group.
// What the P^T kernel needs holds as before: a run's entries are in time order, the pieces of a long
// run are consecutive groups of one wave, a pixel appears in one run per slice.  The sums per pixel
// are the same sums in the same order as with k_fx_pack's lists; only the packing differs (a few
// per cent fewer groups: k_fx_pack closes a group when the next run does not fit).
// Pass 1 (WRITE = false) sorts, stores the sorted keys in ent and counts; pass 2 reads ent and writes.
constexpr int kFbT = 256, kFbMaxS = 4 * kFxT, kFbPer = kFbMaxS / kFbT;
constexpr int kFbMaxGroups = 1280;       // 2048 entries: <= 0.4 groups an entry (runs of 5) x 64 / 50

__device__ __forceinline__ uint64_t fb_exscan(uint64_t v, uint64_t *tmp, uint64_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up((unsigned long long)inc, d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint64_t before = inc - v;
    total = 0;
#pragma unroll
    for (int w = 0; w < kFbT / 64; ++w) {
        if (w < wave) before += tmp[w];
        total += tmp[w];
    }
    __syncthreads();
    return before;
}

template <bool WRITE, int NANG>
__global__ __launch_bounds__(kFbT) void k_fx_build(
    int64_t nslices, uint32_t qmask, const int64_t *__restrict__ slice_k0, int k0_stride,
    const uint16_t *__restrict__ pl, uint32_t *__restrict__ ent, const double *__restrict__ a_tb,
    const double *__restrict__ b_tb, uint32_t *__restrict__ counts, const uint2 *__restrict__ meta,
    const uint32_t *__restrict__ tent_off, uint32_t *__restrict__ gent, double *__restrict__ ga,
    double *__restrict__ gb, uint2 *__restrict__ trun, uint32_t *__restrict__ tent,
    double *__restrict__ ta, double *__restrict__ tb, unsigned int *__restrict__ overflow)
{
    __shared__ uint32_t keys[kFbMaxS];
    __shared__ uint16_t rs[kFbMaxS + 1];
    __shared__ uint64_t tmp[kFbT / 64];
    __shared__ uint32_t rowfirst[64];
    __shared__ uint32_t misc[2];
    __shared__ uint32_t tails[2 * (kFbMaxS / (4 * (kFxMaxLevel + 1)) + 2)];   // (first sorted entry, first tail entry) per tail run
    __shared__ uint32_t stage[WRITE ? 4 * kFbMaxGroups : 4];
    const int64_t s = blockIdx.x;
    if (s >= nslices) return;
    const int t = threadIdx.x;
    // (k0_stride = 1: slice s = [slice_k0[s], slice_k0[s + 1]); 2: a list of (first, end) pairs)
    const int64_t k0 = slice_k0[s * k0_stride];
    const int len = (int)(slice_k0[s * k0_stride + 1] - k0);
    if (!WRITE) {
        int NS = 64;
        while (NS < len) NS <<= 1;
        for (int i = t; i < NS; i += kFbT) {
            uint32_t key = 0xFFFFFFFFu;
            if (i < len) {
                const uint32_t w = pl[k0 + i];
                key = ((w & qmask) << 12) | ((uint32_t)i << 1) | ((w & ~qmask & 0xFFFFu) ? 1u : 0u);
            }
            keys[i] = key;
        }
        __syncthreads();
        for (int k = 2; k <= NS; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < NS / 2; i += kFbT) {
                    const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                    const uint32_t a = keys[lo], b = keys[hi];
                    const bool asc = (lo & k) == 0;
                    if ((a > b) == asc) {
                        keys[lo] = b;
                        keys[hi] = a;
                    }
                }
                __syncthreads();
            }
        for (int i = t; i < len; i += kFbT) ent[k0 + i] = keys[i];
    } else {
        for (int i = t; i < len; i += kFbT) keys[i] = ent[k0 + i];
        __syncthreads();
    }
    // ---- runs: rs[r] = first sorted entry of run r ----
    int nst = 0;
    bool st[kFbPer];
#pragma unroll
    for (int u = 0; u < kFbPer; ++u) {
        const int j = kFbPer * t + u;
        st[u] = j < len && (j == 0 || (keys[j] >> 12) != (keys[j - 1] >> 12));
        nst += st[u] ? 1 : 0;
    }
    uint64_t tot = 0;
    int r0 = (int)fb_exscan((uint64_t)nst, tmp, tot);
    const int nruns = (int)tot;
#pragma unroll
    for (int u = 0; u < kFbPer; ++u)
        if (st[u]) rs[r0++] = (uint16_t)(kFbPer * t + u);
    if (t == 0) {
        rs[nruns] = (uint16_t)len;
        misc[0] = 0;
        misc[1] = 0;
    }
    if (t < 64) rowfirst[t] = 0xFFFFFFFFu;
    __syncthreads();
    // ---- classes and ranks: A = singles | pairs << 12 | triples << 24 | fours << 36,
    //      B = long groups | tail runs << 12 | tail entries << 24 ----
    uint64_t sumA = 0, sumB = 0;
    int L[kFbPer];
#pragma unroll
    for (int u = 0; u < kFbPer; ++u) {
        const int r = kFbPer * t + u;
        L[u] = r < nruns ? (int)rs[r + 1] - (int)rs[r] : 0;
        if (L[u] == 0) continue;
        if (L[u] <= 4) sumA += (uint64_t)1 << (12 * (L[u] - 1));
        else if (L[u] <= 4 * (kFxMaxLevel + 1)) sumB += (uint64_t)((L[u] + 3) / 4);
        else sumB += ((uint64_t)1 << 12) | ((uint64_t)L[u] << 24);
    }
    uint64_t totA = 0, totB = 0;
    uint64_t exA = fb_exscan(sumA, tmp, totA);
    uint64_t exB = fb_exscan(sumB, tmp, totB);
    const int n1 = (int)(totA & 0xFFF), n2 = (int)((totA >> 12) & 0xFFF), n3 = (int)((totA >> 24) & 0xFFF),
              n4 = (int)((totA >> 36) & 0xFFF);
    const int ntr = (int)((totB >> 12) & 0xFFF), nte = (int)(totB >> 24);
    // rows of the long runs: the row length leaves room for the slice's longest run
#pragma unroll
    for (int u = 0; u < kFbPer; ++u)
        if (L[u] > 4 && L[u] <= 4 * (kFxMaxLevel + 1)) atomicMax(&misc[1], (uint32_t)((L[u] - 1) / 4));
    __syncthreads();
    const uint32_t row_len = 64u - misc[1];              // (longest run: misc[1] + 1 groups)
    {
        uint64_t b = exB;
#pragma unroll
        for (int u = 0; u < kFbPer; ++u) {
            if (L[u] > 4 && L[u] <= 4 * (kFxMaxLevel + 1)) {
                const uint32_t uu = (uint32_t)(b & 0xFFF);
                atomicMin(&rowfirst[uu / row_len], uu);
                b += (uint64_t)((L[u] + 3) / 4);
            } else if (L[u] > 4 * (kFxMaxLevel + 1)) {
                b += ((uint64_t)1 << 12) | ((uint64_t)L[u] << 24);
            }
        }
    }
    __syncthreads();
    int pos[kFbPer];
    {
        uint64_t b = exB;
#pragma unroll
        for (int u = 0; u < kFbPer; ++u) {
            pos[u] = 0;
            if (L[u] > 4 && L[u] <= 4 * (kFxMaxLevel + 1)) {
                const uint32_t uu = (uint32_t)(b & 0xFFF), row = uu / row_len;
                pos[u] = (int)(64 * row + uu - rowfirst[row]);
                atomicMax(&misc[0], (uint32_t)(pos[u] + (L[u] + 3) / 4));
                b += (uint64_t)((L[u] + 3) / 4);
            } else if (L[u] > 4 * (kFxMaxLevel + 1)) {
                b += ((uint64_t)1 << 12) | ((uint64_t)L[u] << 24);
            }
        }
    }
    __syncthreads();
    const int GL = (int)misc[0], maxlev = (int)misc[1];
    const int odd2 = n2 & 1;
    const int s1 = n1 > n3 ? n1 - n3 : 0;
    const int x2 = odd2 ? (s1 < 2 ? s1 : 2) : 0;
    const int ng = GL + n4 + n3 + (n2 + 1) / 2 + (s1 - x2 + 3) / 4;
    if (!WRITE) {
        if (t == 0) {
            counts[4 * s] = (uint32_t)ng;
            counts[4 * s + 1] = (uint32_t)ntr;
            counts[4 * s + 2] = (uint32_t)nte;
            counts[4 * s + 3] = (uint32_t)maxlev;
        }
        return;
    }
    if (ng > kFbMaxGroups) {                              // (cannot happen for S <= 2048; never write past the stage)
        if (t == 0) atomicOr(overflow, 1u);
        return;
    }
    for (int i = t; i < 4 * ng; i += kFbT) stage[i] = kFxNull;
    __syncthreads();
    const int64_t g_base = (int64_t)meta[s].x;
    const int64_t tr_base = (int64_t)(meta[s].y & 0x0FFFFFFFu);
    const uint32_t te_base = tent_off[s];
    {
        uint64_t a = exA, b = exB;
        const int G4 = GL, G3 = GL + n4, G2 = G3 + n3, G1 = G2 + (n2 + 1) / 2;
#pragma unroll
        for (int u = 0; u < kFbPer; ++u) {
            if (L[u] == 0) continue;
            const int j0 = (int)rs[kFbPer * t + u];
            auto value = [&](int m) {
                const uint32_t key = keys[j0 + m];
                return (key >> 12) | ((key & 1u) << 15) | (((key >> 1) & 0x7FFu) << 16);
            };
            if (L[u] == 1) {
                const int sr = (int)(a & 0xFFF);
                int g, slot;
                if (sr < n3) {
                    g = G3 + sr;
                    slot = 3;
                } else if (sr - n3 < x2) {
                    g = G2 + n2 / 2;
                    slot = 2 + (sr - n3);
                } else {
                    const int q = sr - n3 - x2;
                    g = G1 + q / 4;
                    slot = q % 4;
                }
                stage[4 * g + slot] = value(0);
                a += 1;
            } else if (L[u] == 2) {
                const int r2 = (int)((a >> 12) & 0xFFF);
                const int g = G2 + r2 / 2, slot = (r2 & 1) * 2;
                stage[4 * g + slot] = value(0);
                stage[4 * g + slot + 1] = value(1);
                a += (uint64_t)1 << 12;
            } else if (L[u] == 3) {
                const int g = G3 + (int)((a >> 24) & 0xFFF);
                for (int m = 0; m < 3; ++m) stage[4 * g + m] = value(m);
                a += (uint64_t)1 << 24;
            } else if (L[u] == 4) {
                const int g = G4 + (int)((a >> 36) & 0xFFF);
                for (int m = 0; m < 4; ++m) stage[4 * g + m] = value(m);
                a += (uint64_t)1 << 36;
            } else if (L[u] <= 4 * (kFxMaxLevel + 1)) {
                for (int m = 0; m < L[u]; ++m)
                    stage[4 * pos[u] + m] = value(m) | ((uint32_t)(m / 4) << 28);
                b += (uint64_t)((L[u] + 3) / 4);
            } else {
                const int tr = (int)((b >> 12) & 0xFFF);
                const uint32_t e0 = te_base + (uint32_t)(b >> 24);
                trun[tr_base + tr] = make_uint2(e0, keys[j0] >> 12);
                tails[2 * tr] = (uint32_t)j0 | ((uint32_t)L[u] << 16);
                tails[2 * tr + 1] = e0;
                b += ((uint64_t)1 << 12) | ((uint64_t)L[u] << 24);
            }
        }
    }
    __syncthreads();
    // the runs kept out of the groups: their entries, in time order
    for (int tr = 0; tr < ntr; ++tr) {
        const int j0 = (int)(tails[2 * tr] & 0xFFFFu), Lr = (int)(tails[2 * tr] >> 16);
        const uint32_t e0 = tails[2 * tr + 1];
        for (int m = t; m < Lr; m += kFbT) {
            const uint32_t key = keys[j0 + m];
            const uint32_t w = (key >> 12) | ((key & 1u) << 15) | (((key >> 1) & 0x7FFu) << 16);
            tent[e0 + m] = w;
            const int64_t src = k0 + (int64_t)((w >> 16) & 0xFFFu);
            if (NANG >= 1) ta[e0 + m] = a_tb[src];
            if (NANG == 2) tb[e0 + m] = b_tb[src];
        }
    }
    for (int i = t; i < 4 * ng; i += kFbT) {
        const uint32_t w = stage[i];
        gent[4 * g_base + i] = w;
        const int64_t src = k0 + (int64_t)((w >> 16) & 0xFFFu);
        if (NANG >= 1) ga[4 * g_base + i] = w == kFxNull ? 0.0 : a_tb[src];
        if (NANG == 2) gb[4 * g_base + i] = w == kFxNull ? 0.0 : b_tb[src];
    }
}

size_t fx_lds_bytes(const cm2_tiles *t, int S)
{
    int vpt = (S + kFxT - 1) / kFxT;
    vpt = vpt <= 2 ? 2 : vpt;
    return sizeof(double) * ((size_t)t->tp * t->pol + 2 * (size_t)vpt * kFxT + 3 * (size_t)kFxMaxChunks);
}

void hot_release(cm2_tiles *t)
{
    void **ptrs[] = {(void **)&t->d_hot_flag, (void **)&t->d_hot_range, (void **)&t->d_hot_tiles,
                     (void **)&t->d_hot_partial, (void **)&t->d_hot_range_tile};
    for (void **q : ptrs) {
        if (*q) (void)cm2::dev_free(*q);
        *q = nullptr;
    }
    t->hot_tile.clear();
    t->hot_chunk0.clear();
}

// one-pixel tiles with at least kHotMin samples, their sample ranges and the scratch of range sums
int hot_plan(cm2_tiles *t, hipStream_t st)
{
    hot_release(t);
    std::vector<uint8_t> flag((size_t)t->ntiles, 0);
    std::vector<int64_t> range, tiles;
    std::vector<int> range_tile;                             // (fused form: which hot tile a range belongs to)
    t->hot_chunk0.assign(1, 0);
    for (int64_t b = 0; b < t->ntiles; ++b) {
        const int64_t n = t->tile_count[(size_t)b];
        if (t->tile_p0[(size_t)b + 1] - t->tile_p0[(size_t)b] != 1 || n < kHotMin) continue;
        flag[(size_t)b] = 1;
        const int64_t c0 = (int64_t)range.size() / 2;
        // (ranges in time order: span after span, kHotChunk consecutive samples of a segment each)
        for (int64_t sp = 0; sp < t->nspans; ++sp) {
            const int64_t a0 = t->seg_off[(size_t)(sp * t->ntiles + b)], a1 = t->seg_off[(size_t)(sp * t->ntiles + b + 1)];
            for (int64_t k = a0; k < a1; k += kHotChunk) {
                range.push_back(k);
                range.push_back(k + kHotChunk < a1 ? k + kHotChunk : a1);
            }
        }
        tiles.push_back(t->tile_p0[(size_t)b]);
        tiles.push_back(c0);
        tiles.push_back((int64_t)range.size() / 2 - c0);
        range_tile.resize(range.size() / 2, (int)t->hot_tile.size());
        t->hot_tile.push_back(b);
        t->hot_chunk0.push_back((int64_t)range.size() / 2);
    }
    if (t->hot_tile.empty()) return 0;
    CM2_HIP(cm2::dev_malloc(&t->d_hot_flag, flag.size()));
    CM2_HIP(cm2::dev_malloc(&t->d_hot_range, sizeof(int64_t) * range.size()));
    CM2_HIP(cm2::dev_malloc(&t->d_hot_tiles, sizeof(int64_t) * tiles.size()));
    CM2_HIP(cm2::dev_malloc(&t->d_hot_partial, sizeof(double) * 3 * (range.size() / 2)));
    CM2_HIP(cm2::upload(t->d_hot_flag, flag.data(), flag.size(), st));
    CM2_HIP(cm2::upload(t->d_hot_range, range.data(), sizeof(int64_t) * range.size(), st));
    CM2_HIP(cm2::upload(t->d_hot_tiles, tiles.data(), sizeof(int64_t) * tiles.size(), st));
    CM2_HIP(cm2::dev_malloc(&t->d_hot_range_tile, sizeof(int) * range_tile.size()));
    CM2_HIP(cm2::upload(t->d_hot_range_tile, range_tile.data(), sizeof(int) * range_tile.size(), st));
    CM2_HIP(hipStreamSynchronize(st));
    return 0;
}

template <int POL, bool HALF>
int hot_launch(const cm2_tiles *t, const double *d_tod_tb, double *d_out, int64_t tile_lo,
               int64_t tile_hi, hipStream_t stream)
{
    // hot tiles inside [tile_lo, tile_hi): they are listed in ascending tile order
    int64_t h0 = 0, h1 = (int64_t)t->hot_tile.size();
    while (h0 < h1 && t->hot_tile[(size_t)h0] < tile_lo) ++h0;
    while (h1 > h0 && t->hot_tile[(size_t)h1 - 1] >= tile_hi) --h1;
    if (h1 <= h0) return 0;
    const int64_t c0 = t->hot_chunk0[(size_t)h0], c1 = t->hot_chunk0[(size_t)h1];
    k_Pt_hot<POL, HALF><<<(unsigned)(c1 - c0), kHotT, 0, stream>>>(
        t->d_hot_range, c0, t->d_pl, HALF ? t->d_half : t->d_cos, t->d_sin, d_tod_tb, t->d_hot_partial);
    CM2_LAUNCH_OK();
    k_hot_combine<POL><<<(unsigned)(h1 - h0), 256, 0, stream>>>(t->d_hot_tiles, h0, t->d_hot_partial, d_out);
    CM2_LAUNCH_OK();
    return 0;
}

void fx_release(cm2_tiles *t)
{
    hot_release(t);
    void **ptrs[] = {(void **)&t->d_fx_slice0, (void **)&t->d_fx_sk, (void **)&t->d_fx_meta, (void **)&t->d_fx_gent,
                     (void **)&t->d_fx_ga, (void **)&t->d_fx_gb, (void **)&t->d_fx_trun,
                     (void **)&t->d_fx_tent, (void **)&t->d_fx_ta, (void **)&t->d_fx_tb,
                     (void **)&t->d_parts, (void **)&t->d_multi,
                     (void **)&t->d_part_buf, (void **)&t->d_fx_fused, (void **)&t->d_fx_count};
    for (void **q : ptrs) {
        if (*q) (void)cm2::dev_free(*q);
        *q = nullptr;
    }
    t->tile_part0.clear();
    t->multi_tile.clear();
    t->nparts = t->part_slots = 0;
    t->part_makespan = 0.0;
    t->fx_S = 0;
    t->fx_ngroups = t->fx_nslices = 0;
}

// a tile that k_Pt_hot takes over (hot_plan): its slices do not count when the slice length is tuned
static bool fx_hot_tile(const cm2_tiles *t, int64_t b)
{
    return t->tile_p0[(size_t)b + 1] - t->tile_p0[(size_t)b] == 1 && t->tile_count[(size_t)b] >= kHotMin;
}

// The slices of the plan for the slice length S, as (first address, end) pairs in the order the
// kernel walks them: tile after tile, a tile's segments span after span (= in time), a segment cut
// where it is longer than S -- into pieces of S with a shorter last one when the plan has one span
// (the tile's whole bucket is one segment: rounds 1-3), into equal pieces otherwise (a segment is
// about one slice long by the choice of the span: cutting 2100 samples into 1856 + 244 would cost a
// whole barrier round for the short piece).  slice0[b] = first slice of tile b.
static void fx_slices(const cm2_tiles *t, int S, std::vector<int64_t> &slice0, std::vector<int64_t> &pairs)
{
    slice0.assign((size_t)t->ntiles + 1, 0);
    pairs.clear();
    for (int64_t b = 0; b < t->ntiles; ++b) {
        slice0[(size_t)b] = (int64_t)pairs.size() / 2;
        for (int64_t sp = 0; sp < t->nspans; ++sp) {
            const int64_t a0 = t->seg_off[(size_t)(sp * t->ntiles + b)], a1 = t->seg_off[(size_t)(sp * t->ntiles + b + 1)];
            if (a1 <= a0) continue;
            int64_t piece = S;
            if (t->nspans > 1) {
                const int64_t np = (a1 - a0 + S - 1) / S;
                piece = ((a1 - a0 + np - 1) / np + 63) / 64 * 64;
                if (piece > S) piece = S;
            }
            for (int64_t k = a0; k < a1; k += piece) {
                pairs.push_back(k);
                pairs.push_back(k + piece < a1 ? k + piece : a1);
            }
        }
    }
    slice0[(size_t)t->ntiles] = (int64_t)pairs.size() / 2;
}

static bool fx_serial()
{
    const char *e = getenv("CM2_FX_BUILD");
    return e && strcmp(e, "serial") == 0;
}

// groups per full slice of S samples and the fraction of slices with more groups than threads,
// from every 8th full slice (k_fx_build's counting pass on ~12 % of the samples): the slice length
// is chosen from this before anything is allocated or written
int fx_estimate(const cm2_tiles *t, int S, hipStream_t st, double *mean_groups, double *over)
{
    *mean_groups = 0.0;
    *over = 0.0;
    std::vector<int64_t> pairs, slice0, all;
    fx_slices(t, S, slice0, all);
    int64_t seen = 0;
    // (one span: the full slices of S samples; several spans: a slice is a segment, all of them count)
    for (int64_t b = 0; b < t->ntiles; ++b) {
        if (fx_hot_tile(t, b)) continue;
        for (int64_t sl = slice0[(size_t)b]; sl < slice0[(size_t)b + 1]; ++sl) {
            const int64_t k = all[(size_t)(2 * sl)], e = all[(size_t)(2 * sl + 1)];
            if (t->nspans == 1 && e - k != S) continue;
            if (seen++ % 8 == 0) {
                pairs.push_back(k);
                pairs.push_back(e);
            }
        }
    }
    const int64_t np = (int64_t)pairs.size() / 2;
    if (np == 0) return 0;
    DevTemp<int64_t> d_pairs;
    DevTemp<uint32_t> ent, d_counts;
    DevTemp<unsigned int> d_overflow;
    CM2_HIP(d_pairs.alloc(pairs.size()));
    CM2_HIP(cm2::upload(d_pairs, pairs.data(), sizeof(int64_t) * pairs.size(), st));
    CM2_HIP(ent.alloc(t->nvalid));
    CM2_HIP(d_counts.alloc(4 * np));
    CM2_HIP(d_overflow.alloc(1));
    k_fx_build<false, 0><<<(unsigned)np, kFbT, 0, st>>>(
        np, t->half ? 0x7FFFu : 0xFFFFu, d_pairs, 2, t->d_pl, ent, nullptr, nullptr, d_counts, nullptr, nullptr,
        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_overflow);
    CM2_LAUNCH_OK();
    std::vector<uint32_t> counts((size_t)(4 * np));
    CM2_HIP(cm2::download(counts.data(), d_counts, sizeof(uint32_t) * counts.size(), st));
    CM2_HIP(hipStreamSynchronize(st));
    double gsum = 0.0;
    int64_t nover = 0;
    for (int64_t i = 0; i < np; ++i) {
        gsum += counts[(size_t)(4 * i)];
        if (counts[(size_t)(4 * i)] > (uint32_t)kFxT) ++nover;
    }
    *mean_groups = gsum / (double)np;
    *over = (double)nover / (double)np;
    return 0;
}

// builds the lists for slices of S samples; *mean_groups = average groups per full slice,
// *over = fraction of slices with more groups than threads
int fx_build(cm2_tiles *t, int S, hipStream_t st, double *mean_groups, double *over)
{
    fx_release(t);
    const int64_t nv = t->nvalid;
    std::vector<int64_t> slice0, k0;                 // k0: (first address, end) of every slice
    fx_slices(t, S, slice0, k0);
    const int64_t nslices = slice0[(size_t)t->ntiles];
    CM2_CHECK(nslices < ((int64_t)1 << 31), "cm2_tiles: too many slices");
    {
        std::vector<uint2> sk((size_t)nslices + 1, make_uint2(0, 0));
        for (int64_t i = 0; i < nslices; ++i)
            sk[(size_t)i] = make_uint2((uint32_t)k0[(size_t)(2 * i)], (uint32_t)(k0[(size_t)(2 * i + 1)] - k0[(size_t)(2 * i)]));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_sk, sizeof(uint2) * sk.size()));
        CM2_HIP(cm2::upload(t->d_fx_sk, sk.data(), sizeof(uint2) * sk.size(), st));
        CM2_HIP(hipStreamSynchronize(st));           // (sk is a local)
    }
    if (k0.empty()) { k0.push_back(0); k0.push_back(0); }
    CM2_HIP(cm2::dev_malloc(&t->d_fx_slice0, sizeof(int64_t) * slice0.size()));
    CM2_HIP(cm2::upload(t->d_fx_slice0, slice0.data(), sizeof(int64_t) * slice0.size(), st));
    std::vector<uint2> meta((size_t)nslices + 1, make_uint2(0, 0));
    std::vector<uint32_t> tent_off((size_t)nslices + 1, 0);
    int64_t ngroups = 0, ntrun = 0, ntent = 0;
    *mean_groups = 0.0;
    *over = 0.0;
    if (nv > 0) {
        DevTemp<int64_t> d_k0;
        DevTemp<uint64_t> keys_in, keys_out;
        DevTemp<uint32_t> vals_in, ent, d_counts, d_tent_off;
        DevTemp<unsigned int> d_overflow;
        DevTemp<char> d_temp;
        CM2_HIP(d_k0.alloc(k0.size()));
        CM2_HIP(cm2::upload(d_k0, k0.data(), sizeof(int64_t) * k0.size(), st));
        // (k_fx_pack, the serial builder of the global order, reads the slices as a cut list: slice s =
        //  [cut[s], cut[s + 1]))
        DevTemp<int64_t> d_k0s;
        std::vector<int64_t> cuts;
        if (fx_serial() || S > kFbMaxS) {
            for (int64_t i = 0; i < nslices; ++i) cuts.push_back(k0[(size_t)(2 * i)]);
            cuts.push_back(nv);
            CM2_HIP(d_k0s.alloc(cuts.size()));
            CM2_HIP(cm2::upload(d_k0s, cuts.data(), sizeof(int64_t) * cuts.size(), st));
        }
        CM2_HIP(ent.alloc(nv));
        CM2_HIP(d_counts.alloc(4 * nslices));
        CM2_HIP(d_overflow.alloc(1));
        CM2_HIP(hipMemsetAsync(d_overflow.p, 0, sizeof(unsigned int), st));
        const uint32_t qmask = t->half ? 0x7FFFu : 0xFFFFu;
        const int pgrid = (int)((nslices + 63) / 64);
        // one workgroup per slice (k_fx_build) unless CM2_FX_BUILD=serial asks for the radix sort and
        // the one-thread-per-slice packer (k_fx_pack): other lists, the same sums
        const bool serial = fx_serial() || S > kFbMaxS;
        if (serial) {
            CM2_HIP(keys_in.alloc(nv));
            CM2_HIP(keys_out.alloc(nv));
            CM2_HIP(vals_in.alloc(nv));
            CM2_CHECK(t->nspans == 1, "cm2_tiles: the serial list builders need the global tile order");
            k_fx_keys<<<grid_for(nv), kBlock, 0, st>>>(nv, t->ntiles, S, qmask, t->d_seg_off,
                                                      t->d_fx_slice0, t->d_pl, keys_in, vals_in);
            CM2_LAUNCH_OK();
            int end_bit = 17;
            while (((int64_t)1 << (end_bit - 16)) <= nslices && end_bit < 64) ++end_bit;
            size_t tb = 0;
            CM2_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in.p, keys_out.p, vals_in.p,
                                                       ent.p, nv, 0, end_bit, st));
            CM2_HIP(d_temp.alloc(tb + 16));
            CM2_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp.p, tb, keys_in.p, keys_out.p, vals_in.p,
                                                       ent.p, nv, 0, end_bit, st));
            k_fx_pack<false, 0><<<pgrid, 64, 0, st>>>(nslices, qmask, d_k0s, ent, nullptr, nullptr, d_counts,
                                                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                   nullptr, nullptr, nullptr);
        } else {
            k_fx_build<false, 0><<<(unsigned)nslices, kFbT, 0, st>>>(
                nslices, qmask, d_k0, 2, t->d_pl, ent, nullptr, nullptr, d_counts, nullptr, nullptr, nullptr,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_overflow);
        }
        CM2_LAUNCH_OK();
        std::vector<uint32_t> counts((size_t)(4 * nslices));
        CM2_HIP(cm2::download(counts.data(), d_counts, sizeof(uint32_t) * counts.size(), st));
        CM2_HIP(hipStreamSynchronize(st));
        int64_t nfull = 0, nover = 0;
        double gsum = 0.0;
        for (int64_t s = 0; s < nslices; ++s) {
            // .y = first tail run | highest level of the slice << 28
            meta[(size_t)s] = make_uint2((uint32_t)ngroups,
                                         (uint32_t)ntrun | (counts[(size_t)(4 * s + 3)] << 28));
            tent_off[(size_t)s] = (uint32_t)ntent;
            ngroups += counts[(size_t)(4 * s)];
            ntrun += counts[(size_t)(4 * s + 1)];
            ntent += counts[(size_t)(4 * s + 2)];
        }
        int64_t ncounted = 0;
        for (int64_t b = 0; b < t->ntiles; ++b) {
            if (fx_hot_tile(t, b)) continue;
            for (int64_t s = slice0[(size_t)b]; s < slice0[(size_t)b + 1]; ++s) {
                ++ncounted;
                if (counts[(size_t)(4 * s)] > (uint32_t)kFxT) ++nover;
                if (t->nspans > 1 || k0[(size_t)(2 * s + 1)] - k0[(size_t)(2 * s)] == S) {
                    ++nfull;
                    gsum += counts[(size_t)(4 * s)];
                }
            }
        }
        meta[(size_t)nslices] = make_uint2((uint32_t)ngroups, (uint32_t)ntrun);
        tent_off[(size_t)nslices] = (uint32_t)ntent;
        CM2_CHECK(ngroups < ((int64_t)1 << 32) && ntent < ((int64_t)1 << 32) &&
                  ntrun < ((int64_t)1 << 28), "cm2_tiles: fixed-order lists exceed their offsets");
        *mean_groups = nfull ? gsum / (double)nfull : 0.0;
        *over = ncounted ? (double)nover / (double)ncounted : 0.0;
        // (+1 group: a slice without groups at the very end still loads "its" group 0)
        const int64_t ng1 = ngroups + 1, nt1 = ntent ? ntent : 1;
        CM2_HIP(cm2::dev_malloc(&t->d_fx_meta, sizeof(uint2) * meta.size()));
        CM2_HIP(cm2::upload(t->d_fx_meta, meta.data(), sizeof(uint2) * meta.size(), st));
        CM2_HIP(d_tent_off.alloc(tent_off.size()));
        CM2_HIP(cm2::upload(d_tent_off, tent_off.data(), sizeof(uint32_t) * tent_off.size(), st));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_gent, sizeof(uint4) * ng1));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_trun, sizeof(uint2) * (ntrun + 1)));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_tent, sizeof(uint32_t) * nt1));
        if (t->pol > 1) {
            CM2_HIP(cm2::dev_malloc(&t->d_fx_ga, sizeof(double) * 4 * ng1));
            CM2_HIP(cm2::dev_malloc(&t->d_fx_ta, sizeof(double) * nt1));
            if (!t->half) {
                CM2_HIP(cm2::dev_malloc(&t->d_fx_gb, sizeof(double) * 4 * ng1));
                CM2_HIP(cm2::dev_malloc(&t->d_fx_tb, sizeof(double) * nt1));
            }
        }
        const uint2 last = make_uint2((uint32_t)ntent, 0);
        CM2_HIP(cm2::upload(t->d_fx_trun + ntrun, &last, sizeof(uint2), st));
#define CM2_FX_PACK(NANG)                                                                       \
    do {                                                                                        \
        if (serial)                                                                             \
            k_fx_pack<true, NANG><<<pgrid, 64, 0, st>>>(                                          \
                nslices, qmask, d_k0s, ent, t->half ? t->d_half : t->d_cos,                      \
                t->half ? nullptr : t->d_sin, nullptr, t->d_fx_meta, d_tent_off,                \
                reinterpret_cast<uint32_t *>(t->d_fx_gent), t->d_fx_ga, t->d_fx_gb, t->d_fx_trun, \
                t->d_fx_tent, t->d_fx_ta, t->d_fx_tb);                                          \
        else                                                                                    \
            k_fx_build<true, NANG><<<(unsigned)nslices, kFbT, 0, st>>>(                           \
                nslices, qmask, d_k0, 2, t->d_pl, ent, t->half ? t->d_half : t->d_cos,              \
                t->half ? nullptr : t->d_sin, nullptr, t->d_fx_meta, d_tent_off,                \
                reinterpret_cast<uint32_t *>(t->d_fx_gent), t->d_fx_ga, t->d_fx_gb, t->d_fx_trun, \
                t->d_fx_tent, t->d_fx_ta, t->d_fx_tb, d_overflow);                              \
    } while (0)
        if (t->pol == 1) CM2_FX_PACK(0);
        else if (t->half) CM2_FX_PACK(1);
        else CM2_FX_PACK(2);
#undef CM2_FX_PACK
        CM2_LAUNCH_OK();
        unsigned int h_over = 0;
        CM2_HIP(cm2::download(&h_over, d_overflow.p, sizeof(h_over), st));
        CM2_HIP(hipStreamSynchronize(st));
        CM2_CHECK(h_over == 0, "cm2_tiles: a slice of %d samples packs into more than %d groups", S,
                  kFbMaxGroups);
    } else {
        CM2_HIP(cm2::dev_malloc(&t->d_fx_meta, sizeof(uint2) * meta.size()));
        CM2_HIP(cm2::upload(t->d_fx_meta, meta.data(), sizeof(uint2) * meta.size(), st));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_gent, sizeof(uint4)));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_trun, sizeof(uint2)));
        CM2_HIP(cm2::dev_malloc(&t->d_fx_tent, sizeof(uint32_t)));
        CM2_HIP(hipStreamSynchronize(st));
    }
    t->fx_S = S;
    t->fx_ngroups = ngroups;
    t->fx_nslices = nslices;
    return 0;
}

// ------------------------------------------------------------------- parts ----------
// Finish time of `items` (samples each, in dispatch order) over the ideal (total / slots).  `slots`
// workgroups are resident and take the next item as one finishes; the kernel is bandwidth bound, so the
// resident workgroups share the chip's rate equally -- but one workgroup alone cannot use more than
// about 1.5 x its share of the full chip (measured at C4 size: a slice takes 3.7 us with 512 workgroups
// resident, 2.5 us with 51), which is what makes a few items left over at the end expensive.
static double parts_makespan(const std::vector<int64_t> &items, int slots)
{
    const double rmax = 1.5;
    std::vector<double> heap;                               // min-heap: finish "virtual time" of the active items
    auto cmp = [](double a, double b2) { return a > b2; };
    double V = 0.0, T = 0.0, total = 0.0;                   // virtual time (work done per active item), real time
    size_t next = 0;
    auto rate = [&]() {
        const double fair = (double)slots / (double)(heap.empty() ? 1 : heap.size());
        return fair < rmax ? fair : rmax;
    };
    for (; next < items.size() && (int)heap.size() < slots; ++next) {
        heap.push_back((double)items[next] + 4096.0);      // (+ zeroing and writing the tile copy)
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
    for (int64_t x : items) total += (double)x + 4096.0;
    while (!heap.empty()) {
        const double vf = heap.front();
        T += (vf - V) / rate();
        V = vf;
        std::pop_heap(heap.begin(), heap.end(), cmp);
        heap.pop_back();
        if (next < items.size()) {
            heap.push_back(V + (double)items[next++] + 4096.0);
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    return total > 0.0 ? T * (double)slots / total : 1.0;
}

// Shares the slices of heavy tiles out to several workgroups (see cm2_tiles.h).  The part length is
// chosen by simulation: a tile of more than 1.1 x target samples is cut into ceil(load / target) parts
// of equal slice counts, for targets between 1.25 and 0.2 of the mean load per resident workgroup; the
// target with the earliest simulated finish wins (fewer parts on a tie).  Parts are dispatched in tile
// order, i.e. by ascending address: dispatched by descending load instead (which scatters the
// workgroups' streams over the buffers) the same parts took 0.45 instead of 0.42 ms at C4 size
// (profiles/r04_uneven_parts.md).  CM2_PT_PARTS=0 keeps one workgroup per tile, CM2_PT_PARTS=<samples>
// fixes the target.
int parts_plan(cm2_tiles *t, hipStream_t st)
{
    if (!t->pt_split || t->fx_nslices == 0) return 0;
    int forced = -1;
    if (const char *e = getenv("CM2_PT_PARTS")) forced = atoi(e);
    if (forced == 0) return 0;
    std::vector<int64_t> slice0, k0;
    fx_slices(t, t->fx_S, slice0, k0);
    // The simulated machine is the MI355X this library is written for (kNumCU), NOT the live device: part
    // boundaries change the order in which a pixel's terms are added, and the header promises that they
    // depend on the plan only -- the same bits on any partition mode or device count.
    // (two workgroups per CU when their LDS fits twice, fx_max_slice)
    const int slots = (fx_lds_bytes(t, t->fx_S) <= 79 * 1024 ? 2 : 1) * kNumCU;
    std::vector<int64_t> load((size_t)t->ntiles, 0);
    int64_t total = 0;
    for (int64_t b = 0; b < t->ntiles; ++b) {
        if (fx_hot_tile(t, b)) continue;
        load[(size_t)b] = t->tile_count[(size_t)b];
        total += load[(size_t)b];
    }
    if (total == 0) return 0;
    auto parts_of = [&](int64_t b, int64_t target) -> int64_t {
        const int64_t ns = slice0[(size_t)b + 1] - slice0[(size_t)b];
        if (load[(size_t)b] * 10 <= target * 11 || ns <= 1) return 1;
        int64_t k = (load[(size_t)b] + target - 1) / target;
        return k < ns ? k : ns;
    };
    auto items_for = [&](int64_t target, std::vector<int64_t> &items) {
        items.clear();
        for (int64_t b = 0; b < t->ntiles; ++b) {
            if (load[(size_t)b] == 0) continue;
            const int64_t k = parts_of(b, target);
            for (int64_t j = 0; j < k; ++j) items.push_back(load[(size_t)b] / k);
        }
    };
    const double per_slot = (double)total / (double)slots;
    int64_t best_target = 0;
    double best = 1e30;
    size_t best_items = 0;
    std::vector<int64_t> items;
    if (forced > 0) {
        best_target = forced;
        items_for(best_target, items);
        best = parts_makespan(items, slots);
    } else {
        // one workgroup per tile is kept unless some split finishes at least 5 % earlier; among the
        // splits the earliest finish, and the fewest parts within 1 % of it
        items_for(INT64_MAX / 16, items);
        const double whole = parts_makespan(items, slots);
        for (int step = 0; step <= 42; ++step) {
            const int64_t target = (int64_t)(per_slot * (1.25 - 0.025 * step)) + 1;
            if (target < 4 * t->fx_S) break;                // (parts of a few slices only: not worth a copy)
            items_for(target, items);
            const double mk = parts_makespan(items, slots);
            if (mk < best - 0.01 || (mk < best + 0.01 && items.size() < best_items)) {
                best = mk;
                best_target = target;
                best_items = items.size();
            }
        }
        if (best > 0.95 * whole) best_target = 0;
    }
    if (best_target == 0) return 0;
    // the parts in tile order (= dispatch order), their scratch slots (split tiles only)
    std::vector<int4> parts;
    std::vector<int64_t> multi;
    t->tile_part0.assign((size_t)t->ntiles + 1, 0);
    int64_t slot = 0;
    for (int64_t b = 0; b < t->ntiles; ++b) {
        t->tile_part0[(size_t)b] = (int64_t)parts.size();
        const int64_t s0 = slice0[(size_t)b], ns = slice0[(size_t)b + 1] - s0;
        const int64_t k = load[(size_t)b] ? parts_of(b, best_target) : 1;
        if (k > 1) {
            t->multi_tile.push_back(b);
            multi.push_back(t->tile_p0[(size_t)b] * t->pol);
            multi.push_back((t->tile_p0[(size_t)b + 1] - t->tile_p0[(size_t)b]) * t->pol);
            multi.push_back(slot);
            multi.push_back(k);
        }
        for (int64_t j = 0; j < k; ++j) {
            const int64_t a = s0 + ns * j / k, e = s0 + ns * (j + 1) / k;
            parts.push_back(make_int4((int)b, (int)(e - a), (int)a, k > 1 ? (int)(slot + j) : -1));
        }
        if (k > 1) slot += k;
    }
    t->tile_part0[(size_t)t->ntiles] = (int64_t)parts.size();
    if (t->multi_tile.empty()) {                            // nothing to split after all
        t->tile_part0.clear();
        return 0;
    }
    t->nparts = (int64_t)parts.size();
    t->part_slots = slot;
    t->part_makespan = best;
    CM2_HIP(cm2::dev_malloc(&t->d_parts, sizeof(int4) * parts.size()));
    CM2_HIP(cm2::dev_malloc(&t->d_multi, sizeof(int64_t) * multi.size()));
    CM2_HIP(cm2::dev_malloc(&t->d_part_buf, sizeof(double) * (size_t)slot * (size_t)t->tp * (size_t)t->pol));
    CM2_HIP(cm2::upload(t->d_parts, parts.data(), sizeof(int4) * parts.size(), st));
    CM2_HIP(cm2::upload(t->d_multi, multi.data(), sizeof(int64_t) * multi.size(), st));
    CM2_HIP(hipStreamSynchronize(st));
    return 0;
}

// The device-side description of the fused form (FxFused) and its arrival counters, after hot_plan: one
// counter per hot tile, in a block of their own padded to 16 bytes (zeroed by one memset in front of every
// launch).  CM2_PT_FUSE=0 keeps the separate kernels.
int fused_plan(cm2_tiles *t, hipStream_t st)
{
    if (const char *e = getenv("CM2_PT_FUSE"))
        if (atoi(e) == 0) return 0;
    const size_t nhot = t->hot_tile.size();
    if (nhot == 0) return 0;
    t->fx_count_bytes = (sizeof(unsigned int) * nhot + 15) / 16 * 16;
    CM2_HIP(cm2::dev_malloc(&t->d_fx_count, t->fx_count_bytes));
    FxFused z;
    memset(&z, 0, sizeof(z));
    z.count = t->d_fx_count;
    z.hot_range = t->d_hot_range;
    z.hot_range_tile = t->d_hot_range_tile;
    z.hot_tiles = t->d_hot_tiles;
    z.hot_partial = t->d_hot_partial;
    z.pl = t->d_pl;
    z.a_tb = t->d_half ? t->d_half : t->d_cos;
    z.b_tb = t->d_sin;
    FxFused *dz = nullptr;
    CM2_HIP(cm2::dev_malloc(&dz, sizeof(FxFused)));
    t->d_fx_fused = dz;
    CM2_HIP(cm2::upload(dz, &z, sizeof(FxFused), st));
    CM2_HIP(hipStreamSynchronize(st));
    return 0;
}

template <int POL, bool HALF, int VPT>
int fx_launch_inst(const cm2_tiles *t, const double *d_tod_tb, double *d_out, int64_t tile_lo,
                   int64_t tile_hi, hipStream_t stream)
{
    // plans with parts (and not the exact order): the workgroups are the parts of the tiles in range, in
    // tile (= address) order; then the copies of the split tiles are added up
    const bool parts = t->d_parts && t->pt_fixed != 2;
    const int64_t q0 = parts ? t->tile_part0[(size_t)tile_lo] : tile_lo;
    const int64_t q1 = parts ? t->tile_part0[(size_t)tile_hi] : tile_hi;
    // fused form: the hot tiles' ranges inside [tile_lo, tile_hi) are further workgroups of this launch
    const bool fused = t->d_fx_fused && t->pt_fixed != 2;
    int64_t hc0 = 0, hc1 = 0;
    if (fused && t->d_hot_flag) {
        int64_t h0 = 0, h1 = (int64_t)t->hot_tile.size();
        while (h0 < h1 && t->hot_tile[(size_t)h0] < tile_lo) ++h0;
        while (h1 > h0 && t->hot_tile[(size_t)h1 - 1] >= tile_hi) --h1;
        hc0 = t->hot_chunk0[(size_t)h0];
        hc1 = t->hot_chunk0[(size_t)h1];
    }
    size_t lds = fx_lds_bytes(t, t->fx_S);
    if (hc1 > hc0 && lds < sizeof(double) * (3 * (size_t)kHotT + 2)) lds = sizeof(double) * (3 * (size_t)kHotT + 2);
    static size_t granted[64] = {0};
    CM2_HIP(ensure_dynamic_lds((const void *)k_Pt_tiles_fixed<POL, HALF, VPT>, lds, granted));
    if (fused) CM2_HIP(hipMemsetAsync(t->d_fx_count, 0, t->fx_count_bytes, stream));
    k_Pt_tiles_fixed<POL, HALF, VPT><<<(int)(q1 - q0 + hc1 - hc0), kFxT, lds, stream>>>(
        t->tp, t->d_tile_p0, (int)tile_lo, t->d_fx_sk, t->d_fx_slice0, t->d_fx_meta,
        t->d_fx_gent, reinterpret_cast<const double2 *>(t->d_fx_ga),
        reinterpret_cast<const double2 *>(t->d_fx_gb), t->d_fx_trun, t->d_fx_tent, t->d_fx_ta,
        t->d_fx_tb, d_tod_tb, d_out,
        t->pt_fixed == 2 ? 0xFFFFFFFFu : (uint32_t)kFxChunkMinDefault,
        t->pt_fixed == 2 ? nullptr : t->d_hot_flag, parts ? t->d_parts : nullptr, (int)q0, t->d_part_buf,
        fused ? static_cast<const FxFused *>(t->d_fx_fused) : nullptr, (int)(q1 - q0), hc0);
    CM2_LAUNCH_OK();
    if (parts) {
        int64_t m0 = 0, m1 = (int64_t)t->multi_tile.size();
        while (m0 < m1 && t->multi_tile[(size_t)m0] < tile_lo) ++m0;
        while (m1 > m0 && t->multi_tile[(size_t)m1 - 1] >= tile_hi) --m1;
        if (m1 > m0) {
            const dim3 cgrid((unsigned)(m1 - m0), (unsigned)(((int64_t)t->tp * t->pol + 255) / 256));
            k_parts_combine<<<cgrid, 256, 0, stream>>>(t->d_multi, m0, (int64_t)t->tp * t->pol,
                                                                     t->d_part_buf, d_out);
            CM2_LAUNCH_OK();
        }
    }
    if (t->pt_fixed != 2 && t->d_hot_flag && !fused)
        return hot_launch<POL, HALF>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
    return 0;
}

template <int POL, bool HALF>
int fx_launch_vpt(const cm2_tiles *t, const double *d_tod_tb, double *d_out, int64_t tile_lo,
                  int64_t tile_hi, hipStream_t stream)
{
    const int vpt = (t->fx_S + kFxT - 1) / kFxT;
    if (vpt <= 2) return fx_launch_inst<POL, HALF, 2>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
    if (vpt == 3) return fx_launch_inst<POL, HALF, 3>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
    return fx_launch_inst<POL, HALF, 4>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
}

}  // namespace

namespace cm2 {

void fx_free(cm2_tiles *t) { fx_release(t); }

bool fx_serial_build() { return fx_serial(); }

// longest slice the kernel can stage beside the tile: 4 values a thread at most, and short enough for
// two workgroups per CU (<= 79 KB each) whenever some slice length allows that
int fx_max_slice(const cm2_tiles *t)
{
    int smax = 4 * kFxT;
    {
        int s2 = smax;
        while (s2 > 2 * kFxT && fx_lds_bytes(t, s2) > 79 * 1024) s2 -= kFxT;
        if (fx_lds_bytes(t, s2) <= 79 * 1024) smax = s2;
    }
    while (smax > 256 && fx_lds_bytes(t, smax) > 159 * 1024) smax -= 256;
    return smax;
}

int fx_parts_info(const cm2_tiles *t, int64_t *h_info)
{
    h_info[0] = t->d_parts ? t->nparts : t->ntiles;
    h_info[1] = (int64_t)t->multi_tile.size();
    h_info[2] = (int64_t)sizeof(double) * (t->part_slots * t->tp * t->pol + (t->hot_chunk0.empty() ? 0 : 3 * t->hot_chunk0.back()));
    h_info[3] = (int64_t)(1000.0 * t->part_makespan + 0.5);
    return 0;
}

int64_t fx_designed_bytes(const cm2_tiles *t)
{
    if (!t || !t->fx_S) return 0;
    const int64_t per_group = 16 + (t->pol > 1 ? (t->half ? 32 : 64) : 0);
    return 8 * t->nvalid + per_group * t->fx_ngroups + 8 * t->fx_nslices;
}

// plan of the fixed-order P^T, built on first use.  Slice length: a slice should fill most of
// the workgroup's 512 groups but rarely more; the number of groups per sample depends on how
// often a pixel is hit twice inside a slice, so it is measured: a trial plan with S = 1536 gives
// the groups per slice, S is then set for ~0.92 x 512 groups and the plan rebuilt
// (CM2_PT_SLICE = samples fixes S).
static int fx_plan_build(const cm2_tiles *tc, hipStream_t st, bool *use);

int fx_plan(const cm2_tiles *tc, hipStream_t st, bool *use)
{
    const int rc = fx_plan_build(tc, st, use);
    // running out of device memory is transient: the failure is not remembered, so that the call may be made
    // again after the host has freed memory (cosmomap2_amd/_hip.py does that once; a build starts from scratch)
    if (rc == CM2_ERR_OUT_OF_MEMORY) const_cast<cm2_tiles *>(tc)->fx_failed = 0;
    return rc;
}

static int fx_plan_build(const cm2_tiles *tc, hipStream_t st, bool *use)
{
    cm2_tiles *t = const_cast<cm2_tiles *>(tc);
    *use = false;
    if (!t->pt_fixed) return 0;
    // The lists are normally built by cm2_tiles_prepare_pt right after the plan (the Python layer
    // and the C demos call it): an application then allocates nothing and is safe to capture or
    // to issue from several host threads.  A plan that was not prepared builds them here, under
    // a lock (two threads applying P^T on one plan would otherwise both build).
    static std::mutex build_lock;
    std::lock_guard<std::mutex> hold(build_lock);
    if (t->fx_failed) {
        set_error("the fixed-order P^T lists of this tile plan could not be built (earlier error); "
                  "cm2_tiles_set_pt_order(t, 0) selects the atomic form");
        return 1;
    }
    struct FailMark { cm2_tiles *t; bool ok; ~FailMark() { if (!ok) { t->fx_failed = 1; t->fx_S = 0; } } } mark{t, false};
    if (t->fx_S == 0) {
        // (4 staged values per thread at most.  Two workgroups per CU need <= 79 KB each: a 2048-pixel
        // IQU tile (48 KB) with four staged values per thread in two buffers (32 KB) would leave ONE
        // workgroup per CU (C5: P^T 0.53 -> 0.64 ms); the slice is kept short enough for two whenever
        // some slice length allows it.)
        int smax = fx_max_slice(t);
        if (fx_lds_bytes(t, smax) > 159 * 1024) {        // the tile alone fills LDS: atomics
            t->pt_fixed = 0;
            mark.ok = true;
            return 0;
        }
        int forced = 0;
        if (const char *e = getenv("CM2_PT_SLICE")) forced = atoi(e);
        double mean = 0.0, over = 0.0;
        if (forced >= 64 && forced <= 4 * kFxT) {
            if (int rc = fx_build(t, forced < smax ? forced : smax, st, &mean, &over)) return rc;
        } else if (t->nspans > 1) {
            // the slices are the segments of the plan's spans (the span length was chosen for segments
            // of ~0.9 smax samples): nothing to tune, S only caps the rare longer segment
            if (int rc = fx_build(t, smax, st, &mean, &over)) return rc;
        } else {
            int S = 1536 < smax ? 1536 : smax;
            auto wanted = [&](int S_now) {
                int want = (int)(0.92 * kFxT * S_now / mean) / 64 * 64;
                if (over > 0.10) want = want < S_now * 7 / 8 ? want : S_now * 7 / 8 / 64 * 64;
                if (want > smax) want = smax;
                if (want < 256) want = 256;
                return want;
            };
            if (!fx_serial() && S <= kFbMaxS) {           // first guess from a sample of the slices
                if (int rc = fx_estimate(t, S, st, &mean, &over)) return rc;
                if (mean > 0.0) S = wanted(S);
            }
            if (int rc = fx_build(t, S, st, &mean, &over)) return rc;
            for (int iter = 0; iter < 3 && mean > 0.0; ++iter) {
                const int want = wanted(S);
                const bool close_enough = want >= S * 15 / 16 && want <= S * 17 / 16 && over <= 0.10;
                if (close_enough || want == S) break;
                S = want;
                if (int rc = fx_build(t, S, st, &mean, &over)) return rc;
            }
        }
    }
    if (!t->d_hot_flag && t->hot_chunk0.empty()) {
        if (int rc = hot_plan(t, st)) return rc;
        if (int rc = parts_plan(t, st)) return rc;
        if (int rc = fused_plan(t, st)) return rc;
    }
    mark.ok = true;
    *use = true;
    return 0;
}

int fx_launch(const cm2_tiles *t, const double *d_tod_tb, double *d_out, int64_t tile_lo,
              int64_t tile_hi, hipStream_t stream)
{
    if (tile_hi <= tile_lo) return 0;
    if (t->pol == 1) return fx_launch_vpt<1, false>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
    if (t->pol == 2)
        return t->half ? fx_launch_vpt<2, true>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream)
                       : fx_launch_vpt<2, false>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
    return t->half ? fx_launch_vpt<3, true>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream)
                   : fx_launch_vpt<3, false>(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
}

}  // namespace cm2
