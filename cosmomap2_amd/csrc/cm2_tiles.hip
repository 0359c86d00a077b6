// cm2_tiles.hip -- tile-bucketed ("TB") TOD order and the LDS-staged pointing kernels
// that work on it.  This is the throughput path of the P^T N^-1 P chain when N has
// off-diagonal terms (so that a time-ordered TOD must exist between P and P^T).
//
// Why: with uniformly random pointing a time-ordered gather P x touches a random 24-B
// map record per sample and a pixel-major P^T touches a random 8-B TOD entry per sample;
// both pay a whole cache line for a few useful bytes (measured 2.0 ms each at 1e8
// samples / nside 256, i.e. 1.4 TB/s algorithmic).  Bucketing the samples by pixel TILE
// (TP pixels, tile map slice = TP*pol*8 B staged in LDS) makes every HBM access of the
// two pointing kernels a coalesced stream:
//
//   TB order   = stable partition of the valid samples by tile (time order inside a tile)
//   pl_tb u16  = pixel index inside the tile,  cos_tb / sin_tb f64     (static, built once)
//   tb_dst u32 = position of time sample t in TB order (0xFFFFFFFF for flagged samples)
//
//   P    : stage the tile of x in LDS, stream the bucket, write d_tb      26 B read + 8 B write
//   P^T  : stream the bucket, ds_add_f64 into the LDS tile, flush tile    26+8 B read
//   time<->TB permutation: time-order-driven; because the partition is stable every tile
//          is a sequential stream, so the scattered 8-B accesses combine in L2
//          (each workgroup owns one contiguous time range).                20 B / sample
//
// P on TB order evaluates the reference's per-sample expression; with both angle arrays
// (CM2_TILE_ANGLES=full) it is bit-identical to the reference loop, in the default half-angle
// storage cos / sin are rebuilt to ~2e-16 absolute, so P agrees to rounding.
// P^T has two forms.  The default (k_Pt_tiles_fixed, CM2_PT_ORDER=fixed) adds the terms of every
// pixel IN TIME ORDER starting from 0, exactly like the reference's serial loop, with no
// atomics: bitwise reproducible from run to run and, with CM2_TILE_ANGLES=full, bit-identical to
// the serial loop.  CM2_PT_ORDER=atomic selects k_Pt_tiles (LDS + global fp64 atomics): the same
// set of terms per pixel in an unspecified order, equal to rounding, not reproducible.
//
// Reference loops replaced: interfaces/linearoperators.py:483-489 (mult_iqu) and :509-516
// (rmult_iqu) and their I / QU variants.
#include "cm2_tiles.h"

#include <hipcub/hipcub.hpp>
#include <cstring>

using namespace cm2;

static uint64_t next_plan_id()
{
    static uint64_t counter = 0;
    return ++counter;
}

// ------------------------------------------------------------------ build -------
// tile of pixel p: uniform tiles of tp pixels, or (p0 != nullptr) the tile whose pixel range
// [p0[b], p0[b + 1]) holds p -- tiles of equal sample count for uneven hit maps
__device__ __forceinline__ uint32_t tile_of(int32_t p, int tp, const int64_t *__restrict__ p0,
                                            uint32_t ntiles)
{
    if (p0 == nullptr) return (uint32_t)(p / tp);
    uint32_t lo = 0, hi = ntiles;                        // largest b with p0[b] <= p
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p0[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_tile_keys(const int32_t *__restrict__ pix, int64_t nt,
                                                    int tp, const int64_t *__restrict__ p0,
                                                    uint32_t ntiles, int64_t npix,
                                                    uint32_t *__restrict__ keys,
                                                    uint32_t *__restrict__ vals,
                                                    unsigned int *__restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int b = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const int32_t p = pix[i];
        if (p < -1 || p >= npix) b = 1;                  // same rule as cm2_pointing_create
        keys[i] = (p < 0 || p >= npix) ? ntiles : tile_of(p, tp, p0, ntiles);
        vals[i] = (uint32_t)i;
    }
    if (b) atomicOr(bad, 1u);
}

// hits of every pixel (flagged / out-of-range samples skipped): input of the balanced tiling.
// A pixel that holds a large share of the samples (5e6 hits on one address: 50 ms of serialised
// global atomics) is counted in LDS: every workgroup keeps a direct-mapped table of 1024 pixels,
// first come first served; a sample whose slot belongs to its pixel is an LDS add, the others go to
// memory as before, and the table is flushed with one global add per slot.  Counts are integers:
// the result does not depend on which path a sample took.
__global__ __launch_bounds__(256) void k_pix_hist(const int32_t *__restrict__ pix, int64_t nt,
                                                   int64_t npix, unsigned int *__restrict__ hits)
{
    __shared__ unsigned int key[1024], cnt[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) {
        key[i] = 0xFFFFFFFFu;
        cnt[i] = 0;
    }
    __syncthreads();
    const int64_t span = (nt + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * span, i1 = i0 + span < nt ? i0 + span : nt;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int32_t p = pix[i];
        if (p < 0 || p >= npix) continue;
        const unsigned int slot = ((unsigned int)p * 2654435761u) >> 22;
        const unsigned int old = atomicCAS(&key[slot], 0xFFFFFFFFu, (unsigned int)p);
        if (old == 0xFFFFFFFFu || old == (unsigned int)p) atomicAdd(&cnt[slot], 1u);
        else atomicAdd(&hits[p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256)
        if (cnt[i]) atomicAdd(&hits[key[i]], cnt[i]);
}

__global__ __launch_bounds__(256) void k_tile_bounds(const uint32_t *__restrict__ keys, int64_t nt,
                                                      int64_t ntiles, int64_t *__restrict__ off)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > ntiles) return;
    int64_t lo = 0, hi = nt;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)b) lo = mid + 1; else hi = mid;
    }
    off[b] = lo;
}

// ---- stable partition by tile without a sort ("multisplit") -----------------------------------
// Every wave owns kSplitChunk consecutive time samples.  Pass 1 (k_tile_rank): per sample its tile
// and its rank among the EARLIER samples of the same tile in the chunk -- 64 samples at a time:
// rank inside the row by comparing with every lower lane, plus the tile's running count in a
// wave-private LDS histogram (read by all lanes of the row, advanced by the last lane of each
// tile) -- and the chunk's count of every tile, stored tile-major.  One exclusive scan over the
// tile-major counts IS the tile order: base[tile][chunk] = first address of that chunk's samples
// of that tile.  Pass 2 (k_tile_place): address = base[tile][chunk] + rank.  The addresses are
// those of a stable sort by tile; nothing here depends on the order in which waves run.
constexpr int kSplitChunk = 8192;                        // samples per wave (rank fits 16 bits)
constexpr int64_t kSplitMaxTiles = 8192;                 // 4 wave histograms of u16 in 64 KB of LDS

// SPANS.  The counts are stored [span][tile][chunk of the span] (G chunks a span, the last span padded
// with empty chunks): the same exclusive scan then yields the order [span][tile][time] -- every span
// of G x 8192 consecutive time samples is partitioned by tile on its own, and segment (span, tile)
// starts at base[(span * ntiles + tile) * G].  G = number of chunks (one span) is the global tile order.
__host__ __device__ inline int64_t cnt_index(int64_t tile, int64_t chunk, int64_t ntiles, int64_t G)
{
    return ((chunk / G) * ntiles + tile) * G + chunk % G;
}

__global__ __launch_bounds__(256) void k_tile_rank(const int32_t *__restrict__ pix, int64_t nt, int tp,
                                                    const int64_t *__restrict__ p0, uint32_t ntiles,
                                                    int64_t npix, int64_t nchunks, int64_t G,
                                                    uint32_t *__restrict__ packed,
                                                    uint32_t *__restrict__ cnt_t,
                                                    unsigned int *__restrict__ bad)
{
    extern __shared__ uint16_t hist_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *hist = hist_all + (size_t)wave * ntiles;
    for (uint32_t b = lane; b < ntiles; b += 64) hist[b] = 0;
    const int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= nchunks) return;
    const int64_t i0 = c * kSplitChunk, i1 = i0 + kSplitChunk < nt ? i0 + kSplitChunk : nt;
    unsigned int b_ = 0;
    for (int64_t i = i0; i < i1; i += 64) {
        const int64_t idx = i + lane;
        const bool in = idx < i1;
        const int32_t p = in ? pix[idx] : -1;
        if (p < -1 || p >= npix) b_ = 1;                 // same rule as cm2_pointing_create
        const bool valid = p >= 0 && p < npix;
        const int tile = valid ? (int)tile_of(p, tp, p0, ntiles) : -1;
        int rank = 0, same = 0;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int tj = __builtin_amdgcn_readlane(tile, j);
            const int eq = tj == tile ? 1 : 0;
            same += eq;
            rank += j < lane ? eq : 0;
        }
        if (valid) {
            const uint32_t r = (uint32_t)hist[tile] + (uint32_t)rank;
            packed[idx] = ((uint32_t)tile << 16) | r;
            if (rank == same - 1) hist[tile] = (uint16_t)(r + 1);
        } else if (in) {
            packed[idx] = 0xFFFFFFFFu;
        }
    }
    for (uint32_t b = lane; b < ntiles; b += 64) cnt_t[cnt_index(b, c, ntiles, G)] = hist[b];
    if (b_) atomicOr(bad, 1u);
}

// off[s] = first address of segment s = span * ntiles + tile (s <= nsegs: the last one is the number of
// valid samples)
__global__ __launch_bounds__(256) void k_tile_offsets(const uint32_t *__restrict__ base, int64_t G,
                                                       int64_t nsegs, int64_t *__restrict__ off)
{
    const int64_t sg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg <= nsegs) off[sg] = base[sg * G];
}

template <int POL, bool HALF>
__global__ __launch_bounds__(256) void k_tile_place(
    int64_t nt, int tp, const int64_t *__restrict__ p0, int64_t ntiles, int64_t G,
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ base,
    const int32_t *__restrict__ pix, const double *__restrict__ c, const double *__restrict__ s,
    uint32_t *__restrict__ tb_dst, uint16_t *__restrict__ pl, double *__restrict__ ctb,
    double *__restrict__ stb)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const uint32_t pk = packed[i];
        if (pk == 0xFFFFFFFFu) {
            tb_dst[i] = kInvalidSample;                  // flagged samples sort behind every tile
            continue;
        }
        const int64_t tile = pk >> 16;
        const uint32_t k = base[cnt_index(tile, i / kSplitChunk, ntiles, G)] + (pk & 0xFFFFu);
        tb_dst[i] = k;
        const int32_t px = pix[i];
        uint16_t w = p0 ? (uint16_t)(px - p0[tile]) : (uint16_t)(px - (int32_t)tile * tp);
        if (POL > 1) {
            if (HALF) {
                const double cv = c[i], sv = s[i];
                if (cv < 0.0) {
                    w |= 0x8000;
                    ctb[k] = sv / (1.0 - cv);
                } else {
                    ctb[k] = sv / (1.0 + cv);
                }
            } else {
                ctb[k] = c[i];
                stb[k] = s[i];
            }
        }
        pl[k] = w;
    }
}

// k_tile_place with the chunk's samples staged in LDS by address: one workgroup per chunk of
// kSplitChunk samples.  The chunk's samples of tile b occupy the addresses base[b][chunk] ..; in LDS
// they get the slots lbase[b] .. (lbase = scan of the chunk's per-tile counts), so that consecutive
// slots are consecutive addresses inside a run and the 2-byte words and half angles leave in pieces
// of 16 entries per tile instead of one scattered store per sample (4.9 -> 2 ms at C4).  Used for the
// one-array forms (pol = 1, half angles): 8192 doubles + 2 x 8192 words + two tables per tile.
template <int POL>
__global__ __launch_bounds__(256) void k_tile_place_staged(
    int64_t nt, int tp, const int64_t *__restrict__ p0, int64_t G, int ntiles,
    const uint32_t *__restrict__ packed, const uint32_t *__restrict__ base,
    const int32_t *__restrict__ pix, const double *__restrict__ c, const double *__restrict__ s,
    uint32_t *__restrict__ tb_dst, uint16_t *__restrict__ pl, double *__restrict__ ctb)
{
    extern __shared__ double sm_p[];
    double *hs = sm_p;                                              // [kSplitChunk] half angles by slot
    uint32_t *gbase = reinterpret_cast<uint32_t *>(hs + (POL > 1 ? kSplitChunk : 0));   // [ntiles]
    uint32_t *lbase = gbase + ntiles;                               // [ntiles + 1]
    uint16_t *ws = reinterpret_cast<uint16_t *>(lbase + ntiles + 1);   // [kSplitChunk] pl words by slot
    uint16_t *ts = ws + kSplitChunk;                                // [kSplitChunk] tile of the slot
    __shared__ uint32_t wsum[4];
    const int64_t ch = blockIdx.x;
    const int64_t i0 = ch * kSplitChunk, i1 = i0 + kSplitChunk < nt ? i0 + kSplitChunk : nt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // per-tile counts of the chunk (next word of the tile-major scan minus this one), scanned
    const int per = (ntiles + 255) / 256;
    uint32_t mine = 0;
    for (int b = t * per; b < (t + 1) * per && b < ntiles; ++b) {
        const int64_t ci = cnt_index(b, ch, ntiles, G);
        const uint32_t g0 = base[ci], g1 = base[ci + 1];
        gbase[b] = g0;
        lbase[b] = g1 - g0;
        mine += g1 - g0;
    }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = inc - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) before += wsum[w];
        total += wsum[w];
    }
    for (int b = t * per; b < (t + 1) * per && b < ntiles; ++b) {
        const uint32_t cnt = lbase[b];
        lbase[b] = before;
        before += cnt;
    }
    if (t == 0) lbase[ntiles] = total;
    __syncthreads();
    // (one workgroup per CU: four samples per thread in flight at a time)
    for (int64_t ib = i0 + t; ib < i1; ib += 4 * 256) {
        uint32_t pk[4];
        int32_t px[4];
        double cv[4], sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = ib + 256 * u;
            const bool in = i < i1;
            pk[u] = in ? packed[i] : 0xFFFFFFFFu;
            px[u] = in ? pix[i] : 0;
            if (POL > 1) {
                cv[u] = in ? c[i] : 1.0;
                sv[u] = in ? s[i] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = ib + 256 * u;
            if (i >= i1) continue;
            if (pk[u] == 0xFFFFFFFFu) {
                tb_dst[i] = kInvalidSample;
                continue;
            }
            const uint32_t tile = pk[u] >> 16, r = pk[u] & 0xFFFFu;
            tb_dst[i] = gbase[tile] + r;
            const uint32_t slot = lbase[tile] + r;
            uint16_t w = p0 ? (uint16_t)(px[u] - p0[tile]) : (uint16_t)(px[u] - (int32_t)tile * tp);
            if (POL > 1) {
                if (cv[u] < 0.0) {
                    w |= 0x8000;
                    hs[slot] = sv[u] / (1.0 - cv[u]);
                } else {
                    hs[slot] = sv[u] / (1.0 + cv[u]);
                }
            }
            ws[slot] = w;
            ts[slot] = (uint16_t)tile;
        }
    }
    __syncthreads();
    for (uint32_t j = t; j < total; j += 256) {
        const uint32_t tile = ts[j];
        const uint32_t k = gbase[tile] + (j - lbase[tile]);
        pl[k] = ws[j];
        if (POL > 1) ctb[k] = hs[j];
    }
}

// 1 if some (cos, sin) pair is not on the unit circle to rounding: then the half-angle form
// would change the operator and the full arrays are kept
__global__ __launch_bounds__(256) void k_unit_circle(int64_t nt, const double *__restrict__ c,
                                                      const double *__restrict__ s,
                                                      unsigned int *__restrict__ off_circle)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const double r = c[i] * c[i] + s[i] * s[i] - 1.0;
        if (!(fabs(r) <= 1e-14)) bad = 1;
    }
    if (bad) atomicOr(off_circle, 1u);
}

template <int POL, bool HALF>
__global__ __launch_bounds__(256) void k_tile_fill(
    int64_t nt, int64_t nvalid, int tp, const int64_t *__restrict__ p0, uint32_t ntiles,
    const uint32_t *__restrict__ tb_src,
    const int32_t *__restrict__ pix, const double *__restrict__ c, const double *__restrict__ s,
    uint32_t *__restrict__ tb_dst, uint16_t *__restrict__ pl, double *__restrict__ ctb,
    double *__restrict__ stb)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nt; k += stride) {
        const uint32_t t = tb_src[k];
        if (k < nvalid) {
            tb_dst[t] = (uint32_t)k;
            const int32_t px = pix[t];
            uint16_t w = p0 ? (uint16_t)(px - p0[tile_of(px, tp, p0, ntiles)]) : (uint16_t)(px % tp);
            if (POL > 1) {
                if (HALF) {
                    const double cv = c[t], sv = s[t];
                    if (cv < 0.0) {
                        w |= 0x8000;
                        ctb[k] = sv / (1.0 - cv);
                    } else {
                        ctb[k] = sv / (1.0 + cv);
                    }
                } else {
                    ctb[k] = c[t];
                    stb[k] = s[t];
                }
            }
            pl[k] = w;
        } else {
            tb_dst[t] = kInvalidSample;      // flagged samples sort behind every tile
        }
    }
}

// (pixel in tile, cos, sin) of TB sample k in either storage form
template <int POL, bool HALF>
__device__ __forceinline__ void tile_sample(const uint16_t *__restrict__ pl,
                                            const double *__restrict__ c,
                                            const double *__restrict__ s, int64_t k, int &q,
                                            double &cc, double &ss)
{
    const uint16_t w = pl[k];
    if (POL == 1) {
        q = w;
        cc = ss = 0.0;
    } else if (HALF) {
        q = w & 0x7FFF;
        const double h = c[k], h2 = h * h, inv = 1.0 / (1.0 + h2);
        cc = (1.0 - h2) * inv;
        ss = (h + h) * inv;
        if (w & 0x8000) cc = -cc;
    } else {
        q = w;
        cc = c[k];
        ss = s[k];
    }
}

constexpr int kSegBatch = 128;           // segments of a work item whose bounds are staged in LDS at a time

// ------------------------------------------------------------------ P (TB) ------
// one workgroup per work item = (tile, spans [sp0, sp1) of its samples, clipped to the addresses
// [k0, k1)): the tile of x is staged once, the item's segments (one per span, ~2000 consecutive
// addresses each; the whole bucket when the plan has one span) are streamed one after the other
template <int POL, bool HALF>
__global__ __launch_bounds__(1024) void k_P_tiles(
    const int64_t *__restrict__ tile_p0, const int32_t *__restrict__ item_tile,
    const int2 *__restrict__ item_span, const int64_t *__restrict__ seg_off, int64_t ntiles,
    const int64_t *__restrict__ item_k0, const int64_t *__restrict__ item_k1,
    const uint16_t *__restrict__ pl, const double *__restrict__ c, const double *__restrict__ s,
    const double *__restrict__ x, double *__restrict__ d_tb)
{
    extern __shared__ double tile[];                    // tp*POL doubles
    const int b = item_tile[blockIdx.x];
    const int64_t p0 = tile_p0[b];
    const int64_t np = tile_p0[b + 1] - p0;
    const int64_t nvals = np * POL;
    const double *xs = x + p0 * POL;
    for (int64_t i = threadIdx.x; i < nvals; i += blockDim.x) tile[i] = xs[i];
    __syncthreads();
    // The item's segments as ONE index space: their bounds are fetched together (a dependent load per
    // segment in front of its samples cost 12 % at C4, and a raster scan -- most segments of a tile
    // empty -- 10 %), scanned in LDS, and sample j of the item is sample j - pre[i] of segment i: no
    // partly filled pass at the end of every segment.  Batches of kSegBatch segments.
    __shared__ int64_t seg_k0[kSegBatch];
    __shared__ uint32_t seg_pre[kSegBatch + 1];
    const int2 spans = item_span[blockIdx.x];
    const int64_t c0 = item_k0[blockIdx.x], c1 = item_k1[blockIdx.x];
    if (spans.y - spans.x == 1) {                        // one segment (always, on the global tile order)
        const int64_t a0 = seg_off[(int64_t)spans.x * ntiles + b], a1 = seg_off[(int64_t)spans.x * ntiles + b + 1];
        const int64_t k0 = a0 > c0 ? a0 : c0, k1 = a1 < c1 ? a1 : c1;
        for (int64_t k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
            int q;
            double cc, ss;
            tile_sample<POL, HALF>(pl, c, s, k, q, cc, ss);
            double r = 0.0;
            if (POL == 1) {
                r += tile[q];
            } else if (POL == 2) {
                r += tile[2 * q] * cc + tile[2 * q + 1] * ss;
            } else {
                r += tile[3 * q] + tile[3 * q + 1] * cc + tile[3 * q + 2] * ss;
            }
            d_tb[k] = r;
        }
        return;
    }
    for (int sp0 = spans.x; sp0 < spans.y; sp0 += kSegBatch) {
        const int ns = spans.y - sp0 < kSegBatch ? spans.y - sp0 : kSegBatch;
        if ((int)threadIdx.x < ns) {
            const int64_t a0 = seg_off[(int64_t)(sp0 + threadIdx.x) * ntiles + b];
            const int64_t a1 = seg_off[(int64_t)(sp0 + threadIdx.x) * ntiles + b + 1];
            const int64_t k0 = a0 > c0 ? a0 : c0, k1 = a1 < c1 ? a1 : c1;
            seg_k0[threadIdx.x] = k0;
            seg_pre[threadIdx.x + 1] = (uint32_t)(k1 > k0 ? k1 - k0 : 0);
        }
        if (threadIdx.x == 0) seg_pre[0] = 0;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < ns; ++i) seg_pre[i + 1] += seg_pre[i];
        __syncthreads();
        const uint32_t total = seg_pre[ns];
        int si = 0;
        for (uint32_t j = threadIdx.x; j < total; j += blockDim.x) {
            while (j >= seg_pre[si + 1]) ++si;
            const int64_t k = seg_k0[si] + (int64_t)(j - seg_pre[si]);
            int q;
            double cc, ss;
            tile_sample<POL, HALF>(pl, c, s, k, q, cc, ss);
            double r = 0.0;
            if (POL == 1) {
                r += tile[q];
            } else if (POL == 2) {
                r += tile[2 * q] * cc + tile[2 * q + 1] * ss;
            } else {
                r += tile[3 * q] + tile[3 * q + 1] * cc + tile[3 * q + 2] * ss;
            }
            d_tb[k] = r;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- P^T (TB) ------
template <int POL, bool HALF>
__global__ __launch_bounds__(1024) void k_Pt_tiles(
    const int64_t *__restrict__ tile_p0, const int32_t *__restrict__ item_tile,
    const int2 *__restrict__ item_span, const int64_t *__restrict__ seg_off, int64_t ntiles,
    const int64_t *__restrict__ item_k0, const int64_t *__restrict__ item_k1,
    const uint16_t *__restrict__ pl, const double *__restrict__ c, const double *__restrict__ s,
    const double *__restrict__ v_tb, double *__restrict__ out)
{
    extern __shared__ double tile[];                    // tp*POL accumulators
    const int b = item_tile[blockIdx.x];
    const int64_t p0 = tile_p0[b];
    const int64_t np = tile_p0[b + 1] - p0;
    const int64_t nvals = np * POL;
    for (int64_t i = threadIdx.x; i < nvals; i += blockDim.x) tile[i] = 0.0;
    __syncthreads();
    __shared__ int64_t seg_b0[kSegBatch], seg_b1[kSegBatch];
    const int2 spans = item_span[blockIdx.x];
    const int64_t clip0 = item_k0[blockIdx.x], clip1 = item_k1[blockIdx.x];
    for (int sp = spans.x; sp < spans.y; ++sp) {
    // (the segment bounds of a batch are fetched together, not one dependent load per segment)
    if ((sp - spans.x) % kSegBatch == 0) {
        __syncthreads();
        const int ns = spans.y - sp < kSegBatch ? spans.y - sp : kSegBatch;
        if ((int)threadIdx.x < ns) {
            seg_b0[threadIdx.x] = seg_off[(int64_t)(sp + threadIdx.x) * ntiles + b];
            seg_b1[threadIdx.x] = seg_off[(int64_t)(sp + threadIdx.x) * ntiles + b + 1];
        }
        __syncthreads();
    }
    const int64_t a0 = seg_b0[(sp - spans.x) % kSegBatch], a1 = seg_b1[(sp - spans.x) % kSegBatch];
    const int64_t k0 = a0 > clip0 ? a0 : clip0, k1 = a1 < clip1 ? a1 : clip1;
    constexpr int U = 4;                     // independent loads in flight per thread
    int64_t k = k0 + threadIdx.x;
    for (; k + (U - 1) * (int64_t)blockDim.x < k1; k += U * (int64_t)blockDim.x) {
        int q[U];
        double v[U], cc[U], ss[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t kk = k + u * (int64_t)blockDim.x;
            v[u] = v_tb[kk];
            tile_sample<POL, HALF>(pl, c, s, kk, q[u], cc[u], ss[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (POL == 1) {
                atomicAdd(&tile[q[u]], v[u]);
            } else if (POL == 2) {
                atomicAdd(&tile[2 * q[u]], v[u] * cc[u]);
                atomicAdd(&tile[2 * q[u] + 1], v[u] * ss[u]);
            } else {
                atomicAdd(&tile[3 * q[u]], v[u]);
                atomicAdd(&tile[3 * q[u] + 1], v[u] * cc[u]);
                atomicAdd(&tile[3 * q[u] + 2], v[u] * ss[u]);
            }
        }
    }
    for (; k < k1; k += blockDim.x) {
        int q;
        double c1, s1;
        tile_sample<POL, HALF>(pl, c, s, k, q, c1, s1);
        const double v = v_tb[k];
        if (POL == 1) {
            atomicAdd(&tile[q], v);
        } else if (POL == 2) {
            atomicAdd(&tile[2 * q], v * c1);
            atomicAdd(&tile[2 * q + 1], v * s1);
        } else {
            atomicAdd(&tile[3 * q], v);
            atomicAdd(&tile[3 * q + 1], v * c1);
            atomicAdd(&tile[3 * q + 2], v * s1);
        }
    }
    }
    __syncthreads();
    double *o = out + p0 * POL;
    for (int64_t i = threadIdx.x; i < nvals; i += blockDim.x) atomicAdd(&o[i], tile[i]);
}

// ------------------------------------------------------- time <-> TB order ------
// each workgroup owns ONE contiguous time range, so that the K sequential tile streams
// it touches stay in its XCD's L2 until their lines are complete
__global__ __launch_bounds__(1024) void k_time_to_tiles(int64_t nt, int64_t chunk,
                                                         const uint32_t *__restrict__ tb_dst,
                                                         const double *__restrict__ in,
                                                         double *__restrict__ out_tb)
{
    const int64_t t0 = (int64_t)blockIdx.x * chunk;
    int64_t t1 = t0 + chunk;
    if (t1 > nt) t1 = nt;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const uint32_t k = tb_dst[t];
        if (k != kInvalidSample) out_tb[k] = in[t];
    }
}

__global__ __launch_bounds__(1024) void k_tiles_to_time(int64_t nt, int64_t chunk,
                                                         const uint32_t *__restrict__ tb_dst,
                                                         const double *__restrict__ in_tb,
                                                         double *__restrict__ out)
{
    const int64_t t0 = (int64_t)blockIdx.x * chunk;
    int64_t t1 = t0 + chunk;
    if (t1 > nt) t1 = nt;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const uint32_t k = tb_dst[t];
        out[t] = (k != kInvalidSample) ? in_tb[k] : 0.0;
    }
}

// Windowed forms of the two permutations.  A per-sample scatter / gather between the orders
// moves 8-byte fragments (a window's samples of one tile are ~20 contiguous entries) and the
// caches write lines back before they are complete: 3.2 GB of HBM writes for 0.8 GB of
// payload.  Here a workgroup owns kPermWin consecutive time samples, reaches their TB
// positions through a list sorted by address (one contiguous run per tile) and stages the
// window in LDS, so that both sides of the copy are streams: 6 + 8 + 8 bytes per sample.
constexpr int kPermWin = 8192, kPermT = 256, kPermPer = kPermWin / kPermT;

template <bool TO_TIME>
__global__ __launch_bounds__(kPermT) void k_perm_windows(int64_t nt, int64_t nwin,
                                                          const uint32_t *__restrict__ lst_k,
                                                          const uint16_t *__restrict__ lst_q,
                                                          const double *__restrict__ in,
                                                          double *__restrict__ out)
{
    __shared__ double win[kPermWin];
    const int per_xcd = (int)((nwin + 7) / 8);
    const int64_t w = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (w >= nwin) return;
    const int64_t t0 = w * kPermWin, base = w * kPermWin;
    const int span = (int)((nt - t0 < kPermWin) ? nt - t0 : kPermWin);
    const int t = threadIdx.x;
    uint32_t kk[kPermPer];
    uint16_t qq[kPermPer];
#pragma unroll
    for (int u = 0; u < kPermPer; ++u) {
        kk[u] = lst_k[base + t + u * kPermT];
        qq[u] = lst_q[base + t + u * kPermT];
    }
    if (TO_TIME) {
        double vv[kPermPer];
#pragma unroll
        for (int u = 0; u < kPermPer; ++u) vv[u] = (kk[u] != kInvalidSample) ? in[kk[u]] : 0.0;
        for (int j = t; j < span; j += kPermT) win[j] = 0.0;      // flagged samples read as 0
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPermPer; ++u)
            if (kk[u] != kInvalidSample) win[qq[u]] = vv[u];
        __syncthreads();
        for (int j = t; j < span; j += kPermT) out[t0 + j] = win[j];
    } else {
        for (int j = t; j < span; j += kPermT) win[j] = in[t0 + j];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPermPer; ++u)
            if (kk[u] != kInvalidSample) out[kk[u]] = win[qq[u]];
    }
}

__global__ __launch_bounds__(256) void k_perm_keys(int64_t nt, int64_t nwin,
                                                    const uint32_t *__restrict__ tb_dst,
                                                    uint64_t *__restrict__ keys,
                                                    uint16_t *__restrict__ vals)
{
    const int64_t total = nwin * kPermWin;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t k = g < nt ? tb_dst[g] : kInvalidSample;
        keys[g] = ((uint64_t)(g / kPermWin) << 32) | (uint64_t)k;
        vals[g] = (uint16_t)(g % kPermWin);
    }
}

__global__ __launch_bounds__(256) void k_perm_unpack(int64_t total, const uint64_t *__restrict__ keys,
                                                      uint32_t *__restrict__ lst_k)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride)
        lst_k[g] = (uint32_t)(keys[g] & 0xFFFFFFFFull);
}

// workgroup size of the two tile kernels: 1024 threads for the gather (0.355 vs 0.38 ms at
// 1e8 samples), 512 for the scatter (no difference); CM2_TILE_BLOCK = 256 / 512 / 1024 sets both
static int tile_block(bool gather = false)
{
    static int forced = -1;
    if (forced < 0) {
        forced = 0;
        if (const char *e = getenv("CM2_TILE_BLOCK")) {
            const int v = atoi(e);
            if (v == 256 || v == 512 || v == 1024) forced = v;
        }
    }
    return forced ? forced : (gather ? 1024 : 512);
}

// ------------------------------------------------------------------ C ABI -------
extern "C" int cm2_tiles_destroy(cm2_tiles *t);

extern "C" int cm2_tiles_destroy(cm2_tiles *t)
{
    if (!t) return 0;
    void *ptrs[] = {t->d_tb_dst, t->d_pl, t->d_cos, t->d_sin, t->d_half, t->d_item_tile, t->d_item_span,
                    t->d_item_k0, t->d_item_k1, t->d_perm_k, t->d_perm_q, t->d_seg_off, t->d_tile_p0};
    for (void *q : ptrs)
        if (q) (void)cm2::dev_free(q);
    cm2::fx_free(t);
    delete t;
    return 0;
}

// The tile order cut in time ([span][tile][time], rounds 4-5) is GONE as an option: measured again in round 5
// with spans sized for the Infinity Cache (profiles/r05_spans_removed.md: C4 step 1.41 ms on the global order
// against 1.43 / 1.48 / 1.57 with spans of 9.4e5 / 1.7e7 / 8.4e6 samples; C5 share 1.99 against 2.05-2.06; at
// C5 whole N^-1 did not move at all) it never paid, and round 4's switch (CM2_TILE_SPAN), its automatic span
// length and its 71 tests were removed.  What remains of it is the GENERAL form of the tables the kernels walk --
// segment (span, tile) with ONE span: a tile's segment is its whole bucket, `seg_off` is the tile offsets,
// every work item is one tile's address range -- which costs nothing and is what the builders were tested on.
extern "C" int cm2_tiles_create(cm2_tiles **out, const int32_t *d_pix, const double *d_cos,
                                const double *d_sin, int64_t nt, int64_t npix, int pol,
                                int tile_pixels, int64_t slice_samples, void *stream_)
{
    CM2_CHECK(out != nullptr, "cm2_tiles_create: out is NULL");
    *out = nullptr;
    CM2_CHECK(pol == 1 || pol == 2 || pol == 3, "cm2_tiles_create: bad pol=%d", pol);
    CM2_CHECK(pol == 1 || (d_cos && d_sin), "cm2_tiles_create: cos/sin required for pol=%d", pol);
    CM2_CHECK(nt > 0 && nt < (int64_t)0xFFFFFFFF, "cm2_tiles_create: nt=%lld out of range",
              (long long)nt);
    CM2_CHECK(tile_pixels >= 64 && tile_pixels <= 65536 && tile_pixels * pol * 8 <= 160 * 1024 - 1024,
              "cm2_tiles_create: tile of %d pixels does not fit LDS / uint16", tile_pixels);
    CM2_CHECK(slice_samples >= 256, "cm2_tiles_create: slice too short");
    hipStream_t stream = as_stream(stream_);
    cm2_tiles *t = new cm2_tiles();
    struct Guard { cm2_tiles *t; ~Guard() { if (t) cm2_tiles_destroy(t); } } guard{t};
    t->nt = nt; t->npix = npix; t->pol = pol; t->tp = tile_pixels;
    t->ntiles = (npix + tile_pixels - 1) / tile_pixels;
    t->plan_id = next_plan_id();
    // order of the per-pixel sums of P^T: fixed (time order, reproducible; default) or atomic
    if (const char *e = getenv("CM2_PT_ORDER"))          // atomic | exact | fixed (default)
        t->pt_fixed = !strcmp(e, "atomic") ? 0 : (!strcmp(e, "exact") ? 2 : 1);
    if (cm2::exact_order_setting() >= 0 && t->pt_fixed) t->pt_fixed = cm2::exact_order_setting() ? 2 : 1;

    // Stable partition of the samples by tile.  Default: the multisplit above (k_tile_rank, one
    // scan, k_tile_place).  With more tiles than its LDS histograms hold, or CM2_TILE_BUILD=sort:
    // a radix sort of (tile, time) pairs and a gather (k_tile_fill).  Same addresses either way.
    bool use_sort = false;
    if (const char *e = getenv("CM2_TILE_BUILD")) use_sort = strcmp(e, "sort") == 0;
    DevTemp<uint32_t> keys_in, keys_out, vals_in, tb_src;      // sort path
    DevTemp<uint32_t> packed, cnt_t;                           // multisplit: tile << 16 | rank; counts -> bases
    const int64_t nchunks = (nt + kSplitChunk - 1) / kSplitChunk;
    int64_t G = nchunks;                                       // chunks per span (nchunks: one span)
    DevTemp<int64_t> d_off;
    DevTemp<char> d_temp;
    DevTemp<unsigned int> d_bad;
    CM2_HIP(d_bad.alloc(1));
    CM2_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), stream));
    std::vector<int64_t> off;
    bool sorted = false;                                       // which path the last partition took
    auto partition_sort = [&](const int64_t *d_p0) -> int {
        if (!keys_in.p) {
            CM2_HIP(keys_in.alloc(nt));
            CM2_HIP(keys_out.alloc(nt));
            CM2_HIP(vals_in.alloc(nt));
            CM2_HIP(tb_src.alloc(nt));
        }
        k_tile_keys<<<grid_for(nt), kBlock, 0, stream>>>(d_pix, nt, tile_pixels, d_p0,
                                                         (uint32_t)t->ntiles, npix, keys_in, vals_in,
                                                         d_bad);
        CM2_LAUNCH_OK();
        int end_bit = 1;
        while (((int64_t)1 << end_bit) <= t->ntiles) ++end_bit;
        size_t tb = 0;
        CM2_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in.p, keys_out.p, vals_in.p,
                                                   tb_src.p, nt, 0, end_bit, stream));
        d_temp.release();
        CM2_HIP(d_temp.alloc(tb + 16));
        CM2_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp.p, tb, keys_in.p, keys_out.p, vals_in.p,
                                                   tb_src.p, nt, 0, end_bit, stream));
        k_tile_bounds<<<(int)((t->ntiles + 1 + kBlock - 1) / kBlock), kBlock, 0, stream>>>(
            keys_out, nt, t->ntiles, d_off);
        CM2_LAUNCH_OK();
        return 0;
    };
    auto partition_split = [&](const int64_t *d_p0) -> int {
        const int64_t ncnt = t->nspans * t->ntiles * G + 1;    // (+1: the scan's last word = nvalid)
        if (!packed.p) CM2_HIP(packed.alloc(nt));
        cnt_t.release();
        CM2_HIP(cnt_t.alloc(ncnt));
        // (the last span is padded to G chunks: the counts of chunks that do not exist stay 0)
        CM2_HIP(hipMemsetAsync(cnt_t.p, 0, sizeof(uint32_t) * ncnt, stream));
        const size_t lds = sizeof(uint16_t) * 4 * (size_t)t->ntiles;
        static size_t granted[64] = {0};
        CM2_HIP(ensure_dynamic_lds((const void *)k_tile_rank, lds, granted));
        k_tile_rank<<<(unsigned)((nchunks + 3) / 4), 256, lds, stream>>>(
            d_pix, nt, tile_pixels, d_p0, (uint32_t)t->ntiles, npix, nchunks, G, packed, cnt_t, d_bad);
        CM2_LAUNCH_OK();
        CM2_CHECK(ncnt < ((int64_t)1 << 31), "cm2_tiles_create: %lld tile x chunk counts", (long long)ncnt);
        size_t tb = 0;
        CM2_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt_t.p, cnt_t.p, (int)ncnt, stream));
        d_temp.release();
        CM2_HIP(d_temp.alloc(tb + 16));
        CM2_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp.p, tb, cnt_t.p, cnt_t.p, (int)ncnt, stream));
        const int64_t nsegs = t->nspans * t->ntiles;
        k_tile_offsets<<<(int)((nsegs + 1 + kBlock - 1) / kBlock), kBlock, 0, stream>>>(cnt_t, G, nsegs, d_off);
        CM2_LAUNCH_OK();
        return 0;
    };
    // `seg`: first address of every segment (span, tile); `off`: samples of every tile, as offsets
    std::vector<int64_t> seg;
    auto partition = [&](const int64_t *d_p0) -> int {
        d_off.release();
        sorted = use_sort || t->ntiles > kSplitMaxTiles || t->ntiles * nchunks + 1 >= ((int64_t)1 << 31);
        G = nchunks;                                        // one span: the global tile order
        t->nspans = (nchunks + G - 1) / G;
        if (t->nspans < 1) t->nspans = 1;
        t->span_samples = G * kSplitChunk;
        const int64_t nsegs = t->nspans * t->ntiles;
        CM2_HIP(d_off.alloc(nsegs + 1));
        if (int rc = sorted ? partition_sort(d_p0) : partition_split(d_p0)) return rc;
        seg.assign((size_t)nsegs + 1, 0);
        CM2_HIP(cm2::download(seg.data(), d_off, sizeof(int64_t) * (nsegs + 1), stream));
        CM2_HIP(hipStreamSynchronize(stream));
        off.assign((size_t)t->ntiles + 1, 0);
        for (int64_t b = 0; b < t->ntiles; ++b) {
            int64_t n = 0;
            for (int64_t sp = 0; sp < t->nspans; ++sp)
                n += seg[(size_t)(sp * t->ntiles + b + 1)] - seg[(size_t)(sp * t->ntiles + b)];
            off[(size_t)b + 1] = off[(size_t)b] + n;
        }
        return 0;
    };
    if (int rc = partition(nullptr)) return rc;
    unsigned int h_bad = 0;
    CM2_HIP(cm2::download(&h_bad, d_bad, sizeof(h_bad), nullptr));
    CM2_CHECK(h_bad == 0, "cm2_tiles_create: a pixel index is outside [-1, npix=%lld)",
              (long long)npix);
    t->tile_p0.assign((size_t)t->ntiles + 1, 0);
    for (int64_t b = 0; b <= t->ntiles; ++b)
        t->tile_p0[(size_t)b] = b * tile_pixels < npix ? b * tile_pixels : npix;
    // The fixed-order P^T gives every tile to ONE workgroup: a hit map that is far from uniform
    // (half of the samples on a tenth of the sky: 0.47 -> 1.7 ms) would leave most of the chip
    // waiting for the heaviest tiles.  When some uniform tile holds over 25 % more than the mean,
    // the pixel ranges are re-cut to equal sample counts (width <= tile_pixels): every pixel is
    // still summed by one workgroup in time order, so results do not change by a bit.
    // Round 4: by default such a hit map keeps the uniform tile width instead, and the fixed-order P^T
    // shares the slices of its heavy tiles out to several workgroups (pt_split; cm2_tiles.h, "PARTS"):
    // narrow dense tiles see many hits per pixel and slice (runs of 5-8 list entries in two level
    // passes) and halve the address runs of the overlap-save kernel.  Only a pixel heavy enough for
    // the hot-tile path is cut out as a tile of its own.  CM2_TILE_BALANCE: 0 = uniform tiles, one
    // workgroup each; 1 / cut = the equal-load cut (also chosen when the exact summation order is
    // asked for: it does not change a bit); parts = split even when the hit map is even.
    const char *bal = getenv("CM2_TILE_BALANCE");
    bool balance = false, split = false;
    double mean_load = 0.0;
    {
        int64_t nmax = 0;
        for (int64_t b = 0; b < t->ntiles; ++b)
            if (off[(size_t)b + 1] - off[(size_t)b] > nmax) nmax = off[(size_t)b + 1] - off[(size_t)b];
        mean_load = (double)off[(size_t)t->ntiles] / (double)(t->ntiles > 0 ? t->ntiles : 1);
        const bool uneven = t->ntiles >= 64 && off[(size_t)t->ntiles] >= (1 << 20) && (double)nmax > 1.25 * mean_load;
        const bool some = off[(size_t)t->ntiles] > 0;
        if (!bal) {
            balance = uneven && t->pt_fixed == 2;      // (CM2_PT_ORDER=exact / cm2_set_exact_order(1), read above)
            split = uneven && !balance;
        } else if (strcmp(bal, "parts") == 0) {
            split = some;
        } else if (strcmp(bal, "cut") == 0 || atoi(bal) != 0) {
            balance = some;
        }
    }
    t->pt_split = split;
    DevTemp<int64_t> d_p0;
    if (balance) {
        DevTemp<unsigned int> d_hits;
        CM2_HIP(d_hits.alloc(npix));
        CM2_HIP(hipMemsetAsync(d_hits, 0, sizeof(unsigned int) * npix, stream));
        k_pix_hist<<<grid_for(nt), kBlock, 0, stream>>>(d_pix, nt, npix, d_hits);
        CM2_LAUNCH_OK();
        std::vector<unsigned int> hits((size_t)npix);
        CM2_HIP(cm2::download(hits.data(), d_hits, sizeof(unsigned int) * npix, stream));
        CM2_HIP(hipStreamSynchronize(stream));
        // as many tiles as before would have had at equal load, 2 % slack so that rounding does
        // not spill a 513th tile; a tile ends when the next pixel would exceed the target or the
        // width tile_pixels (a single pixel heavier than the target is a tile of its own)
        const int64_t nvalid = off[(size_t)t->ntiles];
        // cuts every rank of a sharded run has in common, whatever its own hit map: the group
        // boundaries of cm2_tiles_group_tiles (the pieces of the map that are all-reduced while
        // the next piece is back-projected) -- the uniform tiling's tile boundaries nearest to
        // eighths of the map, a function of npix and tile_pixels alone
        const int64_t ntu = (npix + tile_pixels - 1) / tile_pixels;
        int64_t forced[9];
        for (int c = 0; c <= 8; ++c) {
            forced[c] = (ntu * c / 8) * tile_pixels;
            if (forced[c] > npix || c == 8) forced[c] = npix;
        }
        // The fixed-order P^T keeps 512 workgroups resident (two per CU): tiles of equal load
        // finish in whole rounds of 512, so 737 tiles cost as much as 1024.  The width limit makes
        // a sparse region take more tiles than its load asks for; the target load is therefore
        // lowered (n = 1, 2, 3, 4 times the uniform count) until the cut fits n x the uniform
        // count, with 2 % slack so that rounding does not spill one more tile; a tile ends when the
        // next pixel would exceed the target or the width tile_pixels (a single pixel heavier than
        // the target is a tile of its own).
        std::vector<int64_t> p0v;
        const int64_t base = t->ntiles;
        for (int mult = 1; mult <= 4; ++mult) {
            const int64_t target = (int64_t)(1.02 * (double)nvalid / (double)(base * mult)) + 1;
            int fc = 1;
            p0v.assign(1, 0);
            int64_t acc = 0, start = 0;
            for (int64_t p = 0; p < npix; ++p) {
                const int64_t h = hits[(size_t)p];
                while (fc < 8 && forced[fc] < p) ++fc;
                const bool at_cut = fc < 8 && forced[fc] == p;
                if (p > start && (acc + h > target || p - start >= tile_pixels || at_cut)) {
                    p0v.push_back(p);
                    start = p;
                    acc = 0;
                }
                acc += h;
            }
            p0v.push_back(npix);
            if ((int64_t)p0v.size() - 1 <= base * mult || base % 512 != 0) break;
        }
        t->ntiles = (int64_t)p0v.size() - 1;
        t->tile_p0 = p0v;
        CM2_HIP(d_p0.alloc(p0v.size()));
        CM2_HIP(cm2::upload(d_p0.p, p0v.data(), sizeof(int64_t) * p0v.size(), nullptr));
        if (int rc = partition(d_p0.p)) return rc;
    }
    if (split) {
        int64_t nmax = 0;
        for (int64_t b = 0; b < t->ntiles; ++b)
            if (off[(size_t)b + 1] - off[(size_t)b] > nmax) nmax = off[(size_t)b + 1] - off[(size_t)b];
        const int64_t hot_min = (int64_t)(0.5 * mean_load) > kHotTileMin ? (int64_t)(0.5 * mean_load) : kHotTileMin;
        if (nmax >= hot_min) {                              // (otherwise no pixel can be that heavy)
            DevTemp<unsigned int> d_hits;
            CM2_HIP(d_hits.alloc(npix));
            CM2_HIP(hipMemsetAsync(d_hits, 0, sizeof(unsigned int) * npix, stream));
            k_pix_hist<<<grid_for(nt), kBlock, 0, stream>>>(d_pix, nt, npix, d_hits);
            CM2_LAUNCH_OK();
            std::vector<unsigned int> hits((size_t)npix);
            CM2_HIP(cm2::download(hits.data(), d_hits, sizeof(unsigned int) * npix, stream));
            CM2_HIP(hipStreamSynchronize(stream));
            std::vector<int64_t> p0v(1, 0);
            bool any = false;
            for (int64_t p = 1; p < npix; ++p) {
                const bool hot_here = (int64_t)hits[(size_t)p] >= hot_min, hot_before = (int64_t)hits[(size_t)p - 1] >= hot_min;
                if (p % tile_pixels == 0 || hot_here || hot_before) p0v.push_back(p);
                any = any || hot_here || hot_before;
            }
            p0v.push_back(npix);
            if (any) {
                t->ntiles = (int64_t)p0v.size() - 1;
                t->tile_p0 = p0v;
                CM2_HIP(d_p0.alloc(p0v.size()));
                CM2_HIP(cm2::upload(d_p0.p, p0v.data(), sizeof(int64_t) * p0v.size(), nullptr));
                if (int rc = partition(d_p0.p)) return rc;
                balance = true;                             // (the kernels below read the re-cut boundaries)
            }
        }
    }
    t->balanced = balance;
    CM2_HIP(cm2::dev_malloc(&t->d_tile_p0, sizeof(int64_t) * (t->ntiles + 1)));
    CM2_HIP(cm2::upload(t->d_tile_p0, t->tile_p0.data(), sizeof(int64_t) * (t->ntiles + 1), nullptr));
    t->nvalid = off[t->ntiles];
    auto publish_segments = [&]() -> int {
        t->tile_count.assign((size_t)t->ntiles, 0);
        for (int64_t b = 0; b < t->ntiles; ++b) t->tile_count[(size_t)b] = off[(size_t)b + 1] - off[(size_t)b];
        t->seg_off = seg;
        if (t->d_seg_off) (void)cm2::dev_free(t->d_seg_off);
        t->d_seg_off = nullptr;
        CM2_HIP(cm2::dev_malloc(&t->d_seg_off, sizeof(int64_t) * seg.size()));
        CM2_HIP(hipMemcpyAsync(t->d_seg_off, d_off, sizeof(int64_t) * seg.size(), hipMemcpyDeviceToDevice, stream));
        return 0;
    };
    if (int rc = publish_segments()) return rc;

    const int64_t nv = t->nvalid > 0 ? t->nvalid : 1;
    CM2_HIP(cm2::dev_malloc(&t->d_tb_dst, sizeof(uint32_t) * nt));
    CM2_HIP(cm2::dev_malloc(&t->d_pl, sizeof(uint16_t) * nv));
    // half-angle storage (one double per sample instead of cos and sin) when every pair lies
    // on the unit circle; CM2_TILE_ANGLES=full keeps both arrays
    t->half = false;
    if (pol > 1 && nt > 0 && tile_pixels <= 0x8000) {
        const char *e = getenv("CM2_TILE_ANGLES");
        if (!(e && strcmp(e, "full") == 0)) {
            DevTemp<unsigned int> d_off_circle;
            CM2_HIP(d_off_circle.alloc(1));
            CM2_HIP(hipMemsetAsync(d_off_circle, 0, sizeof(unsigned int), stream));
            k_unit_circle<<<grid_for(nt), kBlock, 0, stream>>>(nt, d_cos, d_sin, d_off_circle);
            CM2_LAUNCH_OK();
            unsigned int h_off = 0;
            CM2_HIP(cm2::download(&h_off, d_off_circle, sizeof(h_off), stream));
            CM2_HIP(hipStreamSynchronize(stream));
            t->half = (h_off == 0);
        }
    }
    if (pol > 1) {
        if (t->half) {
            CM2_HIP(cm2::dev_malloc(&t->d_half, sizeof(double) * nv));
        } else {
            CM2_HIP(cm2::dev_malloc(&t->d_cos, sizeof(double) * nv));
            CM2_HIP(cm2::dev_malloc(&t->d_sin, sizeof(double) * nv));
        }
    }
    auto place = [&]() -> int {
#define CM2_TF(POL, HALF)                                                                      \
    do {                                                                                       \
        if (sorted)                                                                            \
            k_tile_fill<POL, HALF><<<grid_for(nt), kBlock, 0, stream>>>(                       \
                nt, t->nvalid, tile_pixels, balance ? t->d_tile_p0 : nullptr,                  \
                (uint32_t)t->ntiles, tb_src, d_pix, d_cos, d_sin, t->d_tb_dst, t->d_pl,        \
                HALF ? t->d_half : t->d_cos, t->d_sin);                                        \
        else if ((HALF || POL == 1) && t->ntiles <= 4096) {                                    \
            const size_t lds = (POL > 1 ? sizeof(double) * kSplitChunk : 0) +                  \
                               sizeof(uint32_t) * (2 * (size_t)t->ntiles + 1) +                \
                               sizeof(uint16_t) * 2 * kSplitChunk;                             \
            static size_t granted[64] = {0};                                                   \
            CM2_HIP(ensure_dynamic_lds((const void *)k_tile_place_staged<POL>, lds, granted));  \
            k_tile_place_staged<POL><<<(unsigned)nchunks, 256, lds, stream>>>(                 \
                nt, tile_pixels, balance ? t->d_tile_p0 : nullptr, G, (int)t->ntiles,          \
                packed, cnt_t, d_pix, d_cos, d_sin, t->d_tb_dst, t->d_pl, t->d_half);          \
        } else                                                                                 \
            k_tile_place<POL, HALF><<<grid_for(nt), kBlock, 0, stream>>>(                      \
                nt, tile_pixels, balance ? t->d_tile_p0 : nullptr, t->ntiles, G, packed, cnt_t, \
                d_pix, d_cos, d_sin, t->d_tb_dst, t->d_pl, HALF ? t->d_half : t->d_cos,        \
                t->d_sin);                                                                     \
    } while (0)
    if (pol == 1) CM2_TF(1, false);
    else if (pol == 2) { if (t->half) CM2_TF(2, true); else CM2_TF(2, false); }
    else { if (t->half) CM2_TF(3, true); else CM2_TF(3, false); }
#undef CM2_TF
    CM2_LAUNCH_OK();
    return 0;
    };
    if (int rc = place()) return rc;
    // work items: a tile's segments in span order, gathered until they hold >= slice_samples samples; a
    // segment longer than that (one span: the whole bucket) is cut into address ranges
    std::vector<int32_t> it_tile;
    std::vector<int2> it_span;
    std::vector<int64_t> it_k0, it_k1;
    t->tile_item0.assign((size_t)t->ntiles + 1, 0);
    for (int64_t b = 0; b < t->ntiles; ++b) {
        t->tile_item0[(size_t)b] = (int64_t)it_tile.size();
        int64_t sp = 0;
        while (sp < t->nspans) {
            const int64_t a0 = seg[(size_t)(sp * t->ntiles + b)], a1 = seg[(size_t)(sp * t->ntiles + b + 1)];
            if (a1 - a0 > slice_samples) {
                for (int64_t k = a0; k < a1; k += slice_samples) {
                    it_tile.push_back((int32_t)b);
                    it_span.push_back(make_int2((int)sp, (int)sp + 1));
                    it_k0.push_back(k);
                    it_k1.push_back(k + slice_samples < a1 ? k + slice_samples : a1);
                }
                ++sp;
                continue;
            }
            int64_t n = 0, e = sp;
            while (e < t->nspans) {
                const int64_t len = seg[(size_t)(e * t->ntiles + b + 1)] - seg[(size_t)(e * t->ntiles + b)];
                if (len > slice_samples || (n > 0 && n + len > slice_samples)) break;
                n += len;
                ++e;
            }
            if (n > 0) {
                it_tile.push_back((int32_t)b);
                it_span.push_back(make_int2((int)sp, (int)e));
                it_k0.push_back(0);
                it_k1.push_back(INT64_MAX);
            }
            sp = e;
        }
    }
    t->tile_item0[(size_t)t->ntiles] = (int64_t)it_tile.size();
    t->nitems = (int64_t)it_tile.size();
    const int64_t ni = t->nitems > 0 ? t->nitems : 1;
    CM2_HIP(cm2::dev_malloc(&t->d_item_tile, sizeof(int32_t) * ni));
    CM2_HIP(cm2::dev_malloc(&t->d_item_span, sizeof(int2) * ni));
    CM2_HIP(cm2::dev_malloc(&t->d_item_k0, sizeof(int64_t) * ni));
    CM2_HIP(cm2::dev_malloc(&t->d_item_k1, sizeof(int64_t) * ni));
    if (t->nitems) {
        CM2_HIP(cm2::upload(t->d_item_tile, it_tile.data(), sizeof(int32_t) * ni, nullptr));
        CM2_HIP(cm2::upload(t->d_item_span, it_span.data(), sizeof(int2) * ni, nullptr));
        CM2_HIP(cm2::upload(t->d_item_k0, it_k0.data(), sizeof(int64_t) * ni, nullptr));
        CM2_HIP(cm2::upload(t->d_item_k1, it_k1.data(), sizeof(int64_t) * ni, nullptr));
    }
    CM2_HIP(hipStreamSynchronize(stream));
    guard.t = nullptr;
    *out = t;
    return 0;
}

extern "C" int cm2_tiles_info(const cm2_tiles *t, int64_t *h_info)
{
    CM2_CHECK(t && h_info, "cm2_tiles_info: NULL argument");
    h_info[0] = t->nt; h_info[1] = t->nvalid; h_info[2] = t->tp;
    h_info[3] = t->ntiles; h_info[4] = t->nitems; h_info[5] = t->half ? 1 : 0;
    h_info[6] = t->pt_fixed; h_info[7] = (int64_t)t->plan_id;
    h_info[8] = t->fx_S; h_info[9] = cm2::fx_designed_bytes(t);
    h_info[10] = t->nspans; h_info[11] = t->span_samples;
    return 0;
}

extern "C" int cm2_tiles_pt_parts(const cm2_tiles *t, int64_t *h_info)
{
    CM2_CHECK(t && h_info, "cm2_tiles_pt_parts: NULL argument");
    return cm2::fx_parts_info(t, h_info);
}

extern "C" int cm2_tiles_prepare_pt(cm2_tiles *t, void *stream_)
{
    CM2_CHECK(t, "cm2_tiles_prepare_pt: NULL plan");
    bool fixed = false;
    return cm2::fx_plan(t, as_stream(stream_), &fixed);
}

extern "C" int cm2_tiles_set_pt_order(cm2_tiles *t, int fixed)
{
    CM2_CHECK(t, "cm2_tiles_set_pt_order: NULL argument");
    t->pt_fixed = fixed == 2 ? 2 : (fixed ? 1 : 0);
    return 0;
}

extern "C" uint64_t cm2_tiles_plan_id(const cm2_tiles *t) { return t ? t->plan_id : 0; }
extern "C" int64_t cm2_tiles_ntiles(const cm2_tiles *t) { return t ? t->ntiles : 0; }
extern "C" int64_t cm2_tiles_nvalid(const cm2_tiles *t) { return t ? t->nvalid : 0; }
// first tile-order address of every segment (span, tile), [nspans * ntiles + 1] on the device, and the
// samples per span (internal: cm2_noise.hip)
extern "C" const int64_t *cm2_tiles_offsets(const cm2_tiles *t) { return t ? t->d_seg_off : nullptr; }
extern "C" int64_t cm2_tiles_nspans(const cm2_tiles *t) { return t ? t->nspans : 1; }
extern "C" int64_t cm2_tiles_span_samples(const cm2_tiles *t) { return t ? t->span_samples : 0; }

// Tile indices bounding `ngroups` consecutive groups of tiles whose PIXEL boundaries are the same on
// every rank of a sharded run (ranks with different hit maps may have cut their tiles differently):
// group g = tiles [h_tiles[g], h_tiles[g + 1]).
extern "C" int cm2_tiles_group_tiles(const cm2_tiles *t, int ngroups, int64_t *h_tiles)
{
    CM2_CHECK(t && h_tiles && ngroups >= 1, "cm2_tiles_group_tiles: bad argument");
    const int64_t ntu = (t->npix + t->tp - 1) / t->tp;
    for (int g = 0; g <= ngroups; ++g) {
        const int c = g == ngroups ? 8 : (ngroups <= 8 ? (8 * g) / ngroups : -1);
        int64_t pixel;
        if (ngroups > 8) {                     // finer than eighths: only a uniform tiling has the cuts
            CM2_CHECK(!t->balanced, "cm2_tiles_group_tiles: at most 8 groups on a balanced tiling");
            pixel = (ntu * g / ngroups) * t->tp;
        } else {
            pixel = (ntu * c / 8) * t->tp;
        }
        if (pixel > t->npix || g == ngroups) pixel = t->npix;
        // the tile that starts at `pixel` (every tiling has a boundary there)
        int64_t lo = 0, hi = t->ntiles;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (t->tile_p0[(size_t)mid] < pixel) lo = mid + 1; else hi = mid;
        }
        CM2_CHECK(t->tile_p0[(size_t)lo] == pixel, "cm2_tiles_group_tiles: no tile boundary at pixel %lld",
                  (long long)pixel);
        h_tiles[g] = lo;
    }
    return 0;
}

extern "C" int cm2_tiles_pixel_range(const cm2_tiles *t, int64_t tile_lo, int64_t tile_hi,
                                     int64_t *h_p0p1)
{
    CM2_CHECK(t && h_p0p1, "cm2_tiles_pixel_range: NULL argument");
    CM2_CHECK(tile_lo >= 0 && tile_lo <= tile_hi && tile_hi <= t->ntiles,
              "cm2_tiles_pixel_range: tiles [%lld, %lld) outside [0, %lld]", (long long)tile_lo,
              (long long)tile_hi, (long long)t->ntiles);
    h_p0p1[0] = t->tile_p0[(size_t)tile_lo];
    h_p0p1[1] = t->tile_p0[(size_t)tile_hi];
    return 0;
}

extern "C" int cm2_P_tiles_apply(const cm2_tiles *t, const double *d_x, double *d_tod_tb,
                                 void *stream_)
{
    CM2_CHECK(t && d_x && (d_tod_tb || t->nvalid == 0), "cm2_P_tiles_apply: NULL argument");
    if (t->nitems == 0) return 0;
    hipStream_t stream = as_stream(stream_);
    const size_t lds = sizeof(double) * t->tp * t->pol;
#define CM2_PT(POL, HALF)                                                                      \
    k_P_tiles<POL, HALF><<<(int)t->nitems, tile_block(true), lds, stream>>>(                       \
        t->d_tile_p0, t->d_item_tile, t->d_item_span, t->d_seg_off, t->ntiles, t->d_item_k0,   \
        t->d_item_k1, t->d_pl, HALF ? t->d_half : t->d_cos, t->d_sin, d_x, d_tod_tb)
    if (t->pol == 1) CM2_PT(1, false);
    else if (t->pol == 2) { if (t->half) CM2_PT(2, true); else CM2_PT(2, false); }
    else { if (t->half) CM2_PT(3, true); else CM2_PT(3, false); }
#undef CM2_PT
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_Pt_tiles_apply(const cm2_tiles *t, const double *d_tod_tb, double *d_out,
                                  void *stream_)
{
    CM2_CHECK(t && d_out && (d_tod_tb || t->nvalid == 0), "cm2_Pt_tiles_apply: NULL argument");
    hipStream_t stream = as_stream(stream_);
    bool fixed = false;
    if (int rc = cm2::fx_plan(t, stream, &fixed)) return rc;
    if (fixed) return cm2::fx_launch(t, d_tod_tb, d_out, 0, t->ntiles, stream);
    CM2_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * t->npix * t->pol, stream));
    if (t->nitems == 0) return 0;
    const size_t lds = sizeof(double) * t->tp * t->pol;
#define CM2_PTT(POL, HALF)                                                                     \
    k_Pt_tiles<POL, HALF><<<(int)t->nitems, tile_block(), lds, stream>>>(                      \
        t->d_tile_p0, t->d_item_tile, t->d_item_span, t->d_seg_off, t->ntiles, t->d_item_k0,   \
        t->d_item_k1, t->d_pl, HALF ? t->d_half : t->d_cos, t->d_sin, d_tod_tb, d_out)
    if (t->pol == 1) CM2_PTT(1, false);
    else if (t->pol == 2) { if (t->half) CM2_PTT(2, true); else CM2_PTT(2, false); }
    else { if (t->half) CM2_PTT(3, true); else CM2_PTT(3, false); }
#undef CM2_PTT
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_Pt_tiles_apply_range(const cm2_tiles *t, const double *d_tod_tb, double *d_out,
                                        int64_t tile_lo, int64_t tile_hi, void *stream_)
{
    CM2_CHECK(t && d_out && (d_tod_tb || t->nvalid == 0), "cm2_Pt_tiles_apply_range: NULL argument");
    CM2_CHECK(tile_lo >= 0 && tile_lo <= tile_hi && tile_hi <= t->ntiles,
              "cm2_Pt_tiles_apply_range: tiles [%lld, %lld) outside [0, %lld]", (long long)tile_lo,
              (long long)tile_hi, (long long)t->ntiles);
    if (tile_lo == tile_hi) return 0;
    hipStream_t stream = as_stream(stream_);
    bool fixed = false;
    if (int rc = cm2::fx_plan(t, stream, &fixed)) return rc;
    if (fixed) return cm2::fx_launch(t, d_tod_tb, d_out, tile_lo, tile_hi, stream);
    const int64_t p0 = t->tile_p0[(size_t)tile_lo], p1 = t->tile_p0[(size_t)tile_hi];
    CM2_HIP(hipMemsetAsync(d_out + p0 * t->pol, 0, sizeof(double) * (p1 - p0) * t->pol, stream));
    const int64_t i0 = t->tile_item0[(size_t)tile_lo], i1 = t->tile_item0[(size_t)tile_hi];
    if (i1 == i0) return 0;
    const size_t lds = sizeof(double) * t->tp * t->pol;
#define CM2_PTR(POL, HALF)                                                                     \
    k_Pt_tiles<POL, HALF><<<(int)(i1 - i0), tile_block(), lds, stream>>>(                      \
        t->d_tile_p0, t->d_item_tile + i0, t->d_item_span + i0, t->d_seg_off, t->ntiles,      \
        t->d_item_k0 + i0, t->d_item_k1 + i0, t->d_pl, HALF ? t->d_half : t->d_cos, t->d_sin,  \
        d_tod_tb, d_out)
    if (t->pol == 1) CM2_PTR(1, false);
    else if (t->pol == 2) { if (t->half) CM2_PTR(2, true); else CM2_PTR(2, false); }
    else { if (t->half) CM2_PTR(3, true); else CM2_PTR(3, false); }
#undef CM2_PTR
    CM2_LAUNCH_OK();
    return 0;
}

__global__ __launch_bounds__(256) void k_i32_time_to_tiles(int64_t nt,
                                                            const uint32_t *__restrict__ tb_dst,
                                                            const int32_t *__restrict__ in,
                                                            int32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += stride) {
        const uint32_t k = tb_dst[t];
        if (k != kInvalidSample) out[k] = in[t];
    }
}

extern "C" int cm2_i32_time_to_tiles(const cm2_tiles *t, const int32_t *d_time, int32_t *d_tb,
                                     void *stream_)
{
    CM2_CHECK(t && (t->nt == 0 || (d_time && d_tb)), "cm2_i32_time_to_tiles: NULL argument");
    if (t->nt == 0) return 0;
    k_i32_time_to_tiles<<<grid_for(t->nt), kBlock, 0, as_stream(stream_)>>>(t->nt, t->d_tb_dst, d_time,
                                                                          d_tb);
    CM2_LAUNCH_OK();
    return 0;
}

static inline void perm_geometry(int64_t nt, int &blocks, int64_t &chunk)
{
    // one 1024-thread workgroup per CU-slot, each with a contiguous time range
    blocks = kNumCU * 2;
    chunk = (nt + blocks - 1) / blocks;
    chunk = ((chunk + 1023) / 1024) * 1024;
    blocks = (int)((nt + chunk - 1) / chunk);
    if (blocks < 1) blocks = 1;
}

// lists of the windowed permutations (CM2_PERM_WINDOWS=0 keeps the per-sample kernels)
static int perm_lists(const cm2_tiles *tc, hipStream_t st, bool *use)
{
    cm2_tiles *t = const_cast<cm2_tiles *>(tc);         // lazily built cache
    static int enabled = -1;
    if (enabled < 0) {
        const char *e = getenv("CM2_PERM_WINDOWS");
        enabled = e ? atoi(e) : 1;
    }
    *use = false;
    if (!enabled || t->nt < kPermWin) return 0;
    if (!t->d_perm_k) {
        const int64_t nwin = (t->nt + kPermWin - 1) / kPermWin, total = nwin * kPermWin;
        DevTemp<uint64_t> keys_in, keys_out;
        DevTemp<uint16_t> vals_in;
        DevTemp<char> d_temp;
        CM2_HIP(keys_in.alloc(total));
        CM2_HIP(keys_out.alloc(total));
        CM2_HIP(vals_in.alloc(total));
        uint32_t *lk = nullptr;
        uint16_t *lq = nullptr;
        CM2_HIP(cm2::dev_malloc(&lk, sizeof(uint32_t) * total));
        DevTemp<uint32_t> guard_k;
        guard_k.p = lk;
        CM2_HIP(cm2::dev_malloc(&lq, sizeof(uint16_t) * total));
        DevTemp<uint16_t> guard_q;
        guard_q.p = lq;
        k_perm_keys<<<grid_for(total), kBlock, 0, st>>>(t->nt, nwin, t->d_tb_dst, keys_in, vals_in);
        CM2_LAUNCH_OK();
        int end_bit = 33;
        while (((int64_t)1 << (end_bit - 32)) <= nwin && end_bit < 64) ++end_bit;
        size_t tb = 0;
        CM2_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in.p, keys_out.p, vals_in.p, lq,
                                                   total, 0, end_bit, st));
        CM2_HIP(d_temp.alloc(tb + 16));
        CM2_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp.p, tb, keys_in.p, keys_out.p, vals_in.p, lq,
                                                   total, 0, end_bit, st));
        k_perm_unpack<<<grid_for(total), kBlock, 0, st>>>(total, keys_out, lk);
        CM2_LAUNCH_OK();
        CM2_HIP(hipStreamSynchronize(st));
        t->d_perm_k = guard_k.keep();
        t->d_perm_q = guard_q.keep();
        t->nperm_win = nwin;
    }
    *use = true;
    return 0;
}

extern "C" int cm2_tod_time_to_tiles(const cm2_tiles *t, const double *d_time, double *d_tb,
                                     void *stream_)
{
    CM2_CHECK(t && d_time && d_tb, "cm2_tod_time_to_tiles: NULL argument");
    bool windows = false;
    if (int rc = perm_lists(t, as_stream(stream_), &windows)) return rc;
    if (windows) {
        const int grid = (int)(((t->nperm_win + 7) / 8) * 8);
        k_perm_windows<false><<<grid, kPermT, 0, as_stream(stream_)>>>(t->nt, t->nperm_win, t->d_perm_k,
                                                                      t->d_perm_q, d_time, d_tb);
        CM2_LAUNCH_OK();
        return 0;
    }
    int blocks;
    int64_t chunk;
    perm_geometry(t->nt, blocks, chunk);
    k_time_to_tiles<<<blocks, 1024, 0, as_stream(stream_)>>>(t->nt, chunk, t->d_tb_dst, d_time, d_tb);
    CM2_LAUNCH_OK();
    return 0;
}

extern "C" int cm2_tod_tiles_to_time(const cm2_tiles *t, const double *d_tb, double *d_time,
                                     void *stream_)
{
    CM2_CHECK(t && d_time && d_tb, "cm2_tod_tiles_to_time: NULL argument");
    bool windows = false;
    if (int rc = perm_lists(t, as_stream(stream_), &windows)) return rc;
    if (windows) {
        const int grid = (int)(((t->nperm_win + 7) / 8) * 8);
        k_perm_windows<true><<<grid, kPermT, 0, as_stream(stream_)>>>(t->nt, t->nperm_win, t->d_perm_k,
                                                                     t->d_perm_q, d_tb, d_time);
        CM2_LAUNCH_OK();
        return 0;
    }
    int blocks;
    int64_t chunk;
    perm_geometry(t->nt, blocks, chunk);
    k_tiles_to_time<<<blocks, 1024, 0, as_stream(stream_)>>>(t->nt, chunk, t->d_tb_dst, d_tb, d_time);
    CM2_LAUNCH_OK();
    return 0;
}

// device address of the time -> tile-order index (nt entries), for kernels that fuse the
// permutation into their own loads and stores (cm2_noise_apply_tiles)
extern "C" const uint32_t *cm2_tiles_index(const cm2_tiles *t) { return t ? t->d_tb_dst : nullptr; }
extern "C" int64_t cm2_tiles_nt(const cm2_tiles *t) { return t ? t->nt : 0; }
