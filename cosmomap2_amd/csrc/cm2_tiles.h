// cm2_tiles.h -- the tile-bucketed pointing plan shared by cm2_tiles.hip (plan, P, atomic P^T,
// permutations) and cm2_tiles_fixed.hip (fixed-order P^T)
#pragma once
#include "cm2_pixindex.h"

#include <vector>

constexpr int64_t kHotTileMin = 32768;   // samples that make a one-pixel tile a hot tile (cm2_tiles_fixed.hip)

struct cm2_tiles {
    int64_t nt = 0, npix = 0, nvalid = 0;
    int pol = 0;
    int tp = 0;                  // pixels per tile (the largest width when the tiles are balanced)
    bool balanced = false;       // tiles re-cut to equal sample counts (uneven hit map)
    std::vector<int64_t> tile_p0;   // [ntiles+1] first pixel of every tile (host)
    int64_t *d_tile_p0 = nullptr;
    int64_t ntiles = 0, nitems = 0;
    // The tables below keep the general [span][tile][time] form of round 4's span order with ONE span (the
    // global tile order; the option itself was measured out and removed in round 5, cm2_tiles.hip): segment
    // (0, tile b) = the tile's whole bucket, addresses [seg_off[b], seg_off[b + 1]).
    int64_t nspans = 1, span_samples = 0;
    std::vector<int64_t> seg_off;   // [nspans * ntiles + 1] (host)
    int64_t *d_seg_off = nullptr;
    uint32_t *d_tb_dst = nullptr;   // [nt]
    uint16_t *d_pl = nullptr;       // [nvalid]
    double *d_cos = nullptr, *d_sin = nullptr;   // [nvalid]  (full-angle mode)
    // half-angle mode: ONE double per sample, h = sin / (1 + |cos|) (= tan of half the angle
    // folded into [-1, 1]) and the sign of cos in bit 15 of d_pl; cos = +-(1 - h^2)/(1 + h^2),
    // sin = 2h/(1 + h^2) are rebuilt in the kernels (absolute error ~2e-16): 8 bytes less per
    // sample in each of the two tile kernels
    bool half = false;
    double *d_half = nullptr;                    // [nvalid]
    // work items of k_P_tiles / k_Pt_tiles: tile, spans [sp0, sp1) of it, clipped to the addresses
    // [k0, k1) (a segment longer than the slice length is cut into several items)
    int32_t *d_item_tile = nullptr; // [nitems]
    int2 *d_item_span = nullptr;    // [nitems] {sp0, sp1}
    int64_t *d_item_k0 = nullptr;   // [nitems]
    int64_t *d_item_k1 = nullptr;
    std::vector<int64_t> tile_item0;   // [ntiles+1] first work item of every tile (host)
    // address-sorted lists of the windowed permutations (built on first use): for every window
    // of kPermWin consecutive time samples, its samples' TB positions in ascending order and
    // their offsets in the window
    uint32_t *d_perm_k = nullptr;
    uint16_t *d_perm_q = nullptr;
    int64_t nperm_win = 0;
    // identity of this plan for the caches other objects key on it (noise / filter lists): a
    // device address can be reused by a later plan, a plan id cannot
    uint64_t plan_id = 0;
    // fixed-order P^T (cm2_tiles_fixed.hip), built on first use: every tile bucket is cut into
    // slices of fx_S consecutive TB samples; per slice the samples sorted by (pixel, time) are
    // packed into groups of 4 list entries that hold whole runs (= samples of one pixel)
    int pt_fixed = 1;
    int fx_S = 0;
    int fx_failed = 0;       // a build of the fixed-order lists failed: not retried on every apply
    std::vector<int64_t> tile_count;    // [ntiles] valid samples of every tile (host)
    int64_t *d_fx_slice0 = nullptr;     // [ntiles+1] first slice of every tile
    uint2 *d_fx_sk = nullptr;           // [nslices+1] {first TB position, samples} of every slice
    uint2 *d_fx_meta = nullptr;         // [nslices+1] {first group, first tail run | max level << 28}
    uint4 *d_fx_gent = nullptr;         // [ngroups] 4 entries: pl word | offset in slice << 16 | level << 28
    double *d_fx_ga = nullptr, *d_fx_gb = nullptr;   // [4 ngroups] half angle (or cos, sin)
    uint2 *d_fx_trun = nullptr;         // [ntail runs + 1] {first tail entry, pixel in tile}
    uint32_t *d_fx_tent = nullptr;      // [ntail entries] entries of the runs kept out of the groups
    double *d_fx_ta = nullptr, *d_fx_tb = nullptr;
    int64_t fx_ngroups = 0, fx_nslices = 0;
    // hot tiles of the fixed-order P^T: a tile that is ONE pixel with very many samples (a stare at
    // a source; the balanced tiling makes such a pixel a tile of its own) is reduced by many
    // workgroups, each summing a fixed range of kHotChunk consecutive samples of the bucket, and
    // the range sums are added in time order (cm2_tiles_fixed.hip)
    // PARTS of the fixed-order P^T (round 4, cm2_tiles_fixed.hip): on a hit map that is far from
    // uniform the tiles keep their width (only a pixel heavy enough for the hot-tile path becomes a
    // tile of its own) and the SLICES of a heavy tile are shared out to several workgroups, each
    // summing its consecutive slices in time order into its own copy of the tile; k_parts_combine adds
    // the copies in time order.  Part boundaries depend on the plan only: reproducible bit for bit;
    // a regrouped sum, ~1e-16 relative away from the serial one.  pt_split = the plan may do this
    // (set at create: unbalanced hit map, neither the equal-load cut nor the exact order asked for).
    bool pt_split = false;
    std::vector<int64_t> tile_part0;    // [ntiles+1] first part of every tile (empty: no parts)
    std::vector<int64_t> multi_tile;    // tiles with more than one part, ascending
    int4 *d_parts = nullptr;            // [nparts] {tile, slices, first slice, scratch slot or -1}
    int64_t *d_multi = nullptr;         // [nmulti][4] offset in the map, values, first scratch slot, parts
    double *d_part_buf = nullptr;       // [scratch slots][tp * pol]
    int64_t nparts = 0, part_slots = 0;
    double part_makespan = 0.0;         // simulated finish time / ideal, 2 workgroups per CU
    std::vector<int64_t> hot_tile, hot_chunk0;   // tile index, first chunk of every hot tile (+ total)
    uint8_t *d_hot_flag = nullptr;               // [ntiles]
    int64_t *d_hot_range = nullptr;              // [chunks][2] first / one-past-last TB position
    int64_t *d_hot_tiles = nullptr;              // [nhot][3] first pixel, first chunk, chunk count
    double *d_hot_partial = nullptr;             // [chunks][3]
    // FUSED form (round 5): the ranges of a hot tile are work items at the end of the main launch and their
    // sums are added up by the last range to finish (cm2_tiles_fixed.hip, FxFused)
    void *d_fx_fused = nullptr;                  // FxFused (device copy)
    int *d_hot_range_tile = nullptr;             // [chunks] index of the range's tile in d_hot_tiles
    unsigned int *d_fx_count = nullptr;          // [nhot] arrival counters, zeroed before every launch
    size_t fx_count_bytes = 0;
};

namespace cm2 {
int fx_plan(const cm2_tiles *t, hipStream_t st, bool *use);
int fx_launch(const cm2_tiles *t, const double *d_tod_tb, double *d_out, int64_t tile_lo,
              int64_t tile_hi, hipStream_t stream);
void fx_free(cm2_tiles *t);
int64_t fx_designed_bytes(const cm2_tiles *t);
int fx_parts_info(const cm2_tiles *t, int64_t *h_info);   // cm2_tiles_pt_parts
int fx_max_slice(const cm2_tiles *t);      // longest slice (samples) the fixed-order kernel's LDS budget allows
bool fx_serial_build();                    // CM2_FX_BUILD=serial (the reference builders: global tile order only)
}  // namespace cm2

