/*
 * cosmomap2.h -- C ABI of libcosmomap2_hip.so, the MI355X (gfx950) implementation
 * of the COSMOMAP2 PCG map-making hot path.
 *
 * The reference (giuspugl/COSMOMAP2) has no FFI of its own: its native code is a
 * set of C++ loop bodies held as Python strings and JIT-compiled by weave.inline
 * (e.g. interfaces/linearoperators.py:368-382).  Each entry point below replaces
 * one of those loops (or the NumPy/BLAS lines that play the same role) and cites
 * it.  The Python binding a maintainer would add in place of each
 * `inline(code, ...)` call is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure: CM2_ERR_HIP (1) a HIP runtime call
 *     failed, CM2_ERR_ARGUMENT (2) an argument or the state of an object was refused, CM2_ERR_OUT_OF_MEMORY
 *     (3) a device allocation failed even after the library gave back its own cached blocks -- the one
 *     failure a host may answer by freeing device memory of its own and calling again, PROVIDED the entry
 *     point overwrites its outputs (every *_create, *_prepare_*, *_apply; not the in-place updates
 *     cm2_axpy, cm2_scal, cm2_Z_axpy, cm2_panel_gemm(accumulate), cm2_pcg_update_*, cm2_flag_samples, nor
 *     the drivers cm2_pcg, cm2_pcg_sharded, cm2_arnoldi);
 *     cm2_last_error() returns a static, thread-local message for the last failure;
 *   - pointers named d_* are DEVICE pointers, h_* are HOST pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is enqueued on it and the call returns without synchronising unless stated;
 *   - maps are pixel-interleaved doubles [I0,Q0,U0,I1,...] (linearoperators.py:487);
 *     pixel ids are int32, -1 marks a flagged sample (linearoperators.py:372);
 *     all arithmetic is IEEE double, no FMA contraction, reference operand order;
 *   - one context per process per GPU.  Objects may be shared by host threads for APPLICATION
 *     calls once their lazily built parts exist (cm2_tiles_prepare_pt, cm2_noise_prepare_tiles,
 *     cm2_pointing_build_sell); construction and destruction are not thread-safe (the reference is
 *     single-threaded).  Exception: a tile plan whose P^T uses scratch of the plan (tiles
 *     split over workgroups on an uneven hit map, hot one-pixel tiles: cm2_tiles_pt_parts reports
 *     both as "tiles split" / copy bytes > 0) must not run two P^T applications at the same time on
 *     different streams; applications on one stream, or of different plans, are fine.
 */
#ifndef COSMOMAP2_H
#define COSMOMAP2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): cm2_noise_prepare_tiles, cm2_set_exact_order, cm2_pcg_sharded and cm2_tiles_pt_parts added;
 * uneven hit maps keep uniform tiles and split the heavy ones over workgroups (cm2_tiles_pixel_range); cm2_noise_tile_kernel_info reports the one-real-window
 * kernel only; and the defaults that changed behind unchanged signatures since version 1 --
 * cm2_pointing_info fields 3..5 are -1 until the pixel-major copy exists, cm2_tiles_set_pt_order has
 * mode 2 and its default sums runs of more than 256 hits per slice in chunks of 32 terms,
 * cm2_weights_accumulate sums pixels with >= 8192 samples in chunks (both <= 1e-14 from the serial
 * sum; CM2_PT_ORDER=exact / CM2_WEIGHTS_ORDER=exact or cm2_set_exact_order(1) restore the serial
 * order), cm2_pcg calls its callback after the next iteration's work is queued. */
#define CM2_ABI_VERSION 2
/* status codes (added in round 5 behind the same "non-zero on failure" contract: no version change) */
#define CM2_ERR_HIP 1
#define CM2_ERR_ARGUMENT 2
#define CM2_ERR_OUT_OF_MEMORY 3

const char *cm2_last_error(void);
int cm2_abi_version(void);
/* device properties used by the bench harness (name buffer >= 256 bytes) */
int cm2_device_info(int device, char *h_name, int *h_num_cu, double *h_hbm_gib);
/* Device memory of the library.  Every buffer a cm2_* object owns and every temporary of a plan
 * build comes from a per-device cache of released blocks (a released block is handed out again to
 * a request of at most 25 % less; releasing waits for the device exactly like hipFree).  The cache
 * holds at most CM2_DEVICE_CACHE_MB megabytes (environment; default: an eighth of the device's memory,
 * 36 GB on an MI355X; 0 = every release goes straight back to the driver).  Other allocators on the
 * same device (the host framework's, another rank's) do not see these blocks: a host that runs out of
 * device memory should call cm2_release_cached_memory and retry (cosmomap2_amd/device.py does).  cm2_release_cached_memory returns all cached blocks to the
 * driver; cm2_device_memory_info fills h_info[4] = bytes in use by live objects, bytes cached,
 * requests served from the cache, requests that went to the driver. */
int cm2_release_cached_memory(void);
/* Summation order of the per-pixel sums (process-wide; objects created afterwards follow it).  The
 * reference adds every pixel's terms serially in time order (linearoperators.py:509-516,
 * process_ces.py:426-555).  Default (on = 0): that order, except that a pixel with >= 8192 samples
 * (weights) or a run of more than 256 hits inside one slice of a tile (P^T) is summed in fixed chunks
 * -- reproducible, independent of the rest of the hit map, <= 1e-14 relative from the serial sum, so
 * a pixel whose condition number sits within 1e-14 of threshold_cond (process_ces.py:544-550) could
 * be masked differently.  on = 1: the pure serial order everywhere (bit-equal to the reference's
 * loops; a stare at one pixel then costs milliseconds to seconds).  on = -1: back to the environment
 * (CM2_PT_ORDER, CM2_WEIGHTS_ORDER). */
int cm2_set_exact_order(int on);
int cm2_device_memory_info(int64_t *h_info);

/* ------------------------------------------------------------------------- *
 * a1-a3  Pointing matrix  (SparseLO, interfaces/linearoperators.py:326-557)
 * ------------------------------------------------------------------------- */
typedef struct cm2_pointing cm2_pointing;

/* Builds the device-side pointing plan from time-ordered arrays that are
 * already resident in HBM.  The plan KEEPS the three pointers (caller keeps the
 * buffers alive) and adds a pixel-major copy: samples grouped by pixel in time
 * order, 64 pixels per slice, slices sorted by hit count (sliced-ELL), which is
 * what makes P^T a race-free, fixed-order reduction (built lazily, see LIFETIME below).
 * d_cos/d_sin may be NULL for pol == 1.  Replaces SparseLO.__init__
 * (linearoperators.py:527-550); pol not in {1,2,3} fails like the RuntimeError at :549.
 * Synchronises. */
int cm2_pointing_create(cm2_pointing **out, const int32_t *d_pix, const double *d_cos,
                        const double *d_sin, int64_t nt, int64_t npix, int pol,
                        void *stream);
int cm2_pointing_destroy(cm2_pointing *p);
/* LIFETIME: the pixel-major copy is built by its first user (cm2_Pt_apply, cm2_pointing_set_weights,
 * cm2_PtNP_diag_apply) or by cm2_pointing_build_sell, FROM d_pix / d_cos / d_sin: the caller must keep
 * those three buffers alive AND UNCHANGED at least until then (cm2_P_apply reads them for the whole
 * life of the plan).  An operator that only runs on the tile order (cm2_tiles) never builds it. */
int cm2_pointing_build_sell(cm2_pointing *p, void *stream);
/* h_info[0..5] = nt, npix, pol, then valid samples, padded pixel-major length, slices of the
 * pixel-major copy, or -1, -1, -1 while that copy has not been built (this call builds nothing) */
int cm2_pointing_info(const cm2_pointing *p, int64_t *h_info);

/* P x: gather, time order.  d_out[t] = I_p + Q_p cos2phi_t + U_p sin2phi_t, 0 for
 * flagged samples.  Replaces the weave loops at linearoperators.py:368-375
 * (mult), :424-430 (mult_qu), :483-489 (mult_iqu). */
int cm2_P_apply(const cm2_pointing *p, const double *d_x, double *d_out, void *stream);

/* P^T v: scatter-add in sample order per pixel, written as a per-pixel
 * fixed-order reduction.  Replaces linearoperators.py:394-400 (rmult),
 * :447-454 (rmult_qu), :509-516 (rmult_iqu).  Overwrites d_out (pol*npix). */
int cm2_Pt_apply(const cm2_pointing *p, const double *d_v, double *d_out, void *stream);

/* Attach per-sample diagonal noise weights w_t = (N^-1)_tt (time order, nt
 * doubles; NULL = all ones) for the fused matvec below.  Replaces the role of
 * BlockLO.diag (linearoperators.py:677-683). */
int cm2_pointing_set_weights(cm2_pointing *p, const double *d_w, void *stream);

/* Fused P^T diag(w) P x in ONE pass over the pixel-major samples; no TOD vector
 * is materialised.  Same arithmetic, in the same order, as the reference's
 * three-stage chain (P.T*N*P)*x  (tests/test_toeplitz_vector_multiplication.py:24-28,
 * call stack SURVEY 3.2). */
int cm2_PtNP_diag_apply(const cm2_pointing *p, const double *d_x, double *d_out,
                        void *stream);

/* ------------------------------------------------------------------------- *
 * a2-a3 (throughput form)  Tile-bucketed TOD order
 *   Samples grouped by pixel tile (stable, so time order inside a tile); the tile's
 *   slice of the map is staged in LDS, so P and P^T stream HBM with no random access.
 *   The order is internal: TB-ordered TODs are only ever produced and consumed by the entry points of
 *   this section and cm2_noise_apply_tiles / cm2_filter_apply_tiles.  (Round 4's option to cut the
 *   order in time as well, CM2_TILE_SPAN, never paid and was removed in round 5.)
 *   Same loops as above (linearoperators.py:483-489, :509-516).  P^T by default adds every
 *   pixel's terms in time order from 0 like the serial loop (one workgroup per tile, per-slice
 *   lists sorted by (pixel, time), no atomics: bitwise reproducible); cm2_tiles_set_pt_order(t, 0)
 *   or CM2_PT_ORDER=atomic selects LDS + global fp64 atomics (term order not fixed).
 * ------------------------------------------------------------------------- */
typedef struct cm2_tiles cm2_tiles;
int cm2_tiles_create(cm2_tiles **out, const int32_t *d_pix, const double *d_cos,
                     const double *d_sin, int64_t nt, int64_t npix, int pol, int tile_pixels,
                     int64_t slice_samples, void *stream);
int cm2_tiles_destroy(cm2_tiles *t);
/* h_info[0..11] = nt, valid samples (= length of a TB-ordered TOD), tile pixels, tiles, items,
 * 1 if the plan stores one half-angle value per sample instead of cos and sin (done when every
 * (cos, sin) pair is on the unit circle to 1e-14; the kernels rebuild cos = +-(1-h^2)/(1+h^2),
 * sin = 2h/(1+h^2), absolute error ~2e-16, and read 8 bytes less per sample), 1 if P^T sums in
 * fixed (time) order, the plan's id (unique per plan in this process), slice length of the
 * fixed-order lists (0 until the first P^T builds them), the bytes one fixed-order P^T is
 * designed to read (TOD + padded lists), then 1 and the padded sample count (the two fields of the
 * span order removed in round 5: one span = the global tile order) */
int cm2_tiles_info(const cm2_tiles *t, int64_t *h_info);
/* fixed == 1 (default): P^T adds each pixel's terms in time order (the reference's order,
 * reproducible bit for bit), except that a pixel hit more than 256 times inside one slice of a
 * tile's samples (hot pixels: a stare at a source) is summed in chunks of 32 consecutive terms
 * whose sums are then added in time order -- a fixed regrouping, still reproducible bit for bit
 * and independent of the hit map, ~1e-16 relative away from the serial sum;
 * fixed == 2 ("exact", CM2_PT_ORDER=exact): pure time order for every pixel whatever its hit count
 * (one thread walks a hot run: 1e5 hits in one pixel cost milliseconds);
 * fixed == 0 (CM2_PT_ORDER=atomic): LDS / global atomics, term order not fixed */
int cm2_tiles_set_pt_order(cm2_tiles *t, int fixed);
/* Builds the per-slice (pixel, time) lists of the fixed-order P^T now (allocations, sorts, one
 * synchronisation) instead of inside the first cm2_Pt_tiles_apply: afterwards an application only
 * launches kernels, so it may be captured into a graph or issued from several host threads.  A plan
 * that was not prepared builds the lists on its first P^T under a lock.  No-op for the atomic form. */
int cm2_tiles_prepare_pt(cm2_tiles *t, void *stream);
/* Work items of the fixed-order P^T after cm2_tiles_prepare_pt: h_info[0..3] = workgroups per
 * application (= tiles when no tile is split), tiles that are split, bytes of the plan's P^T scratch
 * (tile copies of the split tiles, range sums of hot one-pixel tiles), and
 * 1000 x (simulated finish time of the items on two resident workgroups per CU / ideal). */
int cm2_tiles_pt_parts(const cm2_tiles *t, int64_t *h_info);
/* h_p0p1[0..1] = pixel range [p0, p1) covered by the tiles [tile_lo, tile_hi).  Tiles are uniform
 * (tile_pixels wide).  On an uneven hit map (a uniform tile holding over 25 % more samples than the
 * mean) the fixed-order P^T shares the slices of the heavy tiles out to several workgroups, each
 * summing consecutive slices in time order into its own copy of the tile, and adds the copies in time
 * order: a fixed regrouping of the serial sum (part boundaries depend on the plan only: reproducible
 * bit for bit; ~1e-16 relative away from the serial sum); a pixel with at least 32768 samples and half
 * a tile's mean load becomes a one-pixel tile of its own (summed as described at
 * cm2_tiles_set_pt_order).  CM2_TILE_BALANCE: 0 = uniform tiles, one workgroup each; 1 / cut = pixel
 * ranges re-cut to equal sample counts (never wider than tile_pixels), one workgroup each -- the same
 * bits as the uniform tiling, chosen automatically when the exact order is asked for
 * (cm2_set_exact_order, CM2_PT_ORDER=exact); parts = the default on uneven maps, forced. */
int cm2_tiles_pixel_range(const cm2_tiles *t, int64_t tile_lo, int64_t tile_hi, int64_t *h_p0p1);
/* h_tiles[0..ngroups]: group g = tiles [h_tiles[g], h_tiles[g+1]) -- consecutive pieces of the map
 * for cm2_Pt_tiles_apply_range whose pixel boundaries depend on npix and tile_pixels only, so that
 * every rank of a TOD-sharded run reduces the same pixels (balanced tilings keep these cuts). */
int cm2_tiles_group_tiles(const cm2_tiles *t, int ngroups, int64_t *h_tiles);
/* d_tod_tb[k] = (P x) for the k-th sample in TB order */
int cm2_P_tiles_apply(const cm2_tiles *t, const double *d_x, double *d_tod_tb, void *stream);
/* d_out = P^T v for a TB-ordered v (d_out is overwritten) */
int cm2_Pt_tiles_apply(const cm2_tiles *t, const double *d_tod_tb, double *d_out, void *stream);
/* integer per-sample labels (e.g. ground bins) from time order to TB order; set-up use */
int cm2_i32_time_to_tiles(const cm2_tiles *t, const int32_t *d_time, int32_t *d_tb, void *stream);
/* P^T restricted to the tiles [tile_lo, tile_hi): overwrites the entries of d_out that belong to
 * those tiles (pixels tile_lo*tile_pixels .. min(tile_hi*tile_pixels, npix)) and nothing else, so
 * that a finished part of the map can be reduced across GPUs while the next part is computed. */
int cm2_Pt_tiles_apply_range(const cm2_tiles *t, const double *d_tod_tb, double *d_out,
                             int64_t tile_lo, int64_t tile_hi, void *stream);
/* permutations between time order (nt, flagged samples read as / written with 0) and TB order */
int cm2_tod_time_to_tiles(const cm2_tiles *t, const double *d_time, double *d_tb, void *stream);
int cm2_tod_tiles_to_time(const cm2_tiles *t, const double *d_tb, double *d_time, void *stream);

/* ------------------------------------------------------------------------- *
 * a4-a5  Noise operator N^-1  (ToeplitzLO linearoperators.py:560-602,
 *        BlockLO :627-697, blk_matvec interfaces/blkop.py:178-208)
 * ------------------------------------------------------------------------- */
typedef struct cm2_noise cm2_noise;

#define CM2_TOEPLITZ_AUTO   0   /* direct for short bands, fused FFT, rocFFT beyond its range */
#define CM2_TOEPLITZ_DIRECT 1   /* O(n*lambda), reference summation order  */
#define CM2_TOEPLITZ_FFT    2   /* overlap-save, rocFFT R2C/C2R fp64       */
#define CM2_TOEPLITZ_FUSED  3   /* overlap-save, one kernel, fp64 FFT in LDS (lambda <= 2049) */

/* Block-diagonal with constant diagonal blocks: block b = h_t[b] * I of
 * h_sizes[b] samples (BlockLO offdiag=False, :676-683). */
int cm2_noise_create_diag(cm2_noise **out, const double *h_t, const int64_t *h_sizes,
                          int64_t nblocks);
/* Block-diagonal of symmetric banded Toeplitz blocks: block b has first row
 * h_bands[b*lambda .. b*lambda+lambda-1] (ToeplitzLO.mult :582-595), zero
 * boundary at each block edge -- never circulant, never across blocks. */
int cm2_noise_create_toeplitz(cm2_noise **out, const double *h_bands, int64_t lambda,
                              const int64_t *h_sizes, int64_t nblocks, int method,
                              void *stream);
int cm2_noise_destroy(cm2_noise *n);
/* y = N^-1 v over all blocks (blk_matvec, blkop.py:195-206).  d_out != d_v. */
int cm2_noise_apply(cm2_noise *n, const double *d_v, double *d_out, void *stream);
/* y = N^-1 v with v and y both in the tile-bucketed order of `tiles` (CM2_TOEPLITZ_FUSED
 * operators, and CM2_TOEPLITZ_AUTO ones with lambda <= 2049): the permutation to and from time order
 * is folded into the overlap-save kernel's own loads and stores (ToeplitzLO.mult :582-595 per block,
 * blkop.py:195-206).  d_out_tb != d_in_tb.
 * cm2_noise_prepare_tiles builds the address lists of the (operator, tile plan) pair: it allocates,
 * launches the list builders on `stream` and WAITS for them.  After it, cm2_noise_apply_tiles for the
 * same plan is one kernel launch -- no allocation, no synchronisation, safe inside a stream capture
 * and from several host threads.  Without it the first application prepares the lists itself (under
 * the operator's lock).  An operator keeps the lists of its three most recently used tile plans. */
int cm2_noise_prepare_tiles(cm2_noise *n, const cm2_tiles *tiles, void *stream);
int cm2_noise_apply_tiles(cm2_noise *n, const cm2_tiles *tiles, const double *d_in_tb,
                          double *d_out_tb, void *stream);
/* y = P^T N^-1 P x in ONE call on the tile order (the chain `P.T*N*P` of the reference's scripts,
 * src/test_M2_precond_onto_real_data.py:79-86, SURVEY 8b's cm2_PtNP_apply): cm2_P_tiles_apply,
 * cm2_noise_apply_tiles, cm2_Pt_tiles_apply on `stream`.  d_tb1 != d_tb2: scratch of at least
 * (valid samples of the plan, cm2_tiles_info[1]) doubles each. */
int cm2_PtNP_tiles_apply(const cm2_tiles *tiles, cm2_noise *n, const double *d_x, double *d_y,
                         double *d_tb1, double *d_tb2, void *stream);
/* per-sample diagonal of a constant-diagonal noise operator (BlockLO.diag). */
int cm2_noise_expand_diag(const cm2_noise *n, double *d_w, void *stream);
/* h_info[0..5] = nt, nblocks, lambda (0 for diag), method used, FFT length, 1 if
 * cm2_noise_apply_tiles is available (CM2_TOEPLITZ_FUSED, or CM2_TOEPLITZ_AUTO with lambda <= 2049:
 * an AUTO operator that applies the direct sum on the time order still runs the fused
 * overlap-save kernel on a tile order) */
int cm2_noise_info(const cm2_noise *n, int64_t *h_info);
/* What cm2_noise_apply_tiles runs for this operator (none of this exists in the reference, whose
 * ToeplitzLO.mult is a NumPy loop, interfaces/linearoperators.py:582-595): h_info[0] = complex points
 * per thread of the one-real-window kernel (32), h_info[1] = list format of the most recently used
 * tile plan (1 plain, 2 run-coded lists cut by time, 3 run-coded lists cut by address ("inverse"),
 * 0 = no lists built yet), h_info[2] = window length in samples, h_info[3] = 0 (was: windows across
 * two spans of the span order removed in round 5);
 * *h_bytes_per_sample = HBM bytes per TOD sample the kernel is built to move (lists + gathered windows +
 * results).  Environment switches,
 * read ONCE when the operator is created: CM2_OS_LISTS = auto (default: rc below 768 pixel tiles,
 * inv from there up) | rc | inv | plain, CM2_OS_LIST_BUILD = direct (default) | sort, CM2_OS_FLAT
 * (flat addressing also for buffers below 4 GB). */
int cm2_noise_tile_kernel_info(const cm2_noise *n, int64_t *h_info, double *h_bytes_per_sample);

/* ------------------------------------------------------------------------- *
 * a6-a7  ProcessTimeSamples  (utilities/process_ces.py:58-555)
 * ------------------------------------------------------------------------- */
/* Per-pixel sums of w, w c, w s, w c c, w s s, w s c over the samples of each
 * pixel IN TIME ORDER (fixed-order, race-free).  pol=1 fills d_counts only,
 * pol=2 the last three, pol=3 all six (unused outputs may be NULL).  d_w NULL =
 * ones (process_ces.py:65-66).  Replaces :480-487, :505-514, :527-539 and
 * compute_arrays :125-186.  Synchronises (builds a temporary pixel index).
 * A pixel with 8192 samples or more (a stare at a source; one thread would walk
 * 5e6 samples for 1.9 s) is summed in fixed chunks of 4096 samples -- 256 strided
 * partial sums, a fixed halving tree, the chunks added in time order: reproducible
 * bit for bit, independent of the other pixels, ~1e-16 relative per level from the
 * serial sum (sums of unit weights stay exact).  The environment variable
 * CM2_WEIGHTS_ORDER=exact keeps the serial sum for every pixel. */
int cm2_weights_accumulate(int pol, int64_t nt, int64_t npix, const int32_t *d_pix,
                           const double *d_w, const double *d_cos, const double *d_sin,
                           double *d_counts, double *d_cosine, double *d_sine,
                           double *d_cos2, double *d_sin2, double *d_sincos, void *stream);
/* cos(2 phi_t), sin(2 phi_t) for angles already resident in HBM
 * (process_ces.py:493-494; host arrays use NumPy so that they match the reference
 * bit for bit). */
int cm2_cos_sin_2phi(int64_t nt, const double *d_phi, double *d_cos, double *d_sin,
                     void *stream);
/* keep[p] = 1 for well-conditioned observed pixels: pol=1 counts>0 (:491);
 * pol>=2 |lambda_max/lambda_min| <= threshold of the QU block (:544-550);
 * pol=3 additionally counts>2 (:554-555). */
int cm2_pixel_mask(int pol, int64_t npix, const double *d_counts, const double *d_cos2,
                   const double *d_sin2, const double *d_sincos, double threshold,
                   uint8_t *d_keep, void *stream);
/* old2new[p] = rank of p among kept pixels or -1; *h_new_npix = kept count.
 * Same output as the O(Nold*Nm) search at :205-228, by prefix sum.  Synchronises. */
int cm2_pixel_compact(int64_t npix, const uint8_t *d_keep, int32_t *d_old2new,
                      int64_t *h_new_npix, void *stream);
/* d_out[old2new[p]] = d_in[p] for kept pixels (:218-219, :248-251, :280-286). */
int cm2_compact_f64(int64_t npix, const int32_t *d_old2new, const double *d_in,
                    double *d_out, void *stream);
int cm2_compact_i64(int64_t npix, const int32_t *d_old2new, const int64_t *d_in,
                    int64_t *d_out, void *stream);
/* pix[t] = old2new[pix[t]] in place, -1 stays -1 (flagging_samples :411-418). */
int cm2_flag_samples(int64_t nt, int32_t *d_pix, const int32_t *d_old2new, void *stream);

/* ------------------------------------------------------------------------- *
 * a8-a9  Per-pixel Stokes blocks
 * ------------------------------------------------------------------------- */
/* det and |det|>1e-5 mask of the per-pixel blocks, as the NumPy lines
 * linearoperators.py:792-795 (pol=3) / :820-821 (pol=2) / :789 (pol=1, counts>0). */
int cm2_bd_det_mask(int pol, int64_t npix, const double *d_counts, const double *d_cosine,
                    const double *d_sine, const double *d_cos2, const double *d_sin2,
                    const double *d_sincos, double *d_det, uint8_t *d_mask, void *stream);
/* y = M_BD x: closed-form adjugate/det per masked pixel, 0 elsewhere
 * (BlockDiagonalPreconditionerLO.mult, linearoperators.py:775-841). */
int cm2_bdprecond_apply(int pol, int64_t npix, const double *d_counts,
                        const double *d_cosine, const double *d_sine, const double *d_cos2,
                        const double *d_sin2, const double *d_sincos, const double *d_det,
                        const uint8_t *d_mask, const double *d_x, double *d_y, void *stream);
/* y = (P^T diag(N^-1) P) x per pixel (BlockDiagonalLO.mult, :728-746). */
int cm2_bd_apply(int pol, int64_t npix, const double *d_counts, const double *d_cosine,
                 const double *d_sine, const double *d_cos2, const double *d_sin2,
                 const double *d_sincos, const double *d_x, double *d_y, void *stream);

/* ------------------------------------------------------------------------- *
 * a15-a16  BLAS-1 and the PCG recurrence
 *   (utilities/linear_algebra_funcs.py:31-44; scipy.sparse.linalg.cg as called at
 *    tests/test_2level_preconditioner.py:52, src/test_BD_precond_onto_real_data.py:47)
 * ------------------------------------------------------------------------- */
/* *d_out = sum_i x_i y_i, fixed two-stage tree (bitwise reproducible run to run).
 * d_work: >= cm2_reduce_work_doubles() doubles of scratch. */
int64_t cm2_reduce_work_doubles(void);
int cm2_dot(int64_t n, const double *d_x, const double *d_y, double *d_out, double *d_work,
            void *stream);
int cm2_axpy(int64_t n, double alpha, const double *d_x, double *d_y, void *stream); /* y += alpha x */
int cm2_scal(int64_t n, double alpha, double *d_x, void *stream);
int cm2_xmy(int64_t n, const double *d_x, const double *d_y, double *d_out, void *stream); /* out = x*y */
/* p = z + (rho/rho_prev) p   with rho, rho_prev read from device memory
 * (cg: beta = rho_cur/rho_prev; p *= beta; p += z). */
int cm2_pcg_update_p(int64_t n, const double *d_rho, const double *d_rho_prev,
                     const double *d_z, double *d_p, void *stream);
/* alpha = rho/pq (device scalars); x += alpha p; r -= alpha q; *d_rr = r.r */
int cm2_pcg_update_xr(int64_t n, const double *d_rho, const double *d_pq, const double *d_p,
                      const double *d_q, double *d_x, double *d_r, double *d_rr,
                      double *d_work, void *stream);

/* The whole solve for hosts that are not Python: scipy.sparse.linalg.cg's recurrence (the driver the
 * reference calls: tests/test_2level_preconditioner.py:52, src/test_BD_precond_onto_real_data.py:47)
 * with the operator A and the preconditioner M (may be NULL) as callbacks on device vectors.
 * A callback computes d_out = Op d_in on `stream` and returns 0 on success.  x_is_zero != 0: d_x is
 * overwritten with the start vector 0, otherwise d_x holds x0.  Stops when ||r||_2 < max(atol,
 * rtol ||b||_2), tested before each iteration like scipy does; maxiter < 0 means 10 n.
 * *h_iters = iterations done, *h_info = 0 (converged) or maxiter.  `callback` (may be NULL) is
 * called after every iteration with the iteration number, the iterate (device) and ||r||_2. */
typedef int (*cm2_apply_fn)(void *ctx, const double *d_in, double *d_out, void *stream);
typedef void (*cm2_iter_fn)(void *ctx, int64_t iteration, const double *d_x, double rnorm);
int cm2_pcg(int64_t n, cm2_apply_fn A, void *A_ctx, cm2_apply_fn M, void *M_ctx,
            const double *d_b, double *d_x, int x_is_zero, double rtol, double atol,
            int64_t maxiter, cm2_iter_fn callback, void *cb_ctx, int64_t *h_iters, int *h_info,
            void *stream);

/* (e) The same solve on TOD shards, one process per GPU (the reference is one process: its
 * block independence, interfaces/blkop.py:195-206, is what makes the cut exact).  The library links no
 * collective library: the HOST passes its own reduction -- `reduce` combines `count` doubles at d_vals
 * over all ranks IN PLACE, queued on `stream`, CM2_REDUCE_SUM or CM2_REDUCE_MAX (INTEGRATION.md shows
 * the three-line RCCL form) -- and does the map-sized exchange inside its operator callback.
 *   CM2_LAYOUT_REPLICATED: every rank holds whole map vectors (n_local = n); the A callback applies the
 *     rank's P^T N^-1 P and all-reduces the product; dots are computed redundantly, b.b and ||r||^2 are
 *     MAX-reduced so that every rank takes the same stop decision.
 *   CM2_LAYOUT_ROWS: every vector (d_b, d_x, the callbacks' arguments) is the rank's n_local rows; the A
 *     callback all-gathers its input and reduce-scatters the product; b.b, rho, p.q and ||r||^2 are
 *     SUM-reduced (three 8-byte reductions an iteration); maxiter < 0 means 10 x the summed n_local.
 * Every rank must make the call with the same rtol / atol / maxiter; they return the same *h_iters and
 * *h_info.  Same recurrence and stop rule as cm2_pcg. */
#define CM2_REDUCE_SUM 0
#define CM2_REDUCE_MAX 1
#define CM2_LAYOUT_REPLICATED 0
#define CM2_LAYOUT_ROWS 1
typedef int (*cm2_reduce_fn)(void *ctx, double *d_vals, int64_t count, int op, void *stream);
int cm2_pcg_sharded(int64_t n_local, cm2_apply_fn A, void *A_ctx, cm2_apply_fn M, void *M_ctx,
                    const double *d_b, double *d_x, int x_is_zero, double rtol, double atol,
                    int64_t maxiter, cm2_iter_fn callback, void *cb_ctx, int layout,
                    cm2_reduce_fn reduce, void *reduce_ctx, int64_t *h_iters, int *h_info,
                    void *stream);

/* a13  arnoldi (interfaces/deflationlib.py:17-113) as a C entry point, operator as a callback:
 * modified Gram-Schmidt on r0 = b - A x0 (d_x0 may be NULL = 0) with the reference's early exit
 * (||r0|| < tol ||b|| or < tol: *h_steps = 0, :80-82), its stop rule |v_new[j] h_{j+1,j}| <= tol
 * (:101) and its failure after inner_m steps (returns non-zero, cm2_last_error() =
 * "Convergence not achieved within the Arnoldi algorithm", :111-112; *h_steps = inner_m).
 * d_V: inner_m vectors of n doubles, vector i at d_V + i*n; h_H: (inner_m + 1) x inner_m,
 * row-major, host; on return the first *h_steps vectors and columns are set (build_hess :115-137
 * reads its m x m matrix from the leading block). */
int cm2_arnoldi(int64_t n, cm2_apply_fn A, void *A_ctx, const double *d_b, const double *d_x0,
                double tol, int inner_m, double *d_V, double *h_H, int *h_steps, void *stream);

/* ------------------------------------------------------------------------- *
 * a10-a12  Deflation space and coarse operator
 *   (DeflationLO linearoperators.py:1029-1065, CoarseLO :946-1027,
 *    M2 src/test_M2_precond_onto_real_data.py:98-112)
 *   Z is n x r, ROW-major in HBM (the r entries of one map element contiguous).
 * ------------------------------------------------------------------------- */
int cm2_Zt_apply(int64_t n, int r, const double *d_Z, const double *d_x, double *d_out,
                 double *d_work, void *stream);                 /* out[r] = Z^T x  (:1051-1056) */
int cm2_Z_apply(int64_t n, int r, const double *d_Z, const double *d_y, double *d_out,
                void *stream);

/* w += alpha * Z y for a row-major n x r panel Z: the update step of the Arnoldi
 * orthogonalisation (w -= V h, interfaces/deflationlib.py:88-93 -- the reference's loop of r
 * axpy calls) in one pass over the panel; the order of the r terms is a fixed tree, not the
 * k = 0..r-1 order of cm2_Z_apply. */
int cm2_Z_axpy(int64_t n, int r, const double *d_Z, const double *d_y, double alpha,
               double *d_w, void *stream);                                  /* out[n] = Z y    (:1041-1050) */
/* E[r1 x r2] (row-major) = Z1^T Z2, fp64 MFMA panels when r1,r2 are multiples of
 * 16 (CoarseLO.__init__: dgemm(Z, Az.T), :1019).  d_work >= cm2_gemm_tn_work_doubles(r1,r2). */
int64_t cm2_gemm_tn_work_doubles(int r1, int r2);
int cm2_gemm_tn(int64_t n, int r1, int r2, const double *d_Z1, const double *d_Z2, double *d_E,
                double *d_work, void *stream);
/* C[m x n] (row-major) = A^T B^T with A k x m and B n x k, both row-major: the
 * reference's dgemm(A,B) helper (utilities/linear_algebra_funcs.py:16-29). */
int cm2_gemm_atbt(int64_t m, int64_t n, int64_t k, const double *d_A, const double *d_B,
                  double *d_C, void *stream);
/* d_out[c][r] = d_in[r][c] for a row-major rows x cols matrix: layout change between a set of
 * contiguous map vectors (cols x n) and the row-major n x r panel that DeflationLO streams
 * (the reference keeps Z as a list of column views, linearoperators.py:1059-1062). */
int cm2_transpose(int64_t rows, int64_t cols, const double *d_in, double *d_out, void *stream);
/* d_out[n x rout] (+)= d_P[n x rin] d_W[rin x rout], all row-major, accumulate != 0 adds to d_out:
 * the tall-panel products of the deflation build -- Ritz vectors Z = V y (build_Z,
 * deflationlib.py:183) and A Z = P_{m+1} (H y) through the Arnoldi relation instead of r more
 * applications of A (src/test_M2_precond_onto_real_data.py:98-101); fp64 MFMA for rin = 32,
 * rout = 16 / 32. */
int cm2_panel_gemm(int64_t n, int rin, int rout, const double *d_P, const double *d_W,
                   double *d_out, int accumulate, void *stream);
/* out[r] = M[r x r] (row-major) v  -- E^-1 held as an explicit small matrix */
int cm2_small_matvec(int r, const double *d_M, const double *d_v, double *d_out, void *stream);
/* fused second half of M2 r = M_BD (r - AZ y) + Z y  given y = E^-1 Z^T r:
 * one pass over Z and AZ, then the per-pixel M_BD block. */
int cm2_m2_finish(int pol, int64_t npix, int r, const double *d_Z, const double *d_AZ,
                  const double *d_y, const double *d_res, const double *d_counts,
                  const double *d_cosine, const double *d_sine, const double *d_cos2,
                  const double *d_sin2, const double *d_sincos, const double *d_det,
                  const uint8_t *d_mask, double *d_out, void *stream);

/* ---- f1: sub-scan filtering of a time stream --------------------------------
 * Replaces FilterLO (interfaces/linearoperators.py:94-322).  The time stream is
 * cut into `nseg` chunks (one per CES x detector pair x sub-scan, the loops at
 * :134-140); h_start must be ascending and the chunks must not overlap.
 * Samples outside every chunk are 0 in the output (vec_out = d*0, :130).
 * Flags (pixel id < 0) are read from d_pix at create time and at every apply;
 * they must not change in between (the per-chunk bases below depend on them).
 *
 * order == 0  (FilterLO.mult :129-168): out = d - mean(unflagged d) on every
 *   sample of the chunk; a chunk without unflagged samples (or with a non-finite
 *   mean) stays 0.  h_table_off/h_table are ignored (may be NULL).
 * order  > 0  (globalprocsfilter :286-322): K = order+1 (K <= 8).  h_table holds
 *   one n x K row-major block of normalised Legendre columns for every distinct
 *   chunk length n (get_legendre_polynomials, utilities/linear_algebra_funcs.py:
 *   47-59); h_table_off[s] is the offset (in doubles) of chunk s's block.  Per chunk,
 *   with m the number of unflagged samples:
 *     m <= order : chunk left at 0 (:303-304);
 *     m == n     : out = d - sum_k <L_k,d> L_k with the table columns as they are
 *                  (:317-321);
 *     otherwise  : out = d - Q Q^T d on the unflagged samples and 0 on the flagged
 *                  ones, Q an orthonormal basis of the polynomials of degree <= order
 *                  on the unflagged samples (:307-315, where Q comes from
 *                  qr(legendres[unflagged])).  Here Q is built once at create time
 *                  from the three-term recurrence of the discrete orthogonal
 *                  polynomials of those samples; same projector, and no loss of
 *                  orthogonality when the restricted Legendre block is ill
 *                  conditioned. */
typedef struct cm2_filter cm2_filter;
int cm2_filter_create(cm2_filter **out, int64_t nt, int64_t nseg, const int64_t *h_start,
                      const int64_t *h_len, const int32_t *d_pix, int order,
                      const int64_t *h_table_off, const double *h_table, int64_t table_len,
                      void *stream);
void cm2_filter_destroy(cm2_filter *f);
/* info[7] = nt, nseg, order, samples inside chunks, then (order > 0) the number of
 * chunks skipped / without flags / with flags */
int cm2_filter_info(const cm2_filter *f, int64_t *info);
/* d_out = F d_in (nt doubles each; d_out must not alias d_in) */
int cm2_filter_apply(const cm2_filter *f, const double *d_in, double *d_out, void *stream);

/* The same filter with input and output in the tile-bucketed order of `tiles` (flagged samples
 * have no slot there).  The time stream is cut into windows of <= 8192 samples holding whole
 * chunks; each window is gathered through an address-sorted list, filtered in LDS and
 * scattered back.  *h_done = 0 (and nothing is written) when a chunk is longer than a window:
 * the caller then applies cm2_filter_apply on the time order.  d_out_tb != d_in_tb. */
int cm2_filter_apply_tiles(cm2_filter *f, const cm2_tiles *tiles, const double *d_in_tb,
                           double *d_out_tb, int *h_done, void *stream);

/* ---- f2: ground-template filter  (GroundFilterLO, :24-61) --------------------
 * d_out[t] = d_v[t] - binned[d_bin[t]]   (d_out[t] = d_v[t] where d_bin[t] < 0),
 * the last step of v - G (G^T G)^-1 G^T v once binned = (G^T G)^-1 G^T v has
 * been formed with cm2_Pt_apply and cm2_bdprecond_apply (pol = 1). */
/* d_sums[b] = sum of d_v[t] over the samples with d_bin[t] == b  (G^T v, pol = 1) for
 * nbins <= 8192: LDS histogram per workgroup, then atomic adds -- the order of the terms
 * is not fixed (equal to the serial loop :394-400 to rounding).  More bins: cm2_Pt_apply. */
int cm2_ground_bin_sums(int64_t nt, int nbins, const int32_t *d_bin, const double *d_v,
                        double *d_sums, void *stream);
int cm2_ground_subtract(int64_t nt, const int32_t *d_bin, const double *d_binned,
                        const double *d_v, double *d_out, void *stream);

/* ---- f3: solution vector <-> full-sky HEALPix maps -------------------------------
 * cm2_cutsky_to_fullsky: d_full is pol arrays of nfull doubles one after the other
 * ([I | Q | U]); component k of observed pixel i, d_map[pol*i + k], goes to
 * d_full[k*nfull + d_obspix[i]], all other pixels are 0 (reorganize_map,
 * utilities/healpy_functions.py:47-102).  cm2_fullsky_to_cutsky is the inverse gather
 * (full2cutskymap, utilities/IOfiles.py:377-393).  Pixel ids outside [0, nfull) fail. */
int cm2_cutsky_to_fullsky(int pol, int64_t npix, const int64_t *d_obspix, const double *d_map,
                          int64_t nfull, double *d_full, void *stream);
int cm2_fullsky_to_cutsky(int pol, int64_t npix, const int64_t *d_obspix, const double *d_full,
                          int64_t nfull, double *d_map, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COSMOMAP2_H */
