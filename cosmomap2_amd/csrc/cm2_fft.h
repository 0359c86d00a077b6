// cm2_fft.h -- fused overlap-save Toeplitz application (LDS-resident fp64 FFT), see cm2_fft.hip
#pragma once
#include "cm2_common.h"

#include <cstdlib>
#include <vector>

namespace cm2 {

struct FusedOS;

bool fused_os_supported(int64_t lambda);
// d_bands: [nblocks][lambda] on the device; off: nblocks+1 block offsets (host)
int fused_os_create(FusedOS **out, const double *d_bands, int64_t lambda,
                    const std::vector<int64_t> &off, hipStream_t stream);
int fused_os_apply(const FusedOS *f, const double *d_v, double *d_out, hipStream_t stream);
// same, with input and output TODs addressed through d_idx[t] (tile-bucketed order of the tile
// plan `plan_id`; the address lists are rebuilt when the plan changes)
// d_tile_off: first address of every pixel tile, [ntiles + 1] (NULL = unknown: the lists are then
// sorted instead of written directly)
// ntiles: pixel tiles of the plan (bounds the address runs of a list; 0 = unknown)
// nvalid: doubles in the two tile-order buffers (0 = unknown)
int fused_os_apply_indexed(FusedOS *f, const uint32_t *d_idx, const int64_t *d_tile_off, uint64_t plan_id,
                           int64_t ntiles, int64_t nvalid, const double *d_v, double *d_out, hipStream_t stream);
int64_t fused_os_length(const FusedOS *f);
void fused_os_destroy(FusedOS *f);
// what the tile-order application uses: kernel[0] = 0 segment-pair kernel (cm2_fft.hip), 16 / 32 =
// one-real-window kernel with that many points per thread (cm2_fft_real.hip); kernel[1] = list
// format (1 plain, 2 run-coded, 0 not built yet); returns the HBM bytes per sample it is built to move
double fused_os_tile_info(const FusedOS *f, int *kernel);

// ---- one real window per workgroup (cm2_fft_real.hip) ----
struct RealOS;
int real_os_create(RealOS **out, int pt, const double *d_bands, int64_t lambda,
                   const std::vector<int64_t> &off, hipStream_t stream);
int real_os_apply(const RealOS *f, const double *d_v, double *d_out, hipStream_t stream);
int real_os_apply_indexed(RealOS *f, const uint32_t *d_idx, const int64_t *d_tile_off, uint64_t plan_id,
                          int64_t ntiles, int64_t nvalid, int want_lists, const double *d_v, double *d_out,
                          hipStream_t stream);
double real_os_tile_bytes_per_sample(const RealOS *f);
int real_os_list_mode(const RealOS *f);
int64_t real_os_window(const RealOS *f);
void real_os_destroy(RealOS *f);

}  // namespace cm2
