// cm2_overlap_save.h -- fused overlap-save Toeplitz application (register-resident fp64 FFT, one real
// window per workgroup), see cm2_overlap_save.hip
#pragma once
#include "cm2_common.h"

#include <cstdlib>
#include <vector>

namespace cm2 {

struct FusedOS;

// A tile-bucketed TOD order as the overlap-save kernel needs to know it (cm2_tiles fills one in).
struct OsPlanView {
    const uint32_t *d_idx = nullptr;      // [nt] time sample -> address in the tile order (kInvalidSample: flagged)
    const int64_t *d_tile_off = nullptr;  // [nspans * ntiles + 1] first address of every segment (span, tile)
                                          // (NULL: unknown -- the lists are then sorted, not written directly)
    int64_t nspans = 1;                   // the order is [span][tile][time], a span = span_samples consecutive
    int64_t span_samples = 0;             // time samples (cm2_tiles.h); one span: the global tile order
    uint64_t plan_id = 0;                 // identity of the plan: the lists an operator keeps are keyed on it
    int64_t ntiles = 0;                   // pixel tiles (bounds the address runs of a list; 0 = unknown)
    int64_t nvalid = 0;                   // doubles in the two tile-order buffers (0 = unknown)
};

bool fused_os_supported(int64_t lambda);
// d_bands: [nblocks][lambda] on the device; off: nblocks+1 block offsets (host).  The environment
// switches (CM2_OS_LISTS, CM2_OS_LIST_BUILD, CM2_OS_FLAT) are read here, once.
int fused_os_create(FusedOS **out, const double *d_bands, int64_t lambda,
                    const std::vector<int64_t> &off, hipStream_t stream);
// time order
int fused_os_apply(const FusedOS *f, const double *d_v, double *d_out, hipStream_t stream);
// Build (or find) the address lists of this operator for the tile plan `pv`: allocates, launches the
// list builders on `stream` and waits for them.  An operator keeps the lists of its three most
// recently used plans; safe to call from several host threads.
int fused_os_prepare_indexed(FusedOS *f, const OsPlanView &pv, hipStream_t stream);
// input and output TODs in the tile order of `pv`.  After fused_os_prepare_indexed for the same plan
// this is one kernel launch: no allocation, no synchronisation (graph capture safe).
int fused_os_apply_indexed(FusedOS *f, const OsPlanView &pv, const double *d_v, double *d_out, hipStream_t stream);
int64_t fused_os_length(const FusedOS *f);
void fused_os_destroy(FusedOS *f);
// kernel[0] = complex points per thread of the window kernel (32), kernel[1] = list format of the most
// recently used plan (1 plain, 2 run-coded, 3 inverse, 0 none yet); returns the HBM bytes per sample
// the tile-order application is built to move
double fused_os_tile_info(const FusedOS *f, int *kernel);

}  // namespace cm2
