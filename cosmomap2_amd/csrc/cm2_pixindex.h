// cm2_pixindex.h -- pixel-major index of a time-ordered pointing stream.
//
// A stable radix sort of (pixel id -> sample index) gives, for every pixel, the
// list of its samples IN TIME ORDER.  That list is what turns the reference's
// serial scatter loops (x(pix(i)) += ..., interfaces/linearoperators.py:394-400,
// utilities/process_ces.py:480-487) into race-free per-pixel reductions that add
// the terms in exactly the reference's order.
#pragma once
#include "cm2_common.h"

namespace cm2 {

struct PixIndex {
    int64_t nt = 0, npix = 0, nvalid = 0;
    uint32_t *d_sorted_t = nullptr;  // [nt]   sample ids grouped by pixel, time order inside
    int64_t *d_ptr = nullptr;        // [npix+1] start of each pixel's group; ptr[npix] = nvalid
    void release();
};

// Builds the index on `stream` and synchronises.  Samples with pix == -1 are
// left out.  Fails if a pixel id is >= npix or < -1.
int build_pixindex(PixIndex &ix, const int32_t *d_pix, int64_t nt, int64_t npix,
                   hipStream_t stream);

}  // namespace cm2
