// cm2_pixindex.hip -- stable (pixel -> samples) index; see cm2_pixindex.h
#include "cm2_pixindex.h"

#include <hipcub/hipcub.hpp>

namespace cm2 {

void PixIndex::release()
{
    if (d_sorted_t) (void)cm2::dev_free(d_sorted_t);
    if (d_ptr) (void)cm2::dev_free(d_ptr);
    d_sorted_t = nullptr;
    d_ptr = nullptr;
}

// key = pixel id, flagged samples get key npix so that they sort behind every pixel
__global__ __launch_bounds__(256) void k_make_keys(const int32_t *__restrict__ pix, int64_t nt,
                                                    int64_t npix, uint32_t *__restrict__ keys,
                                                    uint32_t *__restrict__ vals,
                                                    unsigned int *__restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += stride) {
        const int32_t p = pix[i];
        if (p < -1 || p >= npix) atomicAdd(bad, 1u);
        keys[i] = (p < 0 || p >= npix) ? (uint32_t)npix : (uint32_t)p;
        vals[i] = (uint32_t)i;
    }
}

// ptr[p] = first position in the sorted keys whose key is >= p   (p = 0..npix)
__global__ __launch_bounds__(256) void k_lower_bound(const uint32_t *__restrict__ keys,
                                                      int64_t nt, int64_t npix,
                                                      int64_t *__restrict__ ptr)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= npix; p += stride) {
        int64_t lo = 0, hi = nt;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < (uint32_t)p) lo = mid + 1; else hi = mid;
        }
        ptr[p] = lo;
    }
}

int build_pixindex(PixIndex &ix, const int32_t *d_pix, int64_t nt, int64_t npix,
                   hipStream_t stream)
{
    CM2_CHECK(nt >= 0 && nt < (int64_t)0xFFFFFFFF, "nt=%lld out of range (must fit uint32)",
              (long long)nt);
    CM2_CHECK(npix > 0 && npix < (int64_t)0x7FFFFFFF, "npix=%lld out of range", (long long)npix);
    ix.nt = nt;
    ix.npix = npix;
    ix.nvalid = 0;
    CM2_HIP(cm2::dev_malloc(&ix.d_ptr, sizeof(int64_t) * (npix + 1)));
    if (nt == 0) {
        CM2_HIP(hipMemsetAsync(ix.d_ptr, 0, sizeof(int64_t) * (npix + 1), stream));
        CM2_HIP(hipStreamSynchronize(stream));
        return 0;
    }
    DevTemp<uint32_t> keys_in, keys_out, vals_in;
    DevTemp<unsigned int> d_bad;
    DevTemp<char> d_temp;
    CM2_HIP(keys_in.alloc(nt));
    CM2_HIP(keys_out.alloc(nt));
    CM2_HIP(vals_in.alloc(nt));
    CM2_HIP(cm2::dev_malloc(&ix.d_sorted_t, sizeof(uint32_t) * nt));
    CM2_HIP(d_bad.alloc(1));
    CM2_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), stream));
    k_make_keys<<<grid_for(nt), kBlock, 0, stream>>>(d_pix, nt, npix, keys_in, vals_in, d_bad);
    CM2_LAUNCH_OK();

    int end_bit = 1;
    while (((int64_t)1 << end_bit) <= npix) ++end_bit;   // keys take values 0..npix
    size_t temp_bytes = 0;
    CM2_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in.p, keys_out.p,
                                               vals_in.p, ix.d_sorted_t, nt, 0, end_bit, stream));
    CM2_HIP(d_temp.alloc(temp_bytes + 16));
    CM2_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp.p, temp_bytes, keys_in.p, keys_out.p,
                                               vals_in.p, ix.d_sorted_t, nt, 0, end_bit, stream));
    k_lower_bound<<<grid_for(npix + 1), kBlock, 0, stream>>>(keys_out, nt, npix, ix.d_ptr);
    CM2_LAUNCH_OK();

    unsigned int h_bad = 0;
    int64_t h_nvalid = 0;
    CM2_HIP(cm2::download(&h_bad, d_bad, sizeof(h_bad), stream));
    CM2_HIP(cm2::download(&h_nvalid, ix.d_ptr + npix, sizeof(int64_t), stream));
    CM2_HIP(hipStreamSynchronize(stream));
    CM2_CHECK(h_bad == 0, "%u samples have a pixel id outside [-1, npix=%lld)", h_bad,
              (long long)npix);
    ix.nvalid = h_nvalid;
    return 0;
}

}  // namespace cm2
