// cm2_overlap_save.hip -- banded-Toeplitz N^-1 by overlap-save, ONE REAL WINDOW per workgroup.
//
// Reference semantics: ToeplitzLO.mult, interfaces/linearoperators.py:582-595 (symmetric band,
// ZERO boundary at both ends of every block), dispatched per block as interfaces/blkop.py:195-206.
//
// One workgroup (256 threads x 32 complex points in registers, LDS only as the exchange buffer between
// the radix-32 / 16 / 16 passes: 66 KB, two workgroups per CU) transforms one real window of 2N =
// 16384 samples as a complex signal of N = 8192 points
//
//     z[a] = x[2a] + i x[2a+1],   Z = FFT_N(z),
//     Z'[k] = alpha_k Z[k] + i beta_k conj(Z[N-k]),   z' = IFFT_N(Z') = y[2a] + i y[2a+1]
//
// with two real tables per noise block, alpha = (S - D sin(pi k/N)) / N, beta = D cos(pi k/N) / N,
// S, D = (H[k] +- H[k+N]) / 2 and H the band's real, even spectrum on 2N points: no untangling pass;
// the partner bin N-k lives in ONE other thread (k' = N-k is (263 - t, 31 - m) in the digit-reversed
// register layout), so the pairing costs one more plane exchange.  12288 outputs per window (halo
// 2048 >= lambda - 1 on both sides: overlap 1.33).
//
// On the tile-bucketed order the window is reached through address-sorted lists, in one of three
// formats:
//   plain : a 4-byte address and a 2-byte position per entry (plans with more than 2048 tiles);
//   RC    : run-coded -- a window's samples in one pixel tile are consecutive addresses, so a list is
//           a few hundred runs: per entry 2 bytes (position, bit 15 = "a run starts here"), per run
//           one 4-byte word delta = address - slot, staged in LDS; entry s of run r has address
//           delta[r] + s, r from a ballot and a population count.  2.0 + 4 / run length bytes per
//           entry instead of 6.  Two window halves and two result rounds, each sorted by address;
//   inverse : one address-sorted order per window (and per result window), cut by ADDRESS into rounds.
// History (profiles/r03_*): the segment-pair kernel of rounds 1-2 (cm2_fft.hip,
// A + iB packing: 0.89-0.95 ms at C4), a 16-point / four-workgroups-per-CU variant of this kernel
// (1.00-1.06 ms) and a 512-thread x 16-point variant of the same window (1.28 ms) all lost to this one
// (0.75-0.80 ms) and were removed in round 4.  The templates keep the points per thread as a parameter;
// only PT = 32 is instantiated.
// This translation unit is compiled with FMA contraction ON (results are compared with the direct
// sum at 1e-12, not bit for bit).
#include "cm2_overlap_save.h"

#include <hipcub/hipcub.hpp>
#include <cstring>
#include <memory>
#include <mutex>

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

// What was measured and dropped (each was a compile-time switch of this source until round 4; the
// numbers are in profiles/r03_os_knob_builds.jsonl and DESIGN.md section 3.1): (alpha, beta) in
// batches of 8 bins requested in front of the last forward pass, both window halves' gathers in
// flight together, both result rounds' lists requested up front, the second inverse-list round's
// gathers behind the first round's staging, wave priorities, non-temporal gathers and result stores.
// Kept: non-temporal LIST loads (a list is read once).
template <class T> __device__ __forceinline__ T ld_list(const T *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double ld_gather(const double *p) { return *p; }
__device__ __forceinline__ void st_result(double *p, double x) { *p = x; }

constexpr int kT = 256;           // threads per workgroup
constexpr int kHalo = 2048;       // window halo on both sides (>= lambda - 1)

template <int PT>
struct Geo {
    static constexpr int N = kT * PT;                 // complex points
    static constexpr int W = 2 * N;                   // window samples
    static constexpr int HOP = W - 2 * kHalo;         // outputs per window
    static constexpr int RR = PT == 16 ? 1 : 2;       // result rounds through the LDS buffer
    static constexpr int RSLOTS = (PT - 8) / RR;      // register slots per round (m = 4 + j RSLOTS ...)
    static constexpr int RLEN = 512 * RSLOTS;         // outputs per round
    static constexpr int NLIST = 2 + RR;              // lists per window: two window halves, results
    static constexpr int PER = 2 * N + HOP;           // list entries per window
    static constexpr int LDSD = N + N / 32;           // doubles of the exchange buffer
    static constexpr int BLK = PT / 16;               // radix-16 blocks per thread
    __host__ __device__ static constexpr int list_off(int l) { return l <= 2 ? l * N : 2 * N + (l - 2) * RLEN; }
    __host__ __device__ static constexpr int list_len(int l) { return l < 2 ? N : RLEN; }
};

// Diagnostic build only (-DCM2_OS_STAMPS, never in the shipped library): every workgroup records
// s_memtime at its phase boundaries into a buffer of 8 words per window (profiles/scripts/
// os_stamps.py).  Every lane stores: a branch at a phase boundary splits the kernel's one basic
// block and costs the register allocator 70-150 spilled VGPRs, which would time a different kernel.
// TIMING-ONLY builds (wrong results, never shipped; profiles/r05_os_memory_vs_compute.md): -DCM2_OS_TIMING=1 compiles
// the transforms, their LDS exchanges and the pairing out -- what is left is the window's memory side (lists, run
// tables, gathers, staging, result lists, stores) exactly as the kernel issues it.
#ifndef CM2_OS_TIMING
#define CM2_OS_TIMING 0
#endif
#if CM2_OS_TIMING & 1
#define OS_XF(...) do { } while (0)
#else
#define OS_XF(...) do { __VA_ARGS__; } while (0)
#endif

#ifdef CM2_OS_STAMPS
static unsigned long long *g_os_stamps_host = nullptr;      // set by cm2_os_debug_stamps
static unsigned long long *os_stamp_buf()                   // never NULL: a dummy when none is wanted
{
    static unsigned long long *dummy = nullptr;
    if (g_os_stamps_host) return g_os_stamps_host;
    if (!dummy && cm2::dev_malloc(&dummy, sizeof(unsigned long long) * 8 * (1 << 20)) != hipSuccess) abort();
    return dummy;
}
#define OS_STAMP_PARAM , unsigned long long *__restrict__ stamps
#define OS_STAMP_ARG , os_stamp_buf()
#define OS_STAMP(i) (stamps[(int64_t)win * 8 + (i)] = __builtin_amdgcn_s_memtime())
#else
#define OS_STAMP_PARAM
#define OS_STAMP_ARG
#define OS_STAMP(i) do { } while (0)
#endif

struct WinDesc {              // one workgroup's work: HOP (or fewer) outputs of one noise block
    int64_t start, len, lo, hi;
    int32_t blk, pad;
};

struct ListHdr {              // one run-coded list
    uint32_t nvalid;          // entries with a sample (they come first: invalid keys sort last)
    uint32_t nruns;
    int32_t wbase[4];         // run index in front of each wave's first slot (-1: none)
    uint32_t pad[2];
};

// ---- register layouts, as in cm2_fft.hip with 32 -> PT -------------------------------------------
//   P1: a = t + 256 m                      radix-PT pass over stride 256   (n = N)
//   P2: a = (16 (m>>4) + (t>>4)) 256 + (t&15) + 16 (m&15)      radix-16 over stride 16 (n = 256)
//   P3: a = PT t + m                       radix-16 on contiguous points
// padded LDS index padi(a) = a + (a >> 5) split into a per-thread base and a compile-time offset
template <int PT, int L>
__device__ __forceinline__ int reg_base(int t)
{
    if (L == 1) return t + (t >> 5);
    if (L == 2) return (t >> 4) * 264 + (t & 15);
    return PT == 32 ? 33 * t : 16 * t + (t >> 1);
}
template <int L>
__host__ __device__ constexpr int reg_off(int m)
{
    return L == 1 ? 264 * m : (L == 2 ? 4224 * (m >> 4) + 16 * (m & 15) + ((m & 15) >> 1) : m);
}
// where slot m of the array sits after in-place butterflies (dft_sub leaves output m of a radix-R
// block at index brev<R>(m)): PERM = 0 natural, 32 one radix-32 block, 16 radix-16 blocks
template <int PERM>
__host__ __device__ constexpr int reg_slot(int m)
{
    return PERM == 32 ? brev<32>(m) : (PERM == 16 ? 16 * (m >> 4) + brev<16>(m & 15) : m);
}

template <int PT, int FROM, int TO, int PERM>
__device__ __forceinline__ void reg_exchange(double (&a)[PT], double *__restrict__ buf, int t)
{
    double *__restrict__ wp = buf + reg_base<PT, FROM>(t);
    const double *__restrict__ rp = buf + reg_base<PT, TO>(t);
#pragma unroll
    for (int m = 0; m < PT; ++m) wp[reg_off<FROM>(m)] = a[reg_slot<PERM>(m)];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < PT; ++m) a[m] = rp[reg_off<TO>(m)];
    __syncthreads();
}

// in-place decimation-in-frequency butterflies on the sub-block [OFF, OFF + R)
template <int PT, int R, int OFF>
__device__ __forceinline__ void dft_sub(double (&re)[PT], double (&im)[PT])
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

// decimation-in-time counterpart: input m at index brev<R>(m), output natural
template <int PT, int R, int OFF>
__device__ __forceinline__ void dit_sub(double (&re)[PT], double (&im)[PT])
{
#pragma unroll
    for (int h = 1; h <= R / 2; h <<= 1) {
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int i = 0; i < h; ++i) {
                const int a = OFF + blk + i, b = a + h;
                const int tw = i * (32 / (2 * h));
                double tr, ti;
                if (tw == 0) {
                    tr = re[b];
                    ti = im[b];
                } else if (tw == 8) {
                    tr = im[b];
                    ti = -re[b];
                } else {
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
template <int PT, int R, int OFF>
__device__ __forceinline__ void reg_fwd(double (&xr)[PT], double (&xi)[PT], double2 w1)
{
    dft_sub<PT, R, OFF>(xr, xi);
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

// inverse pass on a block: input m (natural index) times conj(w1^m), then the inverse butterfly
// (swap . forward . swap); output m ends at index brev(m)
template <int PT, int R, int OFF>
__device__ __forceinline__ void reg_inv(double (&xr)[PT], double (&xi)[PT], double2 w1)
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
    dft_sub<PT, R, OFF>(xi, xr);
}

// ---- list walks ---------------------------------------------------------------------------------
// MODE 0: time order (addresses computed), 1: plain lists, 2: run-coded lists.
// Slot of entry u of this thread in a list of E entries per thread: every wave owns a contiguous
// range of the list, a wave instruction covers 64 consecutive slots.
template <int E>
__device__ __forceinline__ int slot_of(int t, int u) { return 64 * (E * (t >> 6) + u) + (t & 63); }

// Where the 16-bit word of slot s of a list with E entries per thread is STORED: the words of a
// thread's entries 4i .. 4i+3 share one 8-byte word, word (E/4 wave + i) 64 + lane of the list, so a
// thread fetches its E words with E/4 coalesced 8-byte loads into E/2 registers (one 2-byte load and
// one register per entry before round 4: the 24-32 list words held across the last transform pass
// were what pushed the kernel over 256 VGPRs).  The plan-time kernels write through this map.
__host__ __device__ inline int q_index(int s, int E)
{
    const int row = s >> 6, lane = s & 63, w = row / E, u = row % E;
    return (((E / 4) * w + (u >> 2)) * 64 + lane) * 4 + (u & 3);
}
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
// this thread's E list words, two per register
template <int E>
__device__ __forceinline__ void q_request(const uint16_t *q, int t, uint32_t (&qq)[E / 2])
{
    asm volatile("" : "+v"(t));               // (the lane's list offset is recomputed per list, not kept: see tab_dma)
    const v2u_t *p = reinterpret_cast<const v2u_t *>(q) + (E / 4) * 64 * (t >> 6) + (t & 63);
#pragma unroll
    for (int i = 0; i < E / 4; ++i) {
        const v2u_t v = __builtin_nontemporal_load(p + 64 * i);
        qq[2 * i] = v.x;
        qq[2 * i + 1] = v.y;
    }
}
template <int E2>
__device__ __forceinline__ uint32_t q_word(const uint32_t (&qq)[E2], int u) { return (qq[u >> 1] >> (16 * (u & 1))) & 0xFFFFu; }

struct ListArgs {
    const uint32_t *k;        // plain: addresses of this list
    const uint16_t *q;        // plain / RC: positions of this list
    const ListHdr *hdr;       // RC
    const uint32_t *tab;      // RC: run table of this list in global memory
};

// RC: the run table of a list (rmax words, a multiple of 64; the table is read up to its ALLOCATED
// length, so the request does not wait for the list header) goes from global memory straight into
// LDS (global_load_lds_dword: lane l of a wave writes word l behind the wave-uniform LDS base in M0):
// no VGPR holds a table word.  The data is in LDS once the issuing wave's vmcnt has drained; the
// __syncthreads() that publishes it to the other waves waits for that (the compiler puts
// s_waitcnt vmcnt(0) in front of the barrier).
// At most kTabRows requests per wave (rmax <= 256 kTabRows), unrolled behind wave-uniform tests: with a
// loop of unknown length the compiler cannot count the requests in flight and every later wait for
// an OLDER load becomes s_waitcnt vmcnt(0), i.e. a wait for the table and the list words as well.
constexpr int kTabRows = 8;
__device__ __forceinline__ void tab_dma(const uint32_t *gtab, uint32_t *tab_lds, int rmax, int wave_, int t)
{
    // (the wave index passes through an empty asm statement at every call: the eight wave-uniform tests below
    //  are then scalar compares made where they are used -- shared between the window lists at the top of the
    //  kernel and the result lists at its end they were eight 64-bit masks kept across the whole transform,
    //  16 SGPRs spilled to VGPR lanes in the default instantiation)
    int wave = wave_;
    asm volatile("" : "+s"(wave));
    // (and the thread index through a "+v": the lane's 64-bit table offset is recomputed at every call -- three
    //  VALU instructions -- instead of being kept, or spilled, from the window lists to the result lists)
    asm volatile("" : "+v"(t));
    const uint32_t *g = gtab + 64 * wave + (t & 63);
#pragma unroll
    for (int i = 0; i < kTabRows; ++i)
        if (64 * wave + kT * i < rmax)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + kT * i),
                                             (__attribute__((address_space(3))) void *)(tab_lds + 64 * wave + kT * i),
                                             4, 0, 0);
}

// RC: addresses of E entries from their 16-bit words (bit 15 = run start) and the run table in LDS
template <int E>
__device__ __forceinline__ void rc_decode(const uint32_t (&qq)[E / 2], const uint32_t *__restrict__ tab_lds,
                                          int wbase, uint32_t nvalid, int t, uint32_t (&kk)[E])
{
    int rb = wbase;
    // slot of entry u = s0 + 64 u; s0 passes through an empty asm statement so that the E slot
    // numbers are recomputed here (one add each) instead of being kept live from list to list
    uint32_t s0 = (uint32_t)slot_of<E>(t, 0);
    asm volatile("" : "+v"(s0));
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const bool flag = (qq[u >> 1] & (0x8000u << (16 * (u & 1)))) != 0u;
        const uint64_t mask = __ballot(flag);
        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        int r = rb + below + (flag ? 1 : 0);
        rb += __popcll(mask);
        r = r < 0 ? 0 : r;
        const uint32_t s = s0 + 64u * (uint32_t)u;
        const uint32_t a = tab_lds[r] + s;           // read for every entry: no branch, no wait per entry
        kk[u] = s < nvalid ? a : kInvalidSample;
    }
}

// ---- inverse lists (MODE 3) ----------------------------------------------------------------------
// The lists above are cut by TIME (two window halves, two result rounds), each sorted by address: a
// tile's samples of one window are consecutive addresses, but a half window holds only half of them
// (runs of 16 entries = 128 bytes at C4, 12 = 96 bytes on the result side), and every run ends in
// partly used 64-byte sectors.  An inverse list keeps ONE address-sorted order per window (and one per
// result window) and cuts it by ADDRESS: round j moves slots [j R, (j + 1) R) -- whole runs of 32
// (24) entries -- linearly through the LDS buffer, and every thread picks (or places) its own points
// by slot number: the list stored per POSITION is its slot (u16, 0xFFFF = no sample), read in
// register order.  The slot -> address direction needs only the run table and one bit per slot (a
// run starts here), kept transposed: bit u of word [round][thread] belongs to the thread's u-th slot
// of that round.
struct IListHdr {
    uint32_t nvalid, nruns;
    int32_t wbase[16];        // [round][wave] (4 or 8 waves a workgroup): run index in front of the wave's
                              // first slot of the round
};

// addresses of this thread's E slots of the round that starts at list slot `soff`
template <int E>
__device__ __forceinline__ void idecode(uint32_t fw, const uint32_t *__restrict__ tab_lds, int wbase,
                                        uint32_t nvalid, uint32_t soff, int t, uint32_t (&kk)[E])
{
    int rb = wbase;
    uint32_t s0 = soff + (uint32_t)slot_of<E>(t, 0);
    asm volatile("" : "+v"(s0));
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const bool flag = ((fw >> u) & 1u) != 0u;
        const uint64_t mask = __ballot(flag);
        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        int r = rb + below + (flag ? 1 : 0);
        rb += __popcll(mask);
        r = r < 0 ? 0 : r;
        const uint32_t s = s0 + 64u * (uint32_t)u;
        const uint32_t a = tab_lds[r] + s;
        kk[u] = s < nvalid ? a : kInvalidSample;
    }
}

// ---- the partner exchange and the spectrum product: pairing by HALF PLANES (round 5) ------------------
// Z'[k] = alpha Z[k] + i beta conj(Z[N-k]):  re' = alpha re + beta pim,  im' = alpha im + beta pre.
// Frequency (e, d3) of a thread sits in register slot 16 e + brev16(d3), P3 index m = 16 e + d3.
// Rounds 3-4 published the real plane, read the partner's 32 real parts into registers (64 VGPRs), published the
// imaginary plane and walked the bins with ONE batch of four (alpha, beta) pairs in flight: eight dependent table
// loads a window, seven of them exposed (19 k of the window's 109 k clocks).  Here a thread publishes only its
// UPPER slots (P3 index 16..31), both parts, in the one plane buffer -- real part at 33 t + 16 + j, imaginary
// part at 33 t + j -- and the holder of the LOWER slot of a bin pair (k, N - k) computes BOTH outputs: it has
// Z[k] in registers, reads Z[N-k] from the partner's published slots, writes Z'[N-k] back to the same two words
// (no other thread touches them), and the partner reads its new upper slots back behind one barrier.  No partner
// array in registers: the coefficients travel in TWO batches of eight bin pairs (16 double2 = the partner
// array's 64 registers), the first requested in front of the publication: one exposed round trip instead of
// seven; three barriers instead of four; the same LDS reads, 32 more LDS writes per thread.  (All 32 double2 at
// once, into the upper slots' registers that are dead between publication and read-back, was tried: the
// register allocator spills 100 VGPRs.)  Five same-box alternations of bench.py: N^-1 0.673 against 0.696 ms
// (every pair), step 1.378 against 1.390 (four of five pairs); profiles/r05_os_half_plane_pairing_ab.jsonl.
//   t >= 8: (t, m < 16) pairs with (263 - t, 31 - m): lower <-> upper.
//   t <  8 (frequency digit d1 = 0: 8 threads of wave 0): lower slots pair with LOWER slots (thread 8 - t, slot
//   15 - m; thread 0 with itself, slot 16 - m; bins 0 and N/2 with themselves) and upper with upper (thread
//   7 - t, slot 47 - m).  They also publish their lower slots (xbuf: 2 x 8 x 16 doubles behind the run tables),
//   compute their own 32 bins from the published ORIGINALS (nobody writes to their slots: the partners of
//   threads >= 8 are threads 8..255) with the same two batches of loads -- their "partner" coefficients are their
//   own upper bins' -- and skip the read-back.
template <int PT, int BP>                   // BP: bin pairs per coefficient batch (8; 4 where registers are shortest)
__device__ __forceinline__ void partner_filter_half(double (&zr)[PT], double (&zi)[PT], double *__restrict__ buf,
                                                    double *__restrict__ xbuf, int t,
                                                    const double2 *__restrict__ ab_own,
                                                    const double2 *__restrict__ ab_blk)
{
    static_assert(PT == 32, "half-plane pairing: 32 points per thread");
    constexpr int H = PT / 2;
    double *__restrict__ wp = buf + reg_base<PT, 3>(t);                  // 33 t
    const bool special = t < 8;
    // the coefficients of bin 31 - m: the partner's (general case) or this thread's own upper bin (special case)
    // -- ONE load sequence serves both cases
    const double2 *__restrict__ ab_prt = ab_blk + (special ? t : 263 - t);
    double2 ca[BP], cb[BP];
    auto request = [&](int m0) {
        // The OFFSET passes through an empty asm statement: the loads have no other dependency and would otherwise
        // all be hoisted to one place.  The pointer itself must keep its provenance: a laundered pointer is loaded
        // from with FLAT instructions, and one pending flat load turns every later wait into s_waitcnt vmcnt(0)
        // lgkmcnt(0) (flat loads may return out of order).
        int o = m0 * kT;
        asm volatile("" : "+v"(o));
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            ca[i] = ab_own[o + i * kT];
            cb[i] = ab_prt[(PT - 1) * kT - o - i * kT];
        }
    };
    request(0);                               // arrives behind the publication and its barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < H; ++j) {
        wp[H + j] = zr[reg_slot<16>(H + j)];
        wp[j] = zi[reg_slot<16>(H + j)];
    }
    if (special) {
#pragma unroll
        for (int j = 0; j < H; ++j) {
            xbuf[t * H + j] = zr[reg_slot<16>(j)];
            xbuf[(8 + t) * H + j] = zi[reg_slot<16>(j)];
        }
    }
    __syncthreads();
    if (!special) {
        double *__restrict__ pp = buf + 33 * (263 - t) + 31;             // partner slot 31 - m: re at pp[-m], im at pp[-16 - m]
#pragma unroll
        for (int m0 = 0; m0 < H; m0 += BP) {
            if (m0 > 0) request(m0);
#pragma unroll
            for (int i0 = 0; i0 < BP; i0 += 4) {
                double pre[4], pim[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pre[i] = pp[-(m0 + i0 + i)];
                    pim[i] = pp[-H - (m0 + i0 + i)];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + i0 + i, sl = reg_slot<16>(m);
                    const double a = ca[i0 + i].x, b = ca[i0 + i].y, ap = cb[i0 + i].x, bp = cb[i0 + i].y;
                    const double nr = a * zr[sl] + b * pim[i];
                    const double ni = a * zi[sl] + b * pre[i];
                    const double qr = ap * pre[i] + bp * zi[sl];         // Z'[N-k] = alpha' Z[N-k] + i beta' conj(Z[k])
                    const double qi = ap * pim[i] + bp * zr[sl];
                    zr[sl] = nr;
                    zi[sl] = ni;
                    pp[-m] = qr;
                    pp[-H - m] = qi;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // the eight threads whose partner bins sit in the same half: every own bin from published originals
        // (lower bin m with ca, upper bin 31 - m with cb: the same two batches of loads as the other threads)
        const int tl = t == 0 ? 0 : 8 - t;                               // lower slots' partner thread
        const double *__restrict__ up = buf + 33 * (7 - t) + 47;         // upper: re at up[-m], im at up[-m - 16]
#pragma unroll
        for (int m0 = 0; m0 < H; m0 += BP) {
            if (m0 > 0) request(m0);
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int m = m0 + i, sl = reg_slot<16>(m);
                const int ms = t == 0 ? (m == 0 ? 0 : H - m) : H - 1 - m;
                const double pre_ = xbuf[tl * H + ms], pim_ = xbuf[(8 + tl) * H + ms];
                const double nr = ca[i].x * zr[sl] + ca[i].y * pim_;
                const double ni = ca[i].x * zi[sl] + ca[i].y * pre_;
                zr[sl] = nr;
                zi[sl] = ni;
                const int mu = PT - 1 - m, su = reg_slot<16>(mu);       // the upper bin 31 - m
                const double ure = up[-mu], uim = up[-mu - H];
                // (its own upper originals are read back from the slots it published, so that the upper registers
                //  are dead from the publication on for every lane of the wave: liveness is per register)
                zr[su] = cb[i].x * wp[mu] + cb[i].y * uim;
                zi[su] = cb[i].x * wp[mu - H] + cb[i].y * ure;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if (!special) {
#pragma unroll
        for (int j = 0; j < H; ++j) {
            zr[reg_slot<16>(H + j)] = wp[H + j];
            zi[reg_slot<16>(H + j)] = wp[j];
        }
    }
    __syncthreads();
}

// MODE 0: time order; 1: plain lists; 2: run-coded lists; 3: inverse lists.
// BUF: the TOD buffers are addressed through buffer descriptors of `nbytes` bytes -- an entry
// without a sample carries the address 0xFFFFFFFF, whose byte offset lies outside the descriptor:
// such a load returns 0 and such a store is dropped by the hardware, so the gathers need no
// address clamp and no zeroing select and the result stores no branch (48 exec-masked blocks in
// the flat form).  Flat addressing is kept for buffers of 4 GB and more.

template <int PT, int MODE, bool BUF>
__global__ __launch_bounds__(kT, PT == 16 ? 4 : 2) void k_os_real(
    const WinDesc *__restrict__ wins, int nwin, const double2 *__restrict__ Wtw,
    const double2 *Wtw_inv, const double2 *__restrict__ AB, const uint32_t *__restrict__ lst_k,
    const uint16_t *__restrict__ lst_q, const ListHdr *__restrict__ hdrs,
    const uint32_t *__restrict__ tabs, int rmax, const double *__restrict__ v,
    double *__restrict__ out, uint32_t nbytes, const IListHdr *__restrict__ ihdrs,
    const uint32_t *__restrict__ iflags OS_STAMP_PARAM)
{
    using G = Geo<PT>;
    constexpr int N = G::N, H = PT / 2;
    __amdgpu_buffer_rsrc_t v_rs, o_rs;
    if constexpr (BUF) {
        v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(v), 0, (int)nbytes, 0x00020000);
        o_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)nbytes, 0x00020000);
    }
    // sample at tile-order address k (0xFFFFFFFF: none)
    auto gather = [&](uint32_t k) -> double {
        if constexpr (BUF) {
            const v2u_t r = __builtin_amdgcn_raw_buffer_load_b64(v_rs, k * 8u, 0, 0);
            return __builtin_bit_cast(double, r);
        } else {
            return ld_gather(v + (k != kInvalidSample ? k : 0u));
        }
    };
    auto keep = [&](uint32_t k, double x) -> double {        // value staged for entry k
        if constexpr (BUF) return x;
        return k != kInvalidSample ? x : 0.0;
    };
    extern __shared__ double buf[];
    uint32_t *__restrict__ tab_lds = reinterpret_cast<uint32_t *>(buf + G::LDSD);   // RC: 2 x rmax words
    const int t = threadIdx.x;
    const int per_xcd = (nwin + 7) / 8;
    const int win = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (win >= nwin) return;
    const WinDesc wd = wins[win];
    const int64_t w0 = wd.start - kHalo;            // time of window position 0
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: an SGPR, the header words become scalar loads
    OS_STAMP(0);

    auto list_args = [&](int l) {
        ListArgs la;
        const int64_t e0 = (int64_t)win * G::PER + G::list_off(l);
        // (no null tests: the launcher hands every MODE the arrays it reads -- a test made here is a 64-bit
        //  mask the compiler keeps from the first use of a list to the last, across the whole transform)
        la.k = nullptr;
        la.q = nullptr;
        la.hdr = nullptr;
        la.tab = nullptr;
        if constexpr (MODE == 1) la.k = lst_k + e0;
        if constexpr (MODE == 1 || MODE == 2) la.q = lst_q + e0;
        if constexpr (MODE == 2) {
            la.hdr = hdrs + ((int64_t)win * G::NLIST + l);
            la.tab = tabs + ((int64_t)win * G::NLIST + l) * rmax;
        }
        return la;
    };

    double zr[PT], zi[PT];
    // ---- load the window, one half (N positions) at a time through the LDS buffer --------------
    if constexpr (MODE == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double vv[PT];
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                const int64_t ts = w0 + (int64_t)h * N + t + u * kT;
                vv[u] = (ts >= wd.lo && ts < wd.hi) ? v[ts] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PT; ++u) buf[t + u * kT] = vv[u];
            __syncthreads();
            const double2 *__restrict__ sp = reinterpret_cast<const double2 *>(buf) + t;
#pragma unroll
            for (int m = 0; m < H; ++m) {
                const double2 p = sp[256 * m];
                zr[h * H + m] = p.x;
                zi[h * H + m] = p.y;
            }
            __syncthreads();
        }
    } else if constexpr (MODE == 3) {
        // inverse lists: two rounds of N slots of the window's address-sorted order.  Header, run
        // table, flag words and this thread's slot numbers are requested together (the table is
        // read up to its allocated length: no dependency on the run count).
        const IListHdr *__restrict__ h0 = ihdrs + (int64_t)win * 2;
        const uint32_t *__restrict__ pl = reinterpret_cast<const uint32_t *>(lst_q + (int64_t)win * G::PER);
        uint32_t fw[2], pp[PT];
        {
            tab_dma(tabs + ((int64_t)win * 2) * rmax, tab_lds, rmax, wave, t);
            fw[0] = iflags[((int64_t)win * 2) * 512 + t];
            fw[1] = iflags[((int64_t)win * 2) * 512 + 256 + t];
        }
        const uint32_t nv = h0->nvalid;
        const int wb0 = h0->wbase[wave], wb1 = h0->wbase[4 + wave];
#pragma unroll
        for (int m = 0; m < PT; ++m) pp[m] = pl[t + kT * m];
        __syncthreads();
        // both rounds' gathers are issued before anything is staged: 2 PT loads in flight per thread
        // while the transform's registers are not live yet
        double va[PT], vb[PT];
        {
            uint32_t kk[PT];
            idecode<PT>(fw[0], tab_lds, wb0, nv, 0u, t, kk);
#pragma unroll
            for (int u = 0; u < PT; ++u) va[u] = keep(kk[u], gather(kk[u]));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            uint32_t kk[PT];
            idecode<PT>(fw[1], tab_lds, wb1, nv, (uint32_t)N, t, kk);
#pragma unroll
            for (int u = 0; u < PT; ++u) vb[u] = keep(kk[u], gather(kk[u]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int u = 0; u < PT; ++u) buf[slot_of<PT>(t, u)] = j ? vb[u] : va[u];
            if (j == 1) {
                // the slot numbers again (L2): 32 registers not held across the first round's picks
                int tj = t;
                asm volatile("" : "+v"(tj));
#pragma unroll
                for (int m = 0; m < PT; ++m) pp[m] = pl[tj + kT * m];
            }
            __syncthreads();
            // a point outside this round reads word 0 (one address for all such lanes: no bank conflict)
#pragma unroll
            for (int m = 0; m < PT; ++m) {
                const uint32_t lo = (pp[m] & 0xFFFFu) - (uint32_t)(j * N), hi = (pp[m] >> 16) - (uint32_t)(j * N);
                const bool inl = lo < (uint32_t)N, inh = hi < (uint32_t)N;
                const double x = buf[inl ? lo : 0u], y = buf[inh ? hi : 0u];
                if (j == 0) {
                    zr[m] = inl ? x : 0.0;
                    zi[m] = inh ? y : 0.0;
                } else {
                    zr[m] = inl ? x : zr[m];
                    zi[m] = inh ? y : zi[m];
                }
            }
            __syncthreads();
        }
    } else {
        const ListArgs l0 = list_args(0), l1 = list_args(1);
        uint32_t qa[PT / 2], qb[PT / 2], ka[PT], kb[PT];
        uint32_t nva = 0, nvb = 0;
        int wba = -1, wbb = -1;
        if constexpr (MODE == 2) {
            nva = l0.hdr->nvalid;
            nvb = l1.hdr->nvalid;
            wba = l0.hdr->wbase[wave];
            wbb = l1.hdr->wbase[wave];
            tab_dma(l0.tab, tab_lds, rmax, wave, t);
            tab_dma(l1.tab, tab_lds + rmax, rmax, wave, t);
        }
        q_request<PT>(l0.q, t, qa);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < PT; ++u) ka[u] = ld_list(l0.k + slot_of<PT>(t, u));
        }
        q_request<PT>(l1.q, t, qb);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < PT; ++u) kb[u] = ld_list(l1.k + slot_of<PT>(t, u));
        }
        if constexpr (MODE == 2) {
            __syncthreads();
            OS_STAMP(6);                             // (diagnostic build: lists and run tables have arrived)
            rc_decode<PT>(qa, tab_lds, wba, nva, t, ka);
        }
        __builtin_amdgcn_sched_barrier(0);
        // Both halves' gathers are in flight together (2 PT loads per thread: the transform's registers
        // are not live yet), half a is staged while half b is still on its way: two dependent round
        // trips (lists, gathers) instead of three.  Five same-box alternations of bench.py: step 1.478
        // against 1.507 ms (-1.9 %, every pair); one noisy pair had hidden it earlier in the round.
        double va[PT], vv[PT];
        if constexpr (MODE == 2) rc_decode<PT>(qb, tab_lds + rmax, wbb, nvb, t, kb);
#pragma unroll
        for (int u = 0; u < PT; ++u) va[u] = gather(ka[u]);
#pragma unroll
        for (int u = 0; u < PT; ++u) vv[u] = gather(kb[u]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PT; ++u) buf[q_word(qa, u) & 0x7FFFu] = keep(ka[u], va[u]);
        OS_STAMP(7);                                 // (diagnostic build: the first half's gathers have arrived)
        __syncthreads();
        {
            const double2 *__restrict__ sp = reinterpret_cast<const double2 *>(buf) + t;
#pragma unroll
            for (int m = 0; m < H; ++m) {
                const double2 p = sp[256 * m];
                zr[m] = p.x;
                zi[m] = p.y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PT; ++u) buf[q_word(qb, u) & 0x7FFFu] = keep(kb[u], vv[u]);
        __syncthreads();
        {
            const double2 *__restrict__ sp = reinterpret_cast<const double2 *>(buf) + t;
#pragma unroll
            for (int m = 0; m < H; ++m) {
                const double2 p = sp[256 * m];
                zr[H + m] = p.x;
                zi[H + m] = p.y;
            }
        }
        __syncthreads();
    }

    OS_STAMP(1);
    const double2 w_a = Wtw[t];                      // n = N:   exp(-2 pi i t / N)
    const double2 w_b = Wtw[PT * (t & 15)];          // n = 256: exp(-2 pi i (t & 15) / 256)

    // ---- forward: radix PT, radix 16, radix 16 ----
    OS_XF(reg_fwd<PT, PT, 0>(zr, zi, w_a));
    OS_XF(reg_exchange<PT, 1, 2, PT>(zr, buf, t));
    OS_XF(reg_exchange<PT, 1, 2, PT>(zi, buf, t));
    OS_XF(reg_fwd<PT, 16, 0>(zr, zi, w_b));
    OS_XF(if constexpr (PT == 32) reg_fwd<PT, 16, 16>(zr, zi, w_b));
    OS_XF(reg_exchange<PT, 2, 3, 16>(zr, buf, t));
    OS_XF(reg_exchange<PT, 2, 3, 16>(zi, buf, t));
    const double2 *ab = AB + (int64_t)wd.blk * N + t;
    OS_XF(dft_sub<PT, 16, 0>(zr, zi));
    OS_XF(if constexpr (PT == 32) dft_sub<PT, 16, 16>(zr, zi));
    OS_STAMP(2);
    // ---- pairing with bin N-k and the spectrum product ----
    // (plain lists with flat addressing -- more than 2048 pixel tiles AND buffers of 4 GB and more -- hold 32
    //  address words beside the transform: batches of four bin pairs there, or three VGPRs spill)
    OS_XF(partner_filter_half<PT, (MODE == 1 && !BUF) ? 4 : 8>(zr, zi, buf, reinterpret_cast<double *>(tab_lds + 2 * rmax),
                                                               t, ab, AB + (int64_t)wd.blk * N));
    (void)ab;
    OS_STAMP(3);
    // ---- inverse: radix 16 (decimation in time on the bit-reversed data), radix 16, radix PT ----
    OS_XF(dit_sub<PT, 16, 0>(zi, zr));
    OS_XF(if constexpr (PT == 32) dit_sub<PT, 16, 16>(zi, zr));
    // the inverse passes read their twiddles again (through a second pointer to the same table,
    // so that nothing of the forward passes stays live across the pairing step)
    const double2 w_bi = Wtw_inv[PT * (t & 15)], w_ai = Wtw_inv[t];
    OS_XF(reg_exchange<PT, 3, 2, 0>(zr, buf, t));
    OS_XF(reg_exchange<PT, 3, 2, 0>(zi, buf, t));
    // The result lists of round 0 are requested HERE, in front of the middle inverse pass: a run table
    // travels by LDS-DMA, and while one is in flight every workgroup barrier waits for it (s_waitcnt
    // vmcnt(0) in front of s_barrier) -- so the request is issued right behind a barrier, with the
    // longest barrier-free stretch of arithmetic (the two radix-16 blocks) to hide behind; the 16-bit
    // words are in registers by the end of the last pass.  Both twiddles are made to arrive first
    // (requested two exchanges ago): a wait for a load OLDER than the table requests would be emitted as
    // s_waitcnt vmcnt(0) too -- the compiler does not count LDS-DMA requests behind wave-uniform
    // branches -- and would wait for the lists as well.
    asm volatile("" : : "v"(w_bi.x), "v"(w_bi.y), "v"(w_ai.x), "v"(w_ai.y));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 3) {
        // ---- inverse result list: rounds of RLEN slots of the result window's address-sorted order ----
        constexpr int ER3 = G::RLEN / kT, NP = PT - 8;   // slots per thread and round; points with results
        // (the list number passes through an empty asm statement -- an offset, not a pointer, see
        // partner_filter_half: the requests below have no other dependency and would be hoisted to the top)
        int l1 = 1;
        asm volatile("" : "+s"(l1));
        const IListHdr *h1 = ihdrs + (int64_t)win * 2 + l1;
        const uint32_t *tg = tabs + ((int64_t)win * 2 + l1) * rmax;
        const uint32_t *fg = iflags + ((int64_t)win * 2 + l1) * 512 + t;
        const uint32_t nv1 = h1->nvalid;
        uint32_t fr[G::RR], rp[NP];
        const uint32_t *rl = reinterpret_cast<const uint32_t *>(lst_q + (int64_t)win * G::PER + 2 * N);
        {
            tab_dma(tg, tab_lds + rmax, rmax, wave, t);        // published by the barriers of the next exchange
#pragma unroll
            for (int j = 0; j < G::RR; ++j) fr[j] = fg[256 * j];
            // the slot numbers of this thread's results, in registers by the end of the last pass
            int t0 = t;
            asm volatile("" : "+v"(t0));
#pragma unroll
            for (int m = 0; m < NP; ++m) rp[m] = rl[t0 + kT * m];
        }
        __builtin_amdgcn_sched_barrier(0);
        OS_XF(reg_inv<PT, 16, 0>(zr, zi, w_bi));
        OS_XF(if constexpr (PT == 32) reg_inv<PT, 16, 16>(zr, zi, w_bi));
        OS_XF(reg_exchange<PT, 2, 1, 16>(zr, buf, t));
        OS_XF(reg_exchange<PT, 2, 1, 16>(zi, buf, t));
        OS_XF(reg_inv<PT, PT, 0>(zr, zi, w_ai));   // result slot m at index brev<PT>(m)
        OS_STAMP(4);
#pragma unroll
        for (int j = 0; j < G::RR; ++j) {
            if (j > 0) {
                __syncthreads();                         // the previous round's reads are done
                int tj = t;                              // (the slot numbers again, from L2)
                asm volatile("" : "+v"(tj));
#pragma unroll
                for (int m = 0; m < NP; ++m) rp[m] = rl[tj + kT * m];
            }
            // y[2 (t + 256 m)] = zr, y[.. + 1] = zi for m in [4, PT - 4): each value to its slot of this
            // round, the others to a spare word behind the stage (no branch)
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const uint32_t lo = (rp[m] & 0xFFFFu) - (uint32_t)(j * G::RLEN), hi = (rp[m] >> 16) - (uint32_t)(j * G::RLEN);
                buf[lo < (uint32_t)G::RLEN ? lo : (uint32_t)N] = zr[brev<PT>(m + 4)];
                buf[hi < (uint32_t)G::RLEN ? hi : (uint32_t)N + 1u] = zi[brev<PT>(m + 4)];
            }
            __syncthreads();
            uint32_t ks3[ER3];
            idecode<ER3>(fr[j], tab_lds + rmax, h1->wbase[4 * j + wave], nv1, (uint32_t)(j * G::RLEN), t, ks3);
            double rv[ER3];
#pragma unroll
            for (int u = 0; u < ER3; ++u) rv[u] = buf[slot_of<ER3>(t, u)];
#pragma unroll
            for (int u = 0; u < ER3; ++u) {
                if constexpr (BUF) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, rv[u]), o_rs, ks3[u] * 8u, 0, 0);
                } else {
                    if (ks3[u] != kInvalidSample) st_result(out + ks3[u], rv[u]);
                }
            }
        }
    } else {
    constexpr int ER = G::RLEN / kT;                 // result entries per thread and round
    uint32_t qs[ER / 2], ks[ER];
    uint32_t nvs = 0;
    int wbs = -1;
    auto request_results = [&](int j, auto &qq, auto &kq, uint32_t &nv, int &wb, uint32_t *tdst) {
        if constexpr (MODE != 0) {
            // (the list number passes through an empty asm statement: an offset, not a pointer, see
            // partner_filter_half; the list's addresses stay wave-uniform and its header words scalar loads)
            int lj = 2 + j;
            asm volatile("" : "+s"(lj));
            const ListArgs ls = list_args(lj);
            if constexpr (MODE == 2) {
                nv = ls.hdr->nvalid;
                wb = ls.hdr->wbase[wave];
                tab_dma(ls.tab, tdst, rmax, wave, t);
            }
            q_request<ER>(ls.q, t, qq);
            (void)kq;         // (plain lists: the addresses are fetched behind the last pass, see below)
        }
    };
    // EARLY: the second result round's lists are requested behind the first round's staging barrier, so
    // that they travel while the first round's results are stored (the barrier in front of the second
    // round waits for the stores and the lists together: one round trip less per window).  Five
    // same-box alternations of bench.py: step 1.407 against 1.446 ms (-2.7 %).  Not for plain lists
    // with flat addressing (the 12 list words more spill 22 VGPRs there).
    constexpr bool EARLY = MODE == 2 || (MODE == 1 && BUF);
    uint32_t qs1[ER / 2];
    uint32_t nv1 = 0;
    int wb1 = -1;
    request_results(0, qs, ks, nvs, wbs, tab_lds); // (its run table: published by the barriers of the next exchange)
    __builtin_amdgcn_sched_barrier(0);
    OS_XF(reg_inv<PT, 16, 0>(zr, zi, w_bi));
    OS_XF(if constexpr (PT == 32) reg_inv<PT, 16, 16>(zr, zi, w_bi));
    OS_XF(reg_exchange<PT, 2, 1, 16>(zr, buf, t));
    OS_XF(reg_exchange<PT, 2, 1, 16>(zi, buf, t));
    OS_XF(reg_inv<PT, PT, 0>(zr, zi, w_ai));   // result slot m at index brev<PT>(m)
    // ---- store: y[2 (t + 256 m)] = zr, y[.. + 1] = zi for m in [4, PT - 4), RSLOTS slots a round --
    OS_STAMP(4);
#pragma unroll
    for (int j = 0; j < G::RR; ++j) {
        const uint32_t *tabj = tab_lds + (EARLY && j > 0 ? rmax : 0);
        if (j > 0) {
            __syncthreads();                         // the previous round's reads are done (EARLY: and round j's table is in)
            if constexpr (EARLY) {
#pragma unroll
                for (int i = 0; i < ER / 2; ++i) qs[i] = qs1[i];
                nvs = nv1;
                wbs = wb1;
            } else {
                request_results(j, qs, ks, nvs, wbs, tab_lds);
                if constexpr (MODE == 2) __syncthreads();
            }
        }
        if constexpr (MODE == 2) rc_decode<ER>(qs, tabj, wbs, nvs, t, ks);
        double2 *__restrict__ sp = reinterpret_cast<double2 *>(buf) + t;
#pragma unroll
        for (int mm = 0; mm < G::RSLOTS; ++mm) {
            const int m = 4 + j * G::RSLOTS + mm;
            sp[256 * mm] = make_double2(zr[brev<PT>(m)], zi[brev<PT>(m)]);
        }
        if constexpr (MODE == 1) {
            // plain lists: 24 address words held across the radix-32 pass do not fit beside its 128 data
            // registers (20 spilled VGPRs in the flat form); they are requested here instead
            const ListArgs ls = list_args(2 + j);
#pragma unroll
            for (int u = 0; u < ER; ++u) ks[u] = ld_list(ls.k + slot_of<ER>(t, u));
        }
        __syncthreads();
        if constexpr (EARLY)
            if (j + 1 < G::RR) request_results(j + 1, qs1, ks, nv1, wb1, tab_lds + rmax);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int u = 0; u < ER; ++u) {
                const int e = t + u * kT;
                const int64_t o = (int64_t)j * G::RLEN + e;
                if (o < wd.len) out[wd.start + o] = buf[e];
            }
        } else {
            double rv[ER];
#pragma unroll
            for (int u = 0; u < ER; ++u) rv[u] = buf[q_word(qs, u) & 0x7FFFu];
#pragma unroll
            for (int u = 0; u < ER; ++u) {
                if constexpr (BUF) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, rv[u]), o_rs, ks[u] * 8u, 0, 0);
                } else {
                    if (ks[u] != kInvalidSample) st_result(out + ks[u], rv[u]);
                }
            }
        }
    }
    }
    OS_STAMP(5);
}

// ---- plan-time kernels ----------------------------------------------------------------------------
// entries of the lists of windows [w0, w0 + nw), PER per window, in the order they are stored:
//   list 0 / 1: window positions [0, N) / [N, 2N)        -> value = position within the half
//   list 2 (3): results [0, RLEN) ([RLEN, 2 RLEN))       -> value = position within the round
// key = address in the tile order (0xFFFFFFFF: no sample); a segmented sort then orders every list
// by address.
template <int PT>
__global__ __launch_bounds__(256) void k_real_keys(const WinDesc *__restrict__ wins, int64_t w0,
                                                    int64_t nw, const uint32_t *__restrict__ idx,
                                                    uint32_t *__restrict__ keys,
                                                    uint16_t *__restrict__ vals)
{
    using G = Geo<PT>;
    const int64_t total = nw * G::PER;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const int64_t p = g / G::PER;
        const int e = (int)(g - p * G::PER);
        const WinDesc wd = wins[w0 + p];
        uint32_t k = kInvalidSample;
        int val;
        if (e < 2 * G::N) {
            val = e & (G::N - 1);
            const int64_t ts = wd.start - kHalo + e;
            if (ts >= wd.lo && ts < wd.hi) k = idx[ts];
        } else {
            const int o = e - 2 * G::N;
            val = o % G::RLEN;
            if (o < wd.len) k = idx[wd.start + o];
        }
        keys[g] = k;
        vals[g] = (uint16_t)(val | (k == kInvalidSample ? 0x8000 : 0));   // bit 15: no sample
    }
}

template <int PT>
struct RealListOffset {
    int end;
    __host__ __device__ int operator()(int s) const
    {
        using G = Geo<PT>;
        const int l = s % G::NLIST + end;
        return (s / G::NLIST) * G::PER + (l == G::NLIST ? G::PER : G::list_off(l));
    }
};

// run structure of one sorted list per workgroup: a run starts where the address is not the
// previous address + 1.  One pass over the list in pieces of 256 consecutive entries (coalesced):
// the 16-bit words get their run-start bit, the run table delta[r] = address - slot and the header
// are written.  A list has at most one run per pixel tile (a window's samples in a tile are
// consecutive addresses, and two adjacent tiles' runs can only merge), so the table stride is
// known from the tile count and no counting pass is needed; *max_runs receives the largest count.
template <int PT>
__global__ __launch_bounds__(256) void k_real_rc(int64_t nlists, const uint32_t *__restrict__ lk,
                                                  uint16_t *__restrict__ lq, ListHdr *__restrict__ hdrs,
                                                  uint32_t *__restrict__ tabs, int rmax,
                                                  uint32_t *__restrict__ max_runs)
{
    using G = Geo<PT>;
    __shared__ int wsum[4], vsum[4];
    const int64_t lid = blockIdx.x;
    if (lid >= nlists) return;
    const int l = (int)(lid % G::NLIST);
    const int64_t win = lid / G::NLIST;
    const int64_t e0 = win * G::PER + G::list_off(l);
    const int len = G::list_len(l);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int runs = 0, nvalid = 0;                          // in front of the current piece (uniform)
    for (int s0 = 0; s0 < len; s0 += 256) {
        const int s = s0 + t;
        const uint32_t k = lk[e0 + s];
        const uint32_t prev = s > 0 ? lk[e0 + s - 1] : kInvalidSample;
        const bool valid = k != kInvalidSample;
        const bool flag = valid && (s == 0 || k != prev + 1u);
        const uint64_t fm = __ballot(flag), vm = __ballot(valid);
        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
        if (lane == 0) { wsum[wave] = __popcll(fm); vsum[wave] = __popcll(vm); }
        __syncthreads();
        int wbase = 0, tot = 0, vtot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wbase += wsum[w];
            tot += wsum[w];
            vtot += vsum[w];
        }
        __syncthreads();
        const int r = runs + wbase + below + (flag ? 1 : 0) - 1;     // run of this entry (-1: none yet)
        if (flag) {
            if (r < rmax) tabs[lid * rmax + r] = k - (uint32_t)s;
        }
        const uint16_t q = lq[e0 + s];
        lq[e0 + s] = (uint16_t)((q & 0x7FFFu) | (flag ? 0x8000u : 0u));
        // run index in front of each wave's first slot (waves own len / 4 consecutive slots)
        if (s % (len / 4) == 0) hdrs[lid].wbase[s / (len / 4)] = flag ? r - 1 : r;
        runs += tot;
        nvalid += vtot;
    }
    if (t == 0) {
        hdrs[lid].nvalid = (uint32_t)nvalid;
        hdrs[lid].nruns = (uint32_t)runs;
        atomicMax(max_runs, (uint32_t)runs);
    }
}

// the 16-bit words of every list from slot order (what the segmented sort and k_real_rc leave) to
// the stored order q_index: one workgroup per list, through LDS
template <int PT>
__global__ __launch_bounds__(256) void k_real_qperm(int64_t nlists, uint16_t *__restrict__ lq)
{
    using G = Geo<PT>;
    __shared__ uint16_t stage[G::N];
    const int64_t lid = blockIdx.x;
    if (lid >= nlists) return;
    const int l = (int)(lid % G::NLIST);
    const int64_t e0 = (lid / G::NLIST) * G::PER + G::list_off(l);
    const int len = G::list_len(l), E = len / 256;
    for (int s = threadIdx.x; s < len; s += 256) stage[q_index(s, E)] = lq[e0 + s];
    __syncthreads();
    uint32_t *dst = reinterpret_cast<uint32_t *>(lq + e0);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(stage);
    for (int i = threadIdx.x; i < len / 2; i += 256) dst[i] = src[i];
}

// ---- the lists without a sort ------------------------------------------------------------------
// The tile order is a STABLE partition of the time samples by pixel tile: the samples of one tile
// that fall into any contiguous time range have consecutive addresses, in time order.  A list
// sorted by address is therefore: tiles ascending, inside a tile address - (lowest address of the
// tile in this list).  One workgroup per list: count and lowest address per tile with LDS atomics
// (integer add / min: the result does not depend on their order), a scan over the tiles, then
//   slot(entry) = base[tile] + address - lowest[tile]
// and the entries without a sample behind the valid ones in list order.  The run table falls out
// of the same numbers: every tile with samples starts a run, delta = lowest[tile] - base[tile].
// (k_real_rc merges the runs of two adjacent tiles when their addresses happen to be contiguous;
// this kernel does not: at most one run per tile either way.)
// RC: bit 15 of a word = run start, headers and run tables written; otherwise bit 15 = no sample
// and the addresses go to lk (plain lists).
template <int PT, bool RC>
__global__ __launch_bounds__(256) void k_real_lists(const WinDesc *__restrict__ wins, int64_t nlists,
                                                     const uint32_t *__restrict__ idx,
                                                     const int64_t *__restrict__ tile_off, int ntiles,
                                                     uint16_t *__restrict__ lq, uint32_t *__restrict__ lk,
                                                     ListHdr *__restrict__ hdrs, uint32_t *__restrict__ tabs,
                                                     int rmax, uint32_t *__restrict__ max_runs, int64_t span_samples)
{
    using G = Geo<PT>;
    constexpr int E = G::N / 256;                        // rows of 64 entries a wave handles at most
    extern __shared__ uint32_t sm_l[];
    uint32_t *toff = sm_l;                               // [ntiles + 1] first address of every tile
    uint32_t *cnt = toff + ntiles + 1;                   // [ntiles] entries, then: base slot
    uint32_t *mn = cnt + ntiles;                         // [ntiles] lowest address
    uint32_t *ridx = mn + ntiles;                        // [ntiles] run index of the tile
    uint32_t *misc = ridx + ntiles;                      // [4] waves' scan sums, [4] entries without sample, [4] wbase counts
    uint16_t *stage = reinterpret_cast<uint16_t *>(misc + 12);   // [len] the list's 16-bit words
    const int64_t lid = blockIdx.x;
    if (lid >= nlists) return;
    const int l = (int)(lid % G::NLIST);
    const int64_t win = lid / G::NLIST;
    const int64_t e0 = win * G::PER + G::list_off(l);
    const int len = G::list_len(l), quarter = len / 4, rows = quarter / 64;
    const WinDesc wd = wins[win];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // (a plan ordered [span][tile][time]: the window lies in ONE span -- the caller keeps the windows
    // that straddle two out of this builder -- and the tiles' first addresses are that span's)
    {
        const int64_t ts0 = wd.start - kHalo > wd.lo ? wd.start - kHalo : wd.lo;
        tile_off += span_samples ? (ts0 / span_samples) * ntiles : 0;
    }
    for (int b = t; b <= ntiles; b += 256) toff[b] = (uint32_t)tile_off[b];
    for (int b = t; b < ntiles; b += 256) {
        cnt[b] = 0;
        mn[b] = 0xFFFFFFFFu;
    }
    if (t < 12) misc[t] = 0;
    __syncthreads();
    // ---- pass 1: addresses, tiles, counts ----
    uint32_t a[E];
    uint16_t tl[E];
    int ninv = 0;                                        // entries without a sample of this wave so far
    uint32_t inv_rank[E / 2];                            // (two 16-bit ranks a word)
#pragma unroll
    for (int i = 0; i < E; ++i) {
        a[i] = kInvalidSample;
        tl[i] = 0;
        if (i < rows) {
            const int e = wave * quarter + 64 * i + lane;
            if (l < 2) {
                const int64_t ts = wd.start - kHalo + (int64_t)l * G::N + e;
                if (ts >= wd.lo && ts < wd.hi) a[i] = idx[ts];
            } else {
                const int64_t o = (int64_t)(l - 2) * G::RLEN + e;
                if (o < wd.len) a[i] = idx[wd.start + o];
            }
            const bool valid = a[i] != kInvalidSample;
            if (valid) {
                int lo = 0, hi = ntiles;                 // largest b with toff[b] <= a
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (toff[mid] <= a[i]) lo = mid; else hi = mid;
                }
                tl[i] = (uint16_t)lo;
                atomicAdd(&cnt[lo], 1u);
                atomicMin(&mn[lo], a[i]);
            }
            const uint64_t im = __ballot(!valid);
            const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
            const uint32_t r = (uint32_t)(ninv + below);
            if (i & 1) inv_rank[i / 2] |= r << 16; else inv_rank[i / 2] = r;
            ninv += __popcll(im);
        }
    }
    if (lane == 0) misc[4 + wave] = (uint32_t)ninv;
    __syncthreads();
    // ---- scan over the tiles: base slot and run index (packed: runs << 16 | entries) ----
    const int per = (ntiles + 255) / 256;
    uint32_t mine = 0;
    for (int b = t * per; b < (t + 1) * per && b < ntiles; ++b) mine += cnt[b] | (cnt[b] ? 0x10000u : 0u);
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) misc[wave] = inc;
    __syncthreads();
    uint32_t before = inc - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) before += misc[w];
        total += misc[w];
    }
    const int nvalid = (int)(total & 0xFFFFu), nruns = (int)(total >> 16);
    // (len <= 8192 entries and at most 2^15 tiles: both halves of the packed word are exact)
    for (int b = t * per; b < (t + 1) * per && b < ntiles; ++b) {
        const uint32_t c = cnt[b];
        const uint32_t base = before & 0xFFFFu, r = before >> 16;
        cnt[b] = base;
        ridx[b] = r;
        if (c) {
            if (RC) {
                if ((int)r < rmax) tabs[lid * rmax + r] = mn[b] - base;
#pragma unroll
                for (int w = 1; w < 4; ++w)
                    if ((int)base < w * quarter) atomicAdd(&misc[8 + w], 1u);
            }
            before += c | 0x10000u;
        }
    }
    int inv_before = nvalid;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) inv_before += (int)misc[4 + w];
    __syncthreads();
    // ---- pass 2: every entry to its slot ----
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (i < rows) {
            const int e = wave * quarter + 64 * i + lane;
            const bool valid = a[i] != kInvalidSample;
            int slot;
            uint16_t word = (uint16_t)e;
            if (valid) {
                const uint32_t low = mn[tl[i]];
                slot = (int)(cnt[tl[i]] + (a[i] - low));
                if (RC && a[i] == low) word |= 0x8000u;
            } else {
                slot = inv_before + (int)((inv_rank[i / 2] >> (16 * (i & 1))) & 0xFFFFu);
                if (!RC) word |= 0x8000u;
            }
            stage[q_index(slot, len / 256)] = word;
            if (!RC) lk[e0 + slot] = a[i];
        }
    }
    if (RC && t == 0) {
        ListHdr h;
        h.nvalid = (uint32_t)nvalid;
        h.nruns = (uint32_t)nruns;
        h.wbase[0] = -1;
        for (int w = 1; w < 4; ++w) h.wbase[w] = (int32_t)misc[8 + w] - 1;
        hdrs[lid] = h;
        atomicMax(max_runs, (uint32_t)nruns);
    }
    __syncthreads();
    uint32_t *dst = reinterpret_cast<uint32_t *>(lq + e0);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(stage);
    for (int i = t; i < len / 2; i += 256) dst[i] = src[i];
}

// ---- inverse lists (MODE 3 of k_os_real) ---------------------------------------------------------
// One workgroup per list (l = 0: the 2N window positions, l = 1: the HOP result positions).  Same
// arithmetic as k_real_lists over the whole (result) window: per-tile count and lowest address, scan,
// slot = base[tile] + address - lowest[tile].  Written: the slot of every POSITION (u16, natural
// order, 0xFFFF = no sample), the run table, one run-start bit per slot in the transposed layout the
// kernel reads (bit u of word [round][thread]), the run index in front of every wave's slots of every
// round.
template <int PT>
__global__ __launch_bounds__(256) void k_real_ilists(const WinDesc *__restrict__ wins, int64_t nlists,
                                                      const uint32_t *__restrict__ idx,
                                                      const int64_t *__restrict__ tile_off, int ntiles,
                                                      uint16_t *__restrict__ plist, uint32_t *__restrict__ flags,
                                                      IListHdr *__restrict__ hdrs, uint32_t *__restrict__ tabs,
                                                      int rmax, uint32_t *__restrict__ max_runs, int threads,
                                                      int64_t span_samples)
{
    // threads: workgroup size of the kernel that will read the lists (256: k_os_real, 512: k_os_wide);
    // it fixes the slot order of a round (slot = 64 (E wave + u) + lane, E = round / threads)
    using G = Geo<PT>;
    constexpr int EMAX = 2 * G::N / 256;
    extern __shared__ uint32_t sm_i[];
    uint32_t *toff = sm_i;                               // [ntiles + 1]
    uint32_t *cnt = toff + ntiles + 1;                   // [ntiles] entries, then: base slot
    uint32_t *mn = cnt + ntiles;                         // [ntiles] lowest address
    uint32_t *misc = mn + ntiles;                        // [4] scan sums, [16] wbase counts
    uint32_t *fl = misc + 20;                            // [2 threads] run-start bits
    const int64_t lid = blockIdx.x;
    if (lid >= nlists) return;
    const int l = (int)(lid & 1);
    const int64_t win = lid >> 1;
    const int64_t e0 = win * G::PER + (l ? 2 * G::N : 0);
    const int len = l ? G::HOP : 2 * G::N;               // positions
    const int RL = l ? G::RLEN : G::N;                   // slots a round
    const int rounds = l ? G::RR : 2, rows = len / 256, E = RL / threads, nw = threads / 64;
    const WinDesc wd = wins[win];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {
        const int64_t ts0 = wd.start - kHalo > wd.lo ? wd.start - kHalo : wd.lo;   // (see k_real_lists)
        tile_off += span_samples ? (ts0 / span_samples) * ntiles : 0;
    }
    for (int b = t; b <= ntiles; b += 256) toff[b] = (uint32_t)tile_off[b];
    for (int b = t; b < ntiles; b += 256) {
        cnt[b] = 0;
        mn[b] = 0xFFFFFFFFu;
    }
    if (t < 20) misc[t] = 0;
    for (int i = t; i < 2 * threads; i += 256) fl[i] = 0;
    __syncthreads();
    uint32_t a[EMAX];
    uint16_t tl[EMAX];
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        a[i] = kInvalidSample;
        tl[i] = 0;
        if (i < rows) {
            const int e = 256 * i + t;
            if (l == 0) {
                const int64_t ts = wd.start - kHalo + e;
                if (ts >= wd.lo && ts < wd.hi) a[i] = idx[ts];
            } else {
                if (e < wd.len) a[i] = idx[wd.start + e];
            }
            if (a[i] != kInvalidSample) {
                int lo = 0, hi = ntiles;                 // largest b with toff[b] <= a
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (toff[mid] <= a[i]) lo = mid; else hi = mid;
                }
                tl[i] = (uint16_t)lo;
                atomicAdd(&cnt[lo], 1u);
                atomicMin(&mn[lo], a[i]);
            }
        }
    }
    __syncthreads();
    // scan over the tiles: base slot and run index (packed: runs << 16 | entries)
    const int per = (ntiles + 255) / 256;
    uint32_t mine = 0;
    for (int b = t * per; b < (t + 1) * per && b < ntiles; ++b) mine += cnt[b] | (cnt[b] ? 0x10000u : 0u);
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) misc[wave] = inc;
    __syncthreads();
    uint32_t before = inc - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) before += misc[w];
        total += misc[w];
    }
    const int nvalid = (int)(total & 0xFFFFu), nruns = (int)(total >> 16);
    for (int b = t * per; b < (t + 1) * per && b < ntiles; ++b) {
        const uint32_t c = cnt[b];
        const uint32_t base = before & 0xFFFFu, r = before >> 16;
        cnt[b] = base;
        if (c) {
            if ((int)r < rmax) tabs[lid * rmax + r] = mn[b] - base;
            // the run's first slot: round, then (wave, row, lane) of the kernel's slot order
            const int j = (int)base / RL, sr = (int)base % RL;
            const int wv = sr / (64 * E), rem = sr % (64 * E);
            atomicOr(&fl[threads * j + 64 * wv + (rem & 63)], 1u << (rem >> 6));
            for (int jj = 0; jj < rounds; ++jj)
                for (int w = 0; w < nw; ++w)
                    if ((int)base < jj * RL + w * (RL / nw)) atomicAdd(&misc[4 + nw * jj + w], 1u);
            before += c | 0x10000u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EMAX; ++i)
        if (i < rows) {
            const int e = 256 * i + t;
            uint16_t slot = 0xFFFFu;
            if (a[i] != kInvalidSample) slot = (uint16_t)(cnt[tl[i]] + (a[i] - mn[tl[i]]));
            plist[e0 + e] = slot;
        }
    for (int i = t; i < 2 * threads; i += 256) flags[lid * 2 * threads + i] = fl[i];
    if (t == 0) {
        IListHdr h;
        h.nvalid = (uint32_t)nvalid;
        h.nruns = (uint32_t)nruns;
        for (int k = 0; k < 16; ++k) h.wbase[k] = (int32_t)misc[4 + k] - 1;
        hdrs[lid] = h;
        atomicMax(max_runs, (uint32_t)nruns);
    }
}

// W[t] = exp(-2 pi i t / N)
__global__ void k_real_twiddles(int N, double2 *__restrict__ W)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N) W[t] = make_double2(cospi(2.0 * t / N), -sinpi(2.0 * t / N));
}

// ct[m] = cos(pi m / N), m = 0..N
__global__ void k_real_cos_table(int N, double *__restrict__ ct)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m <= N) ct[m] = cospi((double)m / (double)N);
}

// H[b][k] = a0 + 2 sum_{j>=1} a_j cos(2 pi j k / (2N)),  k = 0..N  (real, even: symmetric band).
// The cosines come from the table ct staged in LDS (cos(pi m / N) for m = j k mod 2N, folded to
// m <= N by the cosine's symmetry) instead of one cospi per term: the sum over j is the same sum in
// the same order (j = lambda - 1 down to 1).  One workgroup: kSpecK bins of one block.
constexpr int kSpecK = 1024;                             // bins per workgroup (4 per thread)
__global__ __launch_bounds__(256) void k_real_spectrum(int nb, int64_t lambda, int N,
                                                        const double *__restrict__ bands,
                                                        const double *__restrict__ ct_g,
                                                        double *__restrict__ Hs)
{
    extern __shared__ double ct[];                       // N + 1 cosines
    constexpr int PER = kSpecK / 256;
    const int chunks = (N + 1 + kSpecK - 1) / kSpecK;
    const int b = blockIdx.x / chunks, c = blockIdx.x % chunks;
    for (int m = threadIdx.x; m <= N; m += 256) ct[m] = ct_g[m];
    __syncthreads();
    const double *band = bands + (int64_t)b * lambda;
    const int mask = 2 * N - 1;                          // N is a power of two
    int k[PER], m[PER];
    double acc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        k[u] = c * kSpecK + u * 256 + (int)threadIdx.x;
        m[u] = (int)(((lambda - 1) * (int64_t)k[u]) & mask);
        acc[u] = 0.0;
    }
    for (int64_t j = lambda - 1; j >= 1; --j) {
        const double a = band[j];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int f = m[u] <= N ? m[u] : 2 * N - m[u];
            acc[u] += a * ct[f];
            m[u] = (m[u] - k[u]) & mask;
        }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (k[u] <= N) Hs[(int64_t)b * (N + 1) + k[u]] = band[0] + 2.0 * acc[u];
}

// AB[b][a] = (alpha_k, beta_k), a = d1 256 + d2 16 + d3 the P3 position holding k = d1 + PT d2 + 16 PT d3
template <int PT>
__global__ __launch_bounds__(256) void k_real_alpha_beta(int nb, const double *__restrict__ Hs,
                                                          double2 *__restrict__ AB)
{
    constexpr int N = Geo<PT>::N;
    const int64_t total = (int64_t)nb * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t b = e / N;
        const int a = (int)(e - b * N);
        const int d1 = a / 256, d2 = (a / 16) % 16, d3 = a % 16;
        const int k = d1 + PT * d2 + 16 * PT * d3;
        const double *h = Hs + b * (N + 1);
        const double hk = h[k], hp = h[N - k];          // H[k + N] = H[N - k]
        const double S = 0.5 * (hk + hp), Dd = 0.5 * (hk - hp);
        const double th = (double)k / (double)N;
        // stored [m][t] (a = PT t + m): the 256 threads read bin m of all of them in one contiguous
        // 4 KB piece -- [t][m] order made every lane of a load touch another cache line
        const int t = a / PT, m = a % PT;
        AB[b * N + (int64_t)m * kT + t] = make_double2((S - Dd * sinpi(th)) / (double)N,
                                                        (Dd * cospi(th)) / (double)N);
    }
}


}  // namespace

namespace cm2 {

constexpr int kPT = 32;                  // complex points per thread of the kernel that ships
using G32 = Geo<kPT>;
constexpr size_t kListCache = 3;         // tile plans whose lists one operator keeps (most recent first)

// Address lists of one set of windows.  A plan ordered [span][tile][time] (cm2_tiles.h) has two sets:
// the windows that lie inside ONE span -- all but one or two per span -- whose lists are written
// directly from that span's segment table (at most one address run per tile), and the windows that
// straddle a span boundary (two runs per tile: their run tables would double the LDS of every
// workgroup of the launch), which get plain lists from the segmented sort and a small launch of
// their own.  A plan with one span has the first set only.
struct OsSet {
    WinDesc *d_wins = nullptr;           // the set's windows (owned unless `borrowed`)
    bool borrowed = false;
    int64_t nwin = 0;
    int mode = 0;                        // 1 plain, 2 run-coded (cut by time), 3 inverse (cut by address)
    uint32_t *d_lst_k = nullptr;         // mode 1: addresses
    uint16_t *d_lst_q = nullptr;         // modes 1, 2: position of every slot; mode 3: slot of every position
    ListHdr *d_hdrs = nullptr;           // mode 2
    uint32_t *d_tabs = nullptr;          // modes 2, 3: run tables
    IListHdr *d_ihdrs = nullptr;         // mode 3
    uint32_t *d_iflags = nullptr;        // mode 3: run-start bits, [list][round][thread]
    int rmax = 0;
    double bytes_per_window = 0.0;
    ~OsSet()
    {
        void *ptrs[] = {borrowed ? nullptr : d_wins, d_lst_k, d_lst_q, d_hdrs, d_tabs, d_ihdrs, d_iflags};
        for (void *q : ptrs)
            if (q) (void)cm2::dev_free(q);
    }
};

// The lists of one (noise operator, tile plan) pair.  Owned by a shared_ptr: an application holds a
// reference while it launches, so a concurrent eviction cannot free lists that a launch is about to
// use (dev_free waits for the device before a block can be handed out again).
struct OsLists {
    uint64_t plan_id = 0;
    OsSet a, b;                          // windows inside one span (or all of them), straddling windows
};

struct FusedOS {
    int64_t nwin = 0;
    int64_t nb = 0;
    std::vector<WinDesc> h_wins;
    WinDesc *d_wins = nullptr;
    double2 *d_AB = nullptr;
    double2 *d_W = nullptr;
    // switches, read ONCE when the operator is created (never on the application path):
    int want_lists = 0;                  // CM2_OS_LISTS = auto (0) | plain (1) | rc (2) | inv (3)
    bool build_sort = false;             // CM2_OS_LIST_BUILD = sort: lists from a segmented sort
    bool flat = false;                   // CM2_OS_FLAT: flat addressing although the buffers are < 4 GB
    int64_t sort_chunk_windows = 0;      // CM2_OS_LIST_CHUNK_PAIRS (test hook: sort in several chunks)
    std::mutex mu;                       // guards `cache`; list builds run under it
    std::vector<std::shared_ptr<OsLists>> cache;
};

void fused_os_destroy(FusedOS *f)
{
    if (!f) return;
    f->cache.clear();
    void *ptrs[] = {f->d_wins, f->d_AB, f->d_W};
    for (void *q : ptrs)
        if (q) (void)cm2::dev_free(q);
    delete f;
}

bool fused_os_supported(int64_t lambda) { return lambda >= 1 && lambda - 1 <= kHalo; }

int64_t fused_os_length(const FusedOS *f) { return f ? G32::N : 0; }

int fused_os_create(FusedOS **out, const double *d_bands, int64_t lambda, const std::vector<int64_t> &off,
                    hipStream_t stream)
{
    using G = G32;
    CM2_CHECK(out != nullptr, "fused_os_create: out is NULL");
    *out = nullptr;
    CM2_CHECK(fused_os_supported(lambda), "fused overlap-save supports lambda <= 2049, got %lld",
              (long long)lambda);
    FusedOS *f = new FusedOS();
    struct Guard { FusedOS *f; ~Guard() { if (f) fused_os_destroy(f); } } guard{f};
    if (const char *e = getenv("CM2_OS_LISTS")) {
        if (!strcmp(e, "plain")) f->want_lists = 1;
        else if (!strcmp(e, "rc")) f->want_lists = 2;
        else if (!strcmp(e, "inv")) f->want_lists = 3;
    }
    if (const char *e = getenv("CM2_OS_LIST_BUILD")) f->build_sort = strcmp(e, "sort") == 0;
    f->flat = getenv("CM2_OS_FLAT") != nullptr;
    if (const char *e = getenv("CM2_OS_LIST_CHUNK_PAIRS")) f->sort_chunk_windows = atoll(e);
    const int64_t nb = (int64_t)off.size() - 1;
    f->nb = nb;
    std::vector<WinDesc> &wins = f->h_wins;
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t s0 = off[b]; s0 < off[b + 1]; s0 += G::HOP) {
            WinDesc wd;
            wd.start = s0;
            wd.len = off[b + 1] - s0 < G::HOP ? off[b + 1] - s0 : G::HOP;
            wd.lo = off[b];
            wd.hi = off[b + 1];
            wd.blk = (int32_t)b;
            wd.pad = 0;
            wins.push_back(wd);
        }
    f->nwin = (int64_t)wins.size();
    CM2_CHECK(f->nwin * 4 < ((int64_t)1 << 31), "fused overlap-save: too many windows (%lld)", (long long)f->nwin);
    CM2_HIP(cm2::dev_malloc(&f->d_wins, sizeof(WinDesc) * (wins.size() ? wins.size() : 1)));
    if (!wins.empty())
        CM2_HIP(cm2::upload(f->d_wins, wins.data(), sizeof(WinDesc) * wins.size(), nullptr));
    CM2_HIP(cm2::dev_malloc(&f->d_AB, sizeof(double2) * (nb > 0 ? nb : 1) * G::N));
    if (nb > 0) {
        DevTemp<double> Hs;
        CM2_HIP(Hs.alloc(nb * (G::N + 1)));
        DevTemp<double> ct;
        CM2_HIP(ct.alloc(G::N + 1));
        k_real_cos_table<<<(G::N + 256) / 256, 256, 0, stream>>>(G::N, ct);
        CM2_LAUNCH_OK();
        const size_t ct_lds = sizeof(double) * (G::N + 1);
        static size_t ct_granted[64] = {0};
        CM2_HIP(ensure_dynamic_lds((const void *)k_real_spectrum, ct_lds, ct_granted));
        const int chunks = (G::N + 1 + kSpecK - 1) / kSpecK;
        k_real_spectrum<<<(int)nb * chunks, 256, ct_lds, stream>>>((int)nb, lambda, G::N, d_bands, ct, Hs);
        CM2_LAUNCH_OK();
        k_real_alpha_beta<kPT><<<grid_for(nb * G::N), kBlock, 0, stream>>>((int)nb, Hs, f->d_AB);
        CM2_LAUNCH_OK();
        CM2_HIP(hipStreamSynchronize(stream));
    }
    CM2_HIP(cm2::dev_malloc(&f->d_W, sizeof(double2) * G::N));
    k_real_twiddles<<<(G::N + 255) / 256, 256, 0, stream>>>(G::N, f->d_W);
    CM2_LAUNCH_OK();
    CM2_HIP(hipStreamSynchronize(stream));
    guard.f = nullptr;
    *out = f;
    return 0;
}

template <int MODE, bool BUF>
static int os_launch_t(const FusedOS *f, const WinDesc *d_wins, int64_t nwin, const OsSet *ls, const double *d_v,
                       double *d_out, uint32_t nbytes, hipStream_t stream)
{
    using G = G32;
    const int rmax = ls ? ls->rmax : 0;
    // (exchange plane, the two run tables of a list pair, 2 KB for the lower slots of the eight self-paired threads
    //  of the half-plane pairing)
    const size_t lds = sizeof(double) * (size_t)G::LDSD + (MODE >= 2 ? sizeof(uint32_t) * 2 * (size_t)rmax : 0) + 2048;
    static size_t granted[64] = {0};
    CM2_HIP(ensure_dynamic_lds((const void *)k_os_real<kPT, MODE, BUF>, lds, granted));
    if (nwin == 0) return 0;
    const int grid = (int)(((nwin + 7) / 8) * 8);          // whole rounds over the 8 XCDs
    k_os_real<kPT, MODE, BUF><<<grid, kT, lds, stream>>>(
        d_wins, (int)nwin, f->d_W, f->d_W, f->d_AB, ls ? ls->d_lst_k : nullptr, ls ? ls->d_lst_q : nullptr,
        ls ? ls->d_hdrs : nullptr, ls ? ls->d_tabs : nullptr, rmax, d_v, d_out, nbytes, ls ? ls->d_ihdrs : nullptr,
        ls ? ls->d_iflags : nullptr OS_STAMP_ARG);
    CM2_LAUNCH_OK();
    return 0;
}

// The kernel instance for a list format, a run-table size and a buffer size.  Buffers below 4 GB are
// addressed through buffer descriptors (BUF), larger ones (or CM2_OS_FLAT) with flat addresses.
static int os_launch(const FusedOS *f, const OsSet *ls, int64_t nvalid, const double *d_v, double *d_out,
                     hipStream_t stream)
{
    if (ls->nwin == 0) return 0;
    const bool buf = nvalid > 0 && nvalid * 8 < (int64_t)0xFFFFFFF0u && !f->flat;
    const uint32_t nbytes = buf ? (uint32_t)(nvalid * 8) : 0u;
    // (the two run tables of a list pair live in LDS beside the 66 KB exchange buffer: 8 rmax bytes)
    if (ls->mode == 1)
        return buf ? os_launch_t<1, true>(f, ls->d_wins, ls->nwin, ls, d_v, d_out, nbytes, stream)
                   : os_launch_t<1, false>(f, ls->d_wins, ls->nwin, ls, d_v, d_out, 0, stream);
    if (ls->mode == 3 && ls->rmax <= 8 * kT)
        return buf ? os_launch_t<3, true>(f, ls->d_wins, ls->nwin, ls, d_v, d_out, nbytes, stream)
                   : os_launch_t<3, false>(f, ls->d_wins, ls->nwin, ls, d_v, d_out, 0, stream);
    if (ls->mode == 2 && ls->rmax <= 8 * kT)
        return buf ? os_launch_t<2, true>(f, ls->d_wins, ls->nwin, ls, d_v, d_out, nbytes, stream)
                   : os_launch_t<2, false>(f, ls->d_wins, ls->nwin, ls, d_v, d_out, 0, stream);
    set_error("fused overlap-save: run table of %d words per list does not fit the kernel", ls->rmax);
    return 2;
}

int fused_os_apply(const FusedOS *f, const double *d_v, double *d_out, hipStream_t stream)
{
    return os_launch_t<0, false>(f, f->d_wins, f->nwin, nullptr, d_v, d_out, 0, stream);
}

// run-table words per list: one run per pixel tile at most (k_real_rc / k_real_lists)
static int os_rmax(int64_t ntiles)
{
    using G = G32;
    int64_t bound = ntiles > 0 ? ntiles : G::N;
    if (bound > G::N) bound = G::N;
    const int rmax = (int)((bound + 63) / 64 * 64);
    return rmax < 64 ? 64 : rmax;
}

// The lists straight from the tile plan's offsets (k_real_lists): no keys, no sort, no temporaries.
static int os_build_lists_direct(const FusedOS *f, OsSet *ls, const OsPlanView &pv, bool want_rc, hipStream_t stream)
{
    using G = G32;
    const int64_t total = ls->nwin * G::PER;
    const int64_t nlists = ls->nwin * G::NLIST;
    const int64_t span = pv.nspans > 1 ? pv.span_samples : 0;
    CM2_HIP(cm2::dev_malloc(&ls->d_lst_q, sizeof(uint16_t) * total));
    const int rmax = os_rmax(pv.ntiles);
    // run-coded lists up to 8 table words per thread (2048 runs a list), plain lists beyond that
    const bool rc = want_rc && rmax <= 8 * kT;
    const size_t lds = sizeof(uint32_t) * (size_t)(4 * pv.ntiles + 1 + 12) + sizeof(uint16_t) * (size_t)G::N;
    static size_t granted[64] = {0};
    DevTemp<uint32_t> d_max;
    CM2_HIP(d_max.alloc(1));
    CM2_HIP(hipMemsetAsync(d_max.p, 0, sizeof(uint32_t), stream));
    if (rc) {
        CM2_HIP(cm2::dev_malloc(&ls->d_hdrs, sizeof(ListHdr) * nlists));
        CM2_HIP(cm2::dev_malloc(&ls->d_tabs, sizeof(uint32_t) * nlists * rmax));
        CM2_HIP(ensure_dynamic_lds((const void *)k_real_lists<kPT, true>, lds, granted));
        k_real_lists<kPT, true><<<(unsigned)nlists, 256, lds, stream>>>(ls->d_wins, nlists, pv.d_idx, pv.d_tile_off,
                                                                       (int)pv.ntiles, ls->d_lst_q, nullptr, ls->d_hdrs,
                                                                       ls->d_tabs, rmax, d_max, span);
    } else {
        CM2_HIP(cm2::dev_malloc(&ls->d_lst_k, sizeof(uint32_t) * total));
        CM2_HIP(ensure_dynamic_lds((const void *)k_real_lists<kPT, false>, lds, granted));
        k_real_lists<kPT, false><<<(unsigned)nlists, 256, lds, stream>>>(ls->d_wins, nlists, pv.d_idx, pv.d_tile_off,
                                                                        (int)pv.ntiles, ls->d_lst_q, ls->d_lst_k, nullptr,
                                                                        nullptr, rmax, d_max, span);
    }
    CM2_LAUNCH_OK();
    uint32_t h_max = 0;
    CM2_HIP(cm2::download(&h_max, d_max.p, sizeof(uint32_t), stream));
    CM2_HIP(hipStreamSynchronize(stream));
    if (rc) {
        CM2_CHECK((int)h_max <= rmax, "fused overlap-save: a list has %u address runs, more than the %d pixel "
                  "tiles allow", h_max, rmax);
        ls->rmax = rmax;
        ls->mode = 2;
        ls->bytes_per_window = 2.0 * G::PER + G::NLIST * (sizeof(ListHdr) + 4.0 * h_max);
    } else {
        ls->mode = 1;
        ls->bytes_per_window = 6.0 * G::PER;
    }
    return 0;
}

// Inverse lists (k_real_ilists): needs the tile offsets and run tables that fit LDS (<= 2048 runs).
static int os_build_ilists(const FusedOS *f, OsSet *ls, const OsPlanView &pv, hipStream_t stream)
{
    using G = G32;
    const int64_t total = ls->nwin * G::PER;
    const int64_t nlists = ls->nwin * 2;
    const int rmax = os_rmax(pv.ntiles);
    CM2_HIP(cm2::dev_malloc(&ls->d_lst_q, sizeof(uint16_t) * total));
    CM2_HIP(cm2::dev_malloc(&ls->d_ihdrs, sizeof(IListHdr) * nlists));
    CM2_HIP(cm2::dev_malloc(&ls->d_iflags, sizeof(uint32_t) * nlists * 2 * kT));
    CM2_HIP(cm2::dev_malloc(&ls->d_tabs, sizeof(uint32_t) * nlists * rmax));
    DevTemp<uint32_t> d_max;
    CM2_HIP(d_max.alloc(1));
    CM2_HIP(hipMemsetAsync(d_max.p, 0, sizeof(uint32_t), stream));
    const size_t lds = sizeof(uint32_t) * (size_t)(3 * pv.ntiles + 1 + 20 + 2 * kT);
    static size_t granted[64] = {0};
    CM2_HIP(ensure_dynamic_lds((const void *)k_real_ilists<kPT>, lds, granted));
    k_real_ilists<kPT><<<(unsigned)nlists, 256, lds, stream>>>(ls->d_wins, nlists, pv.d_idx, pv.d_tile_off, (int)pv.ntiles,
                                                              ls->d_lst_q, ls->d_iflags, ls->d_ihdrs, ls->d_tabs, rmax,
                                                              d_max, kT, pv.nspans > 1 ? pv.span_samples : 0);
    CM2_LAUNCH_OK();
    uint32_t h_max = 0;
    CM2_HIP(cm2::download(&h_max, d_max.p, sizeof(uint32_t), stream));
    CM2_HIP(hipStreamSynchronize(stream));
    CM2_CHECK((int)h_max <= rmax, "fused overlap-save: a list has %u address runs, more than the %d pixel tiles "
              "allow", h_max, rmax);
    ls->rmax = rmax;
    ls->mode = 3;
    ls->bytes_per_window = 2.0 * G::PER + 2 * (sizeof(IListHdr) + 2048.0 + 4.0 * h_max);
    return 0;
}

// The lists from a segmented sort of (address, position) pairs: needs nothing but the index
// (CM2_OS_LIST_BUILD=sort, a plan without tile offsets, more tiles than the direct builders keep in
// LDS, or the windows that straddle two spans of the plan).  `tile_runs`: an upper bound of the
// address runs of a list (0: unknown) -- run-coded lists when want_rc and the bound fits the tables.
static int os_build_lists_sorted(const FusedOS *f, OsSet *ls, const OsPlanView &pv, bool want_rc, int64_t tile_runs,
                                 hipStream_t stream)
{
    using G = G32;
    const int64_t total = ls->nwin * G::PER;
    CM2_HIP(cm2::dev_malloc(&ls->d_lst_k, sizeof(uint32_t) * total));
    CM2_HIP(cm2::dev_malloc(&ls->d_lst_q, sizeof(uint16_t) * total));
    int64_t chunk_w = ((int64_t)1 << 30) / G::PER;             // hipCUB counts items in int
    if (f->sort_chunk_windows > 0 && f->sort_chunk_windows < chunk_w) chunk_w = f->sort_chunk_windows;
    const int64_t cw_max = ls->nwin < chunk_w ? ls->nwin : chunk_w;
    {
        DevTemp<uint32_t> keys_in;
        DevTemp<uint16_t> vals_in;
        DevTemp<char> d_temp;
        CM2_HIP(keys_in.alloc(cw_max * G::PER));
        CM2_HIP(vals_in.alloc(cw_max * G::PER));
        hipcub::CountingInputIterator<int> seg_id(0);
        using OffsetIt = hipcub::TransformInputIterator<int, RealListOffset<kPT>, hipcub::CountingInputIterator<int>>;
        OffsetIt seg_begin(seg_id, RealListOffset<kPT>{0}), seg_end(seg_id, RealListOffset<kPT>{1});
        size_t tb = 0;
        CM2_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(
            nullptr, tb, keys_in.p, ls->d_lst_k, vals_in.p, ls->d_lst_q, (int)(cw_max * G::PER),
            (int)(G::NLIST * cw_max), seg_begin, seg_end, 0, 32, stream));
        CM2_HIP(d_temp.alloc(tb + 16));
        for (int64_t p0 = 0; p0 < ls->nwin; p0 += chunk_w) {
            const int64_t nw = ls->nwin - p0 < chunk_w ? ls->nwin - p0 : chunk_w;
            k_real_keys<kPT><<<grid_for(nw * G::PER), kBlock, 0, stream>>>(ls->d_wins, p0, nw, pv.d_idx, keys_in, vals_in);
            CM2_LAUNCH_OK();
            size_t tbc = tb;
            CM2_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(
                d_temp.p, tbc, keys_in.p, ls->d_lst_k + p0 * G::PER, vals_in.p, ls->d_lst_q + p0 * G::PER,
                (int)(nw * G::PER), (int)(G::NLIST * nw), seg_begin, seg_end, 0, 32, stream));
        }
        CM2_HIP(hipStreamSynchronize(stream));
    }
    ls->mode = 1;
    ls->bytes_per_window = 6.0 * G::PER;
    const int64_t nlists = ls->nwin * G::NLIST;
    if (want_rc) {
        // one run per pixel tile at most (k_real_rc): the run-table stride follows from the tile count;
        // the two window-half tables live in LDS beside the exchange buffer: run-coded lists only
        // up to 8 table words per thread (2048 runs a list), plain lists beyond that
        const int rmax = os_rmax(tile_runs);
        if (rmax <= 8 * kT) {
            DevTemp<uint32_t> d_max;
            CM2_HIP(d_max.alloc(1));
            CM2_HIP(hipMemsetAsync(d_max.p, 0, sizeof(uint32_t), stream));
            CM2_HIP(cm2::dev_malloc(&ls->d_hdrs, sizeof(ListHdr) * nlists));
            CM2_HIP(cm2::dev_malloc(&ls->d_tabs, sizeof(uint32_t) * nlists * rmax));
            k_real_rc<kPT><<<(unsigned)nlists, 256, 0, stream>>>(nlists, ls->d_lst_k, ls->d_lst_q, ls->d_hdrs,
                                                                ls->d_tabs, rmax, d_max);
            CM2_LAUNCH_OK();
            uint32_t h_max = 0;
            CM2_HIP(cm2::download(&h_max, d_max.p, sizeof(uint32_t), stream));
            CM2_HIP(hipStreamSynchronize(stream));
            CM2_CHECK((int)h_max <= rmax, "fused overlap-save: a list has %u address runs, more than the %d pixel "
                      "tiles allow", h_max, rmax);
            (void)cm2::dev_free(ls->d_lst_k);                     // the addresses are now in the run tables
            ls->d_lst_k = nullptr;
            ls->rmax = rmax;
            ls->mode = 2;
            ls->bytes_per_window = 2.0 * G::PER + G::NLIST * (sizeof(ListHdr) + 4.0 * h_max);
        }
    }
    k_real_qperm<kPT><<<(unsigned)nlists, 256, 0, stream>>>(nlists, ls->d_lst_q);
    CM2_LAUNCH_OK();
    CM2_HIP(hipStreamSynchronize(stream));
    return 0;
}

// The lists of `f` for the tile plan `pv`, from the operator's cache or built now (under the
// operator's mutex: two host threads that meet here build once).  List format when the operator was
// created without CM2_OS_LISTS: lists cut by time (mode 2) keep the pick / place side cheap and win
// while a half window's address runs are long (512 tiles at C4: 16 entries); from ~768 tiles up the
// longer runs and whole sectors of the lists cut by address (mode 3) win: C5's 1536 tiles
// 1.24 -> 1.05 ms, the balanced tiling of an uneven hit map (1015 tiles) 0.92 -> 0.87 ms, 512 tiles
// 0.76 -> 0.79 ms (profiles/r03_inverse_lists.md).
static int os_lists_for(FusedOS *f, const OsPlanView &pv, hipStream_t stream, std::shared_ptr<OsLists> *out)
{
    std::lock_guard<std::mutex> lock(f->mu);
    for (size_t i = 0; i < f->cache.size(); ++i)
        if (f->cache[i]->plan_id == pv.plan_id) {
            std::shared_ptr<OsLists> hit = f->cache[i];
            f->cache.erase(f->cache.begin() + (long)i);
            f->cache.insert(f->cache.begin(), hit);
            *out = hit;
            return 0;
        }
    const int want = f->want_lists ? f->want_lists : (pv.ntiles >= 768 ? 3 : 2);
    std::shared_ptr<OsLists> ls = std::make_shared<OsLists>();
    ls->plan_id = pv.plan_id;
    const bool direct = !f->build_sort && pv.d_tile_off && pv.ntiles > 0 && pv.ntiles <= 4096;
    // the two window sets: with several spans (and the direct builders, which look tiles up in ONE
    // span's table) the windows that reach into two spans go to set b
    std::vector<WinDesc> wa, wb;
    if (pv.nspans > 1 && pv.span_samples > 0 && direct) {
        for (const WinDesc &wd : f->h_wins) {
            const int64_t t0 = wd.start - kHalo > wd.lo ? wd.start - kHalo : wd.lo;
            const int64_t t1 = wd.start - kHalo + G32::W < wd.hi ? wd.start - kHalo + G32::W : wd.hi;
            (t0 / pv.span_samples == (t1 - 1) / pv.span_samples ? wa : wb).push_back(wd);
        }
        ls->a.nwin = (int64_t)wa.size();
        ls->b.nwin = (int64_t)wb.size();
        for (int k = 0; k < 2; ++k) {
            OsSet &st = k ? ls->b : ls->a;
            const std::vector<WinDesc> &w = k ? wb : wa;
            CM2_HIP(cm2::dev_malloc(&st.d_wins, sizeof(WinDesc) * (w.size() ? w.size() : 1)));
            if (!w.empty())
                CM2_HIP(cm2::upload(st.d_wins, w.data(), sizeof(WinDesc) * w.size(), stream));
        }
        CM2_HIP(hipStreamSynchronize(stream));               // (wa, wb are locals)
    } else {
        ls->a.d_wins = f->d_wins;
        ls->a.borrowed = true;
        ls->a.nwin = f->nwin;
    }
    CM2_CHECK(f->nwin * G32::NLIST < ((int64_t)1 << 31), "fused overlap-save: too many lists (%lld)",
              (long long)(f->nwin * G32::NLIST));
    if (ls->a.nwin == 0) {
        ls->a.mode = 1;
    } else {
        int rc;
        if (direct) {
            if (want == 3 && os_rmax(pv.ntiles) <= 8 * kT)
                rc = os_build_ilists(f, &ls->a, pv, stream);
            else
                rc = os_build_lists_direct(f, &ls->a, pv, want >= 2, stream);
        } else {
            // (a plan with several spans whose lists are sorted: up to two runs per tile)
            rc = os_build_lists_sorted(f, &ls->a, pv, want >= 2, pv.ntiles * (pv.nspans > 1 ? 2 : 1), stream);
        }
        if (rc) return rc;                                   // (ls frees what it holds)
    }
    if (ls->b.nwin > 0)
        if (int rc = os_build_lists_sorted(f, &ls->b, pv, false, 0, stream)) return rc;
    f->cache.insert(f->cache.begin(), ls);
    while (f->cache.size() > kListCache) f->cache.pop_back();
    *out = ls;
    return 0;
}

int fused_os_prepare_indexed(FusedOS *f, const OsPlanView &pv, hipStream_t stream)
{
    std::shared_ptr<OsLists> ls;
    return os_lists_for(f, pv, stream, &ls);
}

int fused_os_apply_indexed(FusedOS *f, const OsPlanView &pv, const double *d_v, double *d_out, hipStream_t stream)
{
    std::shared_ptr<OsLists> ls;
    if (int rc = os_lists_for(f, pv, stream, &ls)) return rc;
    if (int rc = os_launch(f, &ls->a, pv.nvalid, d_v, d_out, stream)) return rc;
    return os_launch(f, &ls->b, pv.nvalid, d_v, d_out, stream);
}

// kernel[0] = complex points per thread of the window kernel (32), kernel[1] = list format of the
// most recently used plan (1 plain, 2 run-coded, 3 inverse; 0: no lists yet), kernel[2] = windows of
// that plan that straddle two spans (plain lists, a launch of their own); returns the HBM bytes
// per output sample the tile-order kernel is built to move (lists + gathered window + results)
double fused_os_tile_info(const FusedOS *f_, int *kernel)
{
    FusedOS *f = const_cast<FusedOS *>(f_);
    std::shared_ptr<OsLists> ls;
    if (f) {
        std::lock_guard<std::mutex> lock(f->mu);
        if (!f->cache.empty()) ls = f->cache.front();
    }
    if (kernel) {
        kernel[0] = f ? kPT : 0;
        kernel[1] = ls ? ls->a.mode : 0;
        kernel[2] = ls ? (int)ls->b.nwin : 0;
    }
    if (!f) return 0.0;
    const double hop = (double)G32::HOP, win = (double)G32::W;
    const double lists = ls && ls->a.bytes_per_window > 0 ? ls->a.bytes_per_window : 6.0 * (win + hop);
    return (lists + 8.0 * win + 8.0 * hop) / hop;
}

#ifdef CM2_OS_STAMPS
extern "C" int cm2_os_debug_stamps(unsigned long long *d_buf)
{
    g_os_stamps_host = d_buf;
    return 0;
}
#endif

}  // namespace cm2
