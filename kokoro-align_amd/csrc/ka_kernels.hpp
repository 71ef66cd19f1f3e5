// ka_kernels.hpp — gfx950 (CDNA4) device code of the CTC best-path hot path.
//
// What is computed (dense-band form of kokoro_align/align.py:43-109):
//   lab'[p] = p odd ? labels[p/2] : 0;  A_{-1} = {0}, sc_{-1}[0] = 0
//   frame t:  lo = max(0, floor(L*t/T) - B/2), hi = min(lo+B, L)            (align.py:64-65)
//     p in [lo,hi):  c_j = sc_{t-1}[p-j] (+) lp[t, lab'[p]]  (float32 add, then compare)
//                    j even, j>0, lab'[p]==0  ->  c_j = -inf                (align.py:80-81)
//                    j* = first j attaining the max; bp_t[p] = j*           (align.py:83-85)
//   end = highest live position of frame T-1; walk bp back to frame 0       (align.py:99-102)
//
// Layout of the fast path (one 64-lane wavefront owns one lattice):
//   * 1024 slots = 64 lanes x 16 cells; position p lives in slot p mod 1024,
//     lane (p>>4)&63, cell p&15.  The live band is at most 1009 wide, so the 64 blocks
//     [lo>>4, (lo>>4)+63] never alias; as `lo` passes a block its lane is re-labelled
//     for block+64.  No data ever moves when the band slides.
//   * scores live in 16 VGPRs per lane; the three neighbours p-1..p-3 of a lane's first
//     cells come from the previous lane with DPP wave_ror:1.
//   * lane v of a "row" register holds lp[t, v] (V <= 64): one coalesced 256-B load per
//     frame, prefetched 4 frames ahead; blank emission = readfirstlane; the 8 label emissions per
//     lane are gathers lp[t, lab'[p]]: ds_bpermute on the row register in the exact / recompute /
//     workgroup kernels, ds_read_b32 from an LDS copy of the row in the checkpointed forward kernel.
//   * the band [lo,hi) is applied with 16 wave-uniform 64-bit lane masks (one per cell
//     index) held in SGPRs and updated only when lo/hi move.
// Three kernel forms:
//   * checkpointed: forward_ck_kernel keeps scores only and stores the score ring every 32
//     frames; backtrace_rc_kernel recomputes the back-pointers of the 124-cell window below the
//     path, chunk by chunk, walks it and writes all outputs.  Time on gfx950 is proportional
//     to the number of instructions executed: recomputing 3 % of the cells beats comparing and
//     packing all of them.
//   * exact: forward_w16_kernel finds the first move attaining the max with v_cmp_eq -> SGPR
//     lane masks, combines them on the scalar unit and shifts the 2-bit code into a per-lane
//     word with v_addc_co_u32 (16 cells x 2 bit = one dword per lane per frame);
//     backtrace_w16_kernel walks the stored codes, gather_outputs_kernel fills labels/scores.
//   * workgroup (forward_wg4_kernel): four wavefronts per lattice, for latency.
// No MFMA: ~7 flop per cell, nothing to contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ka {

// Pointers read from a descriptor in memory are generic ("flat") to the compiler; everything
// here lives in HBM, so say so: global_load/global_store instead of flat_* (which also tie
// up lgkmcnt).
#define KA_GLOBAL __attribute__((address_space(1)))
typedef KA_GLOBAL const float *gcf32_t;
typedef KA_GLOBAL float *gf32_t;
typedef KA_GLOBAL const int32_t *gci32_t;
typedef KA_GLOBAL int32_t *gi32_t;
typedef KA_GLOBAL const uint32_t *gcu32_t;
typedef KA_GLOBAL uint32_t *gu32_t;
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef KA_GLOBAL const v4i_t *gci4_t;

constexpr int kStatusOk = 0;
constexpr int kStatusEmptyBeam = -1;
constexpr int kStatusBadLabel = -5;
constexpr int kStatusInternal = -8;  // a tile of the tiled form was never fed by the tile below it (a bug, not an input)
constexpr int kStatusNaN = -6;      // a log-prob is NaN (the reference's np.argmax would treat it as a maximum: not reproduced)
constexpr int kFlagZeroLabel = 1;   // meta flags: a transcript label is 0
constexpr int kFlagExact = 2;       // meta flags: the checkpointed path declined this lattice (non-finite log-probs)
constexpr int kFlagDeclined = 4;    // meta flags: declined, and too wide for the exact kernels' ring: no result (KA_ERR_NONFINITE)

constexpr int kSlots = 1024;        // 64 lanes x 16 cells
constexpr int kFastMaxBand = 1009;  // kSlots - 15: widest band the w16 layout can hold
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kCkFrames = 32;       // checkpointed path: frames between stored score rings
constexpr int kRowDepth = 4;        // log-prob rows in flight per wave
static_assert(kRowDepth == 4, "the frame loop is unrolled by the 4 frames of a back-pointer group");

struct Lattice {
    const float *lp;        // [T, ld] log-probs (device)
    const int32_t *labels;  // [S] caller labels (device)
    int32_t *labx;          // [labx_len] 4*label of odd position 2i+1, zero padded (workspace)
    void *bp;               // w16: uint32 [ceil(T/4)][64 blocks][4 frames]; generic: uint8 [T][W]
    float *col;             // generic only: 2 x L float scores followed by 2 x L present bytes
    int32_t *path;          // [T] outputs (device)
    int32_t *lab_out;
    float *sc_out;
    int64_t ld;
    int32_t T, S, L, V;
    int32_t beam, max_move;
    int32_t labx_len, W;    // W = min(beam, L)
    int32_t idx;            // index of this lattice in the caller's batch
    int32_t n_final;        // tiled form: number of tiles alive in the last frame
    // checkpointed forms: the scores after frame 32 (k+1) - 1 are row k of `bp`; position p sits at float index
    // p & ck_mask of its row (one-wavefront form: the 1024-slot ring, mask 1023, pitch 4096; tiled form: a ring that
    // holds every tile the band can touch, or the whole label axis)
    uint32_t ck_mask;
    int32_t ck_pitch;       // bytes
    // chunk-parallel backtrace (ka_parallel_bt.hpp), rows addressed like the checkpoints (position p at p & ck_mask):
    uint8_t *map0;          // [chunk c][ck_pitch / 4]: how far the best path into position p of frame 32c+31 has risen since frame 32c-1
    uint16_t *map1;         // [super-chunk s][ck_pitch / 4]: the same over the 32 chunks of a super-chunk
    int32_t *entry;         // [chunks]: best-path position at the last frame of every chunk, then [super-chunks]: of every super-chunk
    int64_t chunk0;         // index of this lattice's chunk 0 among the chunks of the launch's chunk-parallel lattices
    int32_t par;            // 1: walked back by the chunk-parallel kernels, 0: by one wavefront (backtrace_rc_kernel<.., false>)
    int32_t pad_;
};

// meta[4*idx + {0,1,2,3}] = status, end position, flags (bit0: a transcript label is 0), total score bits
__device__ __forceinline__ int32_t *meta_of(int32_t *meta, int idx) { return meta + 4 * (size_t)idx; }

// ---------------------------------------------------------------------------------------
// label preparation: validate, scale by 4 (ds_bpermute byte address), zero-pad
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_labels_kernel(const Lattice *__restrict__ lats, int32_t *meta)
{
    const Lattice &d = lats[blockIdx.x];
    int bad = 0, zero = 0;
    for (int i = threadIdx.x; i < d.labx_len; i += blockDim.x) {
        int v = 0;
        if (i < d.S) {
            int l = d.labels[i];
            if (l < 0 || l >= d.V) { bad = 1; l = 0; }
            if (l == 0) zero = 1;
            v = l * 4;
        }
        d.labx[i] = v;
    }
    int32_t *m = meta_of(meta, d.idx);
    if (bad) atomicMin(&m[0], kStatusBadLabel);
    if (zero) atomicOr(&m[2], 1);
}

// ---------------------------------------------------------------------------------------
// small helpers (wave64)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float ninf() { return -__builtin_inff(); }

// lane i <- lane i-1, lane 0 <- lane 63 (DPP wave_ror:1)
__device__ __forceinline__ float wave_ror1(float x)
{
    // every lane is written (row_mask = bank_mask = 0xF, wave_ror has no invalid source lanes), so the
    // destination needs no initial value: mov_dpp instead of update_dpp(0, ..) saves a v_mov per call
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x13C, 0xF, 0xF, false));
}
// w = 2*w + mask[lane]
__device__ __forceinline__ uint32_t shl1_in(uint32_t w, uint64_t mask)
{
    uint32_t r;
    uint64_t carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=&s"(carry_out) : "v"(w), "s"(mask));
    return r;
}
// mask[lane] ? b : a
__device__ __forceinline__ float select_by_mask(float a, float b, uint64_t mask)
{
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ float bperm(int byte_addr, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, src)));
}
__device__ __forceinline__ float first_lane(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// Log-prob row loads are issued from inline asm so that hipcc does not track them: with a store
// and a load of different kinds in flight it would wait vmcnt(0) before every row use, draining
// the whole prefetch ring each frame.  The matching counted wait is row_wait<N>() below.
__device__ __forceinline__ float row_load(uint32_t lane_byte_off, const void *row_base /* wave-uniform */)
{
    float r;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(lane_byte_off), "s"(row_base) : "memory");
    return r;
}
// Reload of a loop-carried row register: the destination is TIED to the register's previous (dead) contents, so
// the register allocator has to keep the row in one physical register around the loop.  With a plain output it
// may rotate the loop-carried registers with v_mov copies at the back-edge - copies of registers whose loads
// have not landed (tools/lint_inflight.py checks the compiled code for exactly that).
__device__ __forceinline__ void row_reload(float &r, uint32_t lane_byte_off, const void *row_base /* wave-uniform */)
{
    asm volatile("global_load_dword %0, %1, %2" : "+v"(r) : "v"(lane_byte_off), "s"(row_base) : "memory");
}
// wait until at most N younger vector-memory operations are outstanding, then release `r`
template <int N>
__device__ __forceinline__ void row_wait(float &r)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "i"(N) : "memory");
}

// same with a count that is a constant only after loop unrolling (the switch folds away)
__device__ __forceinline__ void row_wait_n(float &r, int n)
{
    switch (n) {
    case 0: row_wait<0>(r); break;
    case 1: row_wait<1>(r); break;
    case 2: row_wait<2>(r); break;
    case 3: row_wait<3>(r); break;
    case 4: row_wait<4>(r); break;
    case 5: row_wait<5>(r); break;
    case 6: row_wait<6>(r); break;
    case 7: row_wait<7>(r); break;
    case 8: row_wait<8>(r); break;
    case 9: row_wait<9>(r); break;
    case 10: row_wait<10>(r); break;
    case 11: row_wait<11>(r); break;
    case 12: row_wait<12>(r); break;
    case 13: row_wait<13>(r); break;
    case 14: row_wait<14>(r); break;
    case 15: row_wait<15>(r); break;
    case 16: row_wait<16>(r); break;
    case 17: row_wait<17>(r); break;
    case 18: row_wait<18>(r); break;
    case 19: row_wait<19>(r); break;
    case 20: row_wait<20>(r); break;
    case 21: row_wait<21>(r); break;
    case 22: row_wait<22>(r); break;
    case 23: row_wait<23>(r); break;
    case 24: row_wait<24>(r); break;
    case 25: row_wait<25>(r); break;
    case 26: row_wait<26>(r); break;
    case 27: row_wait<27>(r); break;
    case 28: row_wait<28>(r); break;
    case 29: row_wait<29>(r); break;
    case 30: row_wait<30>(r); break;
    case 31: row_wait<31>(r); break;
    default: row_wait<0>(r); break;
    }
}

// Lane masks of the band: m<k> has bit ((p>>4)&63) set for every p in [lo,hi) with p&15 == k.
// A struct of named members (not an array): members can only be addressed with constant
// indices, so the masks stay in SGPR pairs (an array indexed through the switch below is
// turned into a dynamically indexed vector and lands in VGPRs).
struct BandMasks {
    uint64_t m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15;
    template <int K>
    __device__ __forceinline__ uint64_t &at()
    {
        if constexpr (K == 0) return m0;
        else if constexpr (K == 1) return m1;
        else if constexpr (K == 2) return m2;
        else if constexpr (K == 3) return m3;
        else if constexpr (K == 4) return m4;
        else if constexpr (K == 5) return m5;
        else if constexpr (K == 6) return m6;
        else if constexpr (K == 7) return m7;
        else if constexpr (K == 8) return m8;
        else if constexpr (K == 9) return m9;
        else if constexpr (K == 10) return m10;
        else if constexpr (K == 11) return m11;
        else if constexpr (K == 12) return m12;
        else if constexpr (K == 13) return m13;
        else if constexpr (K == 14) return m14;
        else return m15;
    }
};
template <int K>
__device__ __forceinline__ void band_rebuild_one(BandMasks &mk, uint32_t lo, uint32_t hi)
{
    const uint32_t first = (lo + 15u - (uint32_t)K) >> 4;  // ceil((lo-K)/16)
    const uint32_t last = (hi + 15u - (uint32_t)K) >> 4;
    const uint32_t cnt = last - first;
    const uint64_t m = cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull);
    const uint32_t r = first & 63u;
    mk.at<K>() = r ? ((m << r) | (m >> (64u - r))) : m;
    if constexpr (K < 15) band_rebuild_one<K + 1>(mk, lo, hi);
}
__device__ __forceinline__ void band_rebuild(BandMasks &mk, uint32_t lo, uint32_t hi) { band_rebuild_one<0>(mk, lo, hi); }
// One position enters or leaves the band: flip its lane bit in the mask of its cell index (p & 15).
// The 16-way dispatch is hand-written: a binary tree of s_bitcmp1 / s_cbranch_scc1 (10 scalar instructions
// executed per call).  Any C++ formulation (switch, nested ifs, 16 compare-selects) is blown up by the
// compiler's CFG structuriser / 64-bit select lowering to 130-250 scalar instructions per call, and this
// path runs every few frames: it cost 14 % of the forward kernel.
__device__ __forceinline__ void band_toggle(BandMasks &mk, uint32_t p)
{
    const uint64_t bit = 1ull << ((p >> 4) & 63u);
    const uint32_t k = p & 15u;
    asm volatile(
        "s_bitcmp1_b32 %16, 3\n\t"
        "s_cbranch_scc1 .Lka_8_16_%=\n\t"
        "s_bitcmp1_b32 %16, 2\n\t"
        "s_cbranch_scc1 .Lka_4_8_%=\n\t"
        "s_bitcmp1_b32 %16, 1\n\t"
        "s_cbranch_scc1 .Lka_2_4_%=\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_1_2_%=\n\t"
        "s_xor_b64 %0, %0, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_1_2_%=:\n\t"
        "s_xor_b64 %1, %1, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_2_4_%=:\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_3_2_%=\n\t"
        "s_xor_b64 %2, %2, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_3_2_%=:\n\t"
        "s_xor_b64 %3, %3, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_4_8_%=:\n\t"
        "s_bitcmp1_b32 %16, 1\n\t"
        "s_cbranch_scc1 .Lka_6_4_%=\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_5_2_%=\n\t"
        "s_xor_b64 %4, %4, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_5_2_%=:\n\t"
        "s_xor_b64 %5, %5, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_6_4_%=:\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_7_2_%=\n\t"
        "s_xor_b64 %6, %6, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_7_2_%=:\n\t"
        "s_xor_b64 %7, %7, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_8_16_%=:\n\t"
        "s_bitcmp1_b32 %16, 2\n\t"
        "s_cbranch_scc1 .Lka_12_8_%=\n\t"
        "s_bitcmp1_b32 %16, 1\n\t"
        "s_cbranch_scc1 .Lka_10_4_%=\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_9_2_%=\n\t"
        "s_xor_b64 %8, %8, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_9_2_%=:\n\t"
        "s_xor_b64 %9, %9, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_10_4_%=:\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_11_2_%=\n\t"
        "s_xor_b64 %10, %10, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_11_2_%=:\n\t"
        "s_xor_b64 %11, %11, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_12_8_%=:\n\t"
        "s_bitcmp1_b32 %16, 1\n\t"
        "s_cbranch_scc1 .Lka_14_4_%=\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_13_2_%=\n\t"
        "s_xor_b64 %12, %12, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_13_2_%=:\n\t"
        "s_xor_b64 %13, %13, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_14_4_%=:\n\t"
        "s_bitcmp1_b32 %16, 0\n\t"
        "s_cbranch_scc1 .Lka_15_2_%=\n\t"
        "s_xor_b64 %14, %14, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_15_2_%=:\n\t"
        "s_xor_b64 %15, %15, %17\n\t"
        "s_branch .Lka_end_%=\n\t"
        ".Lka_end_%=:\n\t"
        : "+s"(mk.m0), "+s"(mk.m1), "+s"(mk.m2), "+s"(mk.m3), "+s"(mk.m4), "+s"(mk.m5), "+s"(mk.m6), "+s"(mk.m7),
          "+s"(mk.m8), "+s"(mk.m9), "+s"(mk.m10), "+s"(mk.m11), "+s"(mk.m12), "+s"(mk.m13), "+s"(mk.m14), "+s"(mk.m15)
        : "s"(k), "s"(bit)
        : "scc");
}
// bits 0,2,..,2(n-1)
__device__ __forceinline__ uint32_t pair_mask(int n)
{
    n = n < 0 ? 0 : n;
    return n >= 16 ? 0x55555555u : (((1u << (2 * n)) - 1u) & 0x55555555u);
}
// per-lane pair-space mask of the cells of block `blk` that are inside [lo,hi)
__device__ __forceinline__ uint32_t band_pairs(uint32_t lo, uint32_t hi, int blk)
{
    const int p0 = blk * 16;
    return pair_mask((int)hi - p0) & ~pair_mask((int)lo - p0);
}
__device__ __forceinline__ void load_block_labels(gci32_t labx, int blk, int (&la)[8])
{
    gci4_t p = (gci4_t)(labx + (size_t)blk * 8);
    const v4i_t a = p[0], b = p[1];
    la[0] = a.x; la[1] = a.y; la[2] = a.z; la[3] = a.w;
    la[4] = b.x; la[5] = b.y; la[6] = b.z; la[7] = b.w;
}

// 64-bit lane mask of (a == b), ordered compare
__device__ __forceinline__ uint64_t feq(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 1 /*FCMP_OEQ*/); }

// Cell update.  m = max over the allowed moves (one v_max3 [+ v_max]); the back-pointer is the
// FIRST move whose candidate equals m (np.argmax semantics, align.py:83) - found with equality
// compares against m, whose lane masks are combined on the scalar unit.
// Back-pointer code stored per cell (2 bits):
//   label cell (moves {0,1,2,3}): hi = e0|e1,  lo = e0 | (e2 & ~e1)    -> code = 3 - move
//   blank cell (moves {0,1,3}):   hi = e0,     lo = e1                 -> 0 if hi, else 1 if lo, else 3
// The blank code is the two compare masks as they come (no scalar work in the forward kernel, where it
// costs time); the backtrace turns it into 3 - move with a few bit-parallel VALU operations per loaded
// dword (blank_to_uniform) and then decodes every cell with ONE scalar instruction, move = 3 & ~code:
// its scalar chain is what bounds that kernel.
// one blank cell (even position): move 2 is vetoed for blanks (align.py:80-81)
template <int M>
__device__ __forceinline__ void cell_blank(float a0, float a1, float a3, float e, float &m, uint32_t &word)
{
    const float c0 = a0 + e;
    if constexpr (M == 1) {
        m = c0;
        word = (word << 2) | 3u;   // (1,1): move 0
    } else {
        const float c1 = a1 + e;
        if constexpr (M <= 3) {
            m = __builtin_fmaxf(c0, c1);
            word = (shl1_in(word, feq(c0, m)) << 1) | 1u;   // (e0, 1): c1 == m whenever c0 != m
        } else {
            const float c3 = a3 + e;
            m = __builtin_fmaxf(__builtin_fmaxf(c0, c1), c3);
            word = shl1_in(shl1_in(word, feq(c0, m)), feq(c1, m));   // (e0, e1): converted by the backtrace
        }
    }
}
// one label cell (odd position): moves 0..M-1; move 2 vetoed when the label VALUE is 0
template <int M, bool ZL>
__device__ __forceinline__ void cell_label(float a0, float a1, float a2, float a3, float e, float veto,
                                           float &m, uint32_t &word)
{
    const float c0 = a0 + e;
    if constexpr (M == 1) {
        m = c0;
        word = (word << 2) | 3u;   // (1,1): move 0
    } else {
        const float c1 = a1 + e;
        if constexpr (M == 2) {
            m = __builtin_fmaxf(c0, c1);
            const uint64_t e0 = feq(c0, m);
            word = shl1_in((word << 1) | 1u, e0);   // hi=1 always (move < 2), lo = e0
        } else {
            float c2 = a2 + e;
            if constexpr (ZL) c2 = __builtin_fminf(c2, veto);  // veto = -inf where label == 0, else +inf
            if constexpr (M == 3) {
                m = __builtin_fmaxf(__builtin_fmaxf(c0, c1), c2);
                const uint64_t e0 = feq(c0, m), e1 = feq(c1, m);
                // move 2 is the only one left when neither e0 nor e1: code (0,1)
                word = shl1_in(shl1_in(word, e0 | e1), ~e1 | e0);
            } else {
                const float c3 = a3 + e;
                m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(c0, c1), c2), c3);
                const uint64_t e0 = feq(c0, m), e1 = feq(c1, m), e2 = feq(c2, m);
                word = shl1_in(shl1_in(word, e0 | e1), e0 | (e2 & ~e1));
            }
        }
    }
}
// stored dword (16 cells) -> every cell coded as 3 - move: the blank cells (even cells: bit pairs 4i+1,4i)
// go from (e0, e1) to (e0|e1, e0)
__device__ __forceinline__ uint32_t blank_to_uniform(uint32_t x)
{
    const uint32_t h = (x >> 1) & 0x11111111u, l = x & 0x11111111u;
    return (x & 0xCCCCCCCCu) | ((h | l) << 1) | h;
}
// number of positions to step back, from the uniform 2-bit code of a cell
__device__ __forceinline__ int bp_decode(uint32_t code) { return (int)(3u & ~code); }
// a cell's state is live after the frame iff it is in the band and (it moved in from a live
// state = any move > 0, or it stayed on a live state).  "move == 0" per cell, bit-parallel on
// the packed word (bit 2k+1 = hi, bit 2k = lo; even cells are blanks, odd cells labels):
//   blank: move 0 <=> hi;  label: move 0 <=> hi & lo
// Only the pair-bits (even bit positions) of the result mean anything; `band2` has zeros elsewhere.
__device__ __forceinline__ uint32_t live_pairs(uint32_t live2, uint32_t word, uint32_t band2)
{
    const uint32_t stay = (word >> 1) & (word | 0x11111111u);   // 0x1111..: pair-bits of even cells (k = 0,2,4,..)
    return (live2 | ~stay) & band2;
}

// score-only cells (checkpointed path: the back-pointers are recomputed by backtrace_rc_kernel).
// Only the value of the best candidate is needed here, and float32 addition is monotone in each operand:
//   max_j fl(a_j + e) == fl(max_j a_j + e)   bit for bit (also with -inf operands)
// so the emission is added ONCE, after the max over the predecessors: 3-4 instructions per cell instead of
// 5-7.  (The reference's arg-max ties are decided on the sums, align.py:83 - that needs the per-candidate sums
// and is what backtrace_rc_kernel recomputes for the cells around the path.)
// These are the maxima; frame_scores adds the emissions, two cells at a time.
template <int M>
__device__ __forceinline__ float cell_blank_max(float a0, float a1, float a3)
{
    if constexpr (M == 1) return a0;
    if constexpr (M <= 3) return __builtin_fmaxf(a0, a1);
    return __builtin_fmaxf(__builtin_fmaxf(a0, a1), a3);
}
template <int M, bool ZL>
__device__ __forceinline__ float cell_label_max(float a0, float a1, float a2, float a3, float veto)
{
    if constexpr (M == 1) return a0;
    if constexpr (M == 2) return __builtin_fmaxf(a0, a1);
    if constexpr (ZL) a2 = __builtin_fminf(a2, veto);   // veto = -inf where the label value is 0 (move 2 not allowed), else +inf
    if constexpr (M == 3) return __builtin_fmaxf(__builtin_fmaxf(a0, a1), a2);
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a0, a1), a2), a3);
}
// Score registers of the checkpointed kernel: cell k of a lane sits in P[2*(k>>2) + (k&1)][(k>>1)&1], i.e. the
// two blank cells of a group of four share one 64-bit register pair and so do its two label cells - the emission
// is then added to two cells per instruction (v_pk_add_f32: the vector ALU is what bounds this kernel).
#define KA_P(P, k) (P)[2 * ((k) >> 2) + ((k) & 1)][((k) >> 1) & 1]
// cells 4G+3..4G of one frame, in place (descending G: cell k reads the old k-1..k-3), NO band mask
// emission gather: the next frame's log-prob row sits in LDS (one ds_write_b32 per frame), a label cell reads its
// column with ds_read_b32.  ds_bpermute_b32 on the row register does the same without the write, but costs 7.0
// cycles of the CU's LDS pipe per wave-instruction against 3.8 for the read (tools/ubench/lds_rates.hip) - with
// 8 gathers per frame and 32 waves per CU that pipe was 75 % busy with them.
__device__ __forceinline__ float lds_col(const float *row, int byte_addr)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(row) + byte_addr);
}
template <int M, bool ZL, int G>
__device__ __forceinline__ void frame_scores(f32x2 (&P)[8], float h1, float h2, float h3, f32x2 (&E)[4], const float (&vz)[8],
                                             f32x2 e00, const int (&la)[8], const float *next_row)
{
    const float b0 = P[2 * G][0], l0 = P[2 * G + 1][0], b1 = P[2 * G][1], l1 = P[2 * G + 1][1];
    const float p1 = G > 0 ? P[2 * (G > 0 ? G - 1 : 0) + 1][1] : h1;   // cell 4G-1 (label)
    const float p2 = G > 0 ? P[2 * (G > 0 ? G - 1 : 0)][1] : h2;       // cell 4G-2 (blank)
    const float p3 = G > 0 ? P[2 * (G > 0 ? G - 1 : 0) + 1][0] : h3;   // cell 4G-3 (label)
    f32x2 ml, mb;
    ml[1] = cell_label_max<M, ZL>(l1, b1, l0, b0, vz[2 * G + 1]);
    mb[1] = cell_blank_max<M>(b1, l0, p1);
    ml[0] = cell_label_max<M, ZL>(l0, b0, p1, p2, vz[2 * G]);
    mb[0] = cell_blank_max<M>(b0, p1, p3);
    P[2 * G + 1] = ml + E[G];
    P[2 * G] = mb + e00;
    E[G][0] = lds_col(next_row, la[2 * G]);
    E[G][1] = lds_col(next_row, la[2 * G + 1]);
    if constexpr (G > 0) frame_scores<M, ZL, G - 1>(P, h1, h2, h3, E, vz, e00, la, next_row);
}

// -inf into the cells outside the band
template <int K>
__device__ __forceinline__ void mask_scores(f32x2 (&P)[8], BandMasks &mk, float NINF)
{
    KA_P(P, K) = select_by_mask(NINF, KA_P(P, K), mk.at<K>());
    if constexpr (K > 0) mask_scores<K - 1>(P, mk, NINF);
}

// cells 15..0 of one frame, in place (descending k: cell k reads the old k-1..k-3)
// The emission register of a label cell is refilled for the NEXT frame (ds_bpermute of the next
// row) right after the cell has consumed it: one set of 8 emission registers, and a whole frame of
// other work between a gather and its use.
template <int M, bool ZL, int K>
__device__ __forceinline__ void frame_cells(float (&sc)[16], float h1, float h2, float h3, float (&ec)[8],
                                            const float (&vz)[8], float e0, BandMasks &mk, float NINF, uint32_t &word,
                                            const int (&la)[8], float next_row)
{
    const float a0 = sc[K];
    const float a1 = K >= 1 ? sc[K >= 1 ? K - 1 : 0] : h1;
    const float a2 = K >= 2 ? sc[K >= 2 ? K - 2 : 0] : (K == 1 ? h1 : h2);
    const float a3 = K >= 3 ? sc[K >= 3 ? K - 3 : 0] : (K == 2 ? h1 : (K == 1 ? h2 : h3));
    float m;
    if constexpr (K & 1) {
        cell_label<M, ZL>(a0, a1, a2, a3, ec[K >> 1], vz[K >> 1], m, word);
        ec[K >> 1] = bperm(la[K >> 1], next_row);
    } else {
        cell_blank<M>(a0, a1, a3, e0, m, word);
    }
    sc[K] = select_by_mask(NINF, m, mk.at<K>());
    // keep the cells in program order, four at a time: left alone, the scheduler hoists the next frame's
    // gathers and interleaves all 16 cells, which costs ~16 VGPRs and ~60 spilled SGPRs
    if constexpr (K % 4 == 0) __builtin_amdgcn_sched_barrier(0);
    if constexpr (K > 0) frame_cells<M, ZL, K - 1>(sc, h1, h2, h3, ec, vz, e0, mk, NINF, word, la, next_row);
}

// ---------------------------------------------------------------------------------------
// forward DP, one wavefront per lattice
// ---------------------------------------------------------------------------------------
// Exact form: every cell's back-pointer is stored (the checkpointed form is forward_ck below).
template <int M, bool ZL>
__device__ __forceinline__ void forward_w16(const Lattice &d, int32_t *meta)
{
    constexpr int D = kRowDepth;
    const int lane = threadIdx.x;
    // descriptor fields are wave-uniform; say so explicitly so that everything derived from
    // them (band limits, lane masks) is kept on the scalar unit
    const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    const uint32_t B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    const uint32_t halfB = B >> 1;
    const uint32_t dq = L / T, dr = L % T;
    const float NINF = ninf();

    float sc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sc[k] = NINF;
    if (lane == 0) sc[0] = 0.0f;            // virtual state before frame 0 (align.py:57-58)
    uint32_t pres2 = lane == 0 ? 1u : 0u;   // bit 2k: cell k holds a live state
    bool pend_reset = false;                // wave-uniform: some lane was re-labelled for this frame
    bool reset_lane = false;                // per lane: this lane was re-labelled

    int blk = lane;                         // block of 16 positions this lane currently owns
    uint32_t blo = 0;                       // lo >> 4
    int la[8];
    float vz[8];
    gci32_t labx = (gci32_t)d.labx;
    load_block_labels(labx, blk, la);
#pragma unroll
    for (int i = 0; i < 8; ++i) vz[i] = (ZL && la[i] == 0) ? NINF : __builtin_inff();

    uint32_t q = 0, rem = 0;                // floor(L*t/T) and its remainder, advanced per frame
    uint32_t lo = 0, hi = B < L ? B : L;    // band of frame 0
    BandMasks mk;
    band_rebuild(mk, lo, hi);
    uint32_t band2 = band_pairs(lo, hi, blk);

    // lanes >= V read column 0 (a valid address); their value is never selected (labels < V)
    const uint32_t lane_off = (lane < d.V ? (uint32_t)lane : 0u) * 4u;
    const char *lp = reinterpret_cast<const char *>(d.lp);
    const size_t ld = (size_t)d.ld * 4;  // row pitch in bytes
    float rows[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const uint32_t tt = (uint32_t)i < T ? (uint32_t)i : T - 1;
        rows[i] = row_load(lane_off, lp + (size_t)tt * ld);
    }
    // Start-up: the counted wait inside the loop assumes the steady-state number of younger
    // operations (2 stores + 2 loads); the first D rows have fewer behind them, so land them all.
#pragma unroll
    for (int i = 0; i < D; ++i) row_wait<0>(rows[i]);
    // emissions of the label cells: e[i] holds frame t's value until cell 2i+1 has used it, then
    // frame t+1's (see frame_cells); the blank emission is a scalar, double-buffered by frame parity
    float e[8], e0[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = bperm(la[i], rows[0]);
    e0[0] = first_lane(rows[0]);
    float absum = __builtin_fabsf(rows[0]);   // NaN detector: sum over frames of |lp[t, lane]|

    const uint32_t *bp = reinterpret_cast<const uint32_t *>(d.bp);   // wave-uniform row base
    const uint32_t lane_store_off = (uint32_t)lane * 16u;   // back-pointers: [t/4][block][t%4] dwords

    const char *row_ahead = lp + (size_t)(D < T ? D : T - 1) * ld;   // row min(t+D, T-1) of the current frame t
    uint32_t step_thr = dq != 0 ? 0u : T;   // floor(L*t/T) moves in this frame <=> rem + dr >= step_thr
    asm("" : "+s"(step_thr));               // (opaque: one s_cmp + s_cbranch per frame instead of a boolean expression)
    for (uint32_t tb = 0; tb < T; tb += D) {
        // back-pointer words of the 4 frames of this group.  Frames past T leave theirs undefined (the
        // buffer is padded to whole groups); an empty asm output costs nothing, a zero costs a v_mov.
        uint32_t gw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("" : "=v"(gw[i]));
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const uint32_t t = tb + dd;
            if (t < T) {
                // Halos of frame t, then the reset of lanes re-labelled in frame t-1 (they still held the scores of
                // their OLD block, which their right neighbour has just read as its halo).  Done before step A so that
                // A can write the flags of the NEXT frame in place (no per-frame copies of flags and band limits).
                float h1 = wave_ror1(sc[15]), h2 = wave_ror1(sc[14]), h3 = wave_ror1(sc[13]);
                if (__builtin_expect(pend_reset, 0)) {
                    asm volatile("" ::: "memory");  // keep this rare block a real branch (no if-conversion)
                    // a lane re-labelled for this frame holds scores of its OLD block: its new
                    // positions were not live in frame t-1.  Its left halo is valid unless the left
                    // neighbour was re-labelled in the same step (then nobody held those positions).
                    const bool left_reset = __builtin_amdgcn_update_dpp(0, (int)reset_lane, 0x13C, 0xF, 0xF, false) != 0;
                    const bool kill = reset_lane && left_reset;
                    h1 = kill ? NINF : h1;
                    h2 = kill ? NINF : h2;
                    h3 = kill ? NINF : h3;
#pragma unroll
                    for (int k = 0; k < 16; ++k) sc[k] = reset_lane ? NINF : sc[k];
                    pres2 = reset_lane ? 0u : pres2;
                    pend_reset = false;
                }
                // A. band of frame t+1; re-label the lanes whose block has been passed by lo.  The band is a
                // function of floor(L*t/T): nothing to do in the frames where that does not move (every
                // instruction costs issue time here, scalar ones included)
                bool moved = false;
                rem += dr;
                if (__builtin_expect(rem >= step_thr, 0)) {
                    asm volatile("" ::: "memory");  // a real branch: the common frame pays an add, a compare and a jump
                    q += dq;
                    if (rem >= T) { rem -= T; ++q; }
                    if (t + 1 != T) {   // no frame T: keep the last band and labels
                        moved = true;
                        const int32_t dlo = (int32_t)q - (int32_t)halfB;  // signed on purpose: s_max_i32, not a VALU usubsat
                        const uint32_t nlo = (uint32_t)(dlo > 0 ? dlo : 0);
                        if ((nlo >> 4) != blo) {
                            blo = nlo >> 4;
                            const int nb = (int)blo + ((lane - (int)blo) & 63);
                            reset_lane = nb != blk;
                            if (nb != blk) {
                                blk = nb;
                                load_block_labels(labx, blk, la);
                                // consume the loads HERE: otherwise the wait for them lands at the merge
                                // point as an every-frame s_waitcnt vmcnt(0) that also drains the row
                                // prefetches and the back-pointer stores
#pragma unroll
                                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(la[i]));
                            }
                            pend_reset = true;
                        }
                    }
                }
                // B. row t+1 (its emissions are gathered while frame t is computed).  It was issued D-1
                // frames ago; since then (D-2) frames each issued one row load, and the group store that
                // follows frame 4k+3 lies in between unless dd = 3 (rare label reloads only add younger ops)
                if (dd < D - 1) row_wait<D - 1>(rows[(dd + 1) % D]); else row_wait<D - 2>(rows[(dd + 1) % D]);
                const float rn = rows[(dd + 1) % D];
                e0[(dd + 1) & 1] = first_lane(rn);
                absum += __builtin_fabsf(rn);
                // C. frame t
                uint32_t word = 0;
                frame_cells<M, ZL, 15>(sc, h1, h2, h3, e, vz, e0[dd & 1], mk, NINF, word, la, rn);
                gw[dd] = word;
                // live <=> in band and (moved in from a live state, or stayed on a live state)
                pres2 = live_pairs(pres2, word, band2);
                // prefetch the row of frame t+D (the last row again once there is none: never consumed)
                row_reload(rows[dd], lane_off, row_ahead);
                row_ahead += t + D + 1 < T ? ld : 0;
                // D. lane masks of frame t+1
                if (moved) {
                    const int32_t dlo = (int32_t)q - (int32_t)halfB;
                    const uint32_t nlo = (uint32_t)(dlo > 0 ? dlo : 0);
                    const uint32_t nhi = (L - nlo < B) ? L : nlo + B;
                    if (nlo != lo || nhi != hi) {
                        if (nhi - hi <= 1u && nlo - lo <= 1u) {
                            if (nhi != hi) band_toggle(mk, hi);
                            if (nlo != lo) band_toggle(mk, lo);
                        } else {
                            band_rebuild(mk, nlo, nhi);
                        }
                        band2 = band_pairs(nlo, nhi, blk);
                        lo = nlo;
                        hi = nhi;
                        if (ZL && pend_reset) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) vz[i] = la[i] == 0 ? NINF : __builtin_inff();
                        }
                    }
                }
            }
        }
        // one 16-byte store per lane per 4 frames.  saddr (uniform pointer to the group) + voffset
        // (lane*16): no per-lane 64-bit address registers
        const u32x4 words = {gw[0], gw[1], gw[2], gw[3]};
        // (s_nop 1: a store wider than 64 bits reads its data registers for two more wait states, see forward_ck)
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(lane_store_off), "v"(words), "s"(bp + (size_t)tb * 64) : "memory");
    }

    // Drain the row prefetches that are still in flight (the last D frames prefetch clamped rows that are
    // never consumed).  Their destination registers are dead to the compiler after the loop: without this
    // wait it reuses them for the reduction below and a late-landing load overwrites live values.
#pragma unroll
    for (int i = 0; i < D; ++i) row_wait<0>(rows[i]);

    int32_t *m = meta_of(meta, d.idx);
    if (__builtin_amdgcn_ballot_w64((__builtin_bit_cast(uint32_t, absum) & 0x7fffffffu) > 0x7f800000u)) {   // a NaN log-prob
        if (lane == 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusNaN);
        }
        return;
    }
    // terminal state: the HIGHEST live position of frame T-1 (align.py:99-101)
    int best = -1;
    if (pres2) best = blk * 16 + ((31 - __clz((int)pres2)) >> 1);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (best < 0) {
        if (lane == 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusEmptyBeam);
        }
    } else if ((best >> 4) == blk) {
        float v = sc[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) v = (best & 15) == k ? sc[k] : v;
        m[1] = best;
        m[3] = __builtin_bit_cast(int32_t, v);
    }
}

// ---------------------------------------------------------------------------------------
// forward DP of the checkpointed form: scores only, one wavefront per lattice
//
// Same ring, labels, band and row pipeline as forward_w16, but the loop is arranged so that the frame that
// does not step the band - most frames - executes nothing but its own arithmetic and ONE scalar compare
// and branch:
//   * the halos of frame t+1 are taken at the end of frame t and carried in registers, so the re-labelling
//     of lanes, their reset and the band masks all live in one rare block after the cells;
//   * the band mask (a v_cndmask per cell) is applied only where it is needed: in the frame before a band
//     step (the cells that become live must hold -inf), in the first frame of a new band (cells that left
//     it must die) - both inside the rare block, the second by forcing the next frame through it - and in
//     every eighth (narrow gap: fourth) frame; checkpoints are taken there.  In between, cells above hi pick
//     up "leaked" scores from the live cells below them, M-1 cells further per frame; moves only go up, so a
//     leak cannot reach a live cell except around the ring, through the >= 15 dead slots between hi and lo:
//     at most 3 frames x 3 cells + the 3 cells lo reads (7 frames when there are >= 24 dead slots).
// ---------------------------------------------------------------------------------------
template <int M, bool ZL>
__device__ __forceinline__ void forward_ck(const Lattice &d, int32_t *meta)
{
    constexpr int D = kRowDepth;
    const int lane = threadIdx.x;
    const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    const uint32_t B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    const uint32_t halfB = B >> 1;
    const uint32_t dq = L / T, dr = L % T;
    const float NINF = ninf();

    f32x2 P[8];                             // cell k of the lane: KA_P(P, k)
#pragma unroll
    for (int k = 0; k < 16; ++k) KA_P(P, k) = NINF;
    if (lane == 0) KA_P(P, 0) = 0.0f;       // virtual state before frame 0 (align.py:57-58)
    float absum = 0.0f;                     // sum over frames of |lp[t, lane]| (finiteness check)

    int blk = lane;                         // block of 16 positions this lane currently owns
    uint32_t blo = 0;                       // lo >> 4
    int la[8];
    float vz[8];
    gci32_t labx = (gci32_t)d.labx;
    load_block_labels(labx, blk, la);
#pragma unroll
    for (int i = 0; i < 8; ++i) vz[i] = (ZL && la[i] == 0) ? NINF : __builtin_inff();

    uint32_t q = 0, rem = 0;                // floor(L*t/T) and its remainder, advanced per frame
    uint32_t lo = 0, hi = B < L ? B : L;    // band of frame 0
    BandMasks mk;
    band_rebuild(mk, lo, hi);

    const uint32_t lane_off = (lane < d.V ? (uint32_t)lane : 0u) * 4u;
    const char *lp = reinterpret_cast<const char *>(d.lp);
    const size_t ld = (size_t)d.ld * 4;  // row pitch in bytes
    float rows[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const uint32_t tt = (uint32_t)i < T ? (uint32_t)i : T - 1;
        rows[i] = row_load(lane_off, lp + (size_t)tt * ld);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) row_wait<0>(rows[i]);
    f32x2 E[4];                             // emissions of the label cells (see frame_scores)
    float e0[2];                            // blank emission, double-buffered by frame parity
    __shared__ float lrow[64];              // row t+1 while frame t is computed (the workgroup is this one wavefront)
    lrow[lane] = rows[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) E[i >> 1][i & 1] = lds_col(lrow, la[i]);
    e0[0] = first_lane(rows[0]);
    absum = __builtin_fabsf(rows[0]);

    const char *ckp = reinterpret_cast<const char *>(d.bp);
    const char *row_ahead = lp + (size_t)(D < T ? D : T - 1) * ld;   // row min(t+D, T-1) of the current frame t
    const uint32_t thr_real = dq != 0 ? 0u : T;   // floor(L*t/T) moves in this frame <=> rem + dr >= thr_real
    // static mask: frame 8k+7 when the ring has at least 8*(M-1) dead slots (7 unmasked frames leak 7*(M-1) cells and
    // lo reads M-1 below itself), else frame 4k+3.  Checkpoints (frame 32k+31) are masked frames either way.
    const uint32_t mask_every4 = 1024u - (B < L ? B : L) >= 8u * (M - 1) ? 0u : 4u;
    uint32_t thr = thr_real;                // 0 for one frame after a band step: that frame must come through the rare block
    asm("" : "+s"(thr));                    // (opaque: one s_cmp + s_cbranch per frame instead of a boolean expression)
    // halos of frame 0: lane 63's cells 13..15 of the initial state
    float h1 = wave_ror1(KA_P(P, 15)), h2 = wave_ror1(KA_P(P, 14)), h3 = wave_ror1(KA_P(P, 13));
    for (uint32_t tb = 0; tb < T; tb += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const uint32_t t = tb + dd;
            if (t < T) {
                // row t+1 (its emissions are gathered while frame t is computed): issued D-1 frames ago, D-2
                // younger row loads behind it (checkpoint stores and label reloads only add younger operations)
                row_wait<D - 2>(rows[(dd + 1) % D]);
                const float rn = rows[(dd + 1) % D];
                lrow[lane] = rn;                 // (LDS operations of a wave execute in order: the reads of row t are done)
                e0[(dd + 1) & 1] = first_lane(rn);
                absum += __builtin_fabsf(rn);
                const float e0t = e0[dd & 1];
                frame_scores<M, ZL, 3>(P, h1, h2, h3, E, vz, f32x2{e0t, e0t}, la, lrow);
                // prefetch the row of frame t+D (the last row again once there is none: never consumed)
                row_reload(rows[dd], lane_off, row_ahead);
                row_ahead += t + D + 1 < T ? ld : 0;
                if (dd == D - 1 && ((tb | mask_every4) & 4u) != 0) {
                    mask_scores<15>(P, mk, NINF);
                    if (((tb + D) & (kCkFrames - 1)) == 0 && tb + D < T) {
                        // checkpoint (tb+D)/kCkFrames: the scores after frame tb+D-1, [lane][16 cells], 4 KB.  Taken
                        // before the rare block resets re-labelled lanes: their old positions are still inputs
                        // of frame tb+D.
                        const char *ck = ckp + ((size_t)((tb + D) / kCkFrames) - 1) * 4096;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {KA_P(P, 4 * g), KA_P(P, 4 * g + 1), KA_P(P, 4 * g + 2), KA_P(P, 4 * g + 3)};
                            // s_nop 1: the compiler stages all four groups through the same four registers and does not
                            // know that on gfx940+ a store wider than 64 bits still reads its data registers for two
                            // wait states after it has issued (one was not enough: under load the first dword of a
                            // group came out as the next group's - tools/check_batch.py, tests: batch vs oracle)
                            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"((uint32_t)lane * 64u), "v"(v), "s"(ck), "i"(16 * g) : "memory");
                        }
                    }
                }
                h1 = wave_ror1(KA_P(P, 15));
                h2 = wave_ror1(KA_P(P, 14));
                h3 = wave_ror1(KA_P(P, 13));
                rem += dr;
                if (__builtin_expect(rem >= thr, 0)) {
                    asm volatile("" ::: "memory");  // a real branch: the common frame pays an add, a compare and a jump
                    thr = thr_real;
                    if (dd != D - 1 || ((tb | mask_every4) & 4u) == 0) {
                        mask_scores<15>(P, mk, NINF);
                        h1 = wave_ror1(KA_P(P, 15));
                        h2 = wave_ror1(KA_P(P, 14));
                        h3 = wave_ror1(KA_P(P, 13));
                    }
                    if (rem >= thr_real) {
                        q += dq;
                        if (rem >= T) { rem -= T; ++q; }
                        if (t + 1 != T) {   // no frame T: keep the last band and labels
                            const int32_t dlo = (int32_t)q - (int32_t)halfB;  // signed on purpose: s_max_i32, not a VALU usubsat
                            const uint32_t nlo = (uint32_t)(dlo > 0 ? dlo : 0);
                            const uint32_t nhi = (L - nlo < B) ? L : nlo + B;
                            if ((nlo >> 4) != blo) {
                                // re-label the lanes whose block lo has passed.  They held scores of their OLD block,
                                // which their right neighbour has just taken as its halo; their new positions were not
                                // live in frame t.  A lane's left halo is valid unless its left neighbour was
                                // re-labelled in the same step (then nobody held those positions).
                                blo = nlo >> 4;
                                const int nb = (int)blo + ((lane - (int)blo) & 63);
                                const bool reset_lane = nb != blk;
                                if (reset_lane) {
                                    blk = nb;
                                    load_block_labels(labx, blk, la);
#pragma unroll
                                    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(la[i]));
                                }
                                const bool left_reset = __builtin_amdgcn_update_dpp(0, (int)reset_lane, 0x13C, 0xF, 0xF, false) != 0;
                                const bool kill = reset_lane && left_reset;
                                h1 = kill ? NINF : h1;
                                h2 = kill ? NINF : h2;
                                h3 = kill ? NINF : h3;
#pragma unroll
                                for (int k = 0; k < 16; ++k) KA_P(P, k) = reset_lane ? NINF : KA_P(P, k);
                                // emissions of frame t+1 with the new labels (the cells gathered them with the old ones)
#pragma unroll
                                for (int i = 0; i < 8; ++i) E[i >> 1][i & 1] = lds_col(lrow, la[i]);
                                if constexpr (ZL) {
#pragma unroll
                                    for (int i = 0; i < 8; ++i) vz[i] = la[i] == 0 ? NINF : __builtin_inff();
                                }
                            }
                            if (nlo != lo || nhi != hi) {
                                if (nhi - hi <= 1u && nlo - lo <= 1u) {
                                    if (nhi != hi) band_toggle(mk, hi);
                                    if (nlo != lo) band_toggle(mk, lo);
                                } else {
                                    band_rebuild(mk, nlo, nhi);
                                }
                                lo = nlo;
                                hi = nhi;
                                thr = 0;   // frame t+1 is the first of a new band: it must be masked
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) row_wait<0>(rows[i]);   // drain the prefetches (see forward_w16)

    int32_t *m = meta_of(meta, d.idx);
    mask_scores<15>(P, mk, NINF);   // the last frame may have run unmasked
    // Every partial path score is bounded by the sum of all |lp|: if each column's sum stays below 1e30 nothing
    // can have overflowed and every live state has a finite score, so live <=> score > -inf.  Otherwise (an
    // infinity, a NaN, absurd magnitudes) hand the lattice to the exact kernels.
    // (integer test on the bits: the library is built with -fno-honor-nans, under which `!(absum < 1e30f)` is
    //  lowered to an ordered compare that a NaN passes)
    const uint32_t abits = __builtin_bit_cast(uint32_t, absum) & 0x7fffffffu;
    if (__builtin_amdgcn_ballot_w64(abits > 0x7f800000u)) {   // NaN: an explicit error, no path
        if (lane == 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusNaN);
        }
        return;
    }
    if (__builtin_amdgcn_ballot_w64(abits >= __builtin_bit_cast(uint32_t, 1e30f))) {
        if (lane == 0) atomicOr(&m[2], kFlagExact);
        return;
    }
    float sc[16];
    uint32_t pres2 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        sc[k] = KA_P(P, k);
        pres2 |= sc[k] != NINF ? (1u << (2 * k)) : 0u;
    }
    // terminal state: the HIGHEST live position of frame T-1 (align.py:99-101)
    int best = -1;
    if (pres2) best = blk * 16 + ((31 - __clz((int)pres2)) >> 1);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (best < 0) {
        if (lane == 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusEmptyBeam);
        }
    } else if ((best >> 4) == blk) {
        float v = sc[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) v = (best & 15) == k ? sc[k] : v;
        m[1] = best;
        m[3] = __builtin_bit_cast(int32_t, v);
    }
}

#ifndef KA_RC_MIN_WAVES
#define KA_RC_MIN_WAVES 8
#endif
#ifndef KA_FWD_MIN_WAVES
#define KA_FWD_MIN_WAVES 4
#endif
// Two kernels per max_move, launched back to back over the same lattices: ZL = the transcript
// contains label 0 (needs the per-label veto).  A wave whose lattice belongs to the other kernel
// exits at once.  Keeping them apart keeps the veto registers out of the common kernel.
// only_flagged: second pass behind the checkpointed kernels, for the lattices they declined
template <int M, bool ZL>
__global__ __launch_bounds__(64, KA_FWD_MIN_WAVES) void forward_w16_kernel(const Lattice *__restrict__ lats, int32_t *meta, int only_flagged)
{
    const Lattice &d = lats[blockIdx.x];
    const int flags = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2]);
    if (((flags & kFlagZeroLabel) != 0) != ZL) return;
    if (only_flagged && !(flags & kFlagExact)) return;
    forward_w16<M, ZL>(d, meta);
}
template <int M, bool ZL>
__global__ __launch_bounds__(64, KA_FWD_MIN_WAVES) void forward_ck_kernel(const Lattice *__restrict__ lats, int32_t *meta)
{
    const Lattice &d = lats[blockIdx.x];
    const int flags = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2]);
    if (((flags & kFlagZeroLabel) != 0) != ZL) return;
    forward_ck<M, ZL>(d, meta);
}

// ---------------------------------------------------------------------------------------
// forward DP, latency form ("wg4"): one 256-thread workgroup (4 wavefronts) per lattice
//
// Same recurrence, same 1024-slot ring, same back-pointer layout as w16, but each lane owns 4
// consecutive positions instead of 16, so a frame is ~1/4 of the instructions per wavefront.
// The three neighbours of a wave's lane 0 live in the previous wave's lane 63: each wave drops
// them into a double-buffered LDS slot at the end of a frame and the workgroup meets at ONE
// s_barrier per frame.  Used when there are too few lattices to fill the chip with one
// wavefront each (single files, a book's few dozen chapters): per-frame latency is what counts there.
// ---------------------------------------------------------------------------------------
struct BandMasks4 {
    uint64_t m0, m1, m2, m3;
    template <int K>
    __device__ __forceinline__ uint64_t &at()
    {
        if constexpr (K == 0) return m0;
        else if constexpr (K == 1) return m1;
        else if constexpr (K == 2) return m2;
        else return m3;
    }
};
// lanes [x, y) of a 64-bit mask, 0 <= x, y <= 64
__device__ __forceinline__ uint64_t lane_range(uint32_t x, uint32_t y)
{
    if (y <= x) return 0ull;
    const uint32_t n = y - x;
    return (n >= 64u ? ~0ull : ((1ull << n) - 1ull)) << x;
}
// mask of wave `wv` for cell index K: sub-slot u = 64*wv + lane holds position 4u+K (mod 1024)
template <int K>
__device__ __forceinline__ uint64_t band_mask4(uint32_t lo, uint32_t hi, uint32_t wv)
{
    const uint32_t first = (lo + 3u - (uint32_t)K) >> 2;   // ceil((lo-K)/4)
    const uint32_t last = (hi + 3u - (uint32_t)K) >> 2;
    const uint32_t cnt = last - first;                     // <= 253
    const uint32_t a = (first - 64u * wv) & 255u;          // first in-band sub-slot relative to this wave's lane 0
    const uint32_t e1 = a + cnt;                           // one past the last, before wrapping at 256
    uint64_t m = lane_range(a < 64u ? a : 64u, e1 < 64u ? e1 : 64u);
    if (e1 > 256u) {
        const uint32_t w = e1 - 256u;
        m |= lane_range(0u, w < 64u ? w : 64u);
    }
    return m;
}
__device__ __forceinline__ void band_rebuild4(BandMasks4 &mk, uint32_t lo, uint32_t hi, uint32_t wv)
{
    mk.m0 = band_mask4<0>(lo, hi, wv);
    mk.m1 = band_mask4<1>(lo, hi, wv);
    mk.m2 = band_mask4<2>(lo, hi, wv);
    mk.m3 = band_mask4<3>(lo, hi, wv);
}
__device__ __forceinline__ void band_toggle4(BandMasks4 &mk, uint32_t p, uint32_t wv)
{
    const uint32_t u = (p >> 2) & 255u;
    if ((u >> 6) != wv) return;
    const uint64_t bit = 1ull << (u & 63u);
    switch (p & 3u) {
    case 0: mk.m0 ^= bit; break;
    case 1: mk.m1 ^= bit; break;
    case 2: mk.m2 ^= bit; break;
    default: mk.m3 ^= bit; break;
    }
}
__device__ __forceinline__ uint32_t pair_mask4(int n)
{
    n = n < 0 ? 0 : (n > 4 ? 4 : n);
    return ((1u << (2 * n)) - 1u) & 0x55u;
}
template <int M, bool ZL, int K>
__device__ __forceinline__ void frame_cells4(float (&sc)[4], float h1, float h2, float h3, const float (&ec)[2],
                                             const float (&vz)[2], float e0, BandMasks4 &mk, float NINF, uint32_t &word)
{
    const float a0 = sc[K];
    const float a1 = K >= 1 ? sc[K >= 1 ? K - 1 : 0] : h1;
    const float a2 = K >= 2 ? sc[K >= 2 ? K - 2 : 0] : (K == 1 ? h1 : h2);
    const float a3 = K >= 3 ? sc[K >= 3 ? K - 3 : 0] : (K == 2 ? h1 : (K == 1 ? h2 : h3));
    float m;
    if constexpr (K & 1)
        cell_label<M, ZL>(a0, a1, a2, a3, ec[K >> 1], vz[K >> 1], m, word);
    else
        cell_blank<M>(a0, a1, a3, e0, m, word);
    sc[K] = select_by_mask(NINF, m, mk.at<K>());
    if constexpr (K > 0) frame_cells4<M, ZL, K - 1>(sc, h1, h2, h3, ec, vz, e0, mk, NINF, word);
}

template <int M, bool ZL>
__device__ __forceinline__ void forward_wg4(const Lattice &d, int32_t *meta)
{
    constexpr int D = kRowDepth;
    // [frame parity][wave][h1, h2, h3, re-labelled flag] of each wave's lane 63
    __shared__ float s_halo[2][4][4];
    __shared__ int s_best[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = tid & 3;                 // which quarter of its 16-position block this thread owns
    const int sub = tid >> 2;                 // block lane (0..63), like a w16 lane
    const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    const uint32_t B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    const uint32_t halfB = B >> 1;
    const uint32_t dq = L / T, dr = L % T;
    const float NINF = ninf();

    float sc[4] = {NINF, NINF, NINF, NINF};
    if (tid == 0) sc[0] = 0.0f;             // virtual state before frame 0 (align.py:57-58)
    uint32_t pres2 = tid == 0 ? 1u : 0u;    // bit 2k: cell k holds a live state
    bool pend_reset = false, reset_lane = false;

    int blk = sub;
    uint32_t blo = 0;
    gci32_t labx = (gci32_t)d.labx;
    int la[2];
    float vz[2];
    la[0] = labx[(size_t)blk * 8 + 2 * quad];
    la[1] = labx[(size_t)blk * 8 + 2 * quad + 1];
#pragma unroll
    for (int i = 0; i < 2; ++i) vz[i] = (ZL && la[i] == 0) ? NINF : __builtin_inff();

    uint32_t q = 0, rem = 0;
    uint32_t lo = 0, hi = B < L ? B : L;
    BandMasks4 mk;
    band_rebuild4(mk, lo, hi, wv);
    uint32_t band2 = pair_mask4((int)hi - (blk * 16 + 4 * quad)) & ~pair_mask4((int)lo - (blk * 16 + 4 * quad));

    const uint32_t lane_off = (lane < d.V ? (uint32_t)lane : 0u) * 4u;
    const char *lp = reinterpret_cast<const char *>(d.lp);
    const size_t ld = (size_t)d.ld * 4;
    float rows[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const uint32_t tt = (uint32_t)i < T ? (uint32_t)i : T - 1;
        rows[i] = row_load(lane_off, lp + (size_t)tt * ld);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) row_wait<0>(rows[i]);
    float e[2][2], e0[2];
    e[0][0] = bperm(la[0], rows[0]);
    e[0][1] = bperm(la[1], rows[0]);
    e0[0] = first_lane(rows[0]);
    float absum = __builtin_fabsf(rows[0]);   // NaN detector (every wave loads every row)

    const uint32_t *bp = reinterpret_cast<const uint32_t *>(d.bp);
    const uint32_t store_off = (uint32_t)tid * 4u;       // [t/4][block = tid>>2][t%4 = tid&3] dwords: thread tid keeps frame tid&3
    const int prev_wave = (int)((wv + 3u) & 3u);
    uint32_t keep = 0;
    uint32_t step_thr = dq != 0 ? 0u : T;   // floor(L*t/T) moves in this frame <=> rem + dr >= step_thr
    asm("" : "+s"(step_thr));
    const char *row_ahead = lp + (size_t)(D < T ? D : T - 1) * ld;   // row min(t+D, T-1) of the current frame t

    if (lane == 63) {
        s_halo[0][wv][0] = NINF; s_halo[0][wv][1] = NINF; s_halo[0][wv][2] = NINF; s_halo[0][wv][3] = 0.0f;
    }
    __syncthreads();

    for (uint32_t tb = 0; tb < T; tb += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const uint32_t t = tb + dd;
            if (t < T) {
                const int par = (int)(t & 1u);
                // Halos of frame t.  Left neighbours: previous lane (DPP), or for lane 0 the previous wave's lane 63 (LDS).
                // Then the reset of threads re-labelled in frame t-1, ahead of step A (as in forward_w16).
                float h1 = wave_ror1(sc[3]), h2 = wave_ror1(sc[2]), h3 = wave_ror1(sc[1]);
                {
                    const float l1 = s_halo[par][prev_wave][0], l2 = s_halo[par][prev_wave][1], l3 = s_halo[par][prev_wave][2];
                    h1 = lane == 0 ? l1 : h1;
                    h2 = lane == 0 ? l2 : h2;
                    h3 = lane == 0 ? l3 : h3;
                }
                if (__builtin_expect(pend_reset, 0)) {
                    asm volatile("" ::: "memory");
                    bool left_reset = __builtin_amdgcn_update_dpp(0, (int)reset_lane, 0x13C, 0xF, 0xF, false) != 0;
                    if (lane == 0) left_reset = s_halo[par][prev_wave][3] != 0.0f;
                    const bool kill = reset_lane && left_reset;
                    h1 = kill ? NINF : h1;
                    h2 = kill ? NINF : h2;
                    h3 = kill ? NINF : h3;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sc[k] = reset_lane ? NINF : sc[k];
                    pres2 = reset_lane ? 0u : pres2;
                    pend_reset = false;
                    reset_lane = false;
                }
                // A. band of frame t+1; re-label the threads whose block has been passed by lo - only in the frames
                // where floor(L*t/T) moves
                bool moved = false;
                rem += dr;
                if (__builtin_expect(rem >= step_thr, 0)) {
                    asm volatile("" ::: "memory");
                    q += dq;
                    if (rem >= T) { rem -= T; ++q; }
                    if (t + 1 != T) {   // no frame T: keep the last band and labels
                        moved = true;
                        const int32_t dlo = (int32_t)q - (int32_t)halfB;
                        const uint32_t nlo = (uint32_t)(dlo > 0 ? dlo : 0);
                        if ((nlo >> 4) != blo) {
                            blo = nlo >> 4;
                            const int nb = (int)blo + ((sub - (int)blo) & 63);
                            reset_lane = nb != blk;
                            if (nb != blk) {
                                blk = nb;
                                la[0] = labx[(size_t)blk * 8 + 2 * quad];
                                la[1] = labx[(size_t)blk * 8 + 2 * quad + 1];
                                asm volatile("" : "+v"(la[0]), "+v"(la[1]));
                            }
                            pend_reset = true;
                        }
                    }
                }
                // B. emissions of frame t+1
                {
                    if (dd < D - 1) row_wait<D - 1>(rows[(dd + 1) % D]); else row_wait<D - 2>(rows[(dd + 1) % D]);   // as in forward_w16
                    const float rn = rows[(dd + 1) % D];
                    absum += __builtin_fabsf(rn);
                    e[(dd + 1) & 1][0] = bperm(la[0], rn);
                    e[(dd + 1) & 1][1] = bperm(la[1], rn);
                    e0[(dd + 1) & 1] = first_lane(rn);
                }
                // C. frame t
                uint32_t word = 0;
                frame_cells4<M, ZL, 3>(sc, h1, h2, h3, e[dd & 1], vz, e0[dd & 1], mk, NINF, word);
                // hand this wave's top three scores (and whether lane 63 will be re-labelled) to the next wave
                if (lane == 63) {
                    s_halo[par ^ 1][wv][0] = sc[3];
                    s_halo[par ^ 1][wv][1] = sc[2];
                    s_halo[par ^ 1][wv][2] = sc[1];
                    s_halo[par ^ 1][wv][3] = (pend_reset && reset_lane) ? 1.0f : 0.0f;
                }
                {
                    pres2 = live_pairs(pres2, word, band2);
                }
                // 4 threads x 8 bits -> the block's dword (same layout as w16); thread k of the 4 keeps frame 4g+k
                uint32_t x = word << (8 * quad);
                x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
                x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
                keep = quad == dd ? x : keep;
                // prefetch the row of frame t+D (the last row again once there is none: never consumed)
                row_reload(rows[dd], lane_off, row_ahead);
                row_ahead += t + D + 1 < T ? ld : 0;
                // D. lane masks of frame t+1
                if (moved) {
                    const int32_t dlo = (int32_t)q - (int32_t)halfB;
                    const uint32_t nlo = (uint32_t)(dlo > 0 ? dlo : 0);
                    const uint32_t nhi = (L - nlo < B) ? L : nlo + B;
                    if (nlo != lo || nhi != hi) {
                        if ((nhi - hi) + (nlo - lo) <= 6u) {
                            for (uint32_t p = hi; p < nhi; ++p) band_toggle4(mk, p, wv);
                            for (uint32_t p = lo; p < nlo; ++p) band_toggle4(mk, p, wv);
                        } else {
                            band_rebuild4(mk, nlo, nhi, wv);
                        }
                        band2 = pair_mask4((int)nhi - (blk * 16 + 4 * quad)) & ~pair_mask4((int)nlo - (blk * 16 + 4 * quad));
                        lo = nlo;
                        hi = nhi;
                        if (ZL && pend_reset) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) vz[i] = la[i] == 0 ? NINF : __builtin_inff();
                        }
                    }
                }
                // one rendezvous per frame: LDS writes of this frame are visible before anyone reads them in the next
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        // the 256 threads store the 4 frames of the group as one contiguous KB (the last group may be
        // partial: the buffer is padded to whole groups)
        asm volatile("global_store_dword %0, %1, %2" : : "v"(store_off), "v"(keep), "s"(bp + (size_t)tb * 64) : "memory");
    }

    // Drain the row prefetches that are still in flight (the last D frames prefetch clamped rows that are
    // never consumed).  Their destination registers are dead to the compiler after the loop: without this
    // wait it reuses them for the reduction below and a late-landing load overwrites live values.
#pragma unroll
    for (int i = 0; i < D; ++i) row_wait<0>(rows[i]);

    // terminal state: the HIGHEST live position of frame T-1 (align.py:99-101)
    int best = -1;
    if (pres2) best = blk * 16 + 4 * quad + ((31 - __clz((int)pres2)) >> 1);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (__builtin_amdgcn_ballot_w64((__builtin_bit_cast(uint32_t, absum) & 0x7fffffffu) > 0x7f800000u)) best = -2;   // a NaN log-prob
    if (lane == 0) s_best[wv] = best;
    __syncthreads();
    best = s_best[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) best = s_best[w] > best ? s_best[w] : best;
    int32_t *m = meta_of(meta, d.idx);
    if (s_best[0] == -2) {   // (all four waves see the same rows)
        if (tid == 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusNaN);
        }
    } else if (best < 0) {
        if (tid == 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusEmptyBeam);
        }
    } else if ((best >> 2) == blk * 4 + quad) {
        float v = sc[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) v = (best & 3) == k ? sc[k] : v;
        m[1] = best;
        m[3] = __builtin_bit_cast(int32_t, v);
    }
}

template <int M, bool ZL>
__global__ __launch_bounds__(256) void forward_wg4_kernel(const Lattice *__restrict__ lats, int32_t *meta)
{
    const Lattice &d = lats[blockIdx.x];
    const int zl = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2] & 1);
    if ((zl != 0) != ZL) return;
    forward_wg4<M, ZL>(d, meta);
}

// ---------------------------------------------------------------------------------------
// backtrace (best_path), one wavefront per lattice; outputs are gathered by gather_outputs_kernel
//
// The walk is a scalar chain: word = v_readlane(chunk register, lane(p)) -> 2 bits -> p -= move.
// A path moves at most 3 positions per frame, so the frames of a chunk only ever touch the few
// blocks below the position known one chunk earlier: instead of whole 1-KB groups the kernel
// reads a window of kBtBlocks blocks (8 of 64 for 16-frame chunks: 8x less HBM traffic; the
// kernel is HBM- and scalar-unit-bound in batched runs).  The back-pointers are stored
// [t/4][block][t%4], so the window of 4 frames is contiguous (128 B), one dword per lane; the
// next chunk's window is prefetched while the current chunk is walked.
// ---------------------------------------------------------------------------------------
#ifndef KA_BT_CHUNK
#define KA_BT_CHUNK 16
#endif
constexpr int kBtChunk = KA_BT_CHUNK;                  // frames per chunk (16 or 32)
constexpr int kBtReach = 3 * 2 * kBtChunk;             // positions a path can drop over two chunks
constexpr int kBtBlocks = kBtChunk / 2;                // window width in blocks: 16*kBtBlocks >= kBtReach + 16 + 15
constexpr int kBtRegs = kBtChunk * kBtBlocks / 64;     // VGPRs per chunk: one dword per (frame, window block)
constexpr int kBtGroupsPerReg = 16 / kBtBlocks;        // 4-frame groups held by one VGPR
static_assert(kBtChunk == 16 || kBtChunk == 32, "lane layout below");
static_assert(16 * kBtBlocks >= kBtReach + 31, "window too narrow for the prefetch distance");

// first block of the window that covers every position the path can take in the chunk AFTER
// the one that is entered at position p_entry
__device__ __forceinline__ int bt_window(int p_entry)
{
    const int lo = p_entry - kBtReach;
    return (lo > 0 ? lo : 0) >> 4;
}
// Lane layout of a chunk register r[v]: lane = ((g*4 + fr) * kBtBlocks + j) holds the dword of
// frame 4*(v*kBtGroupsPerReg + g) + fr, window block j.
__device__ __forceinline__ uint32_t bt_lane_offset(int w0, int lane)
{
    const int j = lane % kBtBlocks, fr = (lane / kBtBlocks) & 3, g = lane / (4 * kBtBlocks);
    return (uint32_t)g * 1024u + (uint32_t)((w0 + j) & 63) * 16u + (uint32_t)fr * 4u;
}
// (possibly partial) chunk, compiler-tracked loads: used once per lattice for the tail chunk
__device__ __forceinline__ void bt_load_guarded(uint32_t (&r)[kBtRegs], const char *chunk_base, int n, int w0, int lane)
{
    const int fr = (lane / kBtBlocks) & 3, g = lane / (4 * kBtBlocks);
    const uint32_t voff = bt_lane_offset(w0, lane);
#pragma unroll
    for (int v = 0; v < kBtRegs; ++v) {
        const int f = 4 * (v * kBtGroupsPerReg + g) + fr;
        r[v] = f < n ? *(gcu32_t)(chunk_base + (size_t)v * kBtGroupsPerReg * 1024 + voff) : 0u;
    }
}
// full chunk, loads issued from inline asm (not tracked by hipcc: it would drain vmcnt(0) before
// the walk and serialise the prefetch); pair with bt_wait<N>()
__device__ __forceinline__ void bt_load_async(uint32_t (&r)[kBtRegs], const char *chunk_base /* uniform: group t0/4 */, int w0, int lane)
{
    const uint32_t voff = bt_lane_offset(w0, lane);
#pragma unroll
    for (int v = 0; v < kBtRegs; ++v)
        asm volatile("global_load_dword %0, %1, %2" : "=&v"(r[v]) : "v"(voff), "s"(chunk_base + (size_t)v * kBtGroupsPerReg * 1024) : "memory");
}
template <int N>
__device__ __forceinline__ void bt_wait(uint32_t (&r)[8])
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void bt_wait(uint32_t (&r)[2])
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "i"(N) : "memory");
}
// One chunk of the walk.  q = p - 16*w0 is the position relative to the window, so the dependent
// chain per frame is: q>>4 -> |lane group -> v_readlane -> >>2(q&15) -> 3&~code -> q -= move.
template <bool FULL>
__device__ __forceinline__ void bt_walk(const uint32_t (&r)[kBtRegs], int n, int w0, int &p, int &pathv)
{
    int q = p - 16 * w0;
    const int base = 16 * w0;
    uint32_t rr[kBtRegs];
#pragma unroll
    for (int i = 0; i < kBtRegs; ++i) rr[i] = blank_to_uniform(r[i]);
#pragma unroll
    for (int f = kBtChunk - 1; f >= 0; --f) {
        if (FULL || f < n) {
            const int grp = f >> 2;
            const int lane_base = ((grp % kBtGroupsPerReg) * 4 + (f & 3)) * kBtBlocks;
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)rr[grp / kBtGroupsPerReg], (q >> 4) | lane_base);
            asm("v_writelane_b32 %0, %1, %2" : "+v"(pathv) : "s"(q), "i"(f));  // pathv[lane f] = position - base
            q -= bp_decode(w >> ((q * 2) & 31));
        }
    }
    pathv += base;
    p = q + base;
}

__global__ __launch_bounds__(64) void backtrace_w16_kernel(const Lattice *__restrict__ lats, const int32_t *meta, int only_flagged)
{
    const Lattice &d = lats[blockIdx.x];
    const int lane = threadIdx.x;
    if (only_flagged && !(__builtin_amdgcn_readfirstlane(meta[4 * (size_t)d.idx + 2]) & kFlagExact)) return;
    int p = __builtin_amdgcn_readfirstlane(meta[4 * (size_t)d.idx + 1]);
    if (p < 0) return;  // empty beam: status already set by the forward kernel
    const char *bp = reinterpret_cast<const char *>(d.bp);
    gi32_t path = (gi32_t)d.path;
    const int T = __builtin_amdgcn_readfirstlane(d.T);
    uint32_t cur[kBtRegs], nxt[kBtRegs];
    // tail chunk [t0, T): 1..kBtChunk frames, so that every chunk below it is a full one
    int t0 = ((T - 1) / kBtChunk) * kBtChunk;
    int w = bt_window(p + 3 * kBtChunk);   // the tail chunk is entered at the end position itself
    {
        const int n = T - t0;
        bt_load_guarded(cur, bp + (size_t)t0 * 256, n, w, lane);
        const int t1 = t0 > 0 ? t0 - kBtChunk : 0;
        const int wn = bt_window(p);
        bt_load_async(nxt, bp + (size_t)t1 * 256, wn, lane);   // (re-reads chunk 0 when there is no next chunk)
        int pathv = 0;
        bt_walk<false>(cur, n, w, p, pathv);
        if (lane < n) path[t0 + lane] = pathv;
        bt_wait<1>(nxt);                    // younger than the loads: the path store
#pragma unroll
        for (int i = 0; i < kBtRegs; ++i) cur[i] = nxt[i];
        w = wn;
    }
    // full chunks.  Straight-line per iteration: the registers of the in-flight loads (nxt) are
    // not touched by anything between their issue and bt_wait, and never cross the back-edge.
    while (t0 > 0) {
        t0 -= kBtChunk;                     // chunk [t0, t0+kBtChunk) is in cur, window w, entered at p
        const int t1 = t0 > 0 ? t0 - kBtChunk : 0;
        const int wn = bt_window(p);
        bt_load_async(nxt, bp + (size_t)t1 * 256, wn, lane);
        int pathv = 0;
        bt_walk<true>(cur, kBtChunk, w, p, pathv);
        if (lane < kBtChunk) path[t0 + lane] = pathv;
        bt_wait<1>(nxt);
#pragma unroll
        for (int i = 0; i < kBtRegs; ++i) cur[i] = nxt[i];
        w = wn;
    }
}

// ---------------------------------------------------------------------------------------
// checkpointed path, second kernel: back-pointers recomputed around the path, then walked.
//
// forward_ck_kernel left the score ring of every kCkFrames-th frame in HBM.  Going backwards
// chunk by chunk (chunk = the 32 frames after a checkpoint), the path is known at the chunk's
// last frame (position p); it drops at most 3 positions per frame, and a cell only depends on
// positions below it, so the back-pointers the walk will read all lie in the 97 positions below
// p, and they are exact if the chunk is recomputed forward from the checkpoint on the window
// [p-96, p+31]: whatever is wrong at the window's low edge (unknown neighbours) climbs 3 positions
// per frame, exactly as fast as the path can fall.  64 lanes x (one blank + one label cell); same
// float operations in the same order as the forward kernel, so the scores are bit-identical.
// The kernel also writes best_labels / best_scores: the chunk's log-prob rows are in registers
// (lane v = lp[t, v]), so a score is one v_readlane - no second pass over the log-probs.
// ---------------------------------------------------------------------------------------
// 8 frames x 4 bits (blank hi, blank lo, label hi, label lo): blank nibbles (e0, e1) -> (e0|e1, e0)
__device__ __forceinline__ uint32_t rc_blank_to_uniform(uint32_t x)
{
    const uint32_t h = (x >> 3) & 0x11111111u, l = (x >> 2) & 0x11111111u;
    return (x & 0x33333333u) | ((h | l) << 3) | (h << 2);
}
// lanes [first, first+count) of a 64-bit mask, count and first in 0..63: one s_bfm_b64
__device__ __forceinline__ uint64_t lane_field(uint32_t count, uint32_t first)
{
    uint64_t m;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(count), "s"(first));
    return m;
}
// One frame of the window recurrence for max_move = 4, written out so that it costs 21 vector instructions (23 where the
// window touches a band edge) instead of the 29 hipcc makes of the cell_blank / cell_label formulation:
//   * the label cell's two operands from the lane below (label pb-1, blank pb-2) enter their adds through DPP
//     (v_add_f32_dpp .. wave_ror:1) instead of through a v_mov_dpp each;
//   * the blank cell's candidates from the lane below are rotations of ONE sum: score(pb-1) + e0 = ror(sl + e0) and
//     score(pb-3) + e0 = ror of that again - e0 is the same in every lane, so this is the same float add on the same
//     operands, carried out in another lane.  (DPP takes no scalar operand.  Fetching e0 into a VGPR
//     with a second ds_bpermute instead, so that all three lane-below operands go through DPP adds: two vector
//     instructions fewer, and slower - the LDS pipe is shared by the CU's four SIMDs);
//   * the label emission is gathered one frame ahead and waited for at the end of the block;
//   * `open` (the cells the walk can reach lie inside the band at every frame of the chunk, the usual case): the band
//     select is branched over on the scalar unit - the maxima stay in sb / sl as they are.
// Software hazards the assembler does not see to (DPP reads a VGPR written by the previous VALU instruction: 2 wait
// states): sb / sl are last written by the selects, with the s_waitcnt and the next block's gather, v_readfirstlane
// and first add between them and the next DPP read; inside the block every DPP source is written at least three
// instructions earlier (an s_nop fills in where the veto is compiled out).
// `word` takes 4 code bits per frame exactly as cell_blank<4> / cell_label<4> would shift them in.
// (The timing experiments of DESIGN.md 4.7 - builds of this block without its loads, gathers, codes or walk, results wrong by
//  design - were preprocessor variants of this code up to round 2, commit d171691; they are not part of the shipped source.)
// the blank emission as a scalar (v_readfirstlane): one LDS-pipe instruction less per frame - that pipe is shared by
// the CU's four SIMDs and a ds_bpermute holds it for 7 cycles (tools/ubench/lds_rates.hip) - for one more DPP move:
// DPP takes no scalar operand, so score(pb-1) + e0 is formed as ror(sl + e0).
#define KA_RC_E0_OUT "=&s"
#define KA_RC_E0_IN "s"
#define KA_RC_GATHER "ds_bpermute_b32 %[eln], %[lab4], %[rown]\n\tv_readfirstlane_b32 %[e0n], %[rown]\n\t"
#define KA_RC_HEAD                                                                        \
    KA_RC_GATHER                                                                          \
    "v_add_f32 %[t2], %[e0], %[sl]\n\t"                                                   \
    "v_add_f32_dpp %[t4], %[sl], %[el] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_add_f32_dpp %[t5], %[sb], %[el] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_add_f32 %[t0], %[e0], %[sb]\n\t"                                                   \
    "v_add_f32 %[t3], %[sb], %[el]\n\t"                                                   \
    "v_add_f32 %[t7], %[sl], %[el]\n\t"                                                   \
    "v_mov_b32_dpp %[t1], %[t2] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"                \
    "v_max_f32 %[t6], %[t7], %[t3]\n\t"
#define KA_RC_VETO "v_min_f32 %[t4], %[t4], %[veto]\n\t"
// (two instructions between the write of t1 and its DPP read)
#define KA_RC_MAX                                                                         \
    "v_max3_f32 %[sl], %[t6], %[t4], %[t5]\n\t"                                           \
    "s_nop 0\n\t"                                                                         \
    "v_mov_b32_dpp %[t2], %[t1] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"                \
    "v_max3_f32 %[sb], %[t0], %[t1], %[t2]\n\t"
#define KA_RC_COMBINE                                                                     \
    "s_or_b64 %[sx], %[sa], %[sb2]\n\t"                                                   \
    "v_addc_co_u32 %[w], %[sy], %[w], %[w], %[sx]\n\t"                                    \
    "s_andn2_b64 %[sc], %[sc], %[sb2]\n\t"                                                \
    "s_or_b64 %[sc], %[sc], %[sa]\n\t"                                                    \
    "v_addc_co_u32 %[w], %[sy], %[w], %[w], %[sc]\n\t"
#define KA_RC_CODES                                                                       \
    "v_cmp_eq_f32 %[sa], %[t0], %[sb]\n\t"                                                \
    "v_addc_co_u32 %[w], %[sy], %[w], %[w], %[sa]\n\t"                                    \
    "v_cmp_eq_f32 %[sa], %[t1], %[sb]\n\t"                                                \
    "v_addc_co_u32 %[w], %[sy], %[w], %[w], %[sa]\n\t"                                    \
    "v_cmp_eq_f32 %[sa], %[t7], %[sl]\n\t"                                                \
    "v_cmp_eq_f32 %[sb2], %[t3], %[sl]\n\t"                                               \
    "v_cmp_eq_f32 %[sc], %[t4], %[sl]\n\t"                                                \
    KA_RC_COMBINE
// band select, skipped by a scalar branch inside the block when the chunk is open (a branch around two asm blocks made
// hipcc allocate the loop-carried registers differently on the two sides and reconcile them with three v_mov per frame)
#define KA_RC_BAND                                                                        \
    "s_bitcmp1_b32 %[open], 0\n\t"                                                        \
    "s_cbranch_scc1 .Lka_rc_open_%=\n\t"                                                  \
    "v_cndmask_b32 %[sb], %[ninf], %[sb], %[mb]\n\t"                                      \
    "v_cndmask_b32 %[sl], %[ninf], %[sl], %[ml]\n"                                        \
    ".Lka_rc_open_%=:\n\t"                                                                \
    "s_waitcnt lgkmcnt(0)"
#define KA_RC_OUTS                                                                                                              \
    [sb] "+v"(sb), [sl] "+v"(sl), [w] "+v"(word), [eln] "=&v"(el_next), [e0n] KA_RC_E0_OUT(e0_next), [t0] "=&v"(t0), [t1] "=&v"(t1),   \
    [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [sa] "=&s"(sa),             \
    [sb2] "=&s"(sb2), [sc] "=&s"(sc), [sx] "=&s"(sx), [sy] "=&s"(sy)
#define KA_RC_INS                                                                                                               \
    [e0] KA_RC_E0_IN(e0), [el] "v"(el), [rown] "v"(row_next), [lab4] "v"(lab4), [zero] "v"(zero), [open] "s"(open), [mb] "s"(mask_b),   \
    [ml] "s"(mask_l), [ninf] "v"(NINF)
template <bool ZL>
__device__ __forceinline__ void rc_frame4(float &sb, float &sl, uint32_t &word, float e0, float el, float &e0_next, float &el_next,
                                          float row_next, int lab4, int zero, float veto, uint32_t open, uint64_t mask_b,
                                          uint64_t mask_l, float NINF)
{
    float t0, t1, t2, t3, t4, t5, t6, t7;
    uint64_t sa, sb2, sc, sx, sy;
    if constexpr (ZL)
        asm volatile(KA_RC_HEAD KA_RC_VETO KA_RC_MAX KA_RC_CODES KA_RC_BAND : KA_RC_OUTS : KA_RC_INS, [veto] "v"(veto) : "memory", "scc");
    else
        asm volatile(KA_RC_HEAD KA_RC_MAX KA_RC_CODES KA_RC_BAND : KA_RC_OUTS : KA_RC_INS : "memory", "scc");
}
#undef KA_RC_HEAD
#undef KA_RC_GATHER
#undef KA_RC_E0_OUT
#undef KA_RC_E0_IN
#undef KA_RC_VETO
#undef KA_RC_MAX
#undef KA_RC_CODES
#undef KA_RC_COMBINE
#undef KA_RC_BAND
#undef KA_RC_OUTS
#undef KA_RC_INS
constexpr int kRcLanes = 62;   // lanes 62 and 63 are kept at -inf: they are the "nothing below position 0" that
                               // wave_ror hands to lanes 0 and 1 (window width 124 >= 97 + slack)

// Chunks of all lattices of a launch are numbered consecutively (Lattice::chunk0 = a lattice's first): which lattice
// does chunk `g` belong to?  (wave-uniform binary search over the descriptors)
__device__ __forceinline__ int lattice_of_chunk(const Lattice *__restrict__ lats, int n, int64_t g)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (lats[mid].chunk0 <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ int chunks_of(int T) { return (T - 1) / kCkFrames + 1; }

// The walk over one chunk's codes (4 bits per lane and frame, frame f in nibble 7 - f%8 of codes[f/8]; per nibble
// [blank hi, blank lo, label hi, label lo], every cell coded 3 - move): pathv[lane f] = position of frame f relative to
// the window.  One scalar chain per frame: v_readlane -> shift -> 3 & ~code -> subtract.  FULL: all 32 frames, straight
// line (the `f < n` test of the last, partial chunk costs a compare and a taken branch per frame).
template <bool FULL>
__device__ __forceinline__ void rc_walk(const uint32_t (&codes)[kCkFrames / 8], int n, int &qq, int &pathv)
{
#pragma unroll
    for (int f = kCkFrames - 1; f >= 0; --f) {
        if (FULL || f < n) {
            const int sh = 4 * (7 - (f & 7)) + ((qq & 1) ? 0 : 2);
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)codes[f >> 3], qq >> 1) >> sh;
            asm("v_writelane_b32 %0, %1, %2" : "+v"(pathv) : "s"(qq), "i"(f));
            qq -= bp_decode(w);
        }
    }
}
// The same walk, also collecting best_labels / best_scores (align.py:105-107) on the way: the label of the path's cell
// comes out of the window's label register (lane j: 4 * label of position wlo+2j+1; a blank cell: label 0) and the score
// out of the frame's row register (lane v = lp[t0+f, v]), both with v_readlane at a scalar lane index - two vector
// instructions more per frame than gathering the scores afterwards with a ds_bpermute per frame, but the LDS pipe, which
// the CU's four SIMDs share and the frame loop's emission gather needs, is left alone.
template <bool FULL>
__device__ __forceinline__ void rc_walk_out(const uint32_t (&codes)[kCkFrames / 8], const float (&rows)[kCkFrames], int lab4, int n,
                                            int &qq, int &pathv, int &labv, float &scv)
{
#pragma unroll
    for (int f = kCkFrames - 1; f >= 0; --f) {
        if (FULL || f < n) {
            const int j = qq >> 1;
            const int sh = 4 * (7 - (f & 7)) + ((qq & 1) ? 0 : 2);
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)codes[f >> 3], j) >> sh;
            const int col = (qq & 1) ? __builtin_amdgcn_readlane(lab4, j) >> 2 : 0;
            const int sc = __builtin_amdgcn_readlane(__builtin_bit_cast(int, rows[f]), col);
            asm("v_writelane_b32 %0, %1, %2" : "+v"(pathv) : "s"(qq), "i"(f));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(labv) : "s"(col), "i"(f));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(scv) : "s"(sc), "i"(f));
            qq -= bp_decode(w);
        }
    }
}

// PAR = false: one wavefront per lattice walks its chunks from the last to the first (the position a chunk is entered
// at comes out of the chunk above it).  PAR = true: one wavefront per CHUNK, entered at Lattice::entry[chunk], which the
// chunk-parallel backtrace (ka_parallel_bt.hpp) has worked out for every chunk beforehand; grid = all chunks of the launch.
// GO ("gather the outputs"): the walk collects the path only, and best_labels / best_scores are fetched afterwards, lane f
// doing frame t0+f - the label with one ds_bpermute on the window's label register, the score with ONE 4-byte load per frame
// from the row this wavefront read a few microseconds ago (L2 / Infinity Cache).  Four vector instructions per frame fewer
// (two v_readlane + two v_writelane of the walk) for one gather per chunk whose latency other wavefronts cover.  Opt-in
// (ka_debug_set_rc_gather): 27.2 -> 26.1 ms for 8192 lattices alone on the GPU, nothing with several launches in flight, and
// the gathers read 47 GB more per step by the counters.  GO = false, the default, keeps everything in registers.
template <int M, bool ZL, bool PAR, bool GO = false>
__global__ __launch_bounds__(64, KA_RC_MIN_WAVES) void backtrace_rc_kernel(const Lattice *__restrict__ lats, const int32_t *meta, int n_lats)
{
    const int which = PAR ? __builtin_amdgcn_readfirstlane(lattice_of_chunk(lats, n_lats, (int64_t)blockIdx.x)) : (int)blockIdx.x;
    const Lattice &d = lats[which];
    const int lane = threadIdx.x;
    if (!PAR && __builtin_amdgcn_readfirstlane(d.par)) return;   // the chunk-parallel kernels walk this one (a mixed launch runs both)
    const int32_t *mt = meta + 4 * (size_t)d.idx;
    const int flags = __builtin_amdgcn_readfirstlane(mt[2]);
    if (flags & (kFlagExact | kFlagDeclined)) return;        // handled by the exact kernels / not at all
    if (((flags & kFlagZeroLabel) != 0) != ZL) return;       // the other instance's lattice (as in the forward kernels)
    if (__builtin_amdgcn_readfirstlane(mt[0]) != kStatusOk) return;   // rejected (bad label, NaN, empty beam): no path
    int p = __builtin_amdgcn_readfirstlane(mt[1]);
    if (p < 0) return;  // empty beam: status already set by the forward kernel
    const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    const uint32_t B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    const uint32_t halfB = B >> 1;
    const uint32_t dq = L / T, dr = L % T;
    const float NINF = ninf();
    const char *lp = reinterpret_cast<const char *>(d.lp);
    const size_t ldb = (size_t)d.ld * 4;                     // row pitch in bytes
    const uint32_t col_off = (lane < d.V ? (uint32_t)lane : 0u) * 4u;
    gci32_t labx = (gci32_t)d.labx;
    const char *ck = reinterpret_cast<const char *>(d.bp);
    const uint32_t ck_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.ck_mask);
    const size_t ck_pitch = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(d.ck_pitch);
    gi32_t path = (gi32_t)d.path;
    gi32_t lab_out = (gi32_t)d.lab_out;
    gf32_t sc_out = (gf32_t)d.sc_out;

    // floor(L*t/T) and remainder at the start of the last chunk; one chunk back = minus (32*L)/T, (32*L)%T
    uint32_t t0 = ((T - 1) / kCkFrames) * kCkFrames;
    if constexpr (PAR) {
        const uint32_t c = (uint32_t)((int64_t)blockIdx.x - d.chunk0);
        t0 = c * kCkFrames;
        p = __builtin_amdgcn_readfirstlane(d.entry[c]);
    }
    // (64-bit divisions run on the vector unit: tell the compiler the results are wave-uniform)
    const auto uni = [](uint64_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); };
    uint32_t q0 = uni(((uint64_t)L * t0) / T), r0 = uni(((uint64_t)L * t0) % T);
    const uint32_t D32 = uni(((uint64_t)L * kCkFrames) / T), R32 = uni(((uint64_t)L * kCkFrames) % T);
    const float inv_T = 1.0f / (float)T;

    for (;;) {
        const int n = (int)(T - t0 < (uint32_t)kCkFrames ? T - t0 : (uint32_t)kCkFrames);
        const int wlo = __builtin_amdgcn_readfirstlane((p > 96 ? p - 96 : 0) & ~1);
        const int pb = wlo + 2 * lane;                       // this lane's blank position; its label position is pb+1
        // All loads of the chunk are issued here, from inline asm (uniform base + lane offset, and invisible
        // to hipcc, which would otherwise drain them all at the first use): labels, checkpoint, 32 rows.
        // Each is released by a counted wait: at least the loads issued after it are still behind it.
        int lab4;                                            // 4 * label of position pb+1 (zero padded past S)
        asm volatile("global_load_dword %0, %1, %2" : "=v"(lab4) : "v"((uint32_t)lane * 4u), "s"(labx + (wlo >> 1)) : "memory");
        f32x2 ckv = {NINF, NINF};                            // scores of (pb, pb+1) after frame t0-1
        if (t0 != 0) {
            const uint32_t off = ((uint32_t)pb & ck_mask) * 4u;   // (pb is even: both cells of the lane lie in one row slot pair)
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(ckv) : "v"(off), "s"(ck + ((size_t)(t0 / kCkFrames) - 1) * ck_pitch) : "memory");
        }
        float rows[kCkFrames];                               // (frames past T-1, last chunk only, repeat row T-1: never walked)
        {
            const char *rp = lp + (size_t)t0 * ldb;
            if (n == kCkFrames) {
#pragma unroll
                for (int f = 0; f < kCkFrames; ++f) {
                    rows[f] = row_load(col_off, rp);
                    rp += ldb;
                }
            } else {
#pragma unroll
                for (int f = 0; f < kCkFrames; ++f) {
                    rows[f] = row_load(col_off, rp);
                    rp += (f + 1 < n) ? ldb : 0;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(lab4), "+v"(ckv) : "i"(kCkFrames) : "memory");
        const float veto = lab4 == 0 ? NINF : __builtin_inff();
        // scores after frame t0-1, limited to that frame's band
        float sb, sl;
        if (t0 == 0) {
            sb = pb == 0 ? 0.0f : NINF;   // virtual state before frame 0 (align.py:57-58)
            sl = NINF;
        } else {
            const uint32_t qm = r0 >= dr ? q0 - dq : q0 - dq - 1;
            const int32_t dl = (int32_t)qm - (int32_t)halfB;
            const uint32_t lo1 = (uint32_t)(dl > 0 ? dl : 0);
            const uint32_t hi1 = (L - lo1 < B) ? L : lo1 + B;
            sb = ((uint32_t)pb >= lo1 && (uint32_t)pb < hi1 && lane < kRcLanes) ? ckv.x : NINF;
            sl = ((uint32_t)pb + 1 >= lo1 && (uint32_t)pb + 1 < hi1 && lane < kRcLanes) ? ckv.y : NINF;
        }
        // ---- forward over the chunk: scores + back-pointer codes of the window ----
        uint32_t codes[kCkFrames / 8] = {};
        // floor(L*(t0+f)/T) of all frames of the chunk at once, lane f <-> frame f (r0 + 33*dr < 34*T: the host
        // keeps T below 2^26 in this form), and the frames after which it moves as a bit mask: the common frame
        // then pays one s_bitcmp1 + s_cbranch for the band instead of a scalar Bresenham step
        // (x / T for x < 35*T < 2^32 by a float estimate and one correction each way: 7 vector instructions instead of
        //  the ~20 of a general 32-bit division, twice per chunk; lanes above 33 may wrap - their results are not used)
        const auto div_T = [&](uint32_t x) {
            uint32_t qe = (uint32_t)((float)x * inv_T);
            qe -= (qe * T > x) ? 1u : 0u;
            qe += (x - qe * T >= T) ? 1u : 0u;
            return qe;
        };
        const uint32_t qnum = r0 + (uint32_t)lane * dr;
        const uint32_t qa = q0 + (uint32_t)lane * dq + div_T(qnum);
        // band of frame t0+f, relative to the window and clamped to the 2*kRcLanes cells that are computed, as two lane
        // masks: blank wlo+2l in band <=> l in [ceil(x/2), ceil(y/2));  label wlo+2l+1 <=> l in [floor(x/2), floor(y/2))
        uint64_t mask_b = 0, mask_l = 0;
        const auto band_masks = [&](int f) {
            const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)qa, f);
            const int32_t dl = (int32_t)q - (int32_t)halfB;
            const int32_t lo = dl > 0 ? dl : 0;
            const int32_t hi = (L - (uint32_t)lo < B) ? (int32_t)L : lo + (int32_t)B;
            int32_t x = lo - wlo, y = hi - wlo;
            x = x < 0 ? 0 : x;
            y = y < 0 ? 0 : y;
            asm("" : "+s"(x), "+s"(y));   // (keeps the clamp on the scalar unit: no v_med3)
            x = x > 2 * kRcLanes ? 2 * kRcLanes : x;
            y = y > 2 * kRcLanes ? 2 * kRcLanes : y;
            const uint32_t xb = (uint32_t)(x + 1) >> 1, yb = (uint32_t)(y + 1) >> 1, xl = (uint32_t)x >> 1, yl = (uint32_t)y >> 1;
            mask_b = lane_field(yb - xb, xb);
            mask_l = lane_field(yl - xl, xl);
        };
        if constexpr (M == 4) {
            // Does the band cut into the cells the walk can reach, [wlo, p], at any frame t0-1 .. t0+31?  The band only
            // moves up: it does not iff its low edge at the last frame is at or below wlo and its high edge at frame t0-1
            // is above p.  Then nothing is masked (rc_frame4<.., false>): cells of the window above p may lie outside the
            // band and hold anything, as may the ring slots they were loaded from - nothing the walk reads depends on a
            // cell above itself.  (Same argument as for the window's low edge, see the head of this section.)
            bool open = false;
            if (t0 != 0 && n == kCkFrames) {
                const uint32_t q_last = (uint32_t)__builtin_amdgcn_readlane((int)qa, kCkFrames - 1);
                const int32_t dl_last = (int32_t)q_last - (int32_t)halfB;
                const uint32_t qm = r0 >= dr ? q0 - dq : q0 - dq - 1;
                const int32_t dl = (int32_t)qm - (int32_t)halfB;
                const uint32_t lo1 = (uint32_t)(dl > 0 ? dl : 0);
                const uint32_t hi1 = (L - lo1 < B) ? L : lo1 + B;
                // (wlo == 0: the cells below the window do not exist and count as -inf, which only the masked form's idle
                //  lanes 62 and 63 hand to lanes 0 and 1)
                open = wlo > 0 && dl_last <= wlo && (uint32_t)p < hi1;
            }
            const int zero = 0;
            float e0c, elc, e0n, eln;
            row_wait_n(rows[0], kCkFrames - 1);
            asm volatile("ds_bpermute_b32 %0, %2, %4\n\t"
                         "v_readfirstlane_b32 %1, %4\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(elc), "=&s"(e0c) : "v"(lab4), "v"(zero), "v"(rows[0]) : "memory");
            // the frames after which floor(L*t/T) moves, as a bit mask: the common frame then pays one s_bitcmp1 +
            // s_cbranch for the band instead of a scalar Bresenham step
            uint32_t moves = 0;
            const uint32_t open_bit = (uint32_t)__builtin_amdgcn_readfirstlane(open ? 1 : 0);
            if (open_bit) {
                sb = ckv.x;
                sl = ckv.y;
            } else {
                const uint32_t qb = q0 + (uint32_t)(lane + 1) * dq + div_T(qnum + dr);
                moves = (uint32_t)__builtin_amdgcn_ballot_w64(qa != qb) << 1 | 1u;   // bit f: frame f's band differs from frame f-1's (bit 0: set it up)
            }
            asm volatile("s_nop 1" : "+v"(sb), "+v"(sl));   // (a DPP read follows within the next block)
            // ONE loop for both kinds of chunk: the row registers are waited for in one place per frame whichever way the
            // chunk goes (two loops made hipcc copy rows whose loads were in flight)
#pragma unroll
            for (int f = 0; f < kCkFrames; ++f) {
                if ((moves >> f) & 1u) band_masks(f);
                if (f + 1 < kCkFrames) row_wait_n(rows[f + 1], kCkFrames - 2 - f);
                rc_frame4<ZL>(sb, sl, codes[f >> 3], e0c, elc, e0n, eln, rows[f + 1 < kCkFrames ? f + 1 : f], lab4, zero, veto,
                              open_bit, mask_b, mask_l, NINF);
                e0c = e0n;
                elc = eln;
            }
        } else {
            const uint32_t qb = q0 + (uint32_t)(lane + 1) * dq + div_T(qnum + dr);
            const uint32_t moves = (uint32_t)__builtin_amdgcn_ballot_w64(qa != qb);
#pragma unroll
            for (int f = 0; f < kCkFrames; ++f) {
                if (f == 0 || ((moves >> (f > 0 ? f - 1 : 0)) & 1u)) band_masks(f);
                row_wait_n(rows[f], kCkFrames - 1 - f);
                const float el = bperm(lab4, rows[f]);
                const float e0 = first_lane(rows[f]);
                const float L1 = wave_ror1(sl);    // score of pb-1 (lane 0: lane 63's, always -inf)
                const float B1 = wave_ror1(sb);    // score of pb-2
                const float L2 = wave_ror1(L1);    // score of pb-3 (lanes 0, 1: lanes 62, 63's, always -inf)
                uint32_t &word = codes[f >> 3];
                if ((f & 7) == 0) word = 0;
                float mb, ml;
                cell_blank<M>(sb, L1, L2, e0, mb, word);
                cell_label<M, ZL>(sl, sb, L1, B1, el, veto, ml, word);
                sb = select_by_mask(NINF, mb, mask_b);
                sl = select_by_mask(NINF, ml, mask_l);
            }
        }
        // ---- walk back over the chunk: pathv[lane f] = position of frame t0+f, relative to wlo ----
#pragma unroll
        for (int g = 0; g < kCkFrames / 8; ++g) codes[g] = rc_blank_to_uniform(codes[g]);
        int pathv = 0;
        int qq = p - wlo;
        // best_path, best_labels = lab'[best_path], best_scores[t] = lp[t, best_labels[t]] (align.py:105-107), lane f
        // does frame t0+f: collected by the walk itself (rc_walk_out)
        if constexpr (GO) {
            if (n == kCkFrames)
                rc_walk<true>(codes, n, qq, pathv);
            else
                rc_walk<false>(codes, n, qq, pathv);
            // lane f: position wlo + pathv; its label sits in lane (pathv >> 1) of the window's label register (odd positions)
            const int lw = __builtin_amdgcn_ds_bpermute((pathv >> 1) * 4, lab4);
            if (lane < n) {
                const int pos = pathv + wlo;
                const int l4 = (pos & 1) ? lw : 0;
                const float sv = *(gcf32_t)(lp + (size_t)(t0 + (uint32_t)lane) * ldb + (uint32_t)l4);
                path[t0 + lane] = pos;
                lab_out[t0 + lane] = l4 >> 2;
                sc_out[t0 + lane] = sv;
            }
        } else {
            int labv = 0;
            float scv = 0.0f;
            if (n == kCkFrames)
                rc_walk_out<true>(codes, rows, lab4, n, qq, pathv, labv, scv);
            else
                rc_walk_out<false>(codes, rows, lab4, n, qq, pathv, labv, scv);
            if (lane < n) {
                path[t0 + lane] = pathv + wlo;
                lab_out[t0 + lane] = labv;
                sc_out[t0 + lane] = scv;
            }
        }
        p = qq + wlo;
        if (PAR || t0 == 0) break;
        t0 -= kCkFrames;
        q0 -= D32;
        if (r0 < R32) { r0 += T; q0 -= 1; }
        r0 -= R32;
    }
}

// best_labels = lab'[best_path], best_scores[t] = lp[t, best_labels[t]]  (align.py:105-107)
// grid: x = 1024-frame slices (a block strides over them), y = lattice.  only_flagged: behind the
// checkpointed kernels (which write these outputs themselves), for the lattices they declined.
__global__ __launch_bounds__(256) void gather_outputs_kernel(const Lattice *__restrict__ lats, const int32_t *meta, int only_flagged)
{
    const Lattice &d = lats[blockIdx.y];
    if (meta[4 * (size_t)d.idx + 1] < 0) return;
    if (only_flagged && !(meta[4 * (size_t)d.idx + 2] & kFlagExact)) return;
    const int T = d.T;
    gci32_t path = (gci32_t)d.path;
    gci32_t labx = (gci32_t)d.labx;
    gcf32_t lp = (gcf32_t)d.lp;
    const size_t ld = (size_t)d.ld;
    for (int base = blockIdx.x * 1024; base < T; base += gridDim.x * 1024) {
        // three dependent loads per frame (position -> label -> score): keep all four frames of a thread in
        // flight at each stage before anything is stored
        int t[4], pp[4], lab[4];
        float sc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t[i] = base + i * 256 + threadIdx.x;
            pp[i] = t[i] < T ? path[t[i]] : 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) lab[i] = (pp[i] & 1) ? (labx[pp[i] >> 1] >> 2) : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = t[i] < T ? lp[(size_t)t[i] * ld + lab[i]] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (t[i] < T) {
                ((gi32_t)d.lab_out)[t[i]] = lab[i];
                ((gf32_t)d.sc_out)[t[i]] = sc[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// generic path: any beam width, any V, max_move <= 255.  One 256-thread workgroup per
// lattice, score columns double-buffered in global memory (L2-resident), one byte of
// back-pointer per band cell.  Correctness path for argument ranges the w16 layout does
// not cover (beam > 1009 on a longer transcript, V > 64, max_move > 4); not tuned.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void forward_generic_kernel(const Lattice *__restrict__ lats, int32_t *meta)
{
    const Lattice &d = lats[blockIdx.x];
    const int tid = threadIdx.x;
    const int64_t T = d.T, L = d.L, B = d.beam;
    const int M = d.max_move;
    const int64_t W = d.W;
    float *scA = d.col, *scB = d.col + L;
    uint8_t *prA = reinterpret_cast<uint8_t *>(d.col + 2 * L), *prB = prA + L;
    uint8_t *bp = reinterpret_cast<uint8_t *>(d.bp);
    for (int64_t p = tid; p < L; p += 256) { prA[p] = 0; prB[p] = 0; }
    __syncthreads();
    if (tid == 0) { scA[0] = 0.0f; prA[0] = 1; }
    __syncthreads();
    int64_t plo = 0, phi = 1;
    for (int64_t t = 0; t < T; ++t) {
        int64_t lo = (L * t) / T - B / 2;  // host guarantees L, T < 2^31
        lo = lo < 0 ? 0 : lo;
        const int64_t hi = (L - lo < B) ? L : lo + B;
        const float *row = d.lp + (size_t)t * (size_t)d.ld;
        for (int64_t p = lo + tid; p < hi; p += 256) {
            const int lab = (p & 1) ? (d.labx[p >> 1] >> 2) : 0;
            const float e = row[lab];
            float best = ninf();
            int bj = 0;
            for (int j = 0; j < M; ++j) {
                const int64_t u = p - j;
                if (u < 0) break;
                const bool pres = (u >= plo && u < phi) ? prA[u] != 0 : false;
                float c = pres ? scA[u] + e : ninf();
                if (j > 0 && (j & 1) == 0 && lab == 0) c = ninf();
                if (j == 0 || c > best) { best = c; bj = j; }
            }
            const int64_t ub = p - bj;
            prB[p] = (ub >= plo && ub < phi) ? prA[ub] : 0;
            scB[p] = best;
            bp[(size_t)t * (size_t)W + (size_t)(p - lo)] = (uint8_t)bj;
        }
        __syncthreads();
        { float *x = scA; scA = scB; scB = x; }
        { uint8_t *x = prA; prA = prB; prB = x; }
        plo = lo;
        phi = hi;
    }
    // highest live position of the last frame
    __shared__ int64_t s_best;
    if (tid == 0) s_best = -1;
    __syncthreads();
    int64_t mine = -1;
    for (int64_t p = plo + tid; p < phi; p += 256)
        if (prA[p]) mine = p;
    if (mine >= 0) atomicMax((long long *)&s_best, (long long)mine);
    __syncthreads();
    if (tid == 0) {
        int32_t *m = meta_of(meta, d.idx);
        if (s_best < 0) {
            m[1] = -1;
            atomicMin(&m[0], kStatusEmptyBeam);
        } else {
            m[1] = (int32_t)s_best;
            m[3] = __builtin_bit_cast(int32_t, scA[s_best]);
        }
    }
}

__global__ __launch_bounds__(64) void backtrace_generic_kernel(const Lattice *__restrict__ lats, const int32_t *meta)
{
    const Lattice &d = lats[blockIdx.x];
    const int lane = threadIdx.x;
    int64_t p = meta[4 * (size_t)d.idx + 1];
    if (p < 0) return;
    const int64_t T = d.T, L = d.L, B = d.beam, W = d.W;
    const uint8_t *bp = reinterpret_cast<const uint8_t *>(d.bp);
    if (lane == 0) {
        for (int64_t t = T - 1; t >= 0; --t) {
            int64_t lo = (L * t) / T - B / 2;  // host guarantees L, T < 2^31
            lo = lo < 0 ? 0 : lo;
            d.path[t] = (int32_t)p;
            p -= bp[(size_t)t * (size_t)W + (size_t)(p - lo)];
        }
    }
}

// ---------------------------------------------------------------------------------------
// mean-subtracted log-softmax (align.py:116-117), one wavefront per row
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__global__ __launch_bounds__(256) void log_softmax_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          int64_t T, int V, int64_t ld_in, int64_t ld_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float *x = in + (size_t)row * (size_t)ld_in;
    float *y = out + (size_t)row * (size_t)ld_out;
    float s = 0.0f;
    for (int c = lane; c < V; c += 64) s += x[c];
    const float mean = wave_sum(s) / (float)V;
    float z = 0.0f;
    for (int c = lane; c < V; c += 64) z += expf(x[c] - mean);
    const float lz = logf(wave_sum(z));
    for (int c = lane; c < V; c += 64) y[c] = (x[c] - mean) - lz;
}

// ---------------------------------------------------------------------------------------
// LSTM cell update of the log-prob producer (AudioToChar, kokoro_align/train.py:54-65), one time step of one
// layer, both directions: gates = gin[row] + rec, PyTorch gate order (i, f, g, o);
//   c = sigmoid(f)*c + sigmoid(i)*tanh(g);  h = sigmoid(o)*tanh(c)
// gin holds x_t @ W_ih^T + b_ih + b_hh of every frame (one library GEMM per layer), rec = h_{t-1} @ W_hh^T of
// the n sequences still running (one batched library GEMM per step); this kernel is the fused element-wise
// part and scatters h into the layer's output rows.  grid: x = ceil(n*H/256), y = direction (0 fwd, 1 bwd).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_step_kernel(const float *__restrict__ gin, int64_t ldg,
                                                        const float *__restrict__ rec, int64_t rec_dir_stride,
                                                        float *__restrict__ c, float *__restrict__ h, int64_t state_dir_stride,
                                                        float *__restrict__ out, int64_t ldo,
                                                        const int32_t *__restrict__ rows, int64_t rows_dir_stride, int n, int H)
{
    const int dir = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * H) return;
    const int s = (int)(idx / H), j = (int)(idx % H);
    const int64_t row = rows[(size_t)dir * rows_dir_stride + s];
    const float *g = gin + (size_t)row * ldg + (size_t)dir * 4 * H;
    const float *r = rec + (size_t)dir * rec_dir_stride + (size_t)s * 4 * H;
    const float gi = g[j] + r[j], gf = g[H + j] + r[H + j], gg = g[2 * H + j] + r[2 * H + j], go = g[3 * H + j] + r[3 * H + j];
    const float si = 1.0f / (1.0f + expf(-gi)), sf = 1.0f / (1.0f + expf(-gf)), so = 1.0f / (1.0f + expf(-go));
    float *cs = c + (size_t)dir * state_dir_stride + (size_t)s * H;
    float *hs = h + (size_t)dir * state_dir_stride + (size_t)s * H;
    const float cn = sf * cs[j] + si * tanhf(gg);
    const float hn = so * tanhf(cn);
    cs[j] = cn;
    hs[j] = hn;
    out[(size_t)row * ldo + (size_t)dir * H + j] = hn;
}

// ---------------------------------------------------------------------------------------
// One whole LSTM layer, both directions, persistent: a 256-thread workgroup owns 16 sequences of one direction
// for ALL their time steps.  The recurrent product h @ W_hh^T runs on the f32 MFMA (v_mfma_f32_16x16x4_f32,
// exact float32): wave w computes the four gates of hidden units [32w, 32w+32), and its 128 x 128 slice of
// W_hh^T (64 KB) stays in registers for the whole kernel - 256 of the 512 VGPR/AGPRs a wave has at one
// wave per SIMD - so no weight byte is read after start-up.  h lives in LDS (double-buffered, one barrier per
// step); the input projections of a step are loaded before its MFMA loop and added after it.
// Sequences are sorted by length (longest first): a tile runs for its first sequence's length.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float fast_tanh(float x)   // 1 - 2/(1 + e^{2x}): exact limits at +-inf
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681472f));
}
constexpr int kLstmH = 128;
constexpr int kLstmTile = 16;          // sequences per workgroup
constexpr int kLstmLdh = kLstmH + 4;   // LDS row pitch (floats): 16-byte aligned rows, conflict-free b128 A-fragment reads
constexpr int kLstmWAcc = 30;          // k-steps whose 8 W fragments live in AGPRs (240 of 256); the last two sit in VGPRs
// D = A*B + D with B taken straight from an accumulation register: the compiler's own allocation of the builtin
// parked W in AGPRs and copied every fragment through one VGPR (v_accvgpr_read + s_nop + spill reloads) per MFMA.
#define KA_MFMA_ACC(ACC, A, W) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(W))
#define KA_MFMA_VGPR(ACC, A, W) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(W))
// grid.x = 2 * ceil(nseq / 16): workgroup id>>1 = tile (longest sequences first, so the dispatcher starts the
// long tiles first and back-fills the CUs with short ones), id&1 = direction
// XIN = false: `gin` holds the input projections x W_ih^T + b_ih + b_hh of every frame (one library GEMM per layer).
// XIN = true (layer 0, kLstmIn = 40 input features): `gin` IS x [frames, ldg >= 40]; the projection runs inside the step - 80
// more MFMAs (K = 40: 10 k-steps x 8 fragments, W_ih's fragments resident in VGPRs like W_hh's in AGPRs) on top of the 256 of
// h W_hh^T, the bias is the accumulator's initial value - so the [frames, 1024] projection (11 GB for an 8.8-hour book) is
// never written or read, and a step prefetches 10 dwords per lane instead of 32.
constexpr int kLstmIn = 40;
template <bool XIN>
__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(const float *__restrict__ gin, int64_t ldg,
                                                            const float *__restrict__ w_hh, float *__restrict__ out, int64_t ldo,
                                                            const int32_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                                                            int nseq, const float *__restrict__ w_ih, const float *__restrict__ bias)
{
    // h of the tile's 16 sequences, double-buffered; within a row unit k sits at (k&3)*32 + (k>>2), so the 32
    // A operands of a lane (k = 4s + kq, s = 0..31) are contiguous: 8 ds_read_b128 per step
    __shared__ __attribute__((aligned(16))) float s_h[2][kLstmTile][kLstmLdh];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * kLstmTile;
    const int col = lane & 15, kq = lane >> 4;
    const int jbase = 32 * wv;                       // hidden units of this wave: two column tiles of 16
    // W fragments: gate g, column tile ct, k-step s: B[k = 4s+kq][n = col] = W_hh[dir][g*H + jbase + 16ct + col][k]
    float wreg[4][2][32];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int s = 0; s < 32; ++s)
                wreg[g][ct][s] = w_hh[((size_t)dir * 4 * kLstmH + (size_t)g * kLstmH + jbase + 16 * ct + col) * kLstmH + 4 * s + kq];
    // XIN: W_ih's fragments B[k = 4s+kq][n = col] = W_ih[dir][g*H + jbase + 16ct + col][k], the bias of the lane's columns, and the
    // sequence whose x row is this lane's A operand (row `col` of the tile)
    float wih[4][2][kLstmIn / 4], bs[4][2];
    int lenA = 0, rowbaseA = 0;
    if constexpr (XIN) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const size_t wrow = (size_t)dir * 4 * kLstmH + (size_t)g * kLstmH + jbase + 16 * ct + col;
                bs[g][ct] = bias[wrow];
#pragma unroll
                for (int s = 0; s < kLstmIn / 4; ++s) wih[g][ct][s] = w_ih[wrow * kLstmIn + 4 * s + kq];
            }
        const int i = tile0 + col;
        const int l = i < nseq ? seq_len[i] : 0;
        const int o = i < nseq ? seq_off[i] : 0;
        lenA = l;
        rowbaseA = l <= 0 ? 0 : dir == 0 ? o : o + l - 1;
    }
    // the 4 sequences (rows 4kq .. 4kq+3 of the C tile) this lane updates
    int rowbase[4], len[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tile0 + 4 * kq + r;
        const int l = i < nseq ? seq_len[i] : 0;
        const int o = i < nseq ? seq_off[i] : 0;
        len[r] = l;
        rowbase[r] = l <= 0 ? 0 : dir == 0 ? o : o + l - 1;   // empty / padding rows prefetch row 0 (never used)
    }
    const int tile_len = tile0 < nseq ? seq_len[tile0] : 0;   // sorted by length, longest first
    float c[2][4], hreg[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[ct][r] = hreg[ct][r] = 0.0f;
    for (int i = tid; i < 2 * kLstmTile * kLstmLdh; i += 256) (&s_h[0][0][0])[i] = 0.0f;
    __syncthreads();
    const float *gcol = gin + (size_t)dir * 4 * kLstmH + jbase + col;
    float *ocol = out + (size_t)dir * kLstmH + jbase + col;
    int hpos[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) hpos[ct] = ((jbase + 16 * ct + col) & 3) * 32 + ((jbase + 16 * ct + col) >> 2);
    // input projections one step ahead: they are the MFMA's initial accumulator, so they must have landed
    // before the step starts; a finished sequence re-reads its last row (never used)
    auto gin_row = [&](int r, int t) {
        const int k = max(min(t, len[r] - 1), 0);
        return gcol + (size_t)(dir == 0 ? rowbase[r] + k : rowbase[r] - k) * (size_t)ldg;
    };
    auto x_row = [&](int t) {       // XIN: the row of sequence `col` at step t, columns kq, kq+4, ...
        const int k = max(min(t, lenA - 1), 0);
        return gin + (size_t)(dir == 0 ? rowbaseA + k : rowbaseA - k) * (size_t)ldg + kq;
    };
    f32x4 nxt[4][2];
    float xn[kLstmIn / 4];
    if constexpr (XIN) {
        const float *xp = x_row(0);
#pragma unroll
        for (int s = 0; s < kLstmIn / 4; ++s) xn[s] = xp[4 * s];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *gp = gin_row(r, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) nxt[g][ct][r] = gp[(size_t)g * kLstmH + 16 * ct];
        }
    }
    for (int t = 0; t < tile_len; ++t) {
        const int cur = t & 1;
        f32x4 acc[4][2];
        float xa[kLstmIn / 4];
        if constexpr (XIN) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[g][ct] = f32x4{bs[g][ct], bs[g][ct], bs[g][ct], bs[g][ct]};
            const float *xp = x_row(t + 1);
#pragma unroll
            for (int s = 0; s < kLstmIn / 4; ++s) {
                xa[s] = xn[s];
                xn[s] = xp[4 * s];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[g][ct] = nxt[g][ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *gp = gin_row(r, t + 1);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) nxt[g][ct][r] = gp[(size_t)g * kLstmH + 16 * ct];
            }
        }
        const f32x4 *arow = reinterpret_cast<const f32x4 *>(&s_h[cur][col][kq * 32]);
        asm volatile("s_nop 3" ::: "memory");   // VALU-written accumulators -> first MFMA
        if constexpr (XIN) {
#pragma unroll
            for (int s = 0; s < kLstmIn / 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) KA_MFMA_VGPR(acc[g][ct], xa[s], wih[g][ct][s]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 a4 = arow[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * q + u;
                const float a = a4[u];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (s < kLstmWAcc) KA_MFMA_ACC(acc[g][ct], a, wreg[g][ct][s]);
                        else KA_MFMA_VGPR(acc[g][ct], a, wreg[g][ct][s]);
                    }
            }
        }
        asm volatile("s_nop 10" ::: "memory");   // 8-pass MFMA result -> VALU read: 11 wait states
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool act = t < len[r];
                // hardware exp2 / rcp (about 1 ulp each): 4 instructions per sigmoid instead of a ~40-instruction libm call
                const float si = fast_sigmoid(acc[0][ct][r]), sf = fast_sigmoid(acc[1][ct][r]), so = fast_sigmoid(acc[3][ct][r]);
                const float cn = sf * c[ct][r] + si * fast_tanh(acc[2][ct][r]);
                const float hn = so * fast_tanh(cn);
                if (act) {
                    c[ct][r] = cn;
                    hreg[ct][r] = hn;
                    ocol[(size_t)(dir == 0 ? rowbase[r] + t : rowbase[r] - t) * (size_t)ldo + 16 * ct] = hn;
                }
                s_h[cur ^ 1][4 * kq + r][hpos[ct]] = hreg[ct][r];
            }
        __syncthreads();
    }
}
#undef KA_MFMA_ACC
#undef KA_MFMA_VGPR

// ---------------------------------------------------------------------------------------
// Audio front end (kokoro_align/preprocess.py:51-131)
// ---------------------------------------------------------------------------------------
// Mean square of every 256-sample window (preprocess.py:53-54: np.mean(x.reshape(-1, 256)**2, axis=1), float32).
// The split decision compares these levels with a threshold, so the sum is taken in EXACTLY NumPy's order for a
// contiguous float32 row of 256: two halves of 128; in a half, 8 running sums over elements j, j+8, j+16, ...,
// combined as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); then half0 + half1; then / 256.  16 lanes per window: lane
// (h, j) owns running sum j of half h.  grid: 16 windows per 256-thread block.
__global__ __launch_bounds__(256) void window_energy_kernel(const float *__restrict__ x, int64_t n_windows, float *__restrict__ out)
{
    const int64_t w = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15, h = sub >> 3, j = sub & 7;
    float r = 0.0f;
    if (w < n_windows) {
        const float *p = x + (size_t)w * 256 + h * 128 + j;
        const float v0 = p[0];
        r = v0 * v0;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float v = p[8 * k];
            r += v * v;
        }
    }
    // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) within each group of 8 lanes, then the two halves
    r = r + __shfl_xor(r, 1);          // lanes 2i, 2i+1 both hold r[2i] + r[2i+1] (float add is commutative)
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    r = r + __shfl_xor(r, 8);
    if (w < n_windows && sub == 0) out[w] = r / 256.0f;
}

// Windowed frames of the short-time transform, all segments of a recording in one launch:
//   frames[f][k] = window[k] * y_seg[reflect(f_local*hop - n_fft/2 + k)]        (center=True, pad_mode="reflect")
// seg_start / seg_len = first sample and length of every segment, frame_off = first frame row of every segment
// (a segment of len samples has 1 + len/hop frames).  grid: x strides the frames of a segment, y = segment.
__global__ __launch_bounds__(256) void stft_frames_kernel(const float *__restrict__ y, const int64_t *__restrict__ seg_start,
                                                          const int64_t *__restrict__ seg_len, const int64_t *__restrict__ frame_off,
                                                          int n_fft, int hop, const float *__restrict__ window,
                                                          float *__restrict__ frames, int64_t ld)
{
    const int s = blockIdx.y;
    const int64_t len = seg_len[s], start = seg_start[s], f0 = frame_off[s];
    const int64_t nfr = 1 + len / hop;
    const int half = n_fft / 2;
    for (int64_t f = blockIdx.x; f < nfr; f += gridDim.x) {
        float *row = frames + (size_t)(f0 + f) * (size_t)ld;
        for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
            int64_t i = f * hop - half + k;
            i = i < 0 ? -i : i;
            i = i >= len ? 2 * (len - 1) - i : i;
            row[k] = window[k] * y[start + i];
        }
    }
}

// |X|^2 of a transform stored as [n][2*nf] = (real parts | imaginary parts)
__global__ __launch_bounds__(256) void power_kernel(const float *__restrict__ reim, int64_t ld_in, float *__restrict__ power,
                                                    int64_t ld_out, int64_t n, int nf)
{
    const int64_t total = n * (int64_t)nf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nf;
        const int c = (int)(i - r * nf);
        const float re = reim[(size_t)r * ld_in + c], im = reim[(size_t)r * ld_in + nf + c];
        power[(size_t)r * ld_out + c] = re * re + im * im;
    }
}

// 10*log10(max(x, 1e-10)) in place, and the maximum of every segment (AmplitudeToDB("power"), first half).
// segmax must hold -inf on entry.  grid: x strides the rows of a segment, y = segment.
__device__ __forceinline__ void atomic_max_float(float *addr, float v)
{
    // order-preserving integer view: non-negative floats compare as ints, negative ones reversed as unsigned
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __builtin_bit_cast(int, v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __builtin_bit_cast(unsigned int, v));
}
__global__ __launch_bounds__(256) void power_to_db_kernel(float *__restrict__ x, int64_t ld, int cols, const int64_t *__restrict__ frame_off,
                                                          float *__restrict__ segmax)
{
    const int s = blockIdx.y;
    const int64_t r0 = frame_off[s], r1 = frame_off[s + 1];
    const int64_t total = (r1 - r0) * cols;
    float m = -__builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        float *p = x + (size_t)(r0 + r) * (size_t)ld + (i - r * cols);
        const float v = 10.0f * log10f(fmaxf(*p, 1e-10f));
        *p = v;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > -__builtin_inff()) atomic_max_float(&segmax[s], m);
}
// x = max(x, segmax - top_db) (AmplitudeToDB, second half: top_db is relative to the maximum of one call = one segment)
__global__ __launch_bounds__(256) void db_floor_kernel(float *__restrict__ x, int64_t ld, int cols, const int64_t *__restrict__ frame_off,
                                                       const float *__restrict__ segmax, float top_db)
{
    const int s = blockIdx.y;
    const int64_t r0 = frame_off[s], r1 = frame_off[s + 1];
    const int64_t total = (r1 - r0) * cols;
    const float lo = segmax[s] - top_db;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        float *p = x + (size_t)(r0 + r) * (size_t)ld + (i - r * cols);
        *p = fmaxf(*p, lo);
    }
}

// ---------------------------------------------------------------------------------------
// hash generator of synthetic inputs (definition: include/kokoro_align_amd.h)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t idx)
{
    uint64_t z = (seed * 0x9E3779B97F4A7C15ull + idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// blockIdx.y = lattice: lattice i lives at base + i*stride elements and uses seed + i
__global__ __launch_bounds__(256) void hash_logprobs_kernel(float *lp0, int64_t T, int V, int64_t ld, uint64_t seed0,
                                                            int64_t lattice_stride)
{
    float *lp = lp0 + (size_t)blockIdx.y * (size_t)lattice_stride;
    const uint64_t seed = seed0 + blockIdx.y;
    const int64_t n = T * (int64_t)V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / V;
        const int c = (int)(i - t * V);
        const uint64_t h = mix64(seed, (uint64_t)i);
        lp[(size_t)t * (size_t)ld + c] = -8.0f * ((float)(h >> 40) * (1.0f / 16777216.0f));
    }
}
__global__ __launch_bounds__(256) void hash_labels_kernel(int32_t *labels0, int64_t S, int V, uint64_t seed0,
                                                          int64_t lattice_stride)
{
    int32_t *labels = labels0 + (size_t)blockIdx.y * (size_t)lattice_stride;
    const uint64_t seed = seed0 + blockIdx.y;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < S; k += (int64_t)gridDim.x * blockDim.x)
        labels[k] = (int32_t)(1 + mix64(seed ^ 0x4C4142454C53ull, (uint64_t)k) % (uint64_t)(V - 1));
}

}  // namespace ka
