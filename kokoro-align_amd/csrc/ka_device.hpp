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
#pragma once
#include "ka_types.hpp"

namespace ka {

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
// The same reload through a buffer descriptor: the row's byte offset is ONE 32-bit scalar (`soffset`), which a frame advances
// and clamps to the last row with two scalar instructions (s_add_u32, s_min_u32) where the 64-bit pointer of row_reload took
// six (add, compare, two selects, add, add-with-carry) - and a scalar instruction costs this kernel nearly half a vector
// one.  Needs 4 T ld < 2^32.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor over `bytes` bytes at `base` (both wave-uniform; stride 0, 32-bit data format: word 3 = 0x00020000)
__device__ __forceinline__ u32x4 lp_descriptor(const void *base, uint64_t bytes)
{
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t n = bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)n), 0x00020000u};
}
__device__ __forceinline__ void row_reload_buf(float &r, uint32_t lane_byte_off, u32x4 rsrc /* wave-uniform */, uint32_t row_byte_off /* wave-uniform */)
{
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(r) : "v"(lane_byte_off), "s"(rsrc), "s"(row_byte_off) : "memory");
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
// The two label cells of a group of four (positions 4G+1 and 4G+3) with all four moves: the upper one's candidates are l1, b1,
// l0, b0, the lower one's l0, b0, p1, p2 - max(l0, b0) is taken once, and each cell is then ONE v_max3_f32 (five instead of six
// maxima per group, 20 instead of 24 per frame of sixteen cells; max is exact, so the association changes no bit).  With the
// label-0 veto (ZL) the upper cell's l0 goes through a v_min first and nothing is shared.
template <int M, bool ZL>
__device__ __forceinline__ f32x2 label_pair_max(float l1, float b1, float l0, float b0, float p1, float p2, float veto1, float veto0)
{
    if constexpr (M == 4 && !ZL) {
        const float m0 = __builtin_fmaxf(l0, b0);
        return f32x2{__builtin_fmaxf(__builtin_fmaxf(m0, p1), p2), __builtin_fmaxf(__builtin_fmaxf(l1, b1), m0)};
    } else {
        return f32x2{cell_label_max<M, ZL>(l0, b0, p1, p2, veto0), cell_label_max<M, ZL>(l1, b1, l0, b0, veto1)};
    }
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
    ml = label_pair_max<M, ZL>(l1, b1, l0, b0, p1, p2, vz[2 * G + 1], vz[2 * G]);   // {lower, upper}
    mb[1] = cell_blank_max<M>(b1, l0, p1);
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

// lane i takes x of lane i-1, lane 0 takes `first` (DPP wave_shr:1, bound_ctrl off)
__device__ __forceinline__ float wave_shr1(float first, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, x), 0x138, 0xF, 0xF, false));
}

// Which compute unit this wavefront runs on, as an index into a kCuSlots-entry table: XCC_ID (3 bits) | HW_ID's se_id, sh_id,
// cu_id (bits 15:8).  Used for SPEED only (ka_tiled_stream.hpp spreads the tiles that are alive first over the CUs).
__device__ __forceinline__ uint32_t cu_slot()
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
}

}  // namespace ka
