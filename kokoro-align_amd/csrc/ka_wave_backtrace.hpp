// ka_wave_backtrace.hpp — the backtraces: stored back-pointers (backtrace_w16), back-pointers recomputed around the path
// (backtrace_rc, serial and chunk-parallel), the output gather and the generic backtrace.  Included by ka_wave_bt.hip only.
#pragma once
#include "ka_device.hpp"

namespace ka {

#ifndef KA_RC_MIN_WAVES
#define KA_RC_MIN_WAVES 8
#endif

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
__device__ __forceinline__ void backtrace_rc_body(const Lattice *__restrict__ lats, const int32_t *meta, int n_lats)
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

// PAR = false: two kernels per launch (ZL: the transcript contains label 0), each skips the other's lattices - as in the forward kernels
template <int M, bool ZL, bool PAR, bool GO = false>
__global__ __launch_bounds__(64, KA_RC_MIN_WAVES) void backtrace_rc_kernel(const Lattice *__restrict__ lats, const int32_t *meta, int n_lats)
{
    backtrace_rc_body<M, ZL, PAR, GO>(lats, meta, n_lats);
}
// PAR = true (one wavefront per chunk of every chunk-parallel lattice): ONE kernel, the two instances behind a wave-uniform branch -
// a grid of 430 000 workgroups that only look their lattice up and leave cost the corpus launch 0.1 ms; 53 / 54 registers
template <int M>
__global__ __launch_bounds__(64, KA_RC_MIN_WAVES) void backtrace_rc_chunks_kernel(const Lattice *__restrict__ lats, const int32_t *meta, int n_lats)
{
    const Lattice &d = lats[__builtin_amdgcn_readfirstlane(lattice_of_chunk(lats, n_lats, (int64_t)blockIdx.x))];
    const int flags = __builtin_amdgcn_readfirstlane(meta[4 * (size_t)d.idx + 2]);
    if (flags & kFlagZeroLabel)
        backtrace_rc_body<M, true, true>(lats, meta, n_lats);
    else
        backtrace_rc_body<M, false, true>(lats, meta, n_lats);
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


}  // namespace ka
