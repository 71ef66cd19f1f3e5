// ka_tiled.hpp — tile-pipelined forward DP (KA_MODE_TILED): the lattice is cut along the label axis into tiles of
// 256 positions, ONE WAVEFRONT OWNS A TILE, and the tiles of a lattice run as a software pipeline.
//
// Why: the frame axis is a serial chain.  With one wavefront per lattice (forward_ck_kernel) a frame costs that
// wavefront ~70 instructions, so a lone lattice - or a book's few dozen chapters - runs at ~0.22 us per frame on a chip
// that is 99 % idle.  Here a lane owns 4 cells instead of 16 (a frame is ~25 instructions) and the band of a frame is
// spread over the tiles it touches; dependencies only point UP the label axis (cell p reads p, p-1, p-2, p-3 of the
// previous frame, align.py:70-81), so tile b may run any number of frames behind tile b-1: there is no barrier, tile b-1
// publishes its top three cells per frame (a 16-byte "halo" packet in HBM) and tile b consumes them 30-70 frames later.
//   * Tiles are ABSOLUTE: tile b = positions [256 b, 256 b + 256) for the whole run.  It lives from the frame in which the
//     band's upper edge reaches it (t_in) to the frame in which the lower edge has passed it (t_end); nothing is
//     re-labelled and nothing wraps, so any band width works - beam_size >= 2L (the whole lattice, BASELINE configs[4]
//     "tiled DP") is simply every tile alive for all T frames.
//   * One single-wavefront workgroup per tile; an LDS request of 40 KB keeps them at one per SIMD (tools/ubench/census.hip:
//     1024 single-wave workgroups land on 1024 SIMDs), the rest of the grid waits in the dispatcher.  Tiles are drawn from a
//     ticket counter; the host sorts tasks by t_in, so a tile's producer always holds an earlier ticket.
//   * Hand-off (cdna_hip_programming.md Guideline 16, sc1 payload + drained + sc1 flag; all loads of it sc1): halo packets
//     are write-through stores of one lane; a tile publishes "slots < n are complete" once per 32-frame block, n being what
//     its in-order vmcnt wait has already retired - the publish never waits for anything; the consumer polls that word
//     once per block, two blocks ahead of use.  Every slot is written once and read once: no ring, no back-pressure.
//   * Log-prob rows and halo packets are staged through LDS in blocks of 32 frames (LDS-DMA, requested three blocks ahead of
//     their use), so the frame loop reads only LDS.  (Blocks of 16 frames: the per-block work - poll, DMA requests,
//     publish, finiteness sum - was as long as the block's frames; 32: 93 -> 70 ns per frame of a lone tile.)
//   * Scores only, like forward_ck: the score ring is stored every 32 frames (position p at slot p & ck_mask of its
//     checkpoint row) and backtrace_rc_kernel recomputes the back-pointers around the path.
#pragma once
#include "ka_device.hpp"

namespace ka {



typedef uint32_t KA_GLOBAL *gu32w_t;

__device__ __forceinline__ float lds_f32(uint32_t addr) { return *(const __attribute__((address_space(3))) float *)(uintptr_t)addr; }
__device__ __forceinline__ f32x4 lds_f32x4(uint32_t addr) { return *(const __attribute__((address_space(3))) f32x4 *)(uintptr_t)addr; }
// lane i <- lane i-1; lane 0 keeps `first` (DPP wave_shr:1, invalid source lanes keep the old value)
// lanes [a, b) of a 64-bit mask, any a, b (clamped to 0..64)
__device__ __forceinline__ uint64_t tp_lane_range(int32_t a, int32_t b)
{
    a = a < 0 ? 0 : (a > 64 ? 64 : a);
    b = b < 0 ? 0 : (b > 64 ? 64 : b);
    if (b <= a) return 0ull;
    const uint32_t n = (uint32_t)(b - a);
    return (n >= 64u ? ~0ull : ((1ull << n) - 1ull)) << a;
}
struct TpMasks {
    uint64_t m0, m1, m2, m3;   // m<k>: lanes whose cell k (position base + 4 lane + k) is inside the band
};
// band [lo, hi) relative to the tile's first position (may be negative / beyond the tile)
__device__ __forceinline__ void tp_masks(TpMasks &mk, int32_t lo_rel, int32_t hi_rel)
{
    lo_rel = lo_rel < -8 ? -8 : (lo_rel > kTpTile + 8 ? kTpTile + 8 : lo_rel);
    hi_rel = hi_rel < -8 ? -8 : (hi_rel > kTpTile + 8 ? kTpTile + 8 : hi_rel);
    // lanes l with lo_rel <= 4 l + k < hi_rel  <=>  l in [ceil((lo_rel - k) / 4), ceil((hi_rel - k) / 4))
    mk.m0 = tp_lane_range((lo_rel + 3) >> 2, (hi_rel + 3) >> 2);
    mk.m1 = tp_lane_range((lo_rel + 2) >> 2, (hi_rel + 2) >> 2);
    mk.m2 = tp_lane_range((lo_rel + 1) >> 2, (hi_rel + 1) >> 2);
    mk.m3 = tp_lane_range((lo_rel + 0) >> 2, (hi_rel + 0) >> 2);
}
// the position `rel` (relative to the tile's first one; anywhere) enters or leaves the band: flip its lane bit
__device__ __forceinline__ void tp_mask_toggle(TpMasks &mk, int32_t rel)
{
    if (rel < 0 || rel >= kTpTile) return;
    const uint64_t bit = 1ull << ((uint32_t)rel >> 2);
    const uint32_t k = (uint32_t)rel & 3u;
    mk.m0 ^= k == 0 ? bit : 0ull;
    mk.m1 ^= k == 1 ? bit : 0ull;
    mk.m2 ^= k == 2 ? bit : 0ull;
    mk.m3 ^= k == 3 ? bit : 0ull;
}
// -inf into the cell at tile-relative position rel (0..255): S = {cell 0, 2, 1, 3} of lane rel >> 2.  ONE v_cndmask behind
// a two-level scalar branch INSIDE one asm statement (6 instructions executed; as C++ - four selects on masks picked by
// s_cselect, or a switch whose arms are asm statements - hipcc made 24 to 35 of it, with copies at the merges).
__device__ __forceinline__ void tp_kill(f32x4 &S, uint32_t rel, float NINF)
{
    const uint64_t m = 1ull << (rel >> 2);
    float c0 = S[0], c2 = S[1], c1 = S[2], c3 = S[3];
    asm volatile("s_bitcmp1_b32 %[rel], 1\n\t"
                 "s_cbranch_scc1 .Lka_k23_%=\n\t"
                 "s_bitcmp1_b32 %[rel], 0\n\t"
                 "s_cbranch_scc1 .Lka_k1_%=\n\t"
                 "v_cndmask_b32 %[c0], %[c0], %[ninf], %[m]\n\t"
                 "s_branch .Lka_ke_%=\n"
                 ".Lka_k1_%=:\n\t"
                 "v_cndmask_b32 %[c1], %[c1], %[ninf], %[m]\n\t"
                 "s_branch .Lka_ke_%=\n"
                 ".Lka_k23_%=:\n\t"
                 "s_bitcmp1_b32 %[rel], 0\n\t"
                 "s_cbranch_scc1 .Lka_k3_%=\n\t"
                 "v_cndmask_b32 %[c2], %[c2], %[ninf], %[m]\n\t"
                 "s_branch .Lka_ke_%=\n"
                 ".Lka_k3_%=:\n\t"
                 "v_cndmask_b32 %[c3], %[c3], %[ninf], %[m]\n"
                 ".Lka_ke_%=:"
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3)
                 : [rel] "s"(rel), [m] "s"(m), [ninf] "v"(NINF)
                 : "scc");
    S = f32x4{c0, c2, c1, c3};
}
// state of a lane: S = {cell 0, cell 2, cell 1, cell 3} = {blank, blank, label, label} - the two blanks and the two labels
// are register pairs (v_pk_add_f32 of the emissions), and the four registers as they lie ARE the halo packet
__device__ __forceinline__ void tp_mask_state(f32x4 &S, const TpMasks &mk, float NINF)
{
    S[0] = select_by_mask(NINF, S[0], mk.m0);
    S[2] = select_by_mask(NINF, S[2], mk.m1);
    S[1] = select_by_mask(NINF, S[1], mk.m2);
    S[3] = select_by_mask(NINF, S[3], mk.m3);
}

// sc1 (write-through, agent scope) accesses of the hand-off.  The loads are untracked by hipcc like the row loads:
// pair with a counted wait.
template <int OFF>
__device__ __forceinline__ void tp_halo_store(const void *block_base /* uniform */, const f32x4 &pk, uint64_t lane_mask)
{
    // one lane stores: EXEC is narrowed to it and put back as it was (never assumed to be "all lanes": the compiler
    // may have structured the surrounding control flow with lanes parked).  A store wider than 64 bits reads its data
    // registers for two more wait states: the EXEC restore and the s_nop are those.
    uint64_t saved;
    asm volatile("s_nop 4\n\ts_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tglobal_store_dwordx4 %1, %2, %3 offset:%5 sc1\n\ts_mov_b64 exec, %0\n\ts_nop 0"
                 : "=&s"(saved) : "v"(0u), "v"(pk), "s"(block_base), "s"(lane_mask), "i"(OFF) : "memory", "scc");
}
__device__ __forceinline__ void tp_prog_store(gu32w_t word /* uniform */, uint32_t value)
{
    uint64_t saved;
    asm volatile("s_nop 4\n\ts_mov_b64 %0, exec\n\ts_and_b64 exec, exec, 1\n\tglobal_store_dword %1, %2, %3 sc1\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(0u), "v"(value), "s"(word) : "memory", "scc");
}
__device__ __forceinline__ void tp_prog_load(uint32_t &dst, gu32w_t word /* uniform */)
{
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "+v"(dst) : "v"(0u), "s"(word) : "memory");
}
// progress of the tile below must reach `need` leading slots; polled relaxed with a sleep that grows while far away.
// Bounded by a STALL detector: a tile whose producer has not advanced its progress word for ~4 s of wall clock gives up
// (returns false; the lattice gets KA_ERR_INTERNAL) instead of hanging the GPU - this can only be a bug in the hand-off,
// never an input.  The clock restarts whenever the polled word moves: a tile whose producer is healthy but far behind
// (a long lattice with every tile resident, a queue that is time-sliced with another process) waits as long as it takes.
// Hysteresis: a tile that does have to wait waits for `want` >= need (two blocks more): the poll it carries into a block
// start is a block old, so a tile sitting exactly at the limit would pay a poll round trip (~1 us) at every block;
// after one longer wait it stays ahead of its stale information for as long as it is not faster than its producer.
// (diagnostic counters - number of waits, 100 MHz ticks spent in them - live in two LDS words at `stat_lds`)
__device__ __forceinline__ bool tp_wait_progress(gu32w_t word, uint32_t need, uint32_t want, uint32_t have, uint32_t stat_lds)
{
    if (have >= need) return true;
    const uint64_t t0 = wall_clock64();   // 100 MHz
    uint64_t t_moved = t0;
    __attribute__((address_space(3))) uint32_t *st = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds;
    st[0] += 1;
    for (;;) {
        const uint32_t gap = want - have;
        if (gap > 4096u) __builtin_amdgcn_s_sleep(127);
        else if (gap > 256u) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(4);
        uint32_t v = 0;
        tp_prog_load(v, word);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
        const uint32_t now_have = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        const uint64_t now = wall_clock64();
        if (now_have != have) t_moved = now;
        have = now_have;
        if (have >= want) {
            st[1] += (uint32_t)(now - t0);
            return true;
        }
        if (now - t_moved > 400000000ull) return false;
    }
}

template <int M, bool ZL>
struct TpTile {
    // wave-uniform description of the tile and its lattice
    uint32_t T, L, B, dq, dr;
    // The band (align.py:64-65) per BLOCK of 32 frames, not per frame: q0 / r0 = floor(L tb / T) and the remainder at the
    // block's first frame tb, advanced by (32 L) / T, (32 L) % T per block; per block the lanes work out, 34 frames at once,
    // which positions of THIS tile enter or leave the band at which frame (KE / KL) and `ev` gets bit F set when frame
    // tb+F has any to kill.  The common frame pays one s_bitcmp1 + branch, and a tile pays nothing at all for
    // the band steps that do not touch it - 3 of 5 for a 1000-wide band, whose edges are inside a tile for 2 x 256 of the
    // ~1250 steps the tile lives through.  (Rounds 1-2 stepped a Bresenham remainder in every frame and ran ~40 scalar
    // instructions at every step, relevant or not: 161 cycles per frame in cfg2 against 105 where the band never moves.)
    uint32_t q0, r0, dq32, dr32, ev;
    uint32_t KL, KE;        // per lane (VGPRs): the cells to kill, lane l <-> the band step after frame tb - 1 + l (tp_band_block)
    float inv_T;
    __device__ __forceinline__ uint32_t lo_of(uint32_t q) const
    {
        const int32_t d = (int32_t)q - (int32_t)(B >> 1);
        return (uint32_t)(d > 0 ? d : 0);
    }
    __device__ __forceinline__ uint32_t hi_of(uint32_t lo) const { return (L - lo < B) ? L : lo + B; }
    int32_t base, t_in, t_end;
    const char *lp;
    size_t ld;
    uint32_t lane_off;
    const char *halo_in;    // slot j of the lower boundary at halo_in + (j - t_in) * 16
    char *halo_out;         // slot j of the upper boundary at halo_out + (j - t_in) * 16 (the top tile writes to a boundary nobody reads)
    gu32w_t prog_in, prog_out;
    char *ck;               // checkpoint k (scores after frame 32 (k + 1) - 1) at ck + k * ck_pitch
    uint32_t ck_pitch;
    uint32_t ck_off;        // per lane: ((base + 4 lane) & ck_mask) * 4
    // per lane
    f32x4 S;
    int la0, la1;           // 4 * label of cells 1 and 3
    float vz0, vz1;
    float absum;
    // LDS
    uint32_t lds_rows, lds_halo;   // byte addresses of this workgroup's staging rings
    uint32_t lds_stage;            // per lane: where frame 0 of a block drops the lane's four cells (lane 63: the packet row)
    uint32_t lds_packets;          // the packet row: lane 63's cells of frame f at + 16 f
};

// One frame, F = its index in the block.  The LDS reads run TWO frames ahead of their use (an LDS read takes longer than
// half a frame of this loop): In.cur = inputs of this frame (emissions E, e0 and H, the three cells below each lane's
// first cell), In.nxt = raw LDS data of frame t+1 (issued a frame ago, landed by now), and the reads for frame t+2 are
// issued here from `r2_*` / `h2` (byte addresses of row / packet t+2 in LDS).  H of frame t+1 is taken at the end, from
// this frame's final scores and the packet of slot t+1.
struct TpIn {
    f32x2 E;      // emissions of the two label cells
    float e0;     // blank emission
    f32x4 hp;     // packet of the tile below: {cell 0, 2, 1, 3} of the lane below lane 0
};
template <int M, bool ZL, bool GUARDED, int F>
__device__ __forceinline__ void tp_frame(TpTile<M, ZL> &c, uint32_t t, float (&H)[3], TpIn &cur, TpIn &nxt, uint32_t r2_l0, uint32_t r2_l1, uint32_t r2_0,
                                         uint32_t h2, float NINF)
{
    const bool live = !GUARDED || ((int32_t)t >= c.t_in && (int32_t)t < c.t_end);
    if (live) {
        const float b0 = c.S[0], b1 = c.S[1], l0 = c.S[2], l1 = c.S[3];
        f32x2 ml, mb;
        ml = label_pair_max<M, ZL>(l1, b1, l0, b0, H[0], H[1], c.vz1, c.vz0);   // {lower, upper}
        mb[1] = cell_blank_max<M>(b1, l0, H[0]);
        mb[0] = cell_blank_max<M>(b0, H[0], H[2]);
        const f32x2 sl = ml + cur.E, sb = mb + f32x2{cur.e0, cur.e0};
        c.S = f32x4{sb[0], sb[1], sl[0], sl[1]};
        // The band is enforced by KILLING single cells, not by masking all of them: a cell above hi collects "leaked" scores
        // from the live cells under it and must hold -inf at the moment it enters the band (rule i: the positions
        // [hi(t), hi(t+1)) are killed after frame t); a cell that has dropped below lo was live in the last frame of the old
        // band, is still computed in the first frame of the new one and must be dead after it (rule ii: the positions
        // [lo(t-1), lo(t)) are killed after frame t) - from then on it only reads cells below itself, which are dead, and
        // stays -inf by itself.  `ev` marks the frames in which either range meets this tile (tp_band_block).
        if (__builtin_expect((c.ev >> F) & 1u, 0)) {
            asm volatile("" ::: "memory");
            // what to kill was worked out for the whole block (tp_band_block): first tile-relative position | count << 16
            const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)c.KL, F), k1 = (uint32_t)__builtin_amdgcn_readlane((int)c.KE, F + 1);
            for (uint32_t r = k2 & 0xffffu, e = r + (k2 >> 16); r < e; ++r) tp_kill(c.S, r, NINF);   // rule ii: left the band before this frame
            for (uint32_t r = k1 & 0xffffu, e = r + (k1 >> 16); r < e; ++r) tp_kill(c.S, r, NINF);   // rule i: enters it after this frame
        }
    }
    // the three cells below every lane's first cell, for frame t+1 (lane 0: from the packet of the tile below)
    // (the packet's first dword is not needed; it is kept alive up to here so that its register is not recycled - and
    //  the LDS read waited for - earlier)
    asm volatile("" : : "v"(nxt.hp));
    H[0] = wave_shr1(nxt.hp[3], c.S[3]);   // position base + 4 lane - 1 (label)
    H[1] = wave_shr1(nxt.hp[1], c.S[1]);   // - 2 (blank)
    H[2] = wave_shr1(nxt.hp[2], c.S[2]);   // - 3 (label)
    // publish the state after frame t = slot t+1 of the upper boundary (lane 63's four cells)
    // (staged: every lane drops its four cells into this frame's 1-KB row of the LDS staging area - no EXEC change and
    //  no vector-memory instruction per frame; lane 63's go out at the end of the block, tp_publish_block)
    if (live) *(__attribute__((address_space(3))) f32x4 *)(uintptr_t)(c.lds_stage + F * 16) = c.S;
    // LDS reads of frame t+2 (skipped frames read too: they prime the pipeline).  At the END of the frame, behind the branch
    // merge above: hipcc waits with lgkmcnt(0) at every merge (the band visit, the guarded frames), and with the reads at the
    // top of the frame that wait covered reads issued a dozen instructions earlier - every frame stalled for most of an LDS
    // round trip.  Down here the wait of the next frame finds reads that are a whole frame old.
    TpIn far;
    far.E = f32x2{lds_f32(r2_l0), lds_f32(r2_l1)};
    far.e0 = lds_f32(r2_0);
    far.hp = lds_f32x4(h2);
    cur = nxt;
    nxt = far;
}

// Band bookkeeping of the block that starts at frame tb (c.q0 / c.r0 are that frame's floor(L tb / T) and remainder): lane l
// works out floor(L t / T) for frame t = tb - 1 + l (l = 0 .. 33 are used) - x / T for x < 63 T < 2^32 by a float estimate
// and one correction each way, as in backtrace_rc_kernel - and, for the band step between its frame and the next one,
// which positions enter at the top (rule i) or leave at the bottom (rule ii) inside this tile (first position and count,
// packed).  A rule-i step after frame t is dealt with in frame t, a rule-ii step after frame t in frame t+1: bit F of c.ev <=>
// frame tb + F visits the band code, which reads its two kill words with v_readlane.  ~45 vector and a dozen scalar instructions per block; nothing per frame.
template <int M, bool ZL>
__device__ __forceinline__ void tp_band_block(TpTile<M, ZL> &c, uint32_t tb, int lane)
{
    const uint32_t l1 = lane > 0 ? (uint32_t)lane - 1u : 0u;
    const uint32_t x = c.r0 + l1 * c.dr;
    uint32_t qe = (uint32_t)((float)x * c.inv_T);
    qe -= (qe * c.T > x) ? 1u : 0u;
    qe += (x - qe * c.T >= c.T) ? 1u : 0u;
    uint32_t qa = c.q0 + l1 * c.dq + qe;
    const uint32_t q_before = tb == 0 ? c.q0 : (c.r0 >= c.dr ? c.q0 - c.dq : c.q0 - c.dq - 1u);   // frame tb-1 (block 0: no step into frame 0)
    qa = lane == 0 ? q_before : qa;
    // floor(L (t+1) / T): the lane above's value (DPP wave_shl:1; lane 63 keeps its own, it is not used)
    const uint32_t qn = (uint32_t)__builtin_amdgcn_update_dpp((int)qa, (int)qa, 0x130, 0xF, 0xF, false);
    const uint32_t tile_lo = (uint32_t)c.base, tile_hi = (uint32_t)c.base + kTpTile;
    const uint32_t lo_a = c.lo_of(qa), lo_n = c.lo_of(qn);
    const uint32_t hi_a = c.hi_of(lo_a), hi_n = c.hi_of(lo_n);
    const uint32_t t = tb - 1u + (uint32_t)lane;          // (lane 0 of block 0 wraps: its step is void, q_before == q0)
    // rule ii: [lo(t), lo(t+1)) within the tile, killed after frame t+1 (bit l of ev, word read from lane F of KL)
    const uint32_t la = lo_a > tile_lo ? lo_a : tile_lo, lb = lo_n < tile_hi ? lo_n : tile_hi;
    const bool leave = la < lb;
    c.KL = leave ? (la - tile_lo) | ((lb - la) << 16) : 0u;
    // rule i: [hi(t), hi(t+1)) within the tile, killed after frame t (bit l-1 of ev, word read from lane F+1 of KE)
    const uint32_t ea = hi_a > tile_lo ? hi_a : tile_lo, eb = hi_n < tile_hi ? hi_n : tile_hi;
    const bool enter = ea < eb && t + 1u < c.T;
    c.KE = enter ? (ea - tile_lo) | ((eb - ea) << 16) : 0u;
    const uint64_t b_leave = __builtin_amdgcn_ballot_w64(leave), b_enter = __builtin_amdgcn_ballot_w64(enter);
    c.ev = (uint32_t)b_leave | (uint32_t)(b_enter >> 1);
}
// one block further
template <int M, bool ZL>
__device__ __forceinline__ void tp_band_advance(TpTile<M, ZL> &c)
{
    c.q0 += c.dq32;
    c.r0 += c.dr32;
    if (c.r0 >= c.T) { c.r0 -= c.T; ++c.q0; }
}

// the frames of a block.  LDS byte addresses of this block's slot (A[0]) and the next one's (A[1]): row 0 + the lane's
// two label columns, row 0 itself (column 0 = blank), packet 0 - per block, so that a frame adds only an immediate offset
struct TpAddr {
    uint32_t l0, l1, r, h;
};
template <int M, bool ZL, int PITCH, bool GUARDED, int F>
__device__ __forceinline__ void tp_block_frames(TpTile<M, ZL> &c, uint32_t tb, float (&H)[3], TpIn &cur, TpIn &nxt, const TpAddr (&A)[2], float NINF)
{
    // frame t+2 = F+2 of this block, or F+2-16 of the next one
    constexpr int F2 = (F + 2) % kTpBlock, W = (F + 2) / kTpBlock;
    tp_frame<M, ZL, GUARDED, F>(c, tb + F, H, cur, nxt, A[W].l0 + F2 * PITCH, A[W].l1 + F2 * PITCH, A[W].r + F2 * PITCH, A[W].h + F2 * 16, NINF);
    if constexpr (F + 1 < kTpBlock) tp_block_frames<M, ZL, PITCH, GUARDED, F + 1>(c, tb, H, cur, nxt, A, NINF);
}

// end of a block: lane f < kTpBlock fetches what lane 63 staged in frame f and stores it as slot tb + f + 1 (one write-through
// store instruction for the block's packets = 512 contiguous bytes); frames the tile did not compute store nothing
template <int M, bool ZL>
__device__ __forceinline__ void tp_publish_block(TpTile<M, ZL> &c, uint32_t tb, int lane)
{
    const int32_t t = (int32_t)tb + lane;
    if (lane < kTpBlock && t >= c.t_in && t < c.t_end) {
        const f32x4 pk = lds_f32x4(c.lds_packets + (uint32_t)lane * 16u);
        // slot tb of the upper boundary (frame tb + f publishes slot tb + f + 1); worked out here, after the frames: two
        // scalar registers that are not live across them
        const char *out_block = c.halo_out + ((int64_t)tb - (int64_t)c.t_in) * 16;
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:16 sc1\n\ts_nop 1" : : "v"((uint32_t)lane * 16u), "v"(pk), "s"(out_block) : "memory");
    }
}

template <int M, bool ZL>
__device__ __forceinline__ void tp_checkpoint(TpTile<M, ZL> &c, uint32_t t_next /* multiple of 32 */)
{
    const f32x4 v = {c.S[0], c.S[2], c.S[1], c.S[3]};   // cells 0..3 in position order
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(c.ck_off), "v"(v), "s"(c.ck + ((size_t)(t_next / kCkFrames) - 1) * (size_t)c.ck_pitch) : "memory");
}

// ---------------------------------------------------------------------------------------
// one tile, all its frames
// PITCH = bytes between two rows of a staged block in LDS.  CONTIG = false (PITCH 256): rows are staged one by one
// (lane = column; any row stride of the caller's array).  CONTIG = true (PITCH = 4 V; the array's rows are contiguous, V
// columns): a block is copied as it lies in memory, 1 KB per LDS-DMA instruction - 4 (V = 64) or 3 (V = 39) instructions per block instead of 16; an LDS-DMA
// instruction costs the wave ~60 cycles to issue whatever it moves, and with 16 of them the per-block staging took
// longer than the block's frames.
// (tp2_run_tile in ka_tiled2.hpp; the 128-position tiles have a header of their own, ka_tiled_stream.hpp; the one-wavefront tile this header used to end with is gone:
// superseded by the two-wavefront tile in round 3 and removed in round 4.)

}  // namespace ka
