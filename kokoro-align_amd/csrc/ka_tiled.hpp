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
#include "ka_kernels.hpp"

namespace ka {

constexpr int kTpCells = 4;                    // cells per lane
constexpr int kTpTile = 64 * kTpCells;         // positions per tile
constexpr int kTpBlock = 32;                   // frames per staging block (= the checkpoint interval)
constexpr int kTpRing = 4;                     // LDS staging slots: the block being computed, the next one (landed), two more in flight
constexpr int kTpRowBytes = 256;               // LDS pitch of a staged row in the row-by-row staging mode (64 columns)
constexpr int kTpSlotBytes = kTpBlock * 256;   // LDS bytes of a staged block of rows (any mode)
constexpr int kTpStageBytes = 2048;            // publish staging: 512 B of lane 63's packets + the other lanes' scratch
constexpr uint32_t kTpSentinel = 0x7fc0deadu;  // verification fill of the halo region (a NaN: no score is ever NaN)
constexpr uint32_t kTpProgDone = 0x7fffffffu;  // progress word of a finished tile / of "no tile below"
static_assert(kCkFrames % kTpBlock == 0 && kTpBlock <= 32, "checkpoints fall on block ends; a block's packets are published by lanes 0..kTpBlock-1");

struct TileTask {
    int32_t lat;        // index into the launch's Lattice array
    int32_t tile;       // positions [256 tile, 256 tile + 256)
    int32_t t_in;       // first frame whose band reaches into the tile (hi(t) > 256 tile)
    int32_t t_end;      // first frame whose band has left it (lo(t) >= 256 (tile + 1)), or T
    int64_t halo_in;    // halo region byte offset of slot t_in of the boundary BELOW this tile (tile 0: the -inf region)
    int64_t halo_out;   // byte offset of slot t_in of the boundary ABOVE this tile (the top tile has one too: nobody reads it)
    int32_t fill_end;   // last slot of the upper boundary that the tile above reads (its t_end - 1)
    int32_t prog_in;    // progress word of the tile below (word 0 holds kTpProgDone: nothing below tile 0)
    int32_t prog_out;   // progress word of this tile
    int32_t below_end;  // t_end of the tile below (tile 0: INT32_MAX): the slots behind it hold -inf by construction (ka_tiled_narrow.hpp uses it)
};
// per lattice, zeroed before every launch: terminal state by 64-bit atomicMax, arrival counter of the last-frame tiles
struct TileAux {
    unsigned long long best;   // (end position + 1) << 32 | score bits; 0 = no live state
    uint32_t arrived;
    uint32_t pad;
};

typedef uint32_t KA_GLOBAL *gu32w_t;

__device__ __forceinline__ float lds_f32(uint32_t addr) { return *(const __attribute__((address_space(3))) float *)(uintptr_t)addr; }
__device__ __forceinline__ f32x4 lds_f32x4(uint32_t addr) { return *(const __attribute__((address_space(3))) f32x4 *)(uintptr_t)addr; }
// lane i <- lane i-1; lane 0 keeps `first` (DPP wave_shr:1, invalid source lanes keep the old value)
__device__ __forceinline__ float wave_shr1(float first, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, x), 0x138, 0xF, 0xF, false));
}
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
struct TpStats {
    unsigned long long phase[3];   // shader cycles: (wait | check+sum << 32), (progress | requests << 32), (publish+checkpoint)
    unsigned long long wait_ticks, total_ticks, spins, start_tick;   // 100 MHz ticks (ka_engine_set_verify(4): ka_debug_tile_stats)
};
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
        ml[1] = cell_label_max<M, ZL>(l1, b1, l0, b0, c.vz1);
        mb[1] = cell_blank_max<M>(b1, l0, H[0]);
        ml[0] = cell_label_max<M, ZL>(l0, b0, H[0], H[1], c.vz0);
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
// ---------------------------------------------------------------------------------------
// PITCH = bytes between two rows of a staged block in LDS.  CONTIG = false (PITCH 256): rows are staged one by one
// (lane = column; any row stride of the caller's array).  CONTIG = true (PITCH = 4 V; the array's rows are contiguous, V
// columns): a block is copied as it lies in memory, 1 KB per LDS-DMA instruction - 4 (V = 64) or 3 (V = 39) instructions per block instead of 16; an LDS-DMA
// instruction costs the wave ~60 cycles to issue whatever it moves, and with 16 of them the per-block staging took
// longer than the block's frames.
template <int M, bool ZL, int PITCH, bool CONTIG>
__device__ __forceinline__ void tp_run_tile(const Lattice &d, const TileTask &tk, int32_t *meta, char *halo, gu32w_t prog, TileAux *aux,
                                            uint32_t lds_rows, uint32_t lds_halo, int verify, TpStats *stats_out)
{
    const uint32_t stat_lds = lds_halo + kTpRing * kTpBlock * 16 + 16 + kTpStageBytes;   // diagnostic words behind the staging areas
    if (threadIdx.x < 8) ((__attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds)[threadIdx.x] = 0;
    // (ka_engine_set_verify(4)) shader cycles per phase of the block loop: [3] wait for the staged block, [4] its check + finiteness
    // sum, [5] progress store + poll, [6] requests (LDS-DMA issue), [7] publish + checkpoint
    unsigned long long ph = 0;
    auto phase = [&](int w) {
        if (verify & 4) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (w >= 0) ((__attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds)[w] += (uint32_t)(now - ph);
            ph = now;
        }
    };
    if (verify & 4) {   // start stamps: wall clock (100 MHz) and shader clock
        stats_out->start_tick = (unsigned long long)wall_clock64();
        stats_out->total_ticks = __builtin_amdgcn_s_memtime();
    }
    const int lane = threadIdx.x;
    const float NINF = ninf();
    TpTile<M, ZL> c;
    c.T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    c.L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    c.B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    c.dq = c.L / c.T;
    c.dr = c.L % c.T;
    c.base = __builtin_amdgcn_readfirstlane(tk.tile) * kTpTile;
    c.t_in = __builtin_amdgcn_readfirstlane(tk.t_in);
    c.t_end = __builtin_amdgcn_readfirstlane(tk.t_end);
    c.lp = reinterpret_cast<const char *>(d.lp);
    c.ld = (size_t)d.ld * 4;
    c.lane_off = (lane < d.V ? (uint32_t)lane : 0u) * 4u;
    c.halo_in = halo + tk.halo_in;
    c.halo_out = halo + tk.halo_out;
    c.prog_in = prog + tk.prog_in;
    c.prog_out = prog + tk.prog_out;
    c.ck = reinterpret_cast<char *>(d.bp);
    c.ck_pitch = (uint32_t)d.ck_pitch;
    c.ck_off = (((uint32_t)c.base + 4u * (uint32_t)lane) & (uint32_t)d.ck_mask) * 4u;
    c.lds_rows = lds_rows;
    c.lds_halo = lds_halo;
    // publish staging: frame F of a block drops every lane's four cells at lds_stage + 16 F - lane 63's (the packet) into
    // the packet row, the other lanes' into scratch behind it (distinct addresses: no EXEC change, no bank conflict)
    c.lds_packets = lds_halo + kTpRing * kTpBlock * 16 + 16;   // (16 bytes of progress looks sit in between)
    c.lds_stage = lane == 63 ? c.lds_packets : c.lds_packets + kTpBlock * 16 + (uint32_t)lane * 16u;
    static_assert(kTpBlock * 16 + 62 * 16 + (kTpBlock - 1) * 16 + 16 <= kTpStageBytes, "publish staging");
    // band bookkeeping (64-bit divisions once per tile; wave-uniform): the tile's first block, and one block's advance
    const auto uni = [](uint64_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); };
    {
        const uint64_t x = (uint64_t)c.L * (uint64_t)((uint32_t)c.t_in / kTpBlock * kTpBlock);
        c.q0 = uni(x / c.T);
        c.r0 = uni(x % c.T);
        c.dq32 = uni(((uint64_t)c.L * kTpBlock) / c.T);
        c.dr32 = uni(((uint64_t)c.L * kTpBlock) % c.T);
        c.inv_T = 1.0f / (float)c.T;
        c.ev = 0;
        c.KL = c.KE = 0;
    }
    // labels of the lane's two label cells (positions base + 4 lane + 1, + 3); labx is zero padded past S
    {
        gci32_t labx = (gci32_t)d.labx + ((size_t)c.base >> 1) + 2 * (size_t)lane;
        c.la0 = labx[0];
        c.la1 = labx[1];
        c.vz0 = (ZL && c.la0 == 0) ? NINF : __builtin_inff();
        c.vz1 = (ZL && c.la1 == 0) ? NINF : __builtin_inff();
    }
    // state before frame t_in: nothing of the tile is live, except the virtual start state (align.py:57-58)
    c.S = f32x4{NINF, NINF, NINF, NINF};
    if (c.base == 0 && c.t_in == 0 && lane == 0) c.S[0] = 0.0f;
    c.absum = 0.0f;
    tp_halo_store<0>(c.halo_out, c.S, 1ull << 63);   // slot t_in: the state before the tile's first frame

    // ---- staging: block k = frames [16 k, 16 k + 16).  A block's 16 log-prob rows, its 16 halo packets and a look at the
    // progress word of the tile below are fetched by LDS-DMA (global_load_lds: memory -> LDS, no register in between)
    // THREE iterations before the block is computed and are waited for one iteration before: two blocks of latency
    // cover (a block of frames is ~0.7 us of work, a row from HBM ~0.4-1 us away, a write-through packet further),
    // four LDS slots.  Nothing in flight lives in a register, so none of the hazards of asm loads applies here.
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) char *lchar_t;
    const uint32_t last_row = c.T - 1;
    const uint32_t last_slot = (uint32_t)c.t_end - 1;     // this tile reads slots t_in .. t_end - 1
    const uint32_t lds_poll = c.lds_halo + kTpRing * kTpBlock * 16;
    auto ring = [](int32_t k) { return (uint32_t)((k % kTpRing + kTpRing) % kTpRing); };
    constexpr int kRowDmas = !CONTIG ? kTpBlock : (kTpBlock * PITCH + 1023) / 1024;   // LDS-DMA instructions per block of rows
    static_assert(CONTIG || PITCH == kTpRowBytes, "row-by-row staging uses 256-byte rows");
    auto issue_block = [&](int32_t k) {    // k >= 0
        const uint32_t tb = (uint32_t)k * kTpBlock, slot = ring(k);
        lchar_t dst = (lchar_t)(uintptr_t)(c.lds_rows + slot * kTpSlotBytes);
        if constexpr (!CONTIG) {
            const char *rp = c.lp + (size_t)(tb < last_row ? tb : last_row) * c.ld;   // wave-uniform; the lane's column is a 32-bit offset
            if (tb + kTpBlock <= c.T) {      // every row of the block exists (nothing in flight lives in a register: two paths are fine here)
#pragma unroll
                for (int f = 0; f < kTpBlock; ++f) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(rp + c.lane_off), (lptr_t)(dst + f * kTpRowBytes), 4, 0, 0);   // row tb+f: lane = column, 256 B
                    rp += c.ld;
                }
            } else {                         // the lattice's last rows, and blocks requested past them: stop at row T-1
#pragma unroll
                for (int f = 0; f < kTpBlock; ++f) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(rp + c.lane_off), (lptr_t)(dst + f * kTpRowBytes), 4, 0, 0);
                    rp += tb + f < last_row ? c.ld : 0;
                }
            }
        } else {
            // the block as it lies in memory: 16 rows x PITCH bytes from row tb on, 16 bytes per lane and instruction;
            // past the lattice's last row the lanes re-read its last 16 aligned bytes (rows that do not exist are not used)
            const uint32_t first = tb < last_row ? tb : last_row;
            const uint32_t rows_there = c.T - first < (uint32_t)kTpBlock ? c.T - first : (uint32_t)kTpBlock;
            const uint32_t last_chunk = (rows_there * PITCH - 16u) & ~15u;
            const char *bp = c.lp + (size_t)first * PITCH;
#pragma unroll
            for (int j = 0; j < kRowDmas; ++j) {
                uint32_t off = (uint32_t)j * 1024u + (uint32_t)lane * 16u;
                off = off < last_chunk ? off : last_chunk;
                __builtin_amdgcn_global_load_lds((gptr_t)(bp + off), (lptr_t)(dst + j * 1024), 16, 0, 0);
            }
        }
        if (lane < kTpBlock) {       // the block's slots of the lower boundary (clamped to what exists): one lane per slot, 16 B each, write-through data: sc1
            uint32_t s = tb + (uint32_t)lane;
            s = s < (uint32_t)c.t_in ? (uint32_t)c.t_in : (s > last_slot ? last_slot : s);
            __builtin_amdgcn_global_load_lds((gptr_t)(c.halo_in + (size_t)(s - (uint32_t)c.t_in) * 16), (lptr_t)(lchar_t)(uintptr_t)(c.lds_halo + slot * (kTpBlock * 16)), 16, 0, 16);
        }
        if (lane == 0) __builtin_amdgcn_global_load_lds((gptr_t)c.prog_in, (lptr_t)(lchar_t)(uintptr_t)(lds_poll + slot * 4), 4, 0, 16);
    };
    constexpr int kIssued = kRowDmas + 2;   // vector-memory instructions of one issue_block
    // everything issued by the issue_block of TWO iterations ago has landed once at most `younger` younger operations
    // are in flight (vmcnt is an in-order counter; never pass more than were really issued since)
    auto wait_landed = [&](uint32_t younger) {
        asm volatile("s_cmp_ge_u32 %0, %1+4\n\ts_cbranch_scc1 .Lka_w4_%=\n\t"
                     "s_cmp_ge_u32 %0, %1+3\n\ts_cbranch_scc1 .Lka_w3_%=\n\t"
                     "s_cmp_ge_u32 %0, %1+2\n\ts_cbranch_scc1 .Lka_w2_%=\n\t"
                     "s_cmp_ge_u32 %0, %1+1\n\ts_cbranch_scc1 .Lka_w1_%=\n\t"
                     "s_cmp_ge_u32 %0, %1\n\ts_cbranch_scc1 .Lka_w0_%=\n\t"
                     "s_waitcnt vmcnt(0)\n\ts_branch .Lka_we_%=\n"
                     ".Lka_w0_%=:\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lka_we_%=\n"
                     ".Lka_w1_%=:\n\ts_waitcnt vmcnt(%1+1)\n\ts_branch .Lka_we_%=\n"
                     ".Lka_w2_%=:\n\ts_waitcnt vmcnt(%1+2)\n\ts_branch .Lka_we_%=\n"
                     ".Lka_w3_%=:\n\ts_waitcnt vmcnt(%1+3)\n\ts_branch .Lka_we_%=\n"
                     ".Lka_w4_%=:\n\ts_waitcnt vmcnt(%1+4)\n"
                     ".Lka_we_%=:"
                     : : "s"(younger), "i"(kIssued) : "memory", "scc");
    };
    bool stale = false;
    // block k has landed in LDS: the finiteness sum over its rows, and (ka_engine_set_verify(1): the host filled the halo region
    // with a NaN pattern no score can have) no packet this tile is going to consume may still hold that pattern
    auto landed_block = [&](int32_t k) {
        const uint32_t slot = ring(k);
        // (all reads first, then the sum: written as one accumulation chain hipcc waited for every read in turn - 1700
        //  cycles per 16 frames, more than the frames themselves.  16 bytes per lane and read: every KB the staging wrote
        //  is log-probs - whole KBs are written, clamped to the lattice's last row - 8 or 5 reads instead of 32 or 20.)
        const uint32_t r = c.lds_rows + slot * kTpSlotBytes + (uint32_t)lane * 16u;
        constexpr int kReads = !CONTIG ? kTpSlotBytes / 1024 : kRowDmas;
        f32x4 v[kReads];
#pragma unroll
        for (int j = 0; j < kReads; ++j) v[j] = lds_f32x4(r + j * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < kReads; ++j) c.absum += (__builtin_fabsf(v[j][0]) + __builtin_fabsf(v[j][1])) + (__builtin_fabsf(v[j][2]) + __builtin_fabsf(v[j][3]));
        if (verify & 1) {
            const int32_t sidx = k * kTpBlock + (lane & (kTpBlock - 1));
            const f32x4 h = lds_f32x4(c.lds_halo + slot * (kTpBlock * 16) + (uint32_t)(lane & (kTpBlock - 1)) * 16u);
            const bool mine = lane < kTpBlock && sidx >= c.t_in && sidx < c.t_end;
            const bool bad = mine && (__builtin_bit_cast(uint32_t, h[1]) == kTpSentinel || __builtin_bit_cast(uint32_t, h[2]) == kTpSentinel ||
                                      __builtin_bit_cast(uint32_t, h[3]) == kTpSentinel);
            if (__builtin_amdgcn_ballot_w64(bad)) stale = true;
        }
    };
    // slots the tile below must have published before the halo packets of the block starting at frame tb may be fetched
    auto need_for = [&](int32_t k) {
        const uint32_t n = (uint32_t)(k + 1) * kTpBlock;
        return n < (uint32_t)c.t_end ? n : (uint32_t)c.t_end;
    };

    const int32_t kb0 = c.t_in / kTpBlock, kb1 = (c.t_end - 1) / kTpBlock;   // first and last block
    uint32_t tail1 = 0, tail2 = 0;   // stores issued behind the issue_block of the previous iteration / the one before (lower bounds)
    bool fed = true;
    TpIn cur = {f32x2{0.0f, 0.0f}, 0.0f, f32x4{NINF, NINF, NINF, NINF}}, nxt = cur;
    float H[3] = {NINF, NINF, NINF};
    // iteration it: block it+1 has landed (requested two iterations ago), block it+3 is requested, block it is computed.
    // Iterations kb0-3 .. kb0-1 only prime the pipeline.
    for (int32_t it = kb0 - 3; it <= kb1; ++it) {
        const uint32_t tb = (uint32_t)(it * kTpBlock);              // (wraps in the priming iterations of block 0: not used there)
        phase(-1);
        if (it >= kb0 - 1) {
            wait_landed((verify & 2) ? 0u : kIssued + tail1 + tail2);   // younger: one issue_block and the stores behind the last two
            phase(3);
            landed_block(it + 1);
            phase(4);
        }
        // retired by that wait: everything issued before the requests of iteration it-2, i.e. the packets of blocks <= it-3 = slots <= 16 (it-2)
        if (it >= kb0 + 3) tp_prog_store(c.prog_out, tb - 2 * kTpBlock + 1);
        if (it + 3 <= kb1 && fed) {
            // the freshest look at the progress word that has landed is the one requested two iterations ago (with block it+1)
            const uint32_t have = it >= kb0 - 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, lds_f32(lds_poll + ring(it + 1) * 4))) : 0u;
            fed = tp_wait_progress(c.prog_in, need_for(it + 3), need_for(it + 5), have, stat_lds);
        }
        phase(5);
        if (it + 3 >= 0) issue_block(it + 3);
        phase(6);
        tail2 = tail1;
        tail1 = 0;
        if (it < kb0) continue;
        const uint32_t slot = ring(it), nslot = ring(it + 1);
        // (LDS addresses live in vector registers: say so once per block instead of a v_mov per read)
        uint32_t rc = c.lds_rows + slot * kTpSlotBytes, rn = c.lds_rows + nslot * kTpSlotBytes;
        uint32_t hc = c.lds_halo + slot * (kTpBlock * 16), hn = c.lds_halo + nslot * (kTpBlock * 16);
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(rc), "=&v"(rn), "=&v"(hc), "=&v"(hn) : "s"(rc), "s"(rn), "s"(hc), "s"(hn));
        TpAddr A[2] = {{rc + (uint32_t)c.la0, rc + (uint32_t)c.la1, rc, hc}, {rn + (uint32_t)c.la0, rn + (uint32_t)c.la1, rn, hn}};
        asm volatile("" : "+v"(A[0].l0), "+v"(A[0].l1), "+v"(A[1].l0), "+v"(A[1].l1));   // (keep the four sums: no re-add per frame)
        if (it == kb0) {
            // prime the two-frame read pipeline at the block's frames 0 and 1; when the tile starts later in the block,
            // the skipped frames in front of it shift the pipeline along (and take H afresh) exactly like computed ones
            cur.E = f32x2{lds_f32(A[0].l0), lds_f32(A[0].l1)};
            cur.e0 = lds_f32(A[0].r);
            nxt.E = f32x2{lds_f32(A[0].l0 + PITCH), lds_f32(A[0].l1 + PITCH)};
            nxt.e0 = lds_f32(A[0].r + PITCH);
            nxt.hp = lds_f32x4(A[0].h + 16);
            const f32x4 hp = lds_f32x4(A[0].h);
            H[0] = wave_shr1(hp[3], c.S[3]);
            H[1] = wave_shr1(hp[1], c.S[1]);
            H[2] = wave_shr1(hp[2], c.S[2]);
        }
        tp_band_block(c, tb, lane);
        const bool partial = (int32_t)tb < c.t_in || (int32_t)(tb + kTpBlock) > c.t_end;
        if (!partial) {
            const unsigned long long fr0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            tp_block_frames<M, ZL, PITCH, false, 0>(c, tb, H, cur, nxt, A, NINF);
            if (verify & 4) ((__attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds)[2] += (uint32_t)(__builtin_amdgcn_s_memtime() - fr0);
            phase(-1);
            tp_publish_block(c, tb, lane);
            tail1 = 1;
            if ((tb + kTpBlock) % kCkFrames == 0 && tb + kTpBlock < c.T) {
                tp_checkpoint(c, tb + kTpBlock);
                ++tail1;
            }
            phase(7);
        } else {
            tp_block_frames<M, ZL, PITCH, true, 0>(c, tb, H, cur, nxt, A, NINF);
            tp_publish_block(c, tb, lane);
            if ((tb + kTpBlock) % kCkFrames == 0 && (int32_t)(tb + kTpBlock) <= c.t_end && tb + kTpBlock < c.T) tp_checkpoint(c, tb + kTpBlock);
            // (a partial block issued an unknown number of stores: its count stays 0, a lower bound, and the waits that
            //  cover it wait for a store or two more than they must)
        }
        tp_band_advance(c);
    }
    // drain the staging loads still in flight (their registers are dead to the compiler after the loop and would be
    // reused while a load can still land in them).  No register operands here: nothing reads those registers again,
    // and tying them in makes hipcc merge the loop-exit paths with copies of in-flight registers.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- finiteness (as forward_ck: the scores-only form is valid for finite log-probs of sane magnitude).  Flagged
    // before the tile reports itself done, so that whoever closes the lattice sees the flag.
    int32_t *m = meta_of(meta, d.idx);
    if ((!fed || stale) && lane == 0) atomicMin(&m[0], kStatusInternal);
    const uint32_t abits = __builtin_bit_cast(uint32_t, c.absum) & 0x7fffffffu;
    if (__builtin_amdgcn_ballot_w64(abits > 0x7f800000u)) {
        if (lane == 0) atomicMin(&m[0], kStatusNaN);
    } else if (__builtin_amdgcn_ballot_w64(abits >= __builtin_bit_cast(uint32_t, 1e30f))) {
        if (lane == 0) atomicOr(&m[2], d.W <= kFastMaxBand ? kFlagExact : kFlagDeclined);
    }
    // ---- hand the rest of the upper boundary over: after t_end the whole tile is below the band = -inf ----
    {
        const f32x4 dead = {NINF, NINF, NINF, NINF};
        for (int64_t s = (int64_t)c.t_end + 1 + lane; s <= (int64_t)tk.fill_end; s += 64)
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"((uint32_t)((s - c.t_in) * 16)), "v"(dead), "s"(c.halo_out) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tp_prog_store(c.prog_out, kTpProgDone);
    }
    if ((verify & 4) && lane == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const __attribute__((address_space(3))) uint32_t *sw = (const __attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds;
        TpStats st;
        st.spins = sw[0] | ((unsigned long long)((xcc & 0xf) << 16 | (hw & 0xffff))) << 32;   // where the tile ran: XCC, SE/SH/CU/SIMD/wave slot
        st.phase[0] = sw[3] | ((unsigned long long)sw[4] << 32);
        st.phase[1] = sw[5] | ((unsigned long long)sw[6] << 32);
        st.phase[2] = sw[7];
        st.wait_ticks = sw[1] | ((unsigned long long)sw[2] << 32);   // (high half: shader cycles inside the unguarded frame blocks)
        st.start_tick = __builtin_amdgcn_s_memtime() - stats_out->total_ticks;   // (the tile's shader cycles)
        st.total_ticks = wall_clock64() - stats_out->start_tick;
        *stats_out = st;
    }
    // ---- terminal state: the HIGHEST live position of frame T-1 (align.py:99-101), over the tiles alive then ----
    if ((uint32_t)c.t_end == c.T) {
        TpMasks mk;   // (the only full band mask of a tile's life: cells above hi may hold leaked scores)
        const uint32_t q_last = c.L - (c.L + c.T - 1u) / c.T;   // floor(L (T-1) / T) = L - ceil(L / T)
        const uint32_t lo_last = c.lo_of(q_last), hi_last = c.hi_of(lo_last);
        tp_masks(mk, (int32_t)lo_last - c.base, (int32_t)hi_last - c.base);
        tp_mask_state(c.S, mk, NINF);
        const float cell[4] = {c.S[0], c.S[2], c.S[1], c.S[3]};
        unsigned long long key = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cell[k] != NINF) key = ((unsigned long long)(uint32_t)(c.base + 4 * lane + k + 1) << 32) | __builtin_bit_cast(uint32_t, cell[k]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off);
            key = o > key ? o : key;
        }
        if (lane == 0) {
            TileAux *a = aux + d.idx;
            if (key) atomicMax(&a->best, key);
            __threadfence();
            const uint32_t n = atomicAdd(&a->arrived, 1u) + 1u;
            if (n == (uint32_t)d.n_final) {
                __threadfence();
                const unsigned long long best = atomicMax(&a->best, 0ull);
                const int fl = atomicOr(&m[2], 0);
                if (fl & (kFlagExact | kFlagDeclined)) {
                    m[1] = -1;   // declined: the exact kernels redo the lattice (or nobody does: KA_ERR_NONFINITE)
                } else if (best == 0) {
                    m[1] = -1;
                    atomicMin(&m[0], kStatusEmptyBeam);
                } else {
                    m[1] = (int32_t)(best >> 32) - 1;
                    m[3] = (int32_t)(uint32_t)best;
                }
            }
        }
    }
}

// One workgroup (one wavefront) per tile.  A workgroup asks for 40 KB of LDS although it uses 13: at most four fit on
// a CU, i.e. one per SIMD (tools/ubench/census.hip), and the rest of the grid waits in the dispatcher for a tile to
// finish.  The tile a workgroup runs is drawn from a ticket counter, not from its index: tasks are sorted by first frame,
// so whatever order the dispatcher starts workgroups in, a tile's producer holds an earlier ticket and is running or done
// - the earliest unfinished ticket can always run to completion.
constexpr unsigned kTpLdsRequest = 40 * 1024;   // used: 32 KB rows + 2 KB packets + 2 KB publish staging
static_assert(kTpRing * kTpSlotBytes + kTpRing * kTpBlock * 16 + 16 + kTpStageBytes + 64 <= kTpLdsRequest, "LDS budget");
template <int M, int PITCH, bool CONTIG>
__global__ __launch_bounds__(64) void forward_tp_kernel(const Lattice *__restrict__ lats, const TileTask *__restrict__ tasks, int n_tasks,
                                                        int32_t *meta, char *halo, uint32_t *prog, TileAux *aux, uint32_t *ticket, int verify, TpStats *stats)
{
    extern __shared__ __attribute__((aligned(16))) char tp_lds[];
    const uint32_t lds_rows = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&tp_lds[0];
    const uint32_t lds_halo = lds_rows + kTpRing * kTpSlotBytes;
    uint32_t tix = 0;
    if (threadIdx.x == 0) tix = atomicAdd(ticket, 1u);
    tix = (uint32_t)__builtin_amdgcn_readfirstlane((int)tix);
    if (tix >= (uint32_t)n_tasks) return;
    const TileTask &tk = tasks[tix];
    const Lattice &d = lats[__builtin_amdgcn_readfirstlane(tk.lat)];
    const int flags = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2]);
    if (flags & kFlagZeroLabel)
        tp_run_tile<M, true, PITCH, CONTIG>(d, tk, meta, halo, (gu32w_t)prog, aux, lds_rows, lds_halo, verify, stats + tix);
    else
        tp_run_tile<M, false, PITCH, CONTIG>(d, tk, meta, halo, (gu32w_t)prog, aux, lds_rows, lds_halo, verify, stats + tix);
}

}  // namespace ka
