// ka_tiled_narrow.hpp — the tile pipeline with tiles of 128 positions: TWO cells per lane, and up to three wavefronts per tile.
//
// Why: a wavefront that is alone on its SIMD does not issue a vector instruction every four cycles but every ~7 (timing
// variants of the 256-position frame, DESIGN.md section 8: four more s_nop per frame cost 3.4 cycles each, four more
// independent v_max 7.3 each), so a lone lattice's chain of tiles runs at the pace of the frame's INSTRUCTION COUNT.  With
// two cells per lane a frame has one v_max3 for the blank, a v_max and a v_max3 for the label, one v_pk_add_f32 and three DPP
// moves, where the 256-position frame has twice the maxima and adds.  The price: twice as many tiles in the chain, i.e.
// twice the hand-offs and twice the workgroups - which is why the engine uses this form only while the launch's tiles (almost)
// fit the device at once (ka_engine.hip: narrow_tiles_pay).
//
// Two forms:
//   GATHER = true (the default): three wavefronts per tile.  The look-up wavefront turns the staged rows into per-lane
//     emission pairs with the band's kills folded in as -inf; the compute wavefront's frame is 11 instructions with no band
//     code; the feeder keeps the memory protocol.  See the GATHER block below.
//   GATHER = false: two wavefronts as in ka_tiled2.hpp; the compute wavefront reads the staged rows itself and visits the band
//     code (frame of 14 instructions).  Behind ka_debug_set_tile_gather(engine, 0).
// Common to both, and different from ka_tiled2.hpp: the frame loop's LDS instructions are inline asm with hand-counted waits
// (below), the feeder announces a block BEFORE it polls the tile below, and the slots behind the lower tile's t_end are one
// -inf slot (TileTask::below_end) instead of a fill after the loop.  Staging, sc1 packets + progress words, checkpoints
// (position p of row r at r * ck_pitch + (p & ck_mask) * 4 whatever the tile width) are ka_tiled2.hpp's; the host plans the
// tiles with the same formulas for 128 positions.
#pragma once
#include "ka_tiled2.hpp"

namespace ka {


template <int M, bool ZL>
struct TnTile {
    uint32_t T, L, B, dq, dr;
    uint32_t q0, r0, dq32, dr32, ev;
    uint32_t KL, KE;
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
    const char *halo_in;
    char *halo_out;
    gu32w_t prog_in, prog_out;
    char *ck;
    uint32_t ck_pitch;
    uint32_t ck_off;        // per lane: ((base + 2 lane) & ck_mask) * 4
    f32x2 S;                // {blank at base + 2 lane, label at base + 2 lane + 1}
    int la0;                // 4 * label of the lane's label cell
    float vz0;
    float absum;
    uint32_t lds_rows, lds_halo;
    uint32_t lds_stage;
    uint32_t lds_packets;
};
struct TnIn {
    float E;      // emission of the label cell
    float e0;     // blank emission
    f32x4 hp;     // packet of the tile below: its top four cells in position order {top-3, top-2, top-1, top}
};

// -inf into the cell at tile-relative position rel (0..127): cell rel & 1 of lane rel >> 1
__device__ __forceinline__ void tn_kill(f32x2 &S, uint32_t rel, float NINF)
{
    const uint64_t m = 1ull << (rel >> 1);
    float c0 = S[0], c1 = S[1];
    asm volatile("s_bitcmp1_b32 %[rel], 0\n\t"
                 "s_cbranch_scc1 .Lka_n1_%=\n\t"
                 "v_cndmask_b32 %[c0], %[c0], %[ninf], %[m]\n\t"
                 "s_branch .Lka_ne_%=\n"
                 ".Lka_n1_%=:\n\t"
                 "v_cndmask_b32 %[c1], %[c1], %[ninf], %[m]\n"
                 ".Lka_ne_%=:"
                 : [c0] "+v"(c0), [c1] "+v"(c1)
                 : [rel] "s"(rel), [m] "s"(m), [ninf] "v"(NINF)
                 : "scc");
    S = f32x2{c0, c1};
}

// tp_band_block for a tile of 128 positions
template <int M, bool ZL>
__device__ __forceinline__ void tn_band_block(TnTile<M, ZL> &c, uint32_t tb, int lane)
{
    const uint32_t l1 = lane > 0 ? (uint32_t)lane - 1u : 0u;
    const uint32_t x = c.r0 + l1 * c.dr;
    uint32_t qe = (uint32_t)((float)x * c.inv_T);
    qe -= (qe * c.T > x) ? 1u : 0u;
    qe += (x - qe * c.T >= c.T) ? 1u : 0u;
    uint32_t qa = c.q0 + l1 * c.dq + qe;
    const uint32_t q_before = tb == 0 ? c.q0 : (c.r0 >= c.dr ? c.q0 - c.dq : c.q0 - c.dq - 1u);
    qa = lane == 0 ? q_before : qa;
    const uint32_t qn = (uint32_t)__builtin_amdgcn_update_dpp((int)qa, (int)qa, 0x130, 0xF, 0xF, false);
    const uint32_t tile_lo = (uint32_t)c.base, tile_hi = (uint32_t)c.base + kTnTile;
    const uint32_t lo_a = c.lo_of(qa), lo_n = c.lo_of(qn);
    const uint32_t hi_a = c.hi_of(lo_a), hi_n = c.hi_of(lo_n);
    const uint32_t t = tb - 1u + (uint32_t)lane;
    const uint32_t la = lo_a > tile_lo ? lo_a : tile_lo, lb = lo_n < tile_hi ? lo_n : tile_hi;
    const bool leave = la < lb;
    c.KL = leave ? (la - tile_lo) | ((lb - la) << 16) : 0u;
    const uint32_t ea = hi_a > tile_lo ? hi_a : tile_lo, eb = hi_n < tile_hi ? hi_n : tile_hi;
    const bool enter = ea < eb && t + 1u < c.T;
    c.KE = enter ? (ea - tile_lo) | ((eb - ea) << 16) : 0u;
    const uint64_t b_leave = __builtin_amdgcn_ballot_w64(leave), b_enter = __builtin_amdgcn_ballot_w64(enter);
    c.ev = (uint32_t)b_leave | (uint32_t)(b_enter >> 1);
}
template <int M, bool ZL>
__device__ __forceinline__ void tn_band_advance(TnTile<M, ZL> &c)
{
    c.q0 += c.dq32;
    c.r0 += c.dr32;
    if (c.r0 >= c.T) { c.r0 -= c.T; ++c.q0; }
}

// The frame loop's LDS traffic goes through asm statements the compiler cannot see into, with the waits written by hand:
// hipcc waits with lgkmcnt(0) at every branch merge (the band visit is one), i.e. every frame waited for the reads it had
// just issued and took an LDS round trip (~120 cycles) however few instructions it had.  Here the reads run FOUR frames
// ahead, every frame issues exactly four LDS instructions in a fixed order (staging write, label emission, blank emission,
// packet), and the frame that needs the packet of frame t+1 waits with lgkmcnt(8): the eight instructions of the two
// frames in between may still be in flight.  (asm volatile statements keep their order; tools/lint_inflight.py checks that no
//  register is copied between its read and its wait.)
template <int OFF>
__device__ __forceinline__ float tn_lds_f32(uint32_t addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 tn_lds_f32x4(uint32_t addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// rows / packets of frame (block frame) G of the blocks A[0], A[1] describe
template <int PITCH, int G>
__device__ __forceinline__ void tn_read(TnIn &in, const TpAddr (&A)[2])
{
    constexpr int W = G / kTpBlock, R = G % kTpBlock;
    in.E = tn_lds_f32<R * PITCH>(A[W].l0);
    in.e0 = tn_lds_f32<R * PITCH>(A[W].r);
    in.hp = tn_lds_f32x4<R * 16>(A[W].h);
}
template <int N>
__device__ __forceinline__ void tn_wait(TnIn &in)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(in.E), "+v"(in.e0), "+v"(in.hp) : "n"(N));
}
// everything in flight has landed: after the priming reads, and at the end of a block (the barrier that follows waits anyway;
// said here, the four frames read ahead are not "in flight" in the straight-line view of tools/lint_inflight.py either)
__device__ __forceinline__ void tn_wait_all(TnIn (&in)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(in[0].E), "+v"(in[0].e0), "+v"(in[0].hp), "+v"(in[1].E), "+v"(in[1].e0), "+v"(in[1].hp), "+v"(in[2].E), "+v"(in[2].e0),
                   "+v"(in[2].hp), "+v"(in[3].E), "+v"(in[3].e0), "+v"(in[3].hp));
}

// One frame (see tp_frame).  H[0..2] = the three cells below the lane's blank: label and blank of the lane below, label of
// the lane below that.  in[0] = frame t's emissions, in[1] = frame t+1's (its packet is needed here), in[2], in[3]: in flight.
template <int M, bool ZL, int PITCH, bool GUARDED, int F>
__device__ __forceinline__ void tn_frame(TnTile<M, ZL> &c, uint32_t t, float (&H)[3], TnIn (&in)[4], const TpAddr (&A)[2], float NINF)
{
    const bool live = !GUARDED || ((int32_t)t >= c.t_in && (int32_t)t < c.t_end);
    if (live) {
        const float b = c.S[0], l = c.S[1];
        const float ml = cell_label_max<M, ZL>(l, b, H[0], H[1], c.vz0);
        const float mb = cell_blank_max<M>(b, H[0], H[2]);
        c.S = f32x2{mb, ml} + f32x2{in[0].e0, in[0].E};
        if (__builtin_expect((c.ev >> F) & 1u, 0)) {   // the band moves over this tile in this frame (tp_frame explains the two rules)
            asm volatile("" ::: "memory");
            const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)c.KL, F), k1 = (uint32_t)__builtin_amdgcn_readlane((int)c.KE, F + 1);
            for (uint32_t r = k2 & 0xffffu, e = r + (k2 >> 16); r < e; ++r) tn_kill(c.S, r, NINF);   // rule ii
            for (uint32_t r = k1 & 0xffffu, e = r + (k1 >> 16); r < e; ++r) tn_kill(c.S, r, NINF);   // rule i
        }
    }
    tn_wait<8>(in[1]);
    H[0] = wave_shr1(in[1].hp[3], c.S[1]);
    H[1] = wave_shr1(in[1].hp[2], c.S[0]);
    H[2] = wave_shr1(in[1].hp[1], H[0]);
    // the packet of slot t+1 = the top four cells of the tile in position order: lane 62's pair, then lane 63's.  Every lane
    // drops its pair as it lies in its registers (no EXEC change, no copies): lanes 62 and 63 into the two halves of this
    // frame's packet, the others into scratch behind the packet row.  (Skipped frames write too: the wait above counts on
    // four LDS instructions per frame; nobody publishes their rows.)
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(c.lds_stage), "v"(c.S), "n"(F * 16) : "memory");
    TnIn far;
    tn_read<PITCH, F + 4>(far, A);
    in[0] = in[1];
    in[1] = in[2];
    in[2] = in[3];
    in[3] = far;
}
template <int M, bool ZL, int PITCH, bool GUARDED, int F>
__device__ __forceinline__ void tn_block_frames(TnTile<M, ZL> &c, uint32_t tb, float (&H)[3], TnIn (&in)[4], const TpAddr (&A)[2], float NINF)
{
    tn_frame<M, ZL, PITCH, GUARDED, F>(c, tb + F, H, in, A, NINF);
    if constexpr (F + 1 < kTpBlock) tn_block_frames<M, ZL, PITCH, GUARDED, F + 1>(c, tb, H, in, A, NINF);
}
// ---------------------------------------------------------------------------------------
// GATHER form: the FEEDER also looks up the emissions.  For every frame of block it+1 it reads the blank's and the lane's
// label's log-prob from the staged rows and leaves them as a pair {blank, label} per lane in LDS - with -inf in place of
// the emission of every cell the band code would kill after that frame (a sum with -inf IS the kill: rule i / rule ii of
// tp_frame become writes of single words, worked out for the 32 frames of the block at once, lane = frame).  The compute
// wavefront's frame is then: wait, staging write, 7 vector instructions, ONE 8-byte read (which is the second operand of
// the v_pk_add_f32 as it lands) and the packet read - 11 instructions instead of 14, no band code, no branch.  Its reads do
// not cross into the next block (whose pairs the feeder is writing during this iteration): four frames are primed at the
// start of a block, the last four frames read nothing, and the hand-written waits count accordingly.
// LDS: 2 x 16 KB of pairs; the compute wavefront no longer reads rows, so two row slots do (the block being looked up and
// the one landing), and two packet slots - 46.1 KB per workgroup for V = 39, 52.1 KB for V = 64: three per CU.
// ---------------------------------------------------------------------------------------
static_assert(TnLds<256, true, false>::kTotal <= (int)kTpLdsRequest, "LDS budget of the narrow tile");
static_assert(3 * ((TnLds<256, true, true>::kTotal + 511) / 512 * 512) <= 160 * 1024 && 3 * ((TnLds<256, false, true>::kTotal + 511) / 512 * 512) <= 160 * 1024,
              "three look-up workgroups per CU");

struct TgIn {
    f32x2 e;      // {blank emission, label emission} of the lane's two cells, -inf where the cell dies after the frame
    f32x4 hp;     // packet of the tile below
};
template <int G>
__device__ __forceinline__ void tg_read(TgIn &in, uint32_t pairs, uint32_t packets)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(in.e) : "v"(pairs), "n"(G * 512));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(in.hp) : "v"(packets), "n"(G * 16));
}
template <int N>
__device__ __forceinline__ void tg_wait(TgIn &in)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(in.e), "+v"(in.hp) : "n"(N));
}
__device__ __forceinline__ void tg_wait_all(TgIn (&in)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(in[0].e), "+v"(in[0].hp), "+v"(in[1].e), "+v"(in[1].hp), "+v"(in[2].e), "+v"(in[2].hp), "+v"(in[3].e), "+v"(in[3].hp));
}
// LDS instructions frame F of a block issues: the staging write, and the two reads of frame F+4 while that is in the block
constexpr int tg_ops(int F) { return 1 + (F + 4 < kTpBlock ? 2 : 0); }
template <int M, bool ZL, bool GUARDED, int F>
__device__ __forceinline__ void tg_frame(TnTile<M, ZL> &c, uint32_t t, float (&H)[3], TgIn (&in)[4], uint32_t pairs, uint32_t packets, float NINF)
{
    const bool live = !GUARDED || ((int32_t)t >= c.t_in && (int32_t)t < c.t_end);
    if (live) {
        const float b = c.S[0], l = c.S[1];
        const float ml = cell_label_max<M, ZL>(l, b, H[0], H[1], c.vz0);
        const float mb = cell_blank_max<M>(b, H[0], H[2]);
        c.S = f32x2{mb, ml} + in[0].e;
    }
    if constexpr (F + 1 < kTpBlock) {
        // frame F+1's pair and packet: read in frame F-3 - the instructions of frames F-2 and F-1 may still be in flight - or,
        // F + 1 < 4, with the block's first four: behind them came the reads of frames F+2 .. 3 and everything frames 0 .. F-1 issued
        if constexpr (F >= 3) tg_wait<tg_ops(F - 2) + tg_ops(F - 1)>(in[1]);
        else tg_wait<2 * (2 - F) + 3 * F>(in[1]);
        H[0] = wave_shr1(in[1].hp[3], c.S[1]);
        H[1] = wave_shr1(in[1].hp[2], c.S[0]);
        H[2] = wave_shr1(in[1].hp[1], H[0]);
    }
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(c.lds_stage), "v"(c.S), "n"(F * 16) : "memory");
    in[0] = in[1];
    in[1] = in[2];
    in[2] = in[3];
    if constexpr (F + 4 < kTpBlock) tg_read<F + 4>(in[3], pairs, packets);
    else asm volatile("" : "=v"(in[3].e), "=v"(in[3].hp));   // (nothing to read: a fresh value, so that in[2] and in[3] are not one register the next wait "modifies")
}
template <int M, bool ZL, bool GUARDED, int F>
__device__ __forceinline__ void tg_block_frames(TnTile<M, ZL> &c, uint32_t tb, float (&H)[3], TgIn (&in)[4], uint32_t pairs, uint32_t packets, float NINF)
{
    tg_frame<M, ZL, GUARDED, F>(c, tb + F, H, in, pairs, packets, NINF);
    if constexpr (F + 1 < kTpBlock) tg_block_frames<M, ZL, GUARDED, F + 1>(c, tb, H, in, pairs, packets, NINF);
}

template <int M, bool ZL>
__device__ __forceinline__ void tn_publish_block(TnTile<M, ZL> &c, uint32_t tb, int lane)
{
    const int32_t t = (int32_t)tb + lane;
    if (lane < kTpBlock && t >= c.t_in && t < c.t_end) {
        const f32x4 pk = lds_f32x4(c.lds_packets + (uint32_t)lane * 16u);
        const char *out_block = c.halo_out + ((int64_t)tb - (int64_t)c.t_in) * 16;
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:16 sc1\n\ts_nop 1" : : "v"((uint32_t)lane * 16u), "v"(pk), "s"(out_block) : "memory");
    }
}
template <int M, bool ZL>
__device__ __forceinline__ void tn_checkpoint(TnTile<M, ZL> &c, uint32_t t_next /* multiple of 32 */)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 1" : : "v"(c.ck_off), "v"(c.S), "s"(c.ck + ((size_t)(t_next / kCkFrames) - 1) * (size_t)c.ck_pitch) : "memory");
}

template <int M, bool ZL, int PITCH, bool CONTIG, bool GATHER>
__device__ __forceinline__ void tn_run_tile(const Lattice &d, const TileTask &tk, int32_t *meta, char *halo, gu32w_t prog, TileAux *aux,
                                             uint32_t lds_rows, uint32_t lds_halo, int verify, TpStats *stats_out)
{
    typedef __attribute__((address_space(3))) uint32_t *lu32_t;
    const int lane = threadIdx.x & 63;
    // wavefront 0 computes, wavefront 1 feeds (memory: polls, requests, publishes), and in the GATHER form wavefront 2 does the
    // feeder's LDS work (finiteness sum, emission look-up, band): with the look-up the feeder alone took 5100 cycles per
    // block against the compute wavefront's 2500
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool feeder = role == 1, looker = GATHER && role == 2;
    // the compute wavefront is the chain: where it shares a SIMD with other tiles' feeders and look-up wavefronts it issues first
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    typedef TnLds<PITCH, CONTIG, GATHER> Lds;
    constexpr int kRowRing = Lds::kRing, kRowSlot = Lds::kSlot;
    const uint32_t lds_poll = lds_rows + Lds::kPoll;
    const uint32_t lds_stage0 = lds_rows + Lds::kStage;                         // two staging buffers of kTpStageBytes
    const uint32_t stat_lds = lds_rows + Lds::kStat;                            // diagnostic words, then two flag words
    const uint32_t lds_band = lds_rows + Lds::kBand;                            // two buffers of kTp2BandBytes: the band code's kill words and event mask of a block
    const uint32_t lds_pairs = lds_band;                                        // GATHER: two buffers of kTgPairBytes instead
    if (threadIdx.x < 10) ((lu32_t)(uintptr_t)stat_lds)[threadIdx.x] = 0;
    unsigned long long ph = 0;
    auto phase = [&](int w) {
        if (verify & 4) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (w >= 0) ((lu32_t)(uintptr_t)stat_lds)[w] += (uint32_t)(now - ph);
            ph = now;
        }
    };
    if ((verify & 4) && role == 0 && lane == 0) {   // where the compute wavefront runs (the feeder reports its own place below)
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        ((lu32_t)(uintptr_t)stat_lds)[10] = hw & 0xffffu;
    }
    if ((verify & 4) && feeder) {   // start stamps: wall clock (100 MHz) and shader clock
        stats_out->start_tick = (unsigned long long)wall_clock64();
        stats_out->total_ticks = __builtin_amdgcn_s_memtime();
    }
    const float NINF = ninf();
    TnTile<M, ZL> c;
    c.T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    c.L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    c.B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    c.dq = c.L / c.T;
    c.dr = c.L % c.T;
    c.base = __builtin_amdgcn_readfirstlane(tk.tile) * kTnTile;
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
    c.ck_off = (((uint32_t)c.base + (uint32_t)kTnCells * (uint32_t)lane) & (uint32_t)d.ck_mask) * 4u;
    c.lds_rows = lds_rows;
    c.lds_halo = lds_halo;
    constexpr int kStageBytes = Lds::kStageBytes, kScratch = GATHER ? 8 : 16;     // idle lanes' scratch behind the packet rows: bytes per lane
    static_assert(kTpBlock * 16 + 61 * kScratch + (kTpBlock - 1) * 16 + 8 <= kStageBytes, "publish staging");
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
    {
        gci32_t labx = (gci32_t)d.labx + ((size_t)c.base >> 1) + (size_t)lane;   // the lane's ONE label cell: position base + 2 lane + 1
        c.la0 = labx[0];
        c.vz0 = (ZL && c.la0 == 0) ? NINF : __builtin_inff();
    }
    // state before frame t_in: nothing of the tile is live, except the virtual start state (align.py:57-58)
    c.S = f32x2{NINF, NINF};
    if (c.base == 0 && c.t_in == 0 && lane == 0) c.S[0] = 0.0f;
    c.absum = 0.0f;
    c.lds_packets = lds_stage0;
    c.lds_stage = 0;
    // slot t_in of the upper boundary = the state before the tile's first frame: lane 63's cells, all -inf.  The FEEDER
    // stores it: every store the progress word vouches for is in its own in-order vmcnt history.
    if (feeder) tp_halo_store<0>(c.halo_out, f32x4{NINF, NINF, NINF, NINF}, 1ull << 63);

    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) char *lchar_t;
    const uint32_t last_row = c.T - 1;
    // This tile reads slots t_in .. t_end - 1 of the boundary below.  The tile below computes frames < below_end, so its slots
    // <= below_end are states it computed and every slot behind them is -inf BY CONSTRUCTION (it is under the band): it stores
    // ONE of those, slot below_end + 1, with its last block, and this tile reads that one for all of them and never asks for
    // progress beyond it.  (ka_tiled2.hpp fills all the slots behind t_end after its loop and the tile above waits for the
    // "done" word: ~3 us per tile boundary in which the chain's new front - the tile above, when this one dies - stands still.)
    const uint32_t below_end = (uint32_t)__builtin_amdgcn_readfirstlane(tk.below_end);
    const int32_t fill_end = __builtin_amdgcn_readfirstlane(tk.fill_end);      // last slot of the boundary above that the tile above reads
    const uint32_t dead_slot = below_end < 0x7ffffff0u ? below_end + 1u : 0x7fffffffu;
    const uint32_t last_slot = (uint32_t)c.t_end - 1 < dead_slot ? (uint32_t)c.t_end - 1 : dead_slot;
    constexpr int kPkRing = Lds::kPkRing;
    auto ring = [](int32_t k) { return (uint32_t)((k % kPkRing + kPkRing) % kPkRing); };            // packets and poll words
    auto rslot = [](int32_t k) { return (uint32_t)((k % kRowRing + kRowRing) % kRowRing); };        // rows
    constexpr int kRowDmas = Lds::kRowDmas;   // LDS-DMA instructions per block of rows
    static_assert(CONTIG || PITCH == kTpRowBytes, "row-by-row staging uses 256-byte rows");
    auto issue_rows = [&](int32_t k) {    // k >= 0: the log-prob rows of block k (they do not depend on the tile below)
        const uint32_t tb = (uint32_t)k * kTpBlock;
        lchar_t dst = (lchar_t)(uintptr_t)(c.lds_rows + rslot(k) * kRowSlot);
        if constexpr (!CONTIG) {
            const char *rp = c.lp + (size_t)(tb < last_row ? tb : last_row) * c.ld;
            if (tb + kTpBlock <= c.T) {
#pragma unroll
                for (int f = 0; f < kTpBlock; ++f) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(rp + c.lane_off), (lptr_t)(dst + f * kTpRowBytes), 4, 0, 0);
                    rp += c.ld;
                }
            } else {
#pragma unroll
                for (int f = 0; f < kTpBlock; ++f) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(rp + c.lane_off), (lptr_t)(dst + f * kTpRowBytes), 4, 0, 0);
                    rp += tb + f < last_row ? c.ld : 0;
                }
            }
        } else {
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
    };
    auto issue_packets = [&](int32_t k) {    // k >= 0: the tile below's packets of block k, and a look at its progress word
        const uint32_t tb = (uint32_t)k * kTpBlock, slot = ring(k);
        if (lane < kTpBlock) {
            uint32_t s = tb + (uint32_t)lane;
            s = s < (uint32_t)c.t_in ? (uint32_t)c.t_in : (s > last_slot ? last_slot : s);
            __builtin_amdgcn_global_load_lds((gptr_t)(c.halo_in + (size_t)(s - (uint32_t)c.t_in) * 16), (lptr_t)(lchar_t)(uintptr_t)(c.lds_halo + slot * (kTpBlock * 16)), 16, 0, 16);
        }
        if (lane == 0) __builtin_amdgcn_global_load_lds((gptr_t)c.prog_in, (lptr_t)(lchar_t)(uintptr_t)(lds_poll + slot * 4), 4, 0, 16);
    };
    auto issue_block = [&](int32_t k) {
        issue_rows(k);
        issue_packets(k);
    };
    bool stale = false;
    auto check_packets = [&](int32_t k) {      // ka_engine_set_verify(1): a packet of block k (landed) that nobody wrote
        if (verify & 1) {
            const uint32_t slot = ring(k);
            const int32_t sidx = k * kTpBlock + (lane & (kTpBlock - 1));
            const f32x4 h = lds_f32x4(c.lds_halo + slot * (kTpBlock * 16) + (uint32_t)(lane & (kTpBlock - 1)) * 16u);
            const bool mine = lane < kTpBlock && sidx >= c.t_in && sidx < c.t_end;
            const bool bad = mine && (__builtin_bit_cast(uint32_t, h[1]) == kTpSentinel || __builtin_bit_cast(uint32_t, h[2]) == kTpSentinel ||
                                      __builtin_bit_cast(uint32_t, h[3]) == kTpSentinel);
            if (__builtin_amdgcn_ballot_w64(bad)) stale = true;
        }
    };
    auto landed_block = [&](int32_t k) {
        const uint32_t r = c.lds_rows + rslot(k) * kRowSlot + (uint32_t)lane * 16u;
        constexpr int kReads = !CONTIG ? kTpSlotBytes / 1024 : kRowDmas;
        f32x4 v[kReads];
#pragma unroll
        for (int j = 0; j < kReads; ++j) v[j] = lds_f32x4(r + j * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < kReads; ++j) c.absum += (__builtin_fabsf(v[j][0]) + __builtin_fabsf(v[j][1])) + (__builtin_fabsf(v[j][2]) + __builtin_fabsf(v[j][3]));
        if constexpr (!GATHER) check_packets(k);
    };
    auto need_for = [&](int32_t k) {      // progress word the tile below must show before block k's packets are requested (slots < it)
        uint32_t n = (uint32_t)(k + 1) * kTpBlock;
        n = n < (uint32_t)c.t_end ? n : (uint32_t)c.t_end;
        return n < dead_slot + 1u ? n : dead_slot + 1u;
    };

    const int32_t kb0 = c.t_in / kTpBlock, kb1 = (c.t_end - 1) / kTpBlock;   // first and last block
    auto lds_work = [&](int32_t it, uint32_t tb) {
        // block it+1 landed before the last barrier: its finiteness sum
        if (it + 1 >= kb0 && it + 1 <= kb1) landed_block(it + 1);
        // the band bookkeeping of the NEXT block (which positions of the tile enter or leave the band in which frame), for
        // the compute wavefront to pick up after the next barrier: ~60 instructions it does not have to issue
        if (it + 1 >= kb0 && it + 1 <= kb1) {
            tn_band_block(c, tb + kTpBlock, lane);
            if constexpr (!GATHER) {
                const uint32_t bb = lds_band + (uint32_t)((it + 1) & 1) * kTp2BandBytes;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                *(__attribute__((address_space(3))) u32x2 *)(uintptr_t)(bb + (uint32_t)lane * 8u) = u32x2{c.KL, c.KE};
                if (lane == 0) *(lu32_t)(uintptr_t)(bb + 512u) = c.ev;
            } else {
                // the emissions of block it+1, per frame and lane {blank, label} ...
                const uint32_t rows = c.lds_rows + rslot(it + 1) * kRowSlot;
                const uint32_t pb = lds_pairs + (uint32_t)((it + 1) & 1) * kTgPairBytes;
                const uint32_t mine = pb + (uint32_t)lane * 8u, col = rows + (uint32_t)c.la0;
#pragma unroll
                for (int g = 0; g < kTpBlock; g += 8) {      // (eight frames' reads in flight at a time: one by one the loop ran at the LDS latency)
                    f32x2 e[8];
#pragma unroll
                    for (int f = 0; f < 8; ++f) e[f] = f32x2{lds_f32(rows + (g + f) * PITCH), lds_f32(col + (g + f) * PITCH)};
#pragma unroll
                    for (int f = 0; f < 8; ++f) *(__attribute__((address_space(3))) f32x2 *)(uintptr_t)(mine + (g + f) * 512) = e[f];
                }
                // ... and -inf over those of the cells that die after a frame: lane f < 32 deals with frame f - the positions
                // that left the band before it (rule ii: its own KL) and those that enter after it (rule i: KE of lane f+1);
                // position r of the tile is word r of the frame's 512 bytes
                const uint32_t ke = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.KE, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
                uint32_t r2 = c.KL & 0xffffu, n2 = lane < kTpBlock ? c.KL >> 16 : 0u, r1 = ke & 0xffffu, n1 = lane < kTpBlock ? ke >> 16 : 0u;
                const uint32_t fb = pb + (uint32_t)lane * 512u;
                while (__builtin_amdgcn_ballot_w64((n2 | n1) != 0u)) {
                    if (n2) { *(__attribute__((address_space(3))) float *)(uintptr_t)(fb + r2 * 4u) = NINF; ++r2; --n2; }
                    if (n1) { *(__attribute__((address_space(3))) float *)(uintptr_t)(fb + r1 * 4u) = NINF; ++r1; --n1; }
                }
            }
            tn_band_advance(c);
        }
    };
    bool fed = true;
    const TnIn none = {0.0f, 0.0f, f32x4{NINF, NINF, NINF, NINF}};
    TnIn in[4] = {none, none, none, none};
    float H[3] = {NINF, NINF, NINF};
    // Iterations kb0-2, kb0-1 prime the feeder's pipeline; iteration kb1+1 publishes the last block.
    for (int32_t it = kb0 - 2; it <= kb1 + 1; ++it) {
        const uint32_t tb = (uint32_t)(it * kTpBlock);              // (wraps in the priming iterations of block 0: not used there)
        if ((verify & 4) && role == 0) {        // ka_engine_set_verify(4): how long the compute wavefront stands at the barrier
            const unsigned long long b0 = __builtin_amdgcn_s_memtime();
            tp2_barrier();
            ((lu32_t)(uintptr_t)stat_lds)[8] += (uint32_t)(__builtin_amdgcn_s_memtime() - b0);
        } else
            tp2_barrier();
        if (feeder) {
            phase(-1);
            // What the tile above is waiting for comes first: block it-1 is complete in staging buffer (it-1) & 1 - lane f < 32
            // stores the packet of frame f as slot tb-32+f+1 - and is announced as soon as those stores have retired.  In
            // ka_tiled2.hpp the announcement comes at the END of the iteration, behind the poll of the tile below: every block of
            // lag between two tiles is paid once per tile of the chain (the tile above is that far behind when this one dies),
            // and a feeder that spins on ITS producer must not hold back what its consumer needs.  So:
            //   the tile below is known to be far enough (the progress word that landed with block it+1 says so): request block
            //     it+2, publish, ONE wait for both, announce;
            //   else: publish, wait, announce - then poll, request, and wait again at the end of the iteration.
            // (No counted waits: vmcnt orders loads among loads and stores among stores, not one against the other.)
            // GATHER form: the compute wavefront reads nothing of the next block, so the packets of block it+1 are requested in
            // iteration it - ONE block ahead, a block of lag less per tile boundary - and the rows, which do not depend on the tile
            // below, two ahead as before (the look-up wavefront needs block it+1's in this iteration).
            constexpr int kAhead = GATHER ? 1 : 2;
            const bool published = it - 1 >= kb0 && it - 1 <= kb1;
            const bool wanted = it + kAhead <= kb1 && fed;            // that block has packets of the tile below to wait for
            const uint32_t have = wanted && it + kAhead - 1 >= kb0
                                      ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, lds_f32(lds_poll + ring(it + kAhead - 1) * 4))) : 0u;
            const bool ready = !wanted || have >= need_for(it + kAhead);
            const bool request = GATHER ? (it + 1 >= 0 && it + 1 <= kb1) : (it + 2 >= 0 && it + 2 <= kb1 + 1);
            if constexpr (GATHER) {
                if (it + 2 >= 0 && it + 2 <= kb1) issue_rows(it + 2);
                if (ready && request) issue_packets(it + 1);
            } else {
                if (ready && request) issue_block(it + 2);
            }
            phase(6);
            if (published) {
                c.lds_packets = lds_stage0 + (uint32_t)((it - 1) & 1) * kStageBytes;
                tn_publish_block(c, tb - kTpBlock, lane);
                // with the last block, the one -inf slot that stands for everything behind t_end (the tile above reads slots up to
                // its own t_end - 1 = fill_end: none behind t_end when the two end together)
                const bool closing = tb >= (uint32_t)c.t_end && c.t_end <= fill_end;
                if (closing && lane == 0) {
                    const f32x4 dead = {NINF, NINF, NINF, NINF};
                    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"((uint32_t)((c.t_end + 1 - c.t_in) * 16)), "v"(dead), "s"(c.halo_out) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // slots <= tb are in memory - but never vouch for more than the tile's own frames have produced
                tp_prog_store(c.prog_out, (tb < (uint32_t)c.t_end ? tb : (uint32_t)c.t_end) + 1 + (closing ? 1u : 0u));
            }
            phase(7);
            if (!ready) {
                // (no hysteresis: the one-wavefront tile asks for two blocks more than it needs once it has to wait, so that its
                //  frames are not interrupted by a poll per block; here the frames run in the other wavefront)
                fed = tp_wait_progress(c.prog_in, need_for(it + kAhead), need_for(it + kAhead), have, stat_lds);
                phase(5);
                if constexpr (GATHER) {
                    if (request) issue_packets(it + 1);
                } else {
                    if (request) issue_block(it + 2);
                }
            }
            phase(6);
            if constexpr (!GATHER) lds_work(it, tb);
            phase(4);
            // the requests of block it+2 have landed before the barrier: the compute wavefront reads its first rows in the next
            // iteration
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            phase(3);
        } else if (looker) {
            const unsigned long long k0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (it >= kb0 && it <= kb1) check_packets(it);
            lds_work(it, tb);
            if (verify & 4) ((lu32_t)(uintptr_t)stat_lds)[9] += (uint32_t)(__builtin_amdgcn_s_memtime() - k0);   // the look-up wavefront's busy cycles
        } else if (GATHER && it >= kb0 && it <= kb1) {
            if constexpr (GATHER) {
                uint32_t pairs = lds_pairs + (uint32_t)(it & 1) * kTgPairBytes, packets = c.lds_halo + ring(it) * (kTpBlock * 16);
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(pairs), "=&v"(packets) : "s"(pairs), "s"(packets));
                pairs += (uint32_t)lane * 8u;
                asm volatile("" : "+v"(pairs));
                {
                    const uint32_t pk = lds_stage0 + (uint32_t)(it & 1) * kStageBytes;
                    c.lds_stage = lane >= 62 ? pk + (uint32_t)(lane - 62) * 8u : pk + kTpBlock * 16 + (uint32_t)lane * kScratch;
                }
                // the block's first four frames; the others are read four frames ahead (tg_frame)
                TgIn in[4];
                tg_read<0>(in[0], pairs, packets);
                tg_read<1>(in[1], pairs, packets);
                tg_read<2>(in[2], pairs, packets);
                tg_read<3>(in[3], pairs, packets);
                tg_wait<6>(in[0]);      // (the first frame's; the other three land while it runs)
                H[0] = wave_shr1(in[0].hp[3], c.S[1]);   // position base + 2 lane - 1: the label of the lane below (lane 0: the packet's top cell)
                H[1] = wave_shr1(in[0].hp[2], c.S[0]);   // - 2: its blank
                H[2] = wave_shr1(in[0].hp[1], H[0]);     // - 3: the label two lanes below
                const bool partial = (int32_t)tb < c.t_in || (int32_t)(tb + kTpBlock) > c.t_end;
                const unsigned long long fr0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
                if (!partial) {
                    tg_block_frames<M, ZL, false, 0>(c, tb, H, in, pairs, packets, NINF);
                    tg_wait_all(in);
                    if ((tb + kTpBlock) % kCkFrames == 0 && tb + kTpBlock < c.T) tn_checkpoint(c, tb + kTpBlock);
                } else {
                    tg_block_frames<M, ZL, true, 0>(c, tb, H, in, pairs, packets, NINF);
                    tg_wait_all(in);
                    if ((tb + kTpBlock) % kCkFrames == 0 && (int32_t)(tb + kTpBlock) <= c.t_end && tb + kTpBlock < c.T) tn_checkpoint(c, tb + kTpBlock);
                }
                if (verify & 4) ((lu32_t)(uintptr_t)stat_lds)[2] += (uint32_t)(__builtin_amdgcn_s_memtime() - fr0);
            }
        } else if (it >= kb0 && it <= kb1) {
            const uint32_t slot = rslot(it), nslot = rslot(it + 1);
            uint32_t rc = c.lds_rows + slot * kRowSlot, rn = c.lds_rows + nslot * kRowSlot;
            uint32_t hc = c.lds_halo + ring(it) * (kTpBlock * 16), hn = c.lds_halo + ring(it + 1) * (kTpBlock * 16);
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(rc), "=&v"(rn), "=&v"(hc), "=&v"(hn) : "s"(rc), "s"(rn), "s"(hc), "s"(hn));
            TpAddr A[2] = {{rc + (uint32_t)c.la0, 0u, rc, hc}, {rn + (uint32_t)c.la0, 0u, rn, hn}};
            asm volatile("" : "+v"(A[0].l0), "+v"(A[1].l0));
            // where this block's frames drop their packets: lane 63's into the packet row, the others' into scratch behind it
            {
                const uint32_t pk = lds_stage0 + (uint32_t)(it & 1) * kStageBytes;
                c.lds_stage = lane >= 62 ? pk + (uint32_t)(lane - 62) * 8u : pk + kTpBlock * 16 + (uint32_t)lane * kScratch;
            }
            if (it == kb0) {
                // the first four frames' rows and packets (later ones are read four frames ahead, tn_frame); slot t_in's packet
                tn_read<PITCH, 0>(in[0], A);
                tn_read<PITCH, 1>(in[1], A);
                tn_read<PITCH, 2>(in[2], A);
                tn_read<PITCH, 3>(in[3], A);
                tn_wait_all(in);
                const f32x4 hp = in[0].hp;
                H[0] = wave_shr1(hp[3], c.S[1]);   // position base + 2 lane - 1: the label of the lane below (lane 0: the packet's top cell)
                H[1] = wave_shr1(hp[2], c.S[0]);   // - 2: its blank
                H[2] = wave_shr1(hp[1], H[0]);     // - 3: the label two lanes below
            }
            {   // this block's band bookkeeping, left by the feeder
                const uint32_t bb = lds_band + (uint32_t)(it & 1) * kTp2BandBytes;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 kk = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)(bb + (uint32_t)lane * 8u);
                c.KL = kk[0];
                c.KE = kk[1];
                c.ev = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const lu32_t)(uintptr_t)(bb + 512u));
            }
            const bool partial = (int32_t)tb < c.t_in || (int32_t)(tb + kTpBlock) > c.t_end;
            const unsigned long long fr0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (!partial) {
                tn_block_frames<M, ZL, PITCH, false, 0>(c, tb, H, in, A, NINF);
                tn_wait_all(in);
                if ((tb + kTpBlock) % kCkFrames == 0 && tb + kTpBlock < c.T) tn_checkpoint(c, tb + kTpBlock);
            } else {
                tn_block_frames<M, ZL, PITCH, true, 0>(c, tb, H, in, A, NINF);
                tn_wait_all(in);
                if ((tb + kTpBlock) % kCkFrames == 0 && (int32_t)(tb + kTpBlock) <= c.t_end && tb + kTpBlock < c.T) tn_checkpoint(c, tb + kTpBlock);
            }
            if (verify & 4) ((lu32_t)(uintptr_t)stat_lds)[2] += (uint32_t)(__builtin_amdgcn_s_memtime() - fr0);
        }
    }
    // nothing of this workgroup may still be landing in LDS or in a register when it ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int32_t *m = meta_of(meta, d.idx);
    if (feeder && !fed && lane == 0) atomicMin(&m[0], kStatusInternal);
    if (GATHER ? looker : feeder) {
        // ---- finiteness (as forward_ck: the scores-only form is valid for finite log-probs of sane magnitude).  Flagged
        // before the tile reports itself done (barrier below), so that whoever closes the lattice sees the flag.
        if (stale && lane == 0) atomicMin(&m[0], kStatusInternal);
        const uint32_t abits = __builtin_bit_cast(uint32_t, c.absum) & 0x7fffffffu;
        if (__builtin_amdgcn_ballot_w64(abits > 0x7f800000u)) {
            if (lane == 0) atomicMin(&m[0], kStatusNaN);
        } else if (__builtin_amdgcn_ballot_w64(abits >= __builtin_bit_cast(uint32_t, 1e30f))) {
            if (lane == 0) atomicOr(&m[2], d.W <= kFastMaxBand ? kFlagExact : kFlagDeclined);
        }
        __threadfence();
    }
    if (feeder) {
        // (nothing to hand over behind t_end: the one -inf slot went out with the last block)
        tp_prog_store(c.prog_out, kTpProgDone);
        __threadfence();
    }
    tp2_barrier();
    if (feeder) {
        if ((verify & 4) && lane == 0) {
            uint32_t hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const __attribute__((address_space(3))) uint32_t *sw = (const __attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds;
            TpStats st;
            st.spins = sw[0] | ((unsigned long long)((xcc & 0xf) << 16 | (hw & 0xffff))) << 32;
            st.phase[0] = sw[3] | ((unsigned long long)sw[4] << 32);
            st.phase[1] = sw[5] | ((unsigned long long)sw[6] << 32);
            st.phase[2] = sw[7] | ((unsigned long long)sw[10] << 32);   // (high half: HW_ID of the compute wavefront)
            st.extra[0] = sw[8] | ((unsigned long long)sw[2] << 32);    // compute wavefront: cycles at the barrier | cycles inside the frame blocks
            st.extra[1] = sw[9];                                         // look-up wavefront: busy cycles
            st.wait_ticks = sw[1] | ((unsigned long long)sw[2] << 32);
            st.start_tick = __builtin_amdgcn_s_memtime() - stats_out->total_ticks;
            st.total_ticks = wall_clock64() - stats_out->start_tick;
            *stats_out = st;
        }
        return;
    }
    if (looker) return;
    // ---- terminal state: the HIGHEST live position of frame T-1 (align.py:99-101), over the tiles alive then ----
    if ((uint32_t)c.t_end == c.T) {
        // (the only full band mask of a tile's life: cells above hi may hold leaked scores)
        const uint32_t q_last = c.L - (c.L + c.T - 1u) / c.T;   // floor(L (T-1) / T) = L - ceil(L / T)
        const uint32_t lo_last = c.lo_of(q_last), hi_last = c.hi_of(lo_last);
        const float cell[2] = {c.S[0], c.S[1]};
        unsigned long long key = 0;
#pragma unroll
        for (int k = 0; k < kTnCells; ++k) {
            const uint32_t pos = (uint32_t)c.base + (uint32_t)kTnCells * (uint32_t)lane + (uint32_t)k;
            if (pos >= lo_last && pos < hi_last && cell[k] != NINF) key = ((unsigned long long)(pos + 1u) << 32) | __builtin_bit_cast(uint32_t, cell[k]);
        }
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
                    m[1] = -1;   // declined: the exact kernels redo the lattice (or ka_batch_finish hands it to the generic ones)
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

// One workgroup of two wavefronts per 128-position tile; launch parameters as forward_tp2_kernel's, TnLds<..>::kTotal bytes of LDS.
template <int M, int PITCH, bool CONTIG, bool GATHER>
__global__ __launch_bounds__(GATHER ? 192 : 128) void forward_tn_kernel(const Lattice *__restrict__ lats, const TileTask *__restrict__ tasks, int n_tasks,
                                                         int32_t *meta, char *halo, uint32_t *prog, TileAux *aux, uint32_t *ticket, int verify, TpStats *stats)
{
    extern __shared__ __attribute__((aligned(16))) char tp_lds[];
    const uint32_t lds_rows = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&tp_lds[0];
    typedef TnLds<PITCH, CONTIG, GATHER> Lds;
    const uint32_t lds_halo = lds_rows + Lds::kHalo;
    volatile uint32_t *s_ticket = reinterpret_cast<volatile uint32_t *>(&tp_lds[Lds::kStat + 48]);
    if (threadIdx.x == 0) *s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tix = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_ticket);
    __syncthreads();
    if (tix >= (uint32_t)n_tasks) return;
    const TileTask &tk = tasks[tix];
    const Lattice &d = lats[__builtin_amdgcn_readfirstlane(tk.lat)];
    const int flags = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2]);
    if (flags & kFlagZeroLabel)
        tn_run_tile<M, true, PITCH, CONTIG, GATHER>(d, tk, meta, halo, (gu32w_t)prog, aux, lds_rows, lds_halo, verify, stats + tix);
    else
        tn_run_tile<M, false, PITCH, CONTIG, GATHER>(d, tk, meta, halo, (gu32w_t)prog, aux, lds_rows, lds_halo, verify, stats + tix);
}

}  // namespace ka
