// ka_tiled_stream.hpp — the 128-position tile pipeline, round 4: packets that vouch for themselves, no drain between blocks.
//
// A workgroup of three wavefronts per 128-position tile (two cells per lane): the COMPUTE wavefront runs the frames - 11
// instructions each -, the LOOK-UP wavefront turns the staged log-prob rows into per-lane emission pairs with the band's kills
// folded in as -inf (a sum with -inf IS the kill: the compute wavefront has no band code and no branch), the FEEDER moves memory.
// Round 3 had this decomposition too (ka_tiled_narrow.hpp, removed); what changed in round 4 is how a tile talks to its
// neighbours and how the three wavefronts meet; both were what a chain of tiles - a lone lattice, a book's longest chapter -
// actually paid for (profiles/r04_tile_stats_cfg2_before.txt: of the front tile's 3300 cycles per 32-frame
// block 2700 were frames and 600 the code between two blocks; every tile boundary cost 5.3 us of hand-off lag, 79 of them in
// BASELINE configs[1]):
//
//   * The hand-off needs no progress word and no acknowledged store (Guideline 16, R2: "the data is the flag").  The halo
//     region is filled with a NaN sentinel before the launch (scores are never NaN); the lower tile's feeder stores a block's
//     32 packets write-through and goes on - it never waits for them; the upper tile's feeder fetches the block's packets
//     (LDS-DMA, sc1) and looks at them: a packet is there when none of its three words is the sentinel, word by word, so a
//     torn 16-byte store cannot pass.  Until then it fetches again.  One memory round trip between "computed" and "usable"
//     instead of three (store acknowledged -> progress word seen -> packets fetched).
//   * The feeder publishes a block as soon as its last frame's packet appears in the LDS staging buffer (the same sentinel
//     trick inside the workgroup) instead of at the next barrier.
//   * ONE barrier per block, and it no longer sits between two blocks: it comes after frame 27.  The compute wavefront's LDS
//     reads run four frames ahead straight across the block boundary (frames 28..31 read the next block's pairs and packets,
//     which the barrier has just vouched for), so a block no longer starts with eight priming reads and a wait, nor ends with
//     four frames that read nothing.  What is left between two blocks is three address flips and the loop branch.
//   * The checkpoint is stored by the feeder, from the staging buffer where the state after frame 31 lies anyway (every lane
//     drops its pair there each frame): the compute wavefront issues no vector-memory instruction at all.
//
// Epoch k = the time between barrier k-1 and barrier k:
//   compute   frames 28..31 of block k-1, then frames 0..27 of block k
//   look-up   finiteness sum, band bookkeeping and emission pairs of block k+1 (its rows landed before barrier k-1)
//   feeder    publish block k-1 (+ its checkpoint row), request the rows of block k+2, fetch the packets of block k+1 until
//             they are all there, wait for everything in flight
// so at barrier k the pairs and packets of block k+1 are complete, and every buffer that is rewritten during epoch k+1 was last
// read before barrier k (the compute wavefront drains its reads there: the one wait per block, ~an LDS latency minus frame 27).
#pragma once
#include "ka_tiled2.hpp"

namespace ka {

// ---- a tile of 128 positions: two cells per lane ----
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

// The compute wavefront's LDS traffic goes through asm statements the compiler cannot see into, with the waits written by hand:
// hipcc waits with lgkmcnt(0) at every branch merge, i.e. a frame would wait for the reads it has just issued and take an LDS
// round trip (~120 cycles) however few instructions it has.  Here the reads run FOUR frames ahead, every frame issues the same
// three LDS instructions in the same order, and the frame that needs the data of frame t+1 waits with a counted lgkmcnt: the
// instructions of the two frames in between stay in flight.  (asm volatile statements keep their order; tools/lint_inflight.py
// checks that no register is copied between its read and its wait.)
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


// LDS map of a workgroup (bytes from the dynamic block's start)
template <int PITCH, bool CONTIG>
struct TsLds {
    static constexpr int kRowDmas = !CONTIG ? kTpBlock : (kTpBlock * PITCH + 1023) / 1024;
    static constexpr int kSlot = CONTIG ? kRowDmas * 1024 : kTpSlotBytes;        // a block of rows
    static constexpr int kRows = 0;                                              // two row slots
    static constexpr int kPackets = kRows + 2 * kSlot;                           // two blocks of 32 packets of the tile below
    static constexpr int kStageBytes = 1536;                                     // a publish staging buffer: 32 packet rows + the other lanes' scratch, 8 bytes apart
    static constexpr int kStage = kPackets + 2 * kTpBlock * 16;
    static constexpr int kStat = kStage + 2 * kStageBytes;                       // 12 diagnostic words, the ticket at +48
    static constexpr int kPairs = kStat + 64;                                    // two blocks of emission pairs
    static constexpr int kTotal = kPairs + 2 * kTgPairBytes;
};
static_assert(3 * ((TsLds<256, true>::kTotal + 511) / 512 * 512) <= 160 * 1024 && 3 * ((TsLds<256, false>::kTotal + 511) / 512 * 512) <= 160 * 1024,
              "three workgroups per CU");
static_assert(kTpBlock * 16 + 61 * 8 + (kTpBlock - 1) * 16 + 8 <= TsLds<256, true>::kStageBytes, "publish staging");

// one frame of the compute wavefront.  in[0] = frame F's pair, in[1] = frame F+1's pair and packet (waited for here), in[2],
// in[3] in flight.  Every frame issues three LDS instructions - the staging write and the two reads of frame F+4 - in that
// order, except frame 27, which reads first (its reads then have the whole frame to land before the barrier's drain).
template <int M, bool ZL, bool GUARDED, int F>
__device__ __forceinline__ void ts_frame(TnTile<M, ZL> &c, uint32_t t, float (&H)[3], TgIn (&in)[4], uint32_t pairs_cur, uint32_t pairs_nxt, uint32_t pk_cur,
                                         uint32_t pk_nxt)
{
    constexpr bool kEarly = F == 27;
    constexpr int G = F + 4;
    TgIn far;
    if constexpr (kEarly) tg_read<G>(far, pairs_cur, pk_cur);
    const bool live = !GUARDED || ((int32_t)t >= c.t_in && (int32_t)t < c.t_end);
    if (live) {
        const float b = c.S[0], l = c.S[1];
        const float ml = cell_label_max<M, ZL>(l, b, H[0], H[1], c.vz0);
        const float mb = cell_blank_max<M>(b, H[0], H[2]);
        c.S = f32x2{mb, ml} + in[0].e;
    }
    // frame F+1's pair and packet were read in frame F-3 (its last two LDS instructions; frame 31's in frame 27, its first two):
    // the three instructions of each of the frames F-2 and F-1 may still be in flight, and in frame 27 its own two early reads
    tg_wait<kEarly ? 8 : 6>(in[1]);
    H[0] = wave_shr1(in[1].hp[3], c.S[1]);   // position base + 2 lane - 1: the label of the lane below (lane 0: the packet's top cell)
    H[1] = wave_shr1(in[1].hp[2], c.S[0]);   // - 2: its blank
    H[2] = wave_shr1(in[1].hp[1], H[0]);     // - 3: the label two lanes below
    // every lane drops its pair: lanes 62 and 63 into the two halves of this frame's packet (the top four cells in position
    // order), the others into scratch behind the packet rows - after frame 31 that scratch IS the checkpoint
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(c.lds_stage), "v"(c.S), "n"(F * 16) : "memory");
    if constexpr (!kEarly) {
        if constexpr (G < kTpBlock) tg_read<G>(far, pairs_cur, pk_cur);
        else tg_read<G - kTpBlock>(far, pairs_nxt, pk_nxt);
    }
    in[0] = in[1];
    in[1] = in[2];
    in[2] = in[3];
    in[3] = far;
}
template <int M, bool ZL, bool GUARDED, int F, int END>
__device__ __forceinline__ void ts_frames(TnTile<M, ZL> &c, uint32_t tb, float (&H)[3], TgIn (&in)[4], uint32_t pairs_cur, uint32_t pairs_nxt, uint32_t pk_cur,
                                          uint32_t pk_nxt)
{
    ts_frame<M, ZL, GUARDED, F>(c, tb + F, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
    if constexpr (F + 1 < END) ts_frames<M, ZL, GUARDED, F + 1, END>(c, tb, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
}
// the barrier of a block, compute wavefront's side: everything it has read is in its registers, everything it has written is
// in LDS (the reads of frames 28..31, issued in frames 24..27, are the ones this may wait for)
__device__ __forceinline__ void ts_barrier_compute(TgIn (&in)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                 : "+v"(in[0].e), "+v"(in[0].hp), "+v"(in[1].e), "+v"(in[1].hp), "+v"(in[2].e), "+v"(in[2].hp), "+v"(in[3].e), "+v"(in[3].hp)
                 :
                 : "memory");
}
// one block of the compute wavefront: frames 0..27, the barrier, frames 28..31 (which read the NEXT block's buffers)
template <int M, bool ZL, bool GUARDED>
__device__ __forceinline__ void ts_block(TnTile<M, ZL> &c, uint32_t tb, float (&H)[3], TgIn (&in)[4], uint32_t pairs_cur, uint32_t pairs_nxt, uint32_t pk_cur,
                                         uint32_t pk_nxt)
{
    ts_frames<M, ZL, GUARDED, 0, 28>(c, tb, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
    ts_barrier_compute(in);
    ts_frames<M, ZL, GUARDED, 28, kTpBlock>(c, tb, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
}

template <int M, bool ZL, int PITCH, bool CONTIG>
__device__ __forceinline__ void ts_run_tile(const Lattice &d, const TileTask &tk, int32_t *meta, char *halo, TileAux *aux, uint32_t lds0, int verify,
                                            TpStats *stats_out)
{
    typedef __attribute__((address_space(3))) uint32_t *lu32_t;
    typedef TsLds<PITCH, CONTIG> Lds;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // 0 compute, 1 feeder, 2 look-up
    // the compute wavefront is the chain: where it shares a SIMD with other tiles' feeders and look-up wavefronts it issues first
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    const uint32_t lds_rows = lds0 + Lds::kRows, lds_packets = lds0 + Lds::kPackets, lds_stage0 = lds0 + Lds::kStage, stat_lds = lds0 + Lds::kStat,
                   lds_pairs = lds0 + Lds::kPairs;
    constexpr int kRowSlot = Lds::kSlot, kStageBytes = Lds::kStageBytes, kRowDmas = Lds::kRowDmas;
    if (threadIdx.x < 10) ((lu32_t)(uintptr_t)stat_lds)[threadIdx.x] = 0;
    if ((verify & 4) && role == 0 && lane == 0) {   // where the compute wavefront runs (the feeder reports its own place below)
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        ((lu32_t)(uintptr_t)stat_lds)[10] = hw & 0xffffu;
    }
    if ((verify & 4) && role == 1) {   // start stamps: wall clock (100 MHz) and shader clock
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
    c.ck = reinterpret_cast<char *>(d.bp);
    c.ck_pitch = (uint32_t)d.ck_pitch;
    c.ck_off = (((uint32_t)c.base + (uint32_t)kTnCells * (uint32_t)lane) & (uint32_t)d.ck_mask) * 4u;
    c.lds_rows = lds_rows;
    c.lds_halo = lds_packets;
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

    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) char *lchar_t;
    const uint32_t last_row = c.T - 1;
    // This tile reads slots t_in .. t_end - 1 of the boundary below.  The tile below computes frames < below_end, so its slots
    // <= below_end are states it computed and every slot behind them is -inf BY CONSTRUCTION (it is under the band): it stores
    // ONE of those, slot below_end + 1, with its last block, and this tile reads that one for all of them.
    const uint32_t below_end = (uint32_t)__builtin_amdgcn_readfirstlane(tk.below_end);
    const int32_t fill_end = __builtin_amdgcn_readfirstlane(tk.fill_end);      // last slot of the boundary above that the tile above reads
    const uint32_t dead_slot = below_end < 0x7ffffff0u ? below_end + 1u : 0x7fffffffu;
    const uint32_t last_slot = (uint32_t)c.t_end - 1 < dead_slot ? (uint32_t)c.t_end - 1 : dead_slot;
    const int32_t kb0 = c.t_in / kTpBlock, kb1 = (c.t_end - 1) / kTpBlock;   // first and last block

    if (role == 1) {
        // =============================================== the feeder ===============================================
        static_assert(CONTIG || PITCH == kTpRowBytes, "row-by-row staging uses 256-byte rows");
        auto issue_rows = [&](int32_t k) {    // k >= 0: the log-prob rows of block k (they do not depend on the tile below)
            const uint32_t tb = (uint32_t)k * kTpBlock;
            lchar_t dst = (lchar_t)(uintptr_t)(lds_rows + (uint32_t)(k & 1) * kRowSlot);
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
        auto issue_packets = [&](int32_t k) {    // the tile below's packets of block k: slot 32 k + f for frame f
            const uint32_t tb = (uint32_t)k * kTpBlock;
            if (lane < kTpBlock) {
                uint32_t s = tb + (uint32_t)lane;
                s = s < (uint32_t)c.t_in ? (uint32_t)c.t_in : (s > last_slot ? last_slot : s);
                __builtin_amdgcn_global_load_lds((gptr_t)(c.halo_in + (size_t)(s - (uint32_t)c.t_in) * 16), (lptr_t)(lchar_t)(uintptr_t)(lds_packets + (uint32_t)(k & 1) * (kTpBlock * 16)),
                                                 16, 0, 16);
            }
        };
        auto packets_there = [&](int32_t k) {     // (landed) none of the three words a frame uses is the sentinel
            const f32x4 h = lds_f32x4(lds_packets + (uint32_t)(k & 1) * (kTpBlock * 16) + (uint32_t)(lane & (kTpBlock - 1)) * 16u);
            const bool missing = __builtin_bit_cast(uint32_t, h[1]) == kTpSentinel || __builtin_bit_cast(uint32_t, h[2]) == kTpSentinel ||
                                 __builtin_bit_cast(uint32_t, h[3]) == kTpSentinel;
            return __builtin_amdgcn_ballot_w64(missing) == 0ull;
        };
        // slot t_in of the upper boundary = the state before the tile's first frame: lane 63's cells, all -inf
        tp_halo_store<0>(c.halo_out, f32x4{NINF, NINF, NINF, NINF}, 1ull << 63);
        // the last packet row of both staging buffers says "not written yet"
        if (lane < 2) {
            const uint32_t a = lds_stage0 + (uint32_t)lane * kStageBytes + (kTpBlock - 1) * 16 + 8;
            asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(kTpSentinel) : "memory");
        }
        bool fed = true;
        unsigned long long t_wait = 0;
        for (int32_t k = kb0 - 2; k <= kb1 + 1; ++k) {
            // ---- publish block k-1: as soon as the packet of its last frame is in the staging buffer ----
            if (k - 1 >= kb0 && k - 1 <= kb1) {
                const uint32_t tbp = (uint32_t)(k - 1) * kTpBlock;
                const uint32_t stage = lds_stage0 + (uint32_t)((k - 1) & 1) * kStageBytes;
                for (;;) {      // (the compute wavefront writes it four frames after the barrier: ~0.15 us)
                    uint32_t w;
                    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(stage), "n"((kTpBlock - 1) * 16 + 8) : "memory");
                    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)w) != kTpSentinel) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                // lane f < 32 stores the packet of frame f as slot tbp + f + 1; all lanes fetch the checkpoint pair first
                const int32_t t = (int32_t)tbp + lane;
                const f32x4 pk = lds_f32x4(stage + (uint32_t)(lane & (kTpBlock - 1)) * 16u);
                const uint32_t ck_src = lane >= 62 ? stage + (kTpBlock - 1) * 16 + (uint32_t)(lane - 62) * 8u : stage + kTpBlock * 16 + (kTpBlock - 1) * 16 + (uint32_t)lane * 8u;
                const f32x2 ckv = *(const __attribute__((address_space(3))) f32x2 *)(uintptr_t)ck_src;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < kTpBlock && t >= c.t_in && t < c.t_end) {
                    const char *out_block = c.halo_out + ((int64_t)tbp - (int64_t)c.t_in) * 16;
                    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:16 sc1\n\ts_nop 1" : : "v"((uint32_t)lane * 16u), "v"(pk), "s"(out_block) : "memory");
                }
                // with the last block, the one -inf slot that stands for everything behind t_end (the tile above reads slots up to
                // its own t_end - 1 = fill_end: none behind t_end when the two end together)
                if (tbp + kTpBlock >= (uint32_t)c.t_end && c.t_end <= fill_end && lane == 0) {
                    const f32x4 dead = {NINF, NINF, NINF, NINF};
                    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"((uint32_t)((c.t_end + 1 - c.t_in) * 16)), "v"(dead), "s"(c.halo_out) : "memory");
                }
                // the checkpoint row behind this block (the scores after frame tbp + 31), if the tile is alive in that frame
                if ((int32_t)(tbp + kTpBlock) <= c.t_end && tbp + kTpBlock < c.T)
                    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 1" : : "v"(c.ck_off), "v"(ckv), "s"(c.ck + (size_t)(k - 1) * (size_t)c.ck_pitch) : "memory");
                if (lane == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(stage), "v"(kTpSentinel), "n"((kTpBlock - 1) * 16 + 8) : "memory");
            }
            // ---- the rows of block k+2 (for the look-up wavefront in epoch k+1) ----
            if (k + 2 >= 0 && k + 2 <= kb1) issue_rows(k + 2);
            // ---- the packets of block k+1: fetched until every one of them is there ----
            if (k + 1 >= kb0 && k + 1 <= kb1 && fed) {
                issue_packets(k + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!packets_there(k + 1)) {
                    const unsigned long long w0 = wall_clock64();
                    uint32_t spins = 0;
                    for (;;) {
                        // (a fetch takes a memory round trip: no sleep needed between two of them)
                        issue_packets(k + 1);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        ++spins;
                        if (packets_there(k + 1)) break;
                        // stall detector: 4 s without the block appearing can only be a bug in the hand-off, never an input
                        if ((spins & 255u) == 0u && wall_clock64() - w0 > 400000000ull) {
                            fed = false;
                            break;
                        }
                    }
                    if (verify & 4) {
                        ((lu32_t)(uintptr_t)stat_lds)[0] += spins;
                        t_wait += wall_clock64() - w0;
                    }
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            tp2_barrier();
        }
        // nothing of this workgroup may still be landing in LDS when it ends
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int32_t *m = meta_of(meta, d.idx);
        if (!fed && lane == 0) atomicMin(&m[0], kStatusInternal);
        __threadfence();
        tp2_barrier();      // (the three wavefronts leave through the same number of barriers)
        if ((verify & 4) && lane == 0) {
            uint32_t hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const __attribute__((address_space(3))) uint32_t *sw = (const __attribute__((address_space(3))) uint32_t *)(uintptr_t)stat_lds;
            TpStats st;
            st.spins = sw[0] | ((unsigned long long)((xcc & 0xf) << 16 | (hw & 0xffff))) << 32;
            st.phase[0] = st.phase[1] = 0;
            st.phase[2] = (unsigned long long)sw[10] << 32;   // (high half: HW_ID of the compute wavefront)
            st.extra[0] = sw[8] | ((unsigned long long)sw[2] << 32);    // compute wavefront: cycles at the barrier | cycles inside the frame blocks
            st.extra[1] = sw[9];                                         // look-up wavefront: busy cycles
            st.wait_ticks = (uint32_t)t_wait | ((unsigned long long)sw[2] << 32);
            st.start_tick = __builtin_amdgcn_s_memtime() - stats_out->total_ticks;
            st.total_ticks = wall_clock64() - stats_out->start_tick;
            *stats_out = st;
        }
        return;
    }

    if (role == 2) {
        // ============================================== the look-up wavefront ==============================================
        for (int32_t k = kb0 - 2; k <= kb1 + 1; ++k) {
            const unsigned long long k0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            if (k + 1 >= kb0 && k + 1 <= kb1) {
                const uint32_t tbn = (uint32_t)(k + 1) * kTpBlock;
                const uint32_t rows = lds_rows + (uint32_t)((k + 1) & 1) * kRowSlot;
                {   // block k+1 landed before the last barrier: its finiteness sum (all reads first, then the adds)
                    const uint32_t r = rows + (uint32_t)lane * 16u;
                    constexpr int kReads = !CONTIG ? kTpSlotBytes / 1024 : kRowDmas;
                    f32x4 v[kReads];
#pragma unroll
                    for (int j = 0; j < kReads; ++j) v[j] = lds_f32x4(r + j * 1024);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < kReads; ++j) c.absum += (__builtin_fabsf(v[j][0]) + __builtin_fabsf(v[j][1])) + (__builtin_fabsf(v[j][2]) + __builtin_fabsf(v[j][3]));
                }
                tn_band_block(c, tbn, lane);
                // the emissions of block k+1, per frame and lane {blank, label} ...
                const uint32_t pb = lds_pairs + (uint32_t)((k + 1) & 1) * kTgPairBytes;
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
                tn_band_advance(c);
            }
            if (verify & 4) ((lu32_t)(uintptr_t)stat_lds)[9] += (uint32_t)(__builtin_amdgcn_s_memtime() - k0);
            tp2_barrier();
        }
        // ---- finiteness (as forward_ck: the scores-only form is valid for finite log-probs of sane magnitude) ----
        int32_t *m = meta_of(meta, d.idx);
        const uint32_t abits = __builtin_bit_cast(uint32_t, c.absum) & 0x7fffffffu;
        if (__builtin_amdgcn_ballot_w64(abits > 0x7f800000u)) {
            if (lane == 0) atomicMin(&m[0], kStatusNaN);
        } else if (__builtin_amdgcn_ballot_w64(abits >= __builtin_bit_cast(uint32_t, 1e30f))) {
            if (lane == 0) atomicOr(&m[2], d.W <= kFastMaxBand ? kFlagExact : kFlagDeclined);
        }
        __threadfence();
        tp2_barrier();      // (flagged before the compute wavefront closes the lattice)
        return;
    }

    // =================================================== the compute wavefront ===================================================
    auto barrier_timed = [&]() {
        if (verify & 4) {
            const unsigned long long b0 = __builtin_amdgcn_s_memtime();
            tp2_barrier();
            ((lu32_t)(uintptr_t)stat_lds)[8] += (uint32_t)(__builtin_amdgcn_s_memtime() - b0);
        } else
            tp2_barrier();
    };
    barrier_timed();      // barrier kb0-2
    barrier_timed();      // barrier kb0-1: the pairs and packets of block kb0 are there
    float H[3] = {NINF, NINF, NINF};
    TgIn in[4];
    // per-lane LDS addresses: this block's pairs / packets, the next block's, where the frames drop their packets
    uint32_t pairs_cur, pairs_nxt, pk_cur, pk_nxt;
    {
        const uint32_t p0 = lds_pairs + (uint32_t)lane * 8u, p1 = p0 + kTgPairBytes;
        pairs_cur = (kb0 & 1) ? p1 : p0;
        pairs_nxt = (kb0 & 1) ? p0 : p1;
        const uint32_t q0 = lds_packets, q1 = q0 + kTpBlock * 16;
        pk_cur = (kb0 & 1) ? q1 : q0;
        pk_nxt = (kb0 & 1) ? q0 : q1;
        // (addresses of asm LDS instructions live in vector registers)
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(pk_cur), "=&v"(pk_nxt) : "s"(pk_cur), "s"(pk_nxt));
    }
    uint32_t stage_cur, stage_nxt;
    {
        const uint32_t s0 = lane >= 62 ? lds_stage0 + (uint32_t)(lane - 62) * 8u : lds_stage0 + kTpBlock * 16 + (uint32_t)lane * 8u;
        stage_cur = (kb0 & 1) ? s0 + kStageBytes : s0;
        stage_nxt = (kb0 & 1) ? s0 : s0 + kStageBytes;
    }
    // the first four frames of the first block; every later read is issued four frames ahead of its use (ts_frame)
    tg_read<0>(in[0], pairs_cur, pk_cur);
    tg_read<1>(in[1], pairs_cur, pk_cur);
    tg_read<2>(in[2], pairs_cur, pk_cur);
    tg_read<3>(in[3], pairs_cur, pk_cur);
    tg_wait_all(in);
    H[0] = wave_shr1(in[0].hp[3], c.S[1]);
    H[1] = wave_shr1(in[0].hp[2], c.S[0]);
    H[2] = wave_shr1(in[0].hp[1], H[0]);
    const unsigned long long fr0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
    // Three code regions, one after the other - a guarded first block (t_in inside it), the full blocks, a guarded last block -
    // and never two alternative paths that both carry reads in flight to a merge: there hipcc copies registers whose reads
    // have not landed.  The seams drain (twice per tile).
    auto flip = [&]() {      // the double buffers change roles (exchanges of registers the frames do not write)
        uint32_t x = pairs_cur; pairs_cur = pairs_nxt; pairs_nxt = x;
        x = pk_cur; pk_cur = pk_nxt; pk_nxt = x;
        x = stage_cur; stage_cur = stage_nxt; stage_nxt = x;
    };
    int32_t k = kb0;
    const int32_t k_full_end = (c.t_end % kTpBlock) ? kb1 - 1 : kb1;      // last block that lies wholly inside the tile's life
    if (c.t_in % kTpBlock) {
        c.lds_stage = stage_cur;
        ts_block<M, ZL, true>(c, (uint32_t)k * kTpBlock, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
        tg_wait_all(in);
        flip();
        ++k;
    }
    for (; k <= k_full_end; ++k) {
        c.lds_stage = stage_cur;
        ts_block<M, ZL, false>(c, (uint32_t)k * kTpBlock, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
        flip();
    }
    tg_wait_all(in);
    if (k <= kb1) {
        c.lds_stage = stage_cur;
        ts_block<M, ZL, true>(c, (uint32_t)k * kTpBlock, H, in, pairs_cur, pairs_nxt, pk_cur, pk_nxt);
        tg_wait_all(in);
    }
    if (verify & 4) ((lu32_t)(uintptr_t)stat_lds)[2] += (uint32_t)(__builtin_amdgcn_s_memtime() - fr0);
    barrier_timed();      // barrier kb1+1: the feeder publishes the last block
    tp2_barrier();        // the look-up wavefront has flagged what it had to flag

    // ---- terminal state: the HIGHEST live position of frame T-1 (align.py:99-101), over the tiles alive then ----
    int32_t *m = meta_of(meta, d.idx);
    if ((uint32_t)c.t_end == c.T) {
        // (the only full band mask of a tile's life: cells above hi may hold leaked scores)
        const uint32_t q_last = c.L - (c.L + c.T - 1u) / c.T;   // floor(L (T-1) / T) = L - ceil(L / T)
        const uint32_t lo_last = c.lo_of(q_last), hi_last = c.hi_of(lo_last);
        const float cell[2] = {c.S[0], c.S[1]};
        unsigned long long key = 0;
#pragma unroll
        for (int kk = 0; kk < kTnCells; ++kk) {
            const uint32_t pos = (uint32_t)c.base + (uint32_t)kTnCells * (uint32_t)lane + (uint32_t)kk;
            if (pos >= lo_last && pos < hi_last && cell[kk] != NINF) key = ((unsigned long long)(pos + 1u) << 32) | __builtin_bit_cast(uint32_t, cell[kk]);
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

// One workgroup of three wavefronts per 128-position tile; TsLds<..>::kTotal bytes of LDS (at least).  Tickets as in the other
// tile kernels: the tile a workgroup runs is drawn from a counter, tasks are sorted by first frame.
template <int M, int PITCH, bool CONTIG>
__global__ __launch_bounds__(192) void forward_ts_kernel(const Lattice *__restrict__ lats, const TileTask *__restrict__ tasks, int n_tasks, int32_t *meta, char *halo,
                                                         TileAux *aux, uint32_t *ticket, int verify, TpStats *stats, uint32_t *cu_rank)
{
    extern __shared__ __attribute__((aligned(16))) char tp_lds[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&tp_lds[0];
    typedef TsLds<PITCH, CONTIG> Lds;
    volatile uint32_t *s_ticket = reinterpret_cast<volatile uint32_t *>(&tp_lds[Lds::kStat + 48]);
    // Tickets are drawn in order of arrival, and the first tickets are the tiles that are alive first: of the workgroups that
    // start together, the FIRST on each CU draws at once, the second and third wait a little - so the launch's first tiles sit
    // on different CUs instead of wherever the race for the counter put them.  Tiles that share a CU share its LDS pipe (a frame
    // takes 95 cycles beside one other tile, 86 alone), a chain follows its slowest tile, and with 80 tiles alive on 256 CUs
    // a tenth of them had a neighbour (8 chapters: 3.17 ms against 2.94 with one workgroup per CU).  Speed only: any order of
    // tickets is correct.
    const uint32_t my_cu = cu_slot();
    if (threadIdx.x == 0) {
        const uint32_t rank = atomicAdd(&cu_rank[my_cu], 1u);
        for (uint32_t r = 0; r < rank && r < 3u; ++r) __builtin_amdgcn_s_sleep(96);      // ~2.5 us each: a 1024-workgroup grid starts within 0.7 us
        *s_ticket = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    const uint32_t tix = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_ticket);
    __syncthreads();
    if (tix < (uint32_t)n_tasks) {
        const TileTask &tk = tasks[tix];
        const Lattice &d = lats[__builtin_amdgcn_readfirstlane(tk.lat)];
        const int flags = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2]);
        if (flags & kFlagZeroLabel) ts_run_tile<M, true, PITCH, CONTIG>(d, tk, meta, halo, aux, lds0, verify, stats + tix);
        else ts_run_tile<M, false, PITCH, CONTIG>(d, tk, meta, halo, aux, lds0, verify, stats + tix);
    }
    if (threadIdx.x == 0) atomicSub(&cu_rank[my_cu], 1u);      // (the compute wavefront: the last of the three to leave)
}

}  // namespace ka
