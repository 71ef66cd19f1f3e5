// ka_tiled2.hpp — the tile pipeline of ka_tiled.hpp with TWO wavefronts per tile: one computes, one feeds.
//
// Why: a lattice's tiles form a chain, so a lone lattice (and a book: its longest chapter) runs at the speed of ONE
// wavefront's frame loop - and a wavefront that is alone on its SIMD issues one instruction every four cycles, whatever
// the instruction.  In the one-wavefront tile (ka_tiled.hpp, tp_run_tile) a block of 32 frames costs ~4000 cycles of frames
// and ~2700 cycles of everything around them: the wait for the staged block, its finiteness sum, the progress store, the
// poll of the tile below, the LDS-DMA requests of the block three ahead, the publish of the block's halo packets
// (profiles/r03_tile_stats_cfg2_one_wave.txt).  None of that depends on the scores.  Here wavefront 1 of the workgroup (the
// FEEDER) does all of it and wavefront 0 (the COMPUTE wavefront) only runs frames; they meet at one s_barrier per block.
//   iteration `it` (both wavefronts, after the barrier):
//     compute:  frames of block it (LDS rows / packets of blocks it and it+1 - the reads run two frames ahead), drops its
//               per-frame packets into staging buffer it & 1, stores the checkpoint if the block ends on one
//     feeder:   polls the tile below for block it+2 and requests it, publishes block it-1's packets (staging buffer
//               (it-1) & 1), sums block it+1 for the finiteness check, works out block it+1's band bookkeeping, then waits
//               for EVERYTHING it has in flight (s_waitcnt vmcnt(0)) and announces block it-1 in the progress word
//   so the compute wavefront finds blocks it+1 and it+2 landed at barrier it+1, the requests have the whole iteration of the
//   compute wavefront (~1.8 us) to land, and a block is announced one iteration after it was computed.
// Everything of the hand-off protocol (sc1 packets, in-order vmcnt accounting, progress words, bounded stall detector) is
// the feeder's alone, exactly as in the one-wavefront form; the compute wavefront's only vector-memory instruction is the
// checkpoint store.  Same LDS request (40 KB: 4 workgroups per CU), twice the wavefronts.
#pragma once
#include "ka_tiled.hpp"

namespace ka {


// LDS map of a workgroup: kTpRing blocks of rows - as they lie in memory when the rows are contiguous (32 x PITCH bytes, rounded up
// to whole 1-KB LDS-DMA instructions: 5 KB for V = 39), else 32 rows of 256 bytes -, the ring's packets, the poll words, two
// publish staging buffers, the diagnostic words (ticket at +48), two buffers of band words.  27.1 KB for V = 39 with contiguous
// rows: FIVE workgroups per CU when the engine asks for no more (launches whose tiles outnumber the slots), 39.1 KB otherwise.
template <int PITCH, bool CONTIG>
struct Tp2Lds {
    static constexpr int kRowDmas = !CONTIG ? kTpBlock : (kTpBlock * PITCH + 1023) / 1024;
    static constexpr int kSlot = CONTIG ? kRowDmas * 1024 : kTpSlotBytes;
    static constexpr int kHalo = kTpRing * kSlot;
    static constexpr int kTicket = kHalo + kTpRing * kTpBlock * 16 + 16 + kTp2StageBytes + 48;
    static constexpr int kTotal = kHalo + kTpRing * kTpBlock * 16 + 16 + kTp2StageBytes + 64 + 2 * kTp2BandBytes;
};
static_assert(Tp2Lds<256, false>::kTotal <= (int)kTpLdsRequest && Tp2Lds<256, true>::kTotal <= (int)kTpLdsRequest, "four workgroups per CU");
static_assert(5 * ((Tp2Lds<156, true>::kTotal + 511) / 512 * 512) <= 160 * 1024, "five workgroups per CU with V = 39");

// the barrier of an iteration: each side first finishes what the other is going to look at (the compute wavefront its LDS
// writes; the feeder has already waited for its LDS-DMA with a counted vmcnt) - NOT the vmcnt(0) of __syncthreads, which
// would drain the feeder's requests
__device__ __forceinline__ void tp2_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int M, bool ZL, int PITCH, bool CONTIG>
__device__ __forceinline__ void tp2_run_tile(const Lattice &d, const TileTask &tk, int32_t *meta, char *halo, gu32w_t prog, TileAux *aux,
                                             uint32_t lds_rows, uint32_t lds_halo, int verify, TpStats *stats_out)
{
    typedef __attribute__((address_space(3))) uint32_t *lu32_t;
    const int lane = threadIdx.x & 63;
    const bool feeder = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;
    const uint32_t lds_poll = lds_halo + kTpRing * kTpBlock * 16;
    const uint32_t lds_stage0 = lds_poll + 16;                                  // two staging buffers of kTpStageBytes
    const uint32_t stat_lds = lds_stage0 + kTp2StageBytes;                      // diagnostic words, then two flag words
    const uint32_t lds_band = stat_lds + 64;                                    // two buffers of kTp2BandBytes: the band code's kill words and event mask of a block
    if (threadIdx.x < 10) ((lu32_t)(uintptr_t)stat_lds)[threadIdx.x] = 0;
    unsigned long long ph = 0;
    auto phase = [&](int w) {
        if (verify & 4) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (w >= 0) ((lu32_t)(uintptr_t)stat_lds)[w] += (uint32_t)(now - ph);
            ph = now;
        }
    };
    if ((verify & 4) && !feeder && lane == 0) {   // where the compute wavefront runs (the feeder reports its own place below)
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        ((lu32_t)(uintptr_t)stat_lds)[10] = hw & 0xffffu;
    }
    if ((verify & 4) && feeder) {   // start stamps: wall clock (100 MHz) and shader clock
        stats_out->start_tick = (unsigned long long)wall_clock64();
        stats_out->total_ticks = __builtin_amdgcn_s_memtime();
    }
    // the compute wavefront is the chain: where it shares a SIMD with feeders and with other kernels' wavefronts it issues first
    if (!feeder) __builtin_amdgcn_s_setprio(3);
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
    static_assert(kTpBlock * 16 + 62 * 16 + (kTpBlock - 1) * 16 + 16 <= kTpStageBytes, "publish staging");
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
    c.lds_packets = lds_stage0;
    c.lds_stage = 0;
    // slot t_in of the upper boundary = the state before the tile's first frame: lane 63's cells, all -inf.  The FEEDER
    // stores it: every store the progress word vouches for is in its own in-order vmcnt history.
    if (feeder) tp_halo_store<0>(c.halo_out, f32x4{NINF, NINF, NINF, NINF}, 1ull << 63);

    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) char *lchar_t;
    const uint32_t last_row = c.T - 1;
    const uint32_t last_slot = (uint32_t)c.t_end - 1;     // this tile reads slots t_in .. t_end - 1
    auto ring = [](int32_t k) { return (uint32_t)((k % kTpRing + kTpRing) % kTpRing); };
    constexpr int kRowDmas = Tp2Lds<PITCH, CONTIG>::kRowDmas;   // LDS-DMA instructions per block of rows
    constexpr uint32_t kSlot = Tp2Lds<PITCH, CONTIG>::kSlot;     // LDS bytes of a block of rows
    static_assert(CONTIG || PITCH == kTpRowBytes, "row-by-row staging uses 256-byte rows");
    auto issue_block = [&](int32_t k) {    // k >= 0
        const uint32_t tb = (uint32_t)k * kTpBlock, slot = ring(k);
        lchar_t dst = (lchar_t)(uintptr_t)(c.lds_rows + slot * kSlot);
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
        if (lane < kTpBlock) {
            uint32_t s = tb + (uint32_t)lane;
            s = s < (uint32_t)c.t_in ? (uint32_t)c.t_in : (s > last_slot ? last_slot : s);
            __builtin_amdgcn_global_load_lds((gptr_t)(c.halo_in + (size_t)(s - (uint32_t)c.t_in) * 16), (lptr_t)(lchar_t)(uintptr_t)(c.lds_halo + slot * (kTpBlock * 16)), 16, 0, 16);
        }
        if (lane == 0) __builtin_amdgcn_global_load_lds((gptr_t)c.prog_in, (lptr_t)(lchar_t)(uintptr_t)(lds_poll + slot * 4), 4, 0, 16);
    };
    bool stale = false;
    auto landed_block = [&](int32_t k) {
        const uint32_t slot = ring(k);
        const uint32_t r = c.lds_rows + slot * kSlot + (uint32_t)lane * 16u;
        constexpr int kReads = kSlot / 1024;
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
    auto need_for = [&](int32_t k) {
        const uint32_t n = (uint32_t)(k + 1) * kTpBlock;
        return n < (uint32_t)c.t_end ? n : (uint32_t)c.t_end;
    };

    const int32_t kb0 = c.t_in / kTpBlock, kb1 = (c.t_end - 1) / kTpBlock;   // first and last block
    bool fed = true;
    TpIn cur = {f32x2{0.0f, 0.0f}, 0.0f, f32x4{NINF, NINF, NINF, NINF}}, nxt = cur;
    float H[3] = {NINF, NINF, NINF};
    // Iterations kb0-2, kb0-1 prime the feeder's pipeline; iteration kb1+1 publishes the last block.
    for (int32_t it = kb0 - 2; it <= kb1 + 1; ++it) {
        const uint32_t tb = (uint32_t)(it * kTpBlock);              // (wraps in the priming iterations of block 0: not used there)
        if ((verify & 4) && !feeder) {      // the compute wavefront's cycles at the barrier (words 8, 9: as the 128-position tile's)
            const unsigned long long b0 = __builtin_amdgcn_s_memtime();
            tp2_barrier();
            if (it >= kb0 && it <= kb1 + 1) ((lu32_t)(uintptr_t)stat_lds)[8] += (uint32_t)(__builtin_amdgcn_s_memtime() - b0);
        } else
        tp2_barrier();
        if (feeder) {
            phase(-1);
            // block it+2 is requested now (LDS ring slot of block it-2): the tile below must have published its packets (the
            // freshest look at its progress word that has landed came with block it+1)
            if (it + 2 <= kb1 && fed) {
                const uint32_t have = it + 1 >= kb0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, lds_f32(lds_poll + ring(it + 1) * 4))) : 0u;
                // (no hysteresis here: the one-wavefront tile asks for two blocks more than it needs once it has to wait, so that
                //  its frames are not interrupted by a poll per block; here the frames run in the other wavefront)
                fed = tp_wait_progress(c.prog_in, need_for(it + 2), need_for(it + 2), have, stat_lds);
            }
            phase(5);
            if (it + 2 >= 0 && it + 2 <= kb1 + 1) issue_block(it + 2);
            phase(6);
            // block it-1 is complete in staging buffer (it-1) & 1: lane f < 32 stores the packet of frame f as slot tb-32+f+1
            const bool published = it - 1 >= kb0 && it - 1 <= kb1;
            if (published) {
                c.lds_packets = lds_stage0 + (uint32_t)((it - 1) & 1) * kTpStageBytes;
                tp_publish_block(c, tb - kTpBlock, lane);
            }
            phase(7);
            // block it+1 landed before the last barrier: its finiteness sum
            if (it + 1 >= kb0 && it + 1 <= kb1) landed_block(it + 1);
            // the band bookkeeping of the NEXT block (which positions of the tile enter or leave the band in which frame), for
            // the compute wavefront to pick up after the next barrier: ~60 instructions it does not have to issue
            if (it + 1 >= kb0 && it + 1 <= kb1) {
                tp_band_block(c, tb + kTpBlock, lane);
                const uint32_t bb = lds_band + (uint32_t)((it + 1) & 1) * kTp2BandBytes;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                *(__attribute__((address_space(3))) u32x2 *)(uintptr_t)(bb + (uint32_t)lane * 8u) = u32x2{c.KL, c.KE};
                if (lane == 0) *(lu32_t)(uintptr_t)(bb + 512u) = c.ev;
                tp_band_advance(c);
            }
            phase(4);
            // EVERYTHING this wavefront has in flight is waited for, once per iteration: the requests of block it+2 (the compute
            // wavefront reads its first rows in the next iteration) and the packets just published.  No counted wait: a
            // counted vmcnt orders loads among loads and stores among stores, not one against the other - announcing block
            // it-1 behind "at most the requests are outstanding" would rest on stores retiring before younger loads, which
            // is not something the ISA promises.  The requests have had the whole iteration to land, and the feeder is idle
            // for half of it anyway.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            phase(3);
            // slots <= tb are in memory - but never vouch for more than the tile's own frames have produced: the slots behind
            // t_end are filled (with -inf) after the loop, and the final progress word covers those
            if (published) tp_prog_store(c.prog_out, (tb < (uint32_t)c.t_end ? tb : (uint32_t)c.t_end) + 1);
        } else if (it >= kb0 && it <= kb1) {
            const unsigned long long g0 = (verify & 4) ? __builtin_amdgcn_s_memtime() : 0ull;
            const uint32_t slot = ring(it), nslot = ring(it + 1);
            uint32_t rc = c.lds_rows + slot * kSlot, rn = c.lds_rows + nslot * kSlot;
            uint32_t hc = c.lds_halo + slot * (kTpBlock * 16), hn = c.lds_halo + nslot * (kTpBlock * 16);
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(rc), "=&v"(rn), "=&v"(hc), "=&v"(hn) : "s"(rc), "s"(rn), "s"(hc), "s"(hn));
            TpAddr A[2] = {{rc + (uint32_t)c.la0, rc + (uint32_t)c.la1, rc, hc}, {rn + (uint32_t)c.la0, rn + (uint32_t)c.la1, rn, hn}};
            asm volatile("" : "+v"(A[0].l0), "+v"(A[0].l1), "+v"(A[1].l0), "+v"(A[1].l1));
            // where this block's frames drop their packets: lane 63's into the packet row, the others' into scratch behind it
            {
                const uint32_t pk = lds_stage0 + (uint32_t)(it & 1) * kTpStageBytes;
                c.lds_stage = lane == 63 ? pk : pk + kTpBlock * 16 + (uint32_t)lane * 16u;
            }
            if (it == kb0) {
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
                tp_block_frames<M, ZL, PITCH, false, 0>(c, tb, H, cur, nxt, A, NINF);
                if ((tb + kTpBlock) % kCkFrames == 0 && tb + kTpBlock < c.T) tp_checkpoint(c, tb + kTpBlock);
            } else {
                tp_block_frames<M, ZL, PITCH, true, 0>(c, tb, H, cur, nxt, A, NINF);
                if ((tb + kTpBlock) % kCkFrames == 0 && (int32_t)(tb + kTpBlock) <= c.t_end && tb + kTpBlock < c.T) tp_checkpoint(c, tb + kTpBlock);
            }
            if (verify & 4) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                // frames + checkpoint of the block | everything of the iteration that is not frames (set-up, checkpoint)
                ((lu32_t)(uintptr_t)stat_lds)[2] += (uint32_t)(now - fr0);
                ((lu32_t)(uintptr_t)stat_lds)[9] += (uint32_t)(fr0 - g0);
            }
        }
    }
    // nothing of this workgroup may still be landing in LDS or in a register when it ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int32_t *m = meta_of(meta, d.idx);
    if (feeder) {
        // ---- finiteness (as forward_ck: the scores-only form is valid for finite log-probs of sane magnitude).  Flagged
        // before the tile reports itself done (barrier below), so that whoever closes the lattice sees the flag.
        if ((!fed || stale) && lane == 0) atomicMin(&m[0], kStatusInternal);
        const uint32_t abits = __builtin_bit_cast(uint32_t, c.absum) & 0x7fffffffu;
        if (__builtin_amdgcn_ballot_w64(abits > 0x7f800000u)) {
            if (lane == 0) atomicMin(&m[0], kStatusNaN);
        } else if (__builtin_amdgcn_ballot_w64(abits >= __builtin_bit_cast(uint32_t, 1e30f))) {
            if (lane == 0) atomicOr(&m[2], d.W <= kFastMaxBand ? kFlagExact : kFlagDeclined);
        }
        // ---- hand the rest of the upper boundary over: after t_end the whole tile is below the band = -inf ----
        const f32x4 dead = {NINF, NINF, NINF, NINF};
        for (int64_t s = (int64_t)c.t_end + 1 + lane; s <= (int64_t)tk.fill_end; s += 64)
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"((uint32_t)((s - c.t_in) * 16)), "v"(dead), "s"(c.halo_out) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
            st.wait_ticks = sw[1] | ((unsigned long long)sw[2] << 32);
            st.extra[0] = sw[8] | ((unsigned long long)sw[2] << 32);    // compute wavefront: cycles at the barrier | cycles inside the frame blocks (+ checkpoint stores)
            st.extra[1] = sw[9];                                         // ... and between a barrier and the block's first frame
            st.start_tick = __builtin_amdgcn_s_memtime() - stats_out->total_ticks;
            st.total_ticks = wall_clock64() - stats_out->start_tick;
            *stats_out = st;
        }
        return;
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

// One workgroup of TWO wavefronts per tile (40 KB of LDS requested: four workgroups per CU).  Tickets as in
// forward_tp_kernel: the tile a workgroup runs is drawn from a counter, tasks are sorted by first frame.
template <int M, int PITCH, bool CONTIG>
__global__ __launch_bounds__(128) void forward_tp2_kernel(const Lattice *__restrict__ lats, const TileTask *__restrict__ tasks, int n_tasks,
                                                          int32_t *meta, char *halo, uint32_t *prog, TileAux *aux, uint32_t *ticket, int verify, TpStats *stats)
{
    typedef Tp2Lds<PITCH, CONTIG> Lds;
    extern __shared__ __attribute__((aligned(16))) char tp_lds[];
    const uint32_t lds_rows = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&tp_lds[0];
    const uint32_t lds_halo = lds_rows + Lds::kHalo;
    // (the ticket goes through a word of the dynamic LDS block: a static __shared__ variable on top of the request would cost a
    //  workgroup per CU)
    volatile uint32_t *s_ticket = reinterpret_cast<volatile uint32_t *>(&tp_lds[Lds::kTicket]);
    if (threadIdx.x == 0) *s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tix = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_ticket);
    __syncthreads();
    if (tix >= (uint32_t)n_tasks) return;
    const TileTask &tk = tasks[tix];
    const Lattice &d = lats[__builtin_amdgcn_readfirstlane(tk.lat)];
    const int flags = __builtin_amdgcn_readfirstlane(meta_of(meta, d.idx)[2]);
    if (flags & kFlagZeroLabel)
        tp2_run_tile<M, true, PITCH, CONTIG>(d, tk, meta, halo, (gu32w_t)prog, aux, lds_rows, lds_halo, verify, stats + tix);
    else
        tp2_run_tile<M, false, PITCH, CONTIG>(d, tk, meta, halo, (gu32w_t)prog, aux, lds_rows, lds_halo, verify, stats + tix);
}

}  // namespace ka
