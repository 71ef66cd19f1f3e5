// ka_wave_forward.hpp — label preparation and the forward DP with one wavefront per lattice (exact and checkpointed forms),
// plus the generic forward kernel.  Included by ka_wave_fwd.hip only (it holds non-template kernels).
#pragma once
#include "ka_device.hpp"

namespace ka {

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
    // row min(t+D, T-1) of the current frame t, as a byte offset into the lattice's log-probs (forward_ck_kernel has made sure
    // that it fits 32 bits)
    const uint32_t ld32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ld), last_row = (T - 1u) * ld32;
    uint32_t row_ahead = (D < T ? (uint32_t)D : T - 1u) * ld32;
    const u32x4 lp_rsrc = lp_descriptor(d.lp, (uint64_t)T * ld);
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
#ifndef KA_FWD_FLOOR
                absum += __builtin_fabsf(rn);
#endif
                const float e0t = e0[dd & 1];
                frame_scores<M, ZL, 3>(P, h1, h2, h3, E, vz, f32x2{e0t, e0t}, la, lrow);
                // prefetch the row of frame t+D (the last row again once there is none: never consumed)
                row_reload_buf(rows[dd], lane_off, lp_rsrc, row_ahead);
                row_ahead = min(row_ahead + ld32, last_row);
                if (dd == D - 1 && ((tb | mask_every4) & 4u) != 0) {
#ifndef KA_FWD_FLOOR
                    mask_scores<15>(P, mk, NINF);
#endif
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
#ifdef KA_FWD_FLOOR
                // KA_FWD_FLOOR (a timing build, results wrong by design: profiles/r04_forward_floor.json): the recurrence, the row
                // pipeline and the checkpoint stores only - no band (no masks, no band steps, no re-labelling of lanes), no
                // finiteness sum.  What the band handling costs is the difference to the real kernel.
                if (false) {
#else
                rem += dr;
                if (__builtin_expect(rem >= thr, 0)) {
#endif
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
    // the row pipeline addresses the log-probs with 32-bit byte offsets: a lattice of 4 GB or more goes to the exact kernels
    if ((uint64_t)(uint32_t)d.T * (uint64_t)d.ld * 4ull > 0xffffffffull) {
        if (threadIdx.x == 0) atomicOr(&meta_of(meta, d.idx)[2], kFlagExact);
        return;
    }
    forward_ck<M, ZL>(d, meta);
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

}  // namespace ka
