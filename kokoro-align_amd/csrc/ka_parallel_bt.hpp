// ka_parallel_bt.hpp — chunk-parallel backtrace for the checkpointed forms.
//
// backtrace_rc_kernel<.., PAR = false> walks a lattice's 32-frame chunks one after the other, because the position at
// which the best path enters a chunk is only known once the chunk above it has been walked: 5 us per chunk, 8 ms for a
// 50 000-frame lattice - as long as the whole forward pass - with one wavefront busy.  Here every chunk is walked at the
// same time.  What makes that possible is the MAP of a chunk: for EVERY position p the band holds at the chunk's last
// frame, how far the best path into p has risen since the last frame of the chunk before (0..96 positions; for the
// cell that is on the best path of the lattice this is exactly what the walk would find).  Maps need no entry position:
//   1. chunk_map_kernel      one wavefront per (chunk, 408-position segment of its band): the chunk is recomputed forward
//                            from its checkpoint with the reference's exact first-max rule (align.py:83), and every cell
//                            carries along the position its best path had at the chunk's start ("origin"); 8 cells per
//                            lane, 512 positions per wavefront of which the lowest 96 only warm up (what is unknown below
//                            the window climbs 3 positions per frame, exactly as in backtrace_rc).
//   2. compose_maps_kernel   32 chunk maps -> one super-chunk map (1024 frames): one thread per band position.
//   3. chain_entries_kernel  one workgroup per lattice: the end position runs down the super-chunk maps (a few dozen
//                            dependent loads), then every super-chunk runs its own 32 chunk maps: the entry position of
//                            every chunk.
//   4. backtrace_rc_kernel<.., PAR = true>   one wavefront per chunk, all chunks of the launch at once.
// Bit-exact by construction: the maps are made of the same float operations, in the same order, as the forward kernels and
// the reference; the parity suite runs in this form too (KA_MODE_* x parallel backtrace).
#pragma once
#include "ka_device.hpp"

namespace ka {


__device__ __forceinline__ int supers_of(int n_chunks) { return (n_chunks + kSuperChunks - 1) / kSuperChunks; }
__device__ __forceinline__ int chunk_last_frame(int c, int T) { return min(c * kCkFrames + kCkFrames - 1, T - 1); }
// band of frame t (align.py:64-65), 64-bit like the reference
__device__ __forceinline__ void band_of(uint32_t L, uint32_t T, uint32_t B, uint32_t t, uint32_t &lo, uint32_t &hi)
{
    const uint32_t q = (uint32_t)(((uint64_t)L * t) / T);
    const int32_t d = (int32_t)q - (int32_t)(B >> 1);
    lo = (uint32_t)(d > 0 ? d : 0);
    hi = (L - lo < B) ? L : lo + B;
}
__device__ __forceinline__ float select_f(float a, float b, uint64_t mask) { return select_by_mask(a, b, mask); }
__device__ __forceinline__ int select_i(int a, int b, uint64_t mask)
{
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
// a pointer the compiler cannot prove wave-uniform (it came out of a search loop), for an "s" operand of inline asm
// (v_readfirstlane is a VALU write of an SGPR, and a vector-memory instruction needs FIVE wait states before it may
//  read such an SGPR; hipcc pads its own instructions but cannot see into an asm statement - without the s_nop the load
//  used the registers' previous contents: the pointer without its offset)
template <class P>
__device__ __forceinline__ P uniform_ptr(P p)
{
    const uint64_t v = (uint64_t)p;
    uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    asm volatile("s_nop 4" : "+s"(w));
    return (P)w;
}
__device__ __forceinline__ int wave_shr1_i(int first, int x) { return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xF, 0xF, false); }

// one label cell with origin: candidates a_j + e in move order, the first one that attains the maximum wins
template <int M, bool ZL>
__device__ __forceinline__ void cm_label(float a0, float a1, float a2, float a3, int o0, int o1, int o2, int o3, float e, float veto, float &s, int &o)
{
    const float c0 = a0 + e;
    if constexpr (M == 1) {
        s = c0;
        o = o0;
    } else {
        const float c1 = a1 + e;
        if constexpr (M == 2) {
            s = __builtin_fmaxf(c0, c1);
            o = select_i(o1, o0, feq(c0, s));
        } else {
            float c2 = a2 + e;
            if constexpr (ZL) c2 = __builtin_fminf(c2, veto);
            if constexpr (M == 3) {
                s = __builtin_fmaxf(__builtin_fmaxf(c0, c1), c2);
                o = select_i(select_i(o2, o1, feq(c1, s)), o0, feq(c0, s));
            } else {
                const float c3 = a3 + e;
                s = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(c0, c1), c2), c3);
                o = select_i(select_i(select_i(o3, o2, feq(c2, s)), o1, feq(c1, s)), o0, feq(c0, s));
            }
        }
    }
}
// one blank cell: moves {0, 1, 3} (move 2 is vetoed on blanks, align.py:80-81)
template <int M>
__device__ __forceinline__ void cm_blank(float a0, float a1, float a3, int o0, int o1, int o3, float e, float &s, int &o)
{
    const float c0 = a0 + e;
    if constexpr (M == 1) {
        s = c0;
        o = o0;
    } else {
        const float c1 = a1 + e;
        if constexpr (M <= 3) {
            s = __builtin_fmaxf(c0, c1);
            o = select_i(o1, o0, feq(c0, s));
        } else {
            const float c3 = a3 + e;
            s = __builtin_fmaxf(__builtin_fmaxf(c0, c1), c3);
            o = select_i(select_i(o3, o1, feq(c1, s)), o0, feq(c0, s));
        }
    }
}
// Cells per lane of a map wavefront: C = 8 (512 positions, 408 delivered) or 18 (1152 positions, 1048 delivered).  Every
// wavefront recomputes 96 warm-up positions below what it delivers, so the reference's band of 1000 took three wavefronts of 8
// cells - 1536 positions recomputed for ~1010 - and takes ONE of 18: 1152 (round 4; the chunk maps are a third of a book's
// launch).  Bands up to 400 keep C = 8; wider ones than 1048 take several segments of 18.
template <int C>
struct CmGeom {
    static constexpr int kSpan = 64 * C;                    // positions a wavefront recomputes
    static constexpr int kOut = kSpan - kCmWarm - 8;        // positions it delivers (a multiple of 8: segments start on multiples of 8)
    static_assert(C % 2 == 0 && kOut % 8 == 0 && kOut > 0, "a lane holds (blank, label) pairs");
};
static_assert(CmGeom<kCmCells>::kOut == kCmOut && CmGeom<kCmCellsWide>::kOut == kCmOutWide, "ka_types.hpp states the delivered widths for the host");
template <int C>
struct CmMasks {
    uint64_t m[C];   // m[k]: lanes whose cell k (position w0 + C lane + k) is inside the band
};
template <int C>
__device__ __forceinline__ void cm_masks(CmMasks<C> &mk, int32_t lo_rel, int32_t hi_rel)
{
    constexpr int kSpan = CmGeom<C>::kSpan;
    lo_rel = lo_rel < -2 * C ? -2 * C : (lo_rel > kSpan + 2 * C ? kSpan + 2 * C : lo_rel);
    hi_rel = hi_rel < -2 * C ? -2 * C : (hi_rel > kSpan + 2 * C ? kSpan + 2 * C : hi_rel);
#pragma unroll
    for (int k = 0; k < C; ++k) {
        // lanes l with lo_rel <= C l + k < hi_rel: l in [ceil((lo_rel - k) / C), ceil((hi_rel - k) / C))  (C a constant: no division)
        int a = (lo_rel - k + C - 1 + 4 * C) / C - 4, b = (hi_rel - k + C - 1 + 4 * C) / C - 4;     // (+ 4 C: the numerators stay positive)
        a = a < 0 ? 0 : (a > 64 ? 64 : a);
        b = b < 0 ? 0 : (b > 64 ? 64 : b);
        const uint32_t n = b > a ? (uint32_t)(b - a) : 0u;
        mk.m[k] = n == 0 ? 0ull : ((n >= 64u ? ~0ull : ((1ull << n) - 1ull)) << a);
    }
}

// The map of one (chunk c >= 1, segment y) of lattice d: one wavefront of C cells per lane.  The chunk's rows go through LDS
// (`s_rows`, 8 KB: the frame loop stays a loop - unrolled 32 times with the rows in registers it is 280 KB of code per
// instance - and an emission is one ds_read_b32).
template <int M, bool ZL, int C>
__device__ __forceinline__ void chunk_map_task(const Lattice &d, uint32_t c, uint32_t y, int lane, float (*s_rows)[64])
{
    constexpr int kOut = CmGeom<C>::kOut, kPairs = C / 2;
    const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane(d.T);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane(d.L);
    const uint32_t B = (uint32_t)__builtin_amdgcn_readfirstlane(d.beam);
    const float NINF = ninf();
    const uint32_t t0 = c * kCkFrames, te = (uint32_t)chunk_last_frame((int)c, (int)T);
    const auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // band of the chunk's last frame and this wavefront's segment of it
    uint32_t lo_e, hi_e;
    band_of(L, T, B, te, lo_e, hi_e);
    lo_e = uni(lo_e);
    hi_e = uni(hi_e);
    const uint32_t seg_lo = (lo_e & ~7u) + y * kOut;
    if (seg_lo >= hi_e) return;
    const uint32_t w0 = uni(seg_lo > (uint32_t)kCmWarm ? seg_lo - kCmWarm : 0u);   // first position of the window (a multiple of 8); scalar from here on
    const uint32_t ck_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.ck_mask);
    const size_t ck_pitch = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(d.ck_pitch);
    const uint32_t p_lane = w0 + (uint32_t)C * (uint32_t)lane;         // the lane's first position (even: a blank)

    // ---- loads: labels, checkpoint, the chunk's rows (inline asm, one wait for all: as in backtrace_rc) ----
    const char *lp = reinterpret_cast<const char *>(d.lp);
    const size_t ldb = (size_t)d.ld * 4;
    const uint32_t col_off = (lane < d.V ? (uint32_t)lane : 0u) * 4u;
    int la[kPairs];      // 4 * label of the lane's label cells (positions p_lane + 1, 3, ...); labx is zero padded past S
    float sc[C];         // scores of the lane's positions after frame t0 - 1
    {
        gci32_t labx = uniform_ptr((gci32_t)d.labx + (w0 >> 1));
        const char *row = uniform_ptr(reinterpret_cast<const char *>(d.bp) + ((size_t)c - 1) * ck_pitch);
        if constexpr (C == 8) {
            v4i_t lab4;
            f32x4 ck0, ck1;
            const uint32_t off = (p_lane & ck_mask) * 4u;   // (p_lane is a multiple of 8: the 8 cells do not wrap)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(lab4) : "v"((uint32_t)lane * 16u), "s"(labx) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ck0) : "v"(off), "s"(row) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ck1) : "v"(off), "s"(row) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lab4), "+v"(ck0), "+v"(ck1) : : "memory");
            la[0] = lab4.x; la[1] = lab4.y; la[2] = lab4.z; la[3] = lab4.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) { sc[k] = ck0[k]; sc[4 + k] = ck1[k]; }
        } else {
            // a lane's C positions are p_lane .. p_lane + C - 1 with p_lane a multiple of 2 only: pairs of cells (8 bytes, never
            // split by the checkpoint ring's wrap, which is a multiple of 1024 positions), one label per pair
            typedef float f32x2a __attribute__((ext_vector_type(2)));
            f32x2a ck[kPairs];
#pragma unroll
            for (int i = 0; i < kPairs; ++i) {
                asm volatile("global_load_dword %0, %1, %2" : "=v"(la[i]) : "v"(((uint32_t)kPairs * (uint32_t)lane + (uint32_t)i) * 4u), "s"(labx) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(ck[i]) : "v"(((p_lane + 2u * (uint32_t)i) & ck_mask) * 4u), "s"(row) : "memory");
            }
#pragma unroll
            for (int i = 0; i < kPairs; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(la[i]), "+v"(ck[i]) : : "memory");
#pragma unroll
            for (int i = 0; i < kPairs; ++i) { sc[2 * i] = ck[i][0]; sc[2 * i + 1] = ck[i][1]; }
        }
    }
    const int n = (int)(te - t0 + 1);
    {
        float rows[kCkFrames];
        const char *rp = uniform_ptr(lp + (size_t)t0 * ldb);
#pragma unroll
        for (int f = 0; f < kCkFrames; ++f) {
            rows[f] = row_load(col_off, rp);
            rp += (f + 1 < n) ? ldb : 0;
        }
#pragma unroll
        for (int f = 0; f < kCkFrames; ++f) {
            if (f == 0) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            asm volatile("" : "+v"(rows[f]) : : "memory");
            s_rows[f][lane] = rows[f];
        }
    }
    float veto[kPairs];
#pragma unroll
    for (int i = 0; i < kPairs; ++i) veto[i] = (ZL && la[i] == 0) ? NINF : __builtin_inff();

    // ---- state: scores sc[k], origins og[k] (window-relative position at the chunk's start) ----
    uint32_t q, rem;   // floor(L t / T) and remainder of the frame being computed
    const uint32_t dq = L / T, dr = L % T;
    {
        const uint64_t x = (uint64_t)L * (t0 - 1);   // band of frame t0 - 1 limits the checkpoint
        q = uni((uint32_t)(x / T));
        rem = uni((uint32_t)(x % T));
    }
    uint32_t lo, hi;
    {
        const int32_t dl = (int32_t)q - (int32_t)(B >> 1);
        lo = (uint32_t)(dl > 0 ? dl : 0);
        hi = (L - lo < B) ? L : lo + B;
    }
    CmMasks<C> mk;
    cm_masks<C>(mk, (int32_t)lo - (int32_t)w0, (int32_t)hi - (int32_t)w0);
    int og[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        sc[k] = select_f(NINF, sc[k], mk.m[k]);
        og[k] = C * lane + k;
    }
    for (int f = 0; f < n; ++f) {
        // band of frame t0 + f
        rem += dr;
        q += dq;
        if (rem >= T) { rem -= T; ++q; }
        {
            const int32_t dl = (int32_t)q - (int32_t)(B >> 1);
            const uint32_t nlo = (uint32_t)(dl > 0 ? dl : 0);
            const uint32_t nhi = (L - nlo < B) ? L : nlo + B;
            if (nlo != lo || nhi != hi) {
                lo = nlo;
                hi = nhi;
                cm_masks<C>(mk, (int32_t)lo - (int32_t)w0, (int32_t)hi - (int32_t)w0);
            }
        }
        const float e0 = s_rows[f][0];
        float el[kPairs];
#pragma unroll
        for (int i = 0; i < kPairs; ++i) el[i] = lds_col(&s_rows[f][0], la[i]);
        // the three cells below the lane's first one (lane 0: nothing is known below the window)
        const float h1 = wave_shr1(NINF, sc[C - 1]), h2 = wave_shr1(NINF, sc[C - 2]), h3 = wave_shr1(NINF, sc[C - 3]);
        const int g1 = wave_shr1_i(0, og[C - 1]), g2 = wave_shr1_i(0, og[C - 2]), g3 = wave_shr1_i(0, og[C - 3]);
        // cells C-1..0 in place: a cell reads the old values of itself and of the three cells below it
#pragma unroll
        for (int i = kPairs - 1; i >= 0; --i) {
            const int kb = 2 * i, kl = 2 * i + 1;   // blank and label cell of pair i
            const float b_0 = sc[kb], l_0 = sc[kl];
            const int ob_0 = og[kb], ol_0 = og[kl];
            // positions kb-1, kb-2, kb-3 of the lane (negative: the lane below, its last cells)
            const float p1 = i > 0 ? sc[i > 0 ? kb - 1 : 0] : h1, p2 = i > 0 ? sc[i > 0 ? kb - 2 : 0] : h2;
            const float p3 = i > 1 ? sc[i > 1 ? kb - 3 : 0] : (i == 1 ? h1 : h3);
            const int q1 = i > 0 ? og[i > 0 ? kb - 1 : 0] : g1, q2 = i > 0 ? og[i > 0 ? kb - 2 : 0] : g2;
            const int q3 = i > 1 ? og[i > 1 ? kb - 3 : 0] : (i == 1 ? g1 : g3);
            float s;
            int o;
            cm_label<M, ZL>(l_0, b_0, p1, p2, ol_0, ob_0, q1, q2, el[i], veto[i], s, o);
            sc[kl] = select_f(NINF, s, mk.m[kl]);
            og[kl] = o;
            cm_blank<M>(b_0, p1, p3, ob_0, q1, q3, e0, s, o);
            sc[kb] = select_f(NINF, s, mk.m[kb]);
            og[kb] = o;
        }
    }
    // ---- the map: rise of every cell of this segment over the chunk, one byte per position ----
    {
        const uint32_t rel0 = seg_lo - w0;                                   // first delivered window position
        uint8_t *row = d.map0 + (size_t)c * (ck_pitch / 4);
        if constexpr (C == 8) {
            const uint32_t r = (uint32_t)C * (uint32_t)lane;
            const bool mine = r >= rel0 && r < rel0 + kOut && p_lane < hi_e;   // (hi_e need not be a multiple of 8: the last group is cut by the band)
            uint32_t w[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < C; ++k) {
                int rise = (int)r + k - og[k];
                rise = rise < 0 ? 0 : (rise > 255 ? 255 : rise);
                w[k >> 2] |= (uint32_t)rise << (8 * (k & 3));
            }
            if (mine) *reinterpret_cast<uint2 *>(row + (p_lane & ck_mask)) = make_uint2(w[0], w[1]);
        } else {
            // (a lane's cells straddle the segment's edges and, rarely, the ring's wrap: two bytes per pair, each pair by itself;
            //  positions up to the next multiple of 8 behind hi_e are written like the 8-cell form writes them: nobody reads them)
            const uint32_t hi8 = (hi_e + 7u) & ~7u;
#pragma unroll
            for (int i = 0; i < kPairs; ++i) {
                const uint32_t r = (uint32_t)C * (uint32_t)lane + 2u * (uint32_t)i, pos = p_lane + 2u * (uint32_t)i;
                int r0 = (int)r - og[2 * i], r1 = (int)r + 1 - og[2 * i + 1];
                r0 = r0 < 0 ? 0 : (r0 > 255 ? 255 : r0);
                r1 = r1 < 0 ? 0 : (r1 > 255 ? 255 : r1);
                if (r >= rel0 && r < rel0 + kOut && pos < hi8) *reinterpret_cast<uint16_t *>(row + (pos & ck_mask)) = (uint16_t)((uint32_t)r0 | ((uint32_t)r1 << 8));
            }
        }
    }
}

// grid: x = chunk (numbered over all chunk-parallel lattices of the launch), y = segment of the band
// (ONE kernel for transcripts with and without label 0, the two instances of the task behind a wave-uniform branch: as two
//  kernels that each skip the other's lattices, the one with nothing to do still cost 0.13 ms on the corpus launch - 430 000
//  workgroups that look their lattice up and leave - and both instances use the same 104 / 49 registers)
template <int M, int C>
__global__ __launch_bounds__(64) void chunk_map_kernel(const Lattice *__restrict__ lats, const int32_t *meta, int n_lats)
{
    const Lattice &d = lats[__builtin_amdgcn_readfirstlane(lattice_of_chunk(lats, n_lats, (int64_t)blockIdx.x))];
    const int lane = threadIdx.x;
    const int32_t *mt = meta + 4 * (size_t)__builtin_amdgcn_readfirstlane(d.idx);
    const int flags = __builtin_amdgcn_readfirstlane(mt[2]);
    if (flags & (kFlagExact | kFlagDeclined)) return;
    if (__builtin_amdgcn_readfirstlane(mt[0]) != kStatusOk || __builtin_amdgcn_readfirstlane(mt[1]) < 0) return;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((int64_t)blockIdx.x - d.chunk0));
    if (c == 0) return;                                   // nothing lies before chunk 0
    __shared__ float s_rows[kCkFrames][64];
    if (flags & kFlagZeroLabel)
        chunk_map_task<M, true, C>(d, c, (uint32_t)blockIdx.y, lane, s_rows);
    else
        chunk_map_task<M, false, C>(d, c, (uint32_t)blockIdx.y, lane, s_rows);
}

// 32 chunk maps -> one super-chunk map.  grid: x = blocks of 256 band positions, y = super-chunk, z = lattice
__global__ __launch_bounds__(256) void compose_maps_kernel(const Lattice *__restrict__ lats, const int32_t *meta)
{
    const Lattice &d = lats[blockIdx.z];
    if (!d.par) return;
    const int32_t *mt = meta + 4 * (size_t)d.idx;
    if ((mt[2] & (kFlagExact | kFlagDeclined)) || mt[0] != kStatusOk || mt[1] < 0) return;
    const int T = d.T, nck = chunks_of(T), nsup = supers_of(nck);
    const int s = blockIdx.y;
    if (s == 0 || s >= nsup) return;
    const int c_hi = min(s * kSuperChunks + kSuperChunks, nck) - 1, c_lo = s * kSuperChunks;
    uint32_t lo, hi;
    band_of((uint32_t)d.L, (uint32_t)T, (uint32_t)d.beam, (uint32_t)chunk_last_frame(c_hi, T), lo, hi);
    const size_t R = (size_t)(uint32_t)d.ck_pitch / 4;
    const uint32_t mask = d.ck_mask;
    for (uint32_t p = lo + blockIdx.x * 256 + threadIdx.x; p < hi; p += gridDim.x * 256) {
        int q = (int)p;
        for (int c = c_hi; c >= c_lo; --c) {
            q -= d.map0[(size_t)c * R + ((uint32_t)q & mask)];
            q = q < 0 ? 0 : q;
        }
        d.map1[(size_t)s * R + (p & mask)] = (uint16_t)((int)p - q);
    }
}

// one workgroup per lattice: the end position down the super-chunk maps, then every super-chunk down its chunk maps
__global__ __launch_bounds__(256) void chain_entries_kernel(const Lattice *__restrict__ lats, const int32_t *meta)
{
    const Lattice &d = lats[blockIdx.x];
    if (!d.par) return;
    const int32_t *mt = meta + 4 * (size_t)d.idx;
    if ((mt[2] & (kFlagExact | kFlagDeclined)) || mt[0] != kStatusOk || mt[1] < 0) return;
    const int T = d.T, nck = chunks_of(T), nsup = supers_of(nck);
    const size_t R = (size_t)(uint32_t)d.ck_pitch / 4;
    const uint32_t mask = d.ck_mask;
    int32_t *entry = d.entry, *entry1 = d.entry + nck;
    if (threadIdx.x == 0) {
        int p = mt[1];
        for (int s = nsup - 1; s >= 0; --s) {
            entry1[s] = p;
            if (s > 0) {
                p -= d.map1[(size_t)s * R + ((uint32_t)p & mask)];
                p = p < 0 ? 0 : p;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int s = threadIdx.x; s < nsup; s += blockDim.x) {
        int p = entry1[s];
        const int c_hi = min(s * kSuperChunks + kSuperChunks, nck) - 1, c_lo = s * kSuperChunks;
        for (int c = c_hi; c >= c_lo; --c) {
            entry[c] = p;
            if (c > 0) {
                p -= d.map0[(size_t)c * R + ((uint32_t)p & mask)];
                p = p < 0 ? 0 : p;
            }
        }
    }
}

}  // namespace ka
