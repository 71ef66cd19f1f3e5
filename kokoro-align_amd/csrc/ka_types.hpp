// ka_types.hpp — structs and constants shared by the host side (ka_engine.hip) and every device translation unit.
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

// ---- tiled forms (ka_tiled.hpp, ka_tiled2.hpp, ka_tiled_stream.hpp) ----
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
    int32_t below_end;  // t_end of the tile below (tile 0: INT32_MAX): the slots behind it hold -inf by construction (ka_tiled_stream.hpp uses it)
};
// per lattice, zeroed before every launch: terminal state by 64-bit atomicMax, arrival counter of the last-frame tiles
struct TileAux {
    unsigned long long best;   // (end position + 1) << 32 | score bits; 0 = no live state
    uint32_t arrived;
    uint32_t pad;
};

struct TpStats {
    unsigned long long phase[3];   // shader cycles: (wait | check+sum << 32), (progress | requests << 32), (publish+checkpoint)
    unsigned long long wait_ticks, total_ticks, spins, start_tick;   // 100 MHz ticks (ka_engine_set_verify(4): ka_debug_tile_stats)
    unsigned long long extra[2];   // 128-position tiles: (compute wavefront's cycles at the barrier | cycles in its frame blocks << 32), look-up wavefront's busy cycles
};

constexpr unsigned kTpLdsRequest = 40 * 1024;   // used: 32 KB rows + 2 KB packets + 2 KB publish staging
constexpr int kTp2BandBytes = 64 * 8 + 16;           // per block: KL, KE of 64 lanes + the event mask (worked out by the feeder, tp_band_block)
constexpr int kTp2StageBytes = 2 * kTpStageBytes;   // publish staging, double-buffered (the feeder reads block it-1's while block it's is written)
constexpr int kTnCells = 2;
constexpr int kTnTile = 64 * kTnCells;
constexpr int kTgPairBytes = kTpBlock * 64 * 8;   // a block of emission pairs of a 128-position tile (ka_tiled_stream.hpp)

constexpr int kCuSlots = 2048;      // entries of a per-CU table indexed by cu_slot() (ka_device.hpp): XCC_ID (3 bits) | HW_ID's se, sh, cu (8 bits)

// ---- chunk-parallel backtrace (ka_parallel_bt.hpp) ----
constexpr int kCmCells = 8;                      // cells per lane
constexpr int kCmSpan = 64 * kCmCells;           // positions a wavefront recomputes
constexpr int kCmWarm = 96;                      // lowest positions of the window: warm-up only (3 positions x 32 frames)
constexpr int kCmOut = 408;                      // positions a wavefront delivers (a multiple of 8, <= kCmSpan - kCmWarm - 7)
constexpr int kCmCellsWide = 18;                 // ... and the wide form (round 4): 1152 positions recomputed,
constexpr int kCmOutWide = 64 * kCmCellsWide - kCmWarm - 8;   // 1048 delivered: the reference's band of 1000 in ONE wavefront
// which form a launch's map kernel takes, from the widest band among its chunk-parallel lattices: positions a wavefront delivers
inline int cm_out_for(int64_t max_w) { return max_w + 7 <= kCmOut ? kCmOut : kCmOutWide; }
constexpr int kSuperChunks = 32;                 // chunks per super-chunk
static_assert(kCmWarm == 3 * kCkFrames && kCmOut % 8 == 0 && kCmOut + kCmWarm <= kCmSpan, "window geometry");

// ---- the log-prob producer (ka_lstm.hpp) ----
constexpr int kLstmH = 128;
constexpr int kLstmTile = 16;          // sequences per workgroup
constexpr int kLstmIn = 40;           // input features of layer 0 (MFCC coefficients)

}  // namespace ka
