// ka_lstm.hpp — the log-prob producer's LSTM on the device (AudioToChar, kokoro_align/train.py:54-65).  Included by ka_misc.hip only.
#pragma once
#include "ka_types.hpp"

namespace ka {

// ---------------------------------------------------------------------------------------
// LSTM cell update of the log-prob producer (AudioToChar, kokoro_align/train.py:54-65), one time step of one
// layer, both directions: gates = gin[row] + rec, PyTorch gate order (i, f, g, o);
//   c = sigmoid(f)*c + sigmoid(i)*tanh(g);  h = sigmoid(o)*tanh(c)
// gin holds x_t @ W_ih^T + b_ih + b_hh of every frame (one library GEMM per layer), rec = h_{t-1} @ W_hh^T of
// the n sequences still running (one batched library GEMM per step); this kernel is the fused element-wise
// part and scatters h into the layer's output rows.  grid: x = ceil(n*H/256), y = direction (0 fwd, 1 bwd).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_step_kernel(const float *__restrict__ gin, int64_t ldg,
                                                        const float *__restrict__ rec, int64_t rec_dir_stride,
                                                        float *__restrict__ c, float *__restrict__ h, int64_t state_dir_stride,
                                                        float *__restrict__ out, int64_t ldo,
                                                        const int32_t *__restrict__ rows, int64_t rows_dir_stride, int n, int H)
{
    const int dir = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * H) return;
    const int s = (int)(idx / H), j = (int)(idx % H);
    const int64_t row = rows[(size_t)dir * rows_dir_stride + s];
    const float *g = gin + (size_t)row * ldg + (size_t)dir * 4 * H;
    const float *r = rec + (size_t)dir * rec_dir_stride + (size_t)s * 4 * H;
    const float gi = g[j] + r[j], gf = g[H + j] + r[H + j], gg = g[2 * H + j] + r[2 * H + j], go = g[3 * H + j] + r[3 * H + j];
    const float si = 1.0f / (1.0f + expf(-gi)), sf = 1.0f / (1.0f + expf(-gf)), so = 1.0f / (1.0f + expf(-go));
    float *cs = c + (size_t)dir * state_dir_stride + (size_t)s * H;
    float *hs = h + (size_t)dir * state_dir_stride + (size_t)s * H;
    const float cn = sf * cs[j] + si * tanhf(gg);
    const float hn = so * tanhf(cn);
    cs[j] = cn;
    hs[j] = hn;
    out[(size_t)row * ldo + (size_t)dir * H + j] = hn;
}

// ---------------------------------------------------------------------------------------
// One whole LSTM layer, both directions, persistent: a 256-thread workgroup owns 16 sequences of one direction
// for ALL their time steps.  The recurrent product h @ W_hh^T runs on the f32 MFMA (v_mfma_f32_16x16x4_f32,
// exact float32): wave w computes the four gates of hidden units [32w, 32w+32), and its 128 x 128 slice of
// W_hh^T (64 KB) stays in registers for the whole kernel - 256 of the 512 VGPR/AGPRs a wave has at one
// wave per SIMD - so no weight byte is read after start-up.  h lives in LDS (double-buffered, one barrier per
// step); the input projections of a step are loaded before its MFMA loop and added after it.
// Sequences are sorted by length (longest first): a tile runs for its first sequence's length.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float fast_tanh(float x)   // 1 - 2/(1 + e^{2x}): exact limits at +-inf
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681472f));
}
constexpr int kLstmLdh = kLstmH + 4;   // LDS row pitch (floats): 16-byte aligned rows, conflict-free b128 A-fragment reads
constexpr int kLstmWAcc = 30;          // k-steps whose 8 W fragments live in AGPRs (240 of 256); the last two sit in VGPRs
// D = A*B + D with B taken straight from an accumulation register: the compiler's own allocation of the builtin
// parked W in AGPRs and copied every fragment through one VGPR (v_accvgpr_read + s_nop + spill reloads) per MFMA.
#define KA_MFMA_ACC(ACC, A, W) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(W))
#define KA_MFMA_VGPR(ACC, A, W) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(W))
// grid.x = 2 * ceil(nseq / 16): workgroup id>>1 = tile (longest sequences first, so the dispatcher starts the
// long tiles first and back-fills the CUs with short ones), id&1 = direction
// XIN = false: `gin` holds the input projections x W_ih^T + b_ih + b_hh of every frame (one library GEMM per layer).
// XIN = true (layer 0, kLstmIn = 40 input features): `gin` IS x [frames, ldg >= 40]; the projection runs inside the step - 80
// more MFMAs (K = 40: 10 k-steps x 8 fragments, W_ih's fragments resident in VGPRs like W_hh's in AGPRs) on top of the 256 of
// h W_hh^T, the bias is the accumulator's initial value - so the [frames, 1024] projection (11 GB for an 8.8-hour book) is
// never written or read, and a step prefetches 10 dwords per lane instead of 32.
template <bool XIN>
__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(const float *__restrict__ gin, int64_t ldg,
                                                            const float *__restrict__ w_hh, float *__restrict__ out, int64_t ldo,
                                                            const int32_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len,
                                                            int nseq, const float *__restrict__ w_ih, const float *__restrict__ bias)
{
    // h of the tile's 16 sequences, double-buffered; within a row unit k sits at (k&3)*32 + (k>>2), so the 32
    // A operands of a lane (k = 4s + kq, s = 0..31) are contiguous: 8 ds_read_b128 per step
    __shared__ __attribute__((aligned(16))) float s_h[2][kLstmTile][kLstmLdh];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int dir = blockIdx.x & 1;
    const int tile0 = (blockIdx.x >> 1) * kLstmTile;
    const int col = lane & 15, kq = lane >> 4;
    const int jbase = 32 * wv;                       // hidden units of this wave: two column tiles of 16
    // W fragments: gate g, column tile ct, k-step s: B[k = 4s+kq][n = col] = W_hh[dir][g*H + jbase + 16ct + col][k]
    float wreg[4][2][32];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int s = 0; s < 32; ++s)
                wreg[g][ct][s] = w_hh[((size_t)dir * 4 * kLstmH + (size_t)g * kLstmH + jbase + 16 * ct + col) * kLstmH + 4 * s + kq];
    // XIN: W_ih's fragments B[k = 4s+kq][n = col] = W_ih[dir][g*H + jbase + 16ct + col][k], the bias of the lane's columns, and the
    // sequence whose x row is this lane's A operand (row `col` of the tile)
    float wih[4][2][kLstmIn / 4], bs[4][2];
    int lenA = 0, rowbaseA = 0;
    if constexpr (XIN) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const size_t wrow = (size_t)dir * 4 * kLstmH + (size_t)g * kLstmH + jbase + 16 * ct + col;
                bs[g][ct] = bias[wrow];
#pragma unroll
                for (int s = 0; s < kLstmIn / 4; ++s) wih[g][ct][s] = w_ih[wrow * kLstmIn + 4 * s + kq];
            }
        const int i = tile0 + col;
        const int l = i < nseq ? seq_len[i] : 0;
        const int o = i < nseq ? seq_off[i] : 0;
        lenA = l;
        rowbaseA = l <= 0 ? 0 : dir == 0 ? o : o + l - 1;
    }
    // the 4 sequences (rows 4kq .. 4kq+3 of the C tile) this lane updates
    int rowbase[4], len[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tile0 + 4 * kq + r;
        const int l = i < nseq ? seq_len[i] : 0;
        const int o = i < nseq ? seq_off[i] : 0;
        len[r] = l;
        rowbase[r] = l <= 0 ? 0 : dir == 0 ? o : o + l - 1;   // empty / padding rows prefetch row 0 (never used)
    }
    const int tile_len = tile0 < nseq ? seq_len[tile0] : 0;   // sorted by length, longest first
    float c[2][4], hreg[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[ct][r] = hreg[ct][r] = 0.0f;
    for (int i = tid; i < 2 * kLstmTile * kLstmLdh; i += 256) (&s_h[0][0][0])[i] = 0.0f;
    __syncthreads();
    const float *gcol = gin + (size_t)dir * 4 * kLstmH + jbase + col;
    float *ocol = out + (size_t)dir * kLstmH + jbase + col;
    int hpos[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) hpos[ct] = ((jbase + 16 * ct + col) & 3) * 32 + ((jbase + 16 * ct + col) >> 2);
    // input projections one step ahead: they are the MFMA's initial accumulator, so they must have landed
    // before the step starts; a finished sequence re-reads its last row (never used)
    auto gin_row = [&](int r, int t) {
        const int k = max(min(t, len[r] - 1), 0);
        return gcol + (size_t)(dir == 0 ? rowbase[r] + k : rowbase[r] - k) * (size_t)ldg;
    };
    auto x_row = [&](int t) {       // XIN: the row of sequence `col` at step t, columns kq, kq+4, ...
        const int k = max(min(t, lenA - 1), 0);
        return gin + (size_t)(dir == 0 ? rowbaseA + k : rowbaseA - k) * (size_t)ldg + kq;
    };
    f32x4 nxt[4][2];
    float xn[kLstmIn / 4];
    if constexpr (XIN) {
        const float *xp = x_row(0);
#pragma unroll
        for (int s = 0; s < kLstmIn / 4; ++s) xn[s] = xp[4 * s];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *gp = gin_row(r, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) nxt[g][ct][r] = gp[(size_t)g * kLstmH + 16 * ct];
        }
    }
    for (int t = 0; t < tile_len; ++t) {
        const int cur = t & 1;
        f32x4 acc[4][2];
        float xa[kLstmIn / 4];
        if constexpr (XIN) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[g][ct] = f32x4{bs[g][ct], bs[g][ct], bs[g][ct], bs[g][ct]};
            const float *xp = x_row(t + 1);
#pragma unroll
            for (int s = 0; s < kLstmIn / 4; ++s) {
                xa[s] = xn[s];
                xn[s] = xp[4 * s];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[g][ct] = nxt[g][ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *gp = gin_row(r, t + 1);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) nxt[g][ct][r] = gp[(size_t)g * kLstmH + 16 * ct];
            }
        }
        const f32x4 *arow = reinterpret_cast<const f32x4 *>(&s_h[cur][col][kq * 32]);
        asm volatile("s_nop 3" ::: "memory");   // VALU-written accumulators -> first MFMA
        if constexpr (XIN) {
#pragma unroll
            for (int s = 0; s < kLstmIn / 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) KA_MFMA_VGPR(acc[g][ct], xa[s], wih[g][ct][s]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 a4 = arow[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * q + u;
                const float a = a4[u];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (s < kLstmWAcc) KA_MFMA_ACC(acc[g][ct], a, wreg[g][ct][s]);
                        else KA_MFMA_VGPR(acc[g][ct], a, wreg[g][ct][s]);
                    }
            }
        }
        asm volatile("s_nop 10" ::: "memory");   // 8-pass MFMA result -> VALU read: 11 wait states
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool act = t < len[r];
                // hardware exp2 / rcp (about 1 ulp each): 4 instructions per sigmoid instead of a ~40-instruction libm call
                const float si = fast_sigmoid(acc[0][ct][r]), sf = fast_sigmoid(acc[1][ct][r]), so = fast_sigmoid(acc[3][ct][r]);
                const float cn = sf * c[ct][r] + si * fast_tanh(acc[2][ct][r]);
                const float hn = so * fast_tanh(cn);
                if (act) {
                    c[ct][r] = cn;
                    hreg[ct][r] = hn;
                    ocol[(size_t)(dir == 0 ? rowbase[r] + t : rowbase[r] - t) * (size_t)ldo + 16 * ct] = hn;
                }
                s_h[cur ^ 1][4 * kq + r][hpos[ct]] = hreg[ct][r];
            }
        __syncthreads();
    }
}
#undef KA_MFMA_ACC
#undef KA_MFMA_VGPR


}  // namespace ka
