/*
 * oracle/ctc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the reference's CTC best-path
 * dynamic program and backtrace, kokoro_align/align.py:43-109 (`ctc_best_path`)
 * with its helper `flush_determined_path` (align.py:21-40).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product path (kokoro-align_amd/) never does.
 *
 * Parity pin: proven bit-identical to the imported reference on the golden
 * vectors under tests/golden/ (made by tests/golden/make_golden.py, which
 * imports /root/reference in the dev container) — see tests/test_oracle_golden.py.
 *
 * The reference keeps a compacted list of live states per frame and a
 * back-index into the previous list; this restatement is the dense-band form
 * of the same recurrence:
 *
 *   lab'[p] = p odd ? labels[p/2] : 0                       (align.py:46-50)
 *   A_{-1} = {0}, sc_{-1}[0] = 0f                           (align.py:57-58)
 *   for t in 0..T-1:
 *     lo = max(0, floor(L*t/T) - floor(B/2)); hi = min(lo+B, L)   (align.py:64-65)
 *     for p in [lo,hi):  e = lp[t, lab'[p]]
 *       for j in 0..M-1: u = p-j
 *         present_j = u in A_{t-1}                          (align.py:71-76)
 *         c_j = present_j ? fl32(sc_{t-1}[u] + e) : -inf    (align.py:77)
 *         if j>0 and j even and lab'[p]==0: c_j = -inf      (align.py:80-81)
 *       j* = first j attaining max_j c_j                    (align.py:83, np.argmax)
 *       p in A_t  <=>  present_{j*}                         (align.py:84,87)
 *       sc_t[p] = c_{j*}; bp_t[p] = j*                      (align.py:85,89-90)
 *   end = max A_{T-1}; empty -> error                       (align.py:99-101)
 *   for t = T-1..0: path[t] = p; p -= bp_t[p]               (align.py:21-40,102)
 *   best_labels = lab'[path]; best_scores[t] = lp[t, best_labels[t]]  (align.py:105-107)
 *
 * The periodic flush (align.py:95-96) only bounds the reference's memory; its
 * output equals one full backtrace from the terminal node, which is what is
 * done here.
 *
 * Preconditions shared with the product: log-probs contain no NaN and no +inf
 * (np.argmax would treat NaN as the maximum; not reproduced). -inf is handled
 * exactly (a state can be live with score -inf).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KAO_OK 0
#define KAO_EMPTY_BEAM (-1)
#define KAO_BAD_ARGS (-2)
#define KAO_NOMEM (-3)

/* band window of frame t, align.py:64-65 (Python big-int floor division). */
static void band(int64_t L, int64_t T, int64_t B, int64_t t, int64_t *lo, int64_t *hi)
{
    int64_t l = (int64_t)(((__int128)L * t) / T) - B / 2;
    if (l < 0) l = 0;
    int64_t h = l + B;
    if (h > L) h = L;
    *lo = l;
    *hi = h;
}

/*
 * Returns 0, or KAO_EMPTY_BEAM when frame T-1 has no live state (the reference
 * raises ValueError from np.argmax of an empty array, align.py:101).
 * `total_score` receives sc_{T-1}[end] (the cumulative float32 score of the
 * chosen terminal state; not returned by the reference, exposed for tests).
 */
int kao_ctc_best_path_f32(const float *lp, int64_t T, int32_t V, int64_t ld,
                          const int32_t *labels, int64_t S,
                          int32_t beam_size, int32_t max_move,
                          int32_t *path, int32_t *labels_out, float *scores_out,
                          float *total_score, int64_t *end_pos)
{
    if (T < 0 || S < 0 || V <= 0 || ld < V || beam_size < 0 || max_move < 1 || max_move > 255)
        return KAO_BAD_ARGS;
    for (int64_t s = 0; s < S; ++s)
        if (labels[s] < 0 || labels[s] >= V) return KAO_BAD_ARGS;
    const int64_t L = 2 * S + 1;
    const int64_t B = beam_size;
    const int M = max_move;
    if (T == 0) {
        /* reference: beams is empty -> beams[-1] raises IndexError; surfaced as bad args */
        return KAO_BAD_ARGS;
    }
    int64_t W = B < L ? B : L; /* widest band */
    if (W < 1) W = 1;

    float *sc_a = (float *)malloc(sizeof(float) * (size_t)(L + 1));
    float *sc_b = (float *)malloc(sizeof(float) * (size_t)(L + 1));
    uint8_t *pr_a = (uint8_t *)calloc((size_t)(L + 1), 1);
    uint8_t *pr_b = (uint8_t *)calloc((size_t)(L + 1), 1);
    uint8_t *bp = (uint8_t *)malloc((size_t)T * (size_t)W);
    int64_t *los = (int64_t *)malloc(sizeof(int64_t) * (size_t)T);
    if (!sc_a || !sc_b || !pr_a || !pr_b || !bp || !los) {
        free(sc_a); free(sc_b); free(pr_a); free(pr_b); free(bp); free(los);
        return KAO_NOMEM;
    }
    float *psc = sc_a, *csc = sc_b;
    uint8_t *ppr = pr_a, *cpr = pr_b;
    /* virtual state before frame 0, align.py:57-58 */
    int64_t plo = 0, phi = 1;
    psc[0] = 0.0f;
    ppr[0] = 1;

    for (int64_t t = 0; t < T; ++t) {
        int64_t lo, hi;
        band(L, T, B, t, &lo, &hi);
        los[t] = lo;
        const float *row = lp + (size_t)t * (size_t)ld;
        uint8_t *bprow = bp + (size_t)t * (size_t)W;
        for (int64_t p = lo; p < hi; ++p) {
            const int32_t lab = (p & 1) ? labels[p >> 1] : 0;
            const float e = row[lab];
            float best = -INFINITY;
            int bj = 0;
            for (int j = 0; j < M; ++j) {
                const int64_t u = p - j;
                if (u < 0) break;
                const int pres = (u >= plo && u < phi) ? ppr[u] : 0;
                float c = -INFINITY;
                if (pres) {
                    volatile float s = psc[u] + e; /* float32 add, then compare */
                    c = s;
                }
                if (j > 0 && (j % 2) == 0 && lab == 0) c = -INFINITY;
                if (j == 0 || c > best) { best = c; bj = j; } /* first max wins */
            }
            const int64_t ub = p - bj;
            cpr[p] = (ub >= plo && ub < phi) ? ppr[ub] : 0;
            csc[p] = best;
            bprow[p - lo] = (uint8_t)bj;
        }
        { float *tf = psc; psc = csc; csc = tf; }
        { uint8_t *tu = ppr; ppr = cpr; cpr = tu; }
        plo = lo;
        phi = hi;
    }

    /* terminal: highest live position of the last frame, align.py:99-101 */
    int64_t end = -1;
    for (int64_t p = phi - 1; p >= plo; --p)
        if (ppr[p]) { end = p; break; }
    int rc = KAO_OK;
    if (end < 0) {
        rc = KAO_EMPTY_BEAM;
    } else {
        if (total_score) *total_score = psc[end];
        if (end_pos) *end_pos = end;
        int64_t p = end;
        for (int64_t t = T - 1; t >= 0; --t) {
            path[t] = (int32_t)p;
            const int32_t lab = (p & 1) ? labels[p >> 1] : 0;
            labels_out[t] = lab;
            scores_out[t] = lp[(size_t)t * (size_t)ld + lab];
            p -= bp[(size_t)t * (size_t)W + (size_t)(p - los[t])];
        }
    }
    free(sc_a); free(sc_b); free(pr_a); free(pr_b); free(bp); free(los);
    return rc;
}

/* band cell count sum_t (hi_t - lo_t): used by bench.py for bytes-per-frame accounting. */
int64_t kao_band_cells(int64_t T, int64_t S, int32_t beam_size)
{
    const int64_t L = 2 * S + 1;
    int64_t n = 0;
    for (int64_t t = 0; t < T; ++t) {
        int64_t lo, hi;
        band(L, T, beam_size, t, &lo, &hi);
        n += hi - lo;
    }
    return n;
}

/*
 * Bit-reproducible synthetic inputs (SURVEY.md §8d "hash generator"):
 *   lp[t,c]   = -8 * u24(mix(seed, t*V + c))      (24-bit uniform, exact in float32)
 *   labels[k] = 1 + mix(seed ^ LABEL_SALT, k) % (V-1)
 * mix = splitmix64 finaliser of (seed * GOLDEN + idx + 1) * GOLDEN.
 */
static inline uint64_t kao_mix(uint64_t seed, uint64_t idx)
{
    uint64_t z = (seed * 0x9E3779B97F4A7C15ull + idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void kao_hash_logprobs_f32(float *lp, int64_t T, int32_t V, int64_t ld, uint64_t seed)
{
    for (int64_t t = 0; t < T; ++t)
        for (int32_t c = 0; c < V; ++c) {
            const uint64_t h = kao_mix(seed, (uint64_t)t * (uint64_t)V + (uint64_t)c);
            lp[(size_t)t * (size_t)ld + c] = -8.0f * ((float)(h >> 40) * (1.0f / 16777216.0f));
        }
}

void kao_hash_labels_i32(int32_t *labels, int64_t S, int32_t V, uint64_t seed)
{
    for (int64_t k = 0; k < S; ++k)
        labels[k] = (int32_t)(1 + kao_mix(seed ^ 0x4C4142454C53ull, (uint64_t)k) % (uint64_t)(V - 1));
}
