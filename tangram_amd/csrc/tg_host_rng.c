/* Host helper: the initial logits of a mapper, bit for bit what the reference draws.
 *
 * `Mapper.__init__` (tangram/mapping_optimizer.py:147-157; MapperConstrained :473-490) seeds NumPy's global generator and
 * calls `np.random.normal(0, 1, (n_cells, n_spots))`: MT19937 + the legacy polar Box-Muller method, one value at a time --
 * 2.6 s for the 2.6e8 logits of a 26 431 x 9 852 problem, longer than its 1 000 training iterations on the GPU (2.0 s).
 * This file reproduces that stream exactly (same 32-bit words, same 53-bit doubles, same rejection loop, same libm log / sqrt,
 * same caching of the second value of a pair, same final generator state) in chunks: the MT19937 words of a chunk are produced
 * sequentially by vectorisable whole-state twists, the candidates are accepted / transformed by all host threads.
 *
 * Not part of the device ABI (include/tangram_hip.h); bound by tangram_amd/host_rng.py, which falls back to NumPy itself -- the
 * same bits, slower -- when this library is not built.   gcc -O3 -fopenmp -shared -fPIC (no -ffast-math, no -march: no FMA). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MT_N 624
#define MT_M 397
#define UPPER 0x80000000U
#define LOWER 0x7fffffffU
#define MAGIC 0x9908b0dfU

/* one twist of the whole state (numpy random/src/mt19937/mt19937.c: mt19937_gen); three loops without loop-carried reads */
static void mt_twist(uint32_t* mt) {
    int k;
    for (k = 0; k < MT_N - MT_M; k++) {
        const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);
        mt[k] = mt[k + MT_M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MAGIC);
    }
    for (; k < MT_N - 1; k++) {
        const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);
        mt[k] = mt[k + (MT_M - MT_N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MAGIC);
    }
    {
        const uint32_t y = (mt[MT_N - 1] & UPPER) | (mt[0] & LOWER);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MAGIC);
    }
}
static inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}
/* the next n tempered words of the stream into w[]; key / pos like NumPy's state (pos == 624: a twist is due) */
static void mt_words(uint32_t* key, int* pos, uint32_t* w, int64_t n) {
    int64_t i = 0;
    int p = *pos;
    while (i < n) {
        if (p >= MT_N) { mt_twist(key); p = 0; }
        int64_t take = MT_N - p;
        if (take > n - i) take = n - i;
        for (int64_t j = 0; j < take; j++) w[i + j] = mt_temper(key[p + j]);
        i += take; p += (int)take;
    }
    *pos = p;
}
/* ... or just advance the state by n words */
static void mt_skip(uint32_t* key, int* pos, int64_t n) {
    int p = *pos;
    while (n > 0) {
        if (p >= MT_N) { mt_twist(key); p = 0; }
        int64_t take = MT_N - p;
        if (take > n) take = n;
        n -= take; p += (int)take;
    }
    *pos = p;
}
/* candidate c of a chunk: two doubles in [0, 1) from four words (random_double: (a >> 5, b >> 6)), mapped to (-1, 1) */
static inline void cand_xy(const uint32_t* w, int64_t c, double* x1, double* x2, double* r2) {
    const uint32_t a1 = w[4 * c] >> 5, b1 = w[4 * c + 1] >> 6, a2 = w[4 * c + 2] >> 5, b2 = w[4 * c + 3] >> 6;
    *x1 = 2.0 * ((a1 * 67108864.0 + b1) / 9007199254740992.0) - 1.0;
    *x2 = 2.0 * ((a2 * 67108864.0 + b2) / 9007199254740992.0) - 1.0;
    *r2 = (*x1) * (*x1) + (*x2) * (*x2);
}

#ifndef CHUNK_CAND
#define CHUNK_CAND (1 << 18)                 /* candidates per chunk: 1 M words = 4 MB per thread */
#define ROUND_CHUNKS 4096                    /* chunks per round: 2.5 KB of snapshot each (tests build tiny values of both) */
#endif
#define MAX_THREADS 256

typedef struct { uint32_t key[MT_N]; int pos; } mt_snap;

/* out[0 .. n): n draws of legacy normal(0, 1) as float32 (what `.astype(np.float32)` gives); the generator state
 * (key[624], *pos, *has_gauss, *gauss) is read and left exactly as NumPy would leave it.  out == NULL: draws discarded.
 * returns 0, or -1 when out of memory.
 * Rounds of up to a few thousand chunks: (0) one thread walks the generator over the round and keeps its state at every chunk
 * start (twists only: 0.3 ns per word); (A) all threads, chunk by chunk: the chunk's words from its snapshot, accepted candidates
 * counted; prefix sum; (B) the words again, the accepted candidates transformed and written where the sequential loop would have
 * put them.  The generator is finally set to the snapshot of the last chunk used plus the words the reference would have consumed. */
int tg_legacy_normal_f32(uint32_t* key, int* pos, int* has_gauss, double* gauss, float* out, int64_t n, int n_threads) {
    int64_t i = 0;
    if (n <= 0) return 0;
    if (*has_gauss) {                                            /* the cached second value of an earlier pair comes first */
        if (out) out[0] = (float)(*gauss);
        *has_gauss = 0; *gauss = 0.0;
        i = 1;
    }
    const int64_t rest = n - i, pairs = (rest + 1) / 2;          /* accepted candidates still needed; an odd rest caches one value */
    if (pairs == 0) return 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > MAX_THREADS) n_threads = MAX_THREADS;
    const int64_t max_chunks = ROUND_CHUNKS;
    mt_snap* snap = (mt_snap*)malloc((size_t)max_chunks * sizeof(mt_snap));
    int64_t* cnt = (int64_t*)malloc((size_t)(max_chunks + 1) * sizeof(int64_t));
    {                                                            /* no more threads (and word buffers) than chunks */
        const int64_t all_chunks = ((int64_t)((double)pairs * 1.2733 * 1.002) + 8192 + CHUNK_CAND - 1) / CHUNK_CAND;
        if (all_chunks < n_threads) n_threads = (int)all_chunks;
    }
    uint32_t* wbuf = (uint32_t*)malloc((size_t)n_threads * CHUNK_CAND * 4 * sizeof(uint32_t));
    if (!snap || !cnt || !wbuf) { free(snap); free(cnt); free(wbuf); return -1; }
    int64_t done = 0;                                            /* accepted so far */
    while (done < pairs) {
        int64_t cand_total = (int64_t)((double)(pairs - done) * 1.2733 * 1.002) + 8192;      /* acceptance rate pi / 4 */
        int64_t nch = (cand_total + CHUNK_CAND - 1) / CHUNK_CAND, last_cand = cand_total - (nch - 1) * (int64_t)CHUNK_CAND;
        if (nch > max_chunks) { nch = max_chunks; last_cand = CHUNK_CAND; }       /* (a full round; the rest follows in the next one) */
        for (int64_t c = 0; c < nch; c++) {                      /* (0) */
            memcpy(snap[c].key, key, sizeof(snap[c].key));
            snap[c].pos = *pos;
            mt_skip(key, pos, 4 * (c == nch - 1 ? last_cand : (int64_t)CHUNK_CAND));
        }
        int T = n_threads;
        if (cand_total < 65536 && CHUNK_CAND >= 65536) T = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
#endif
        for (int64_t c = 0; c < nch; c++) {                      /* (A) */
#ifdef _OPENMP
            uint32_t* w = wbuf + (size_t)omp_get_thread_num() * CHUNK_CAND * 4;
#else
            uint32_t* w = wbuf;
#endif
            const int64_t nc = c == nch - 1 ? last_cand : (int64_t)CHUNK_CAND;
            mt_snap st = snap[c];
            mt_words(st.key, &st.pos, w, 4 * nc);
            int64_t acc = 0;
            for (int64_t k = 0; k < nc; k++) {
                double x1, x2, r2;
                cand_xy(w, k, &x1, &x2, &r2);
                acc += !(r2 >= 1.0 || r2 == 0.0);
            }
            cnt[c + 1] = acc;
        }
        cnt[0] = 0;
        for (int64_t c = 0; c < nch; c++) cnt[c + 1] += cnt[c];
        int64_t end_chunk = -1, end_cand = -1;                   /* where the reference's loop stops, if inside this round */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
#endif
        for (int64_t c = 0; c < nch; c++) {                      /* (B) */
            int64_t k = done + cnt[c];                           /* global index of the chunk's first accepted candidate */
            if (k >= pairs) continue;
#ifdef _OPENMP
            uint32_t* w = wbuf + (size_t)omp_get_thread_num() * CHUNK_CAND * 4;
#else
            uint32_t* w = wbuf;
#endif
            const int64_t nc = c == nch - 1 ? last_cand : (int64_t)CHUNK_CAND;
            mt_snap st = snap[c];
            mt_words(st.key, &st.pos, w, 4 * nc);
            for (int64_t q = 0; q < nc && k < pairs; q++) {
                double x1, x2, r2;
                cand_xy(w, q, &x1, &x2, &r2);
                if (r2 >= 1.0 || r2 == 0.0) continue;
                const double f = sqrt(-2.0 * log(r2) / r2);
                const int64_t o = i + 2 * k;
                if (out) out[o] = (float)(f * x2);               /* legacy_gauss returns f * x2 first and keeps f * x1 */
                if (o + 1 < n) { if (out) out[o + 1] = (float)(f * x1); }
                else { *has_gauss = 1; *gauss = f * x1; }        /* (only the one chunk that owns the last pair gets here) */
                if (k == pairs - 1) { end_chunk = c; end_cand = q; }
                k++;
            }
        }
        if (done + cnt[nch] >= pairs) {                          /* finished inside this round: rewind to the words really consumed */
            memcpy(key, snap[end_chunk].key, sizeof(snap[end_chunk].key));
            *pos = snap[end_chunk].pos;
            mt_skip(key, pos, 4 * (end_cand + 1));
            done = pairs;
        } else done += cnt[nch];
    }
    free(snap); free(cnt); free(wbuf);
    return 0;
}
