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

#define CHUNK_CAND (1 << 22)                 /* candidates per chunk: 16 M words = 64 MB */
#define MAX_THREADS 256

/* out[0 .. n): n draws of legacy normal(0, 1) as float32 (what `.astype(np.float32)` gives); the generator state
 * (key[624], *pos, *has_gauss, *gauss) is read and left exactly as NumPy would leave it.  out == NULL: draws discarded.
 * returns 0, or -1 when out of memory. */
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
    uint32_t* w = (uint32_t*)malloc((size_t)CHUNK_CAND * 4 * sizeof(uint32_t));
    if (!w) return -1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > MAX_THREADS) n_threads = MAX_THREADS;
    int64_t done = 0;                                            /* accepted so far */
    while (done < pairs) {
        int64_t cand = (int64_t)((double)(pairs - done) * 1.2733 * 1.02) + 4096;     /* acceptance rate pi / 4 */
        if (cand > CHUNK_CAND) cand = CHUNK_CAND;
        uint32_t key0[MT_N];
        int pos0 = *pos;
        memcpy(key0, key, sizeof(key0));
        mt_words(key, pos, w, 4 * cand);
        int64_t count[MAX_THREADS + 1];
        int64_t last_c = -1;                                     /* the last candidate the reference would have consumed */
        int T = n_threads, nt_used = 1;
        if (cand < 65536) T = 1;
        count[0] = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
            const int tid = 0, nt = 1;
#endif
            const int64_t c0 = cand * tid / nt, c1 = cand * (tid + 1) / nt;
            int64_t acc = 0;
            for (int64_t c = c0; c < c1; c++) {
                double x1, x2, r2;
                cand_xy(w, c, &x1, &x2, &r2);
                acc += !(r2 >= 1.0 || r2 == 0.0);
            }
            count[tid + 1] = acc;
#ifdef _OPENMP
#pragma omp barrier
#pragma omp single
#endif
            { for (int t = 0; t < nt; t++) count[t + 1] += count[t]; nt_used = nt; }
            /* (implicit barrier after single) */
            int64_t k = done + count[tid];                       /* global index of this thread's first accepted candidate */
            for (int64_t c = c0; c < c1 && k < pairs; c++) {
                double x1, x2, r2;
                cand_xy(w, c, &x1, &x2, &r2);
                if (r2 >= 1.0 || r2 == 0.0) continue;
                const double f = sqrt(-2.0 * log(r2) / r2);
                const int64_t o = i + 2 * k;
                if (out) out[o] = (float)(f * x2);               /* legacy_gauss returns f * x2 first and keeps f * x1 */
                if (o + 1 < n) { if (out) out[o + 1] = (float)(f * x1); }
                else { *has_gauss = 1; *gauss = f * x1; }        /* (only the one thread that owns the last pair gets here) */
                if (k == pairs - 1) last_c = c;
                k++;
            }
        }
        const int64_t accepted = count[nt_used];
        if (done + accepted >= pairs) {                          /* finished inside this chunk: rewind to the words really consumed */
            memcpy(key, key0, sizeof(key0));
            *pos = pos0;
            mt_skip(key, pos, 4 * (last_c + 1));
            done = pairs;
        } else done += accepted;
    }
    free(w);
    return 0;
}
