/*
 * CPU ORACLE (plain C + OpenMP) for the covisitation builder -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: the reference ships no covisitation builder; this restates SPEC-COVIS
 *  a third time, independently of covis_oracle.py, and is checked
 * against it in tests/test_covis_oracle.py. Related reference code: the session self-join of
 * src/matrix_factorization/torch_trainer.py:198-223; type-weight vectors src/baseline/aid_weight.py:34,82 and
 * src/covisitation/inference.py:72; output contract src/covisitation/inference.py:19-35,87-111.
 *
 * Algorithm: (1) per session window, scalar O(n^2) expansion with a "first valid (i, j) per (aid_x, aid_y)"
 * table -> records (x, y, type_y, filter bits, time extra); (2) counting sort of the records by aid_x;
 * (3) per aid_x: sort by aid_y, sum runs, weight per kind, partial selection of the k best (W desc, aid_y asc).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint32_t x, y; uint32_t ch; uint32_t extra; } rec_t;   /* ch = type_y | fbits << 2 */

#define KIND_TIME 0
#define KIND_TYPE 1   /* param = 3 weights */
#define KIND_FILTER 2 /* param[0] = 9-bit mask, bit (type_x * 3 + type_y) */

static int64_t expand_window(const uint32_t* aid, const int32_t* ts, const uint8_t* type, int64_t lo, int64_t hi,
                             int window, int max_gap, int64_t t0, int64_t t1, const uint32_t* fmask, int n_filters,
                             rec_t* out /* NULL: count only */) {
    int n = (int)(hi - lo);
    if (n > window) { lo = hi - window; n = window; }
    if (n < 2) return 0;
    int cls[32];
    int first[32][32];       /* first valid (i << 5 | j) per (class_x, class_y), -1 = none */
    uint32_t fb[32][32];
    for (int i = 0; i < n; ++i) {
        cls[i] = i;
        for (int j = 0; j < i; ++j) if (aid[lo + j] == aid[lo + i]) { cls[i] = j; break; }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { first[i][j] = -1; fb[i][j] = 0; }
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            if (aid[lo + i] == aid[lo + j]) continue;
            int64_t dt = (int64_t)ts[lo + i] - (int64_t)ts[lo + j];
            if (dt < 0) dt = -dt;
            if (dt > max_gap) continue;
            const int ci = cls[i], cj = cls[j];
            if (first[ci][cj] < 0) first[ci][cj] = (i << 5) | j;
            const int bit = type[lo + i] * 3 + type[lo + j];
            for (int f = 0; f < n_filters; ++f) fb[ci][cj] |= ((fmask[f] >> bit) & 1u) << f;
        }
    }
    int64_t cnt = 0;
    for (int ci = 0; ci < n; ++ci) {
        if (cls[ci] != ci) continue;
        for (int cj = 0; cj < n; ++cj) {
            const int e = first[ci][cj];
            if (e < 0) continue;
            if (out) {
                const int i = e >> 5, j = e & 31;
                rec_t r;
                r.x = aid[lo + ci];
                r.y = aid[lo + cj];
                r.ch = (uint32_t)type[lo + j] | (fb[ci][cj] << 2);
                r.extra = t1 > t0 ? (uint32_t)((196608ll * ((int64_t)ts[lo + i] - t0)) / (t1 - t0)) : 0u;
                out[cnt] = r;
            }
            ++cnt;
        }
    }
    return cnt;
}

static int cmp_y(const void* a, const void* b) {
    const uint32_t ya = ((const rec_t*)a)->y, yb = ((const rec_t*)b)->y;
    return ya < yb ? -1 : (ya > yb);
}

typedef struct { uint64_t w; uint32_t y; } cand_t;
static int better(const cand_t* a, const cand_t* b) { return a->w > b->w || (a->w == b->w && a->y < b->y); }

/* kinds: kind_group[j] in {TIME, TYPE, FILTER}; kind_param[j*3 .. j*3+2].
 * outputs [n_kinds][n_aids][k]; returns P (deduped ordered pairs of the all-ones expansion), or -1. */
int64_t covis_topk_c(const uint32_t* aid, const int32_t* ts, const uint8_t* type, const int64_t* sess_off, int64_t n_sess,
                     uint32_t n_aids, int window, int max_gap, int64_t t0, int64_t t1, int n_kinds, const int32_t* kind_group,
                     const int32_t* kind_param, int k, uint32_t* out_y, uint64_t* out_w, int32_t* out_n, int threads) {
    if (window > 32 || k > 64) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    uint32_t fmask[8];
    int filt_of_kind[64], n_filters = 0;
    for (int j = 0; j < n_kinds; ++j) {
        filt_of_kind[j] = -1;
        if (kind_group[j] == KIND_FILTER) { filt_of_kind[j] = n_filters; fmask[n_filters++] = (uint32_t)kind_param[j * 3]; }
    }
    int64_t* rec_off = (int64_t*)malloc((size_t)(n_sess + 1) * sizeof(int64_t));
    if (!rec_off) return -1;
    rec_off[0] = 0;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t s = 0; s < n_sess; ++s)
        rec_off[s + 1] = expand_window(aid, ts, type, sess_off[s], sess_off[s + 1], window, max_gap, t0, t1, fmask, n_filters, NULL);
    for (int64_t s = 0; s < n_sess; ++s) rec_off[s + 1] += rec_off[s];
    const int64_t P = rec_off[n_sess];
    rec_t* recs = (rec_t*)malloc((size_t)(P ? P : 1) * sizeof(rec_t));
    rec_t* sorted = (rec_t*)malloc((size_t)(P ? P : 1) * sizeof(rec_t));
    int64_t* xoff = (int64_t*)calloc((size_t)n_aids + 1, sizeof(int64_t));
    if (!recs || !sorted || !xoff) { free(rec_off); free(recs); free(sorted); free(xoff); return -1; }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t s = 0; s < n_sess; ++s)
        expand_window(aid, ts, type, sess_off[s], sess_off[s + 1], window, max_gap, t0, t1, fmask, n_filters, recs + rec_off[s]);
    /* counting sort by aid_x */
    for (int64_t i = 0; i < P; ++i) xoff[recs[i].x + 1]++;
    for (uint32_t x = 0; x < n_aids; ++x) xoff[x + 1] += xoff[x];
    {
        int64_t* cur = (int64_t*)malloc((size_t)n_aids * sizeof(int64_t));
        if (!cur) { free(rec_off); free(recs); free(sorted); free(xoff); return -1; }
        memcpy(cur, xoff, (size_t)n_aids * sizeof(int64_t));
        for (int64_t i = 0; i < P; ++i) sorted[cur[recs[i].x]++] = recs[i];
        free(cur);
    }
    memset(out_n, 0, (size_t)n_kinds * n_aids * sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t x = 0; x < (int64_t)n_aids; ++x) {
        const int64_t b = xoff[x], e = xoff[x + 1];
        if (b == e) continue;
        qsort(sorted + b, (size_t)(e - b), sizeof(rec_t), cmp_y);
        cand_t top[16][64];
        int ntop[16];
        for (int j = 0; j < n_kinds; ++j) ntop[j] = 0;
        int64_t i = b;
        while (i < e) {
            const uint32_t y = sorted[i].y;
            uint64_t c[3] = {0, 0, 0}, fc[8] = {0}, cnt = 0, ext = 0;
            for (; i < e && sorted[i].y == y; ++i) {
                c[sorted[i].ch & 3u]++;
                for (int f = 0; f < n_filters; ++f) fc[f] += (sorted[i].ch >> (2 + f)) & 1u;
                cnt++;
                ext += sorted[i].extra;
            }
            for (int j = 0; j < n_kinds; ++j) {
                cand_t cd;
                cd.y = y;
                if (kind_group[j] == KIND_TIME) cd.w = 65536ull * cnt + ext;
                else if (kind_group[j] == KIND_TYPE)
                    cd.w = 65536ull * (c[0] * (uint64_t)kind_param[j * 3] + c[1] * (uint64_t)kind_param[j * 3 + 1] + c[2] * (uint64_t)kind_param[j * 3 + 2]);
                else cd.w = 65536ull * fc[filt_of_kind[j]];
                if (cd.w == 0) continue;
                /* sorted insertion into the k-list */
                int n = ntop[j];
                if (n == k && !better(&cd, &top[j][k - 1])) continue;
                int pos = n < k ? n : k - 1;
                while (pos > 0 && better(&cd, &top[j][pos - 1])) { top[j][pos] = top[j][pos - 1]; --pos; }
                top[j][pos] = cd;
                if (n < k) ntop[j] = n + 1;
            }
        }
        for (int j = 0; j < n_kinds; ++j) {
            out_n[(size_t)j * n_aids + x] = ntop[j];
            for (int q = 0; q < ntop[j]; ++q) {
                out_y[((size_t)j * n_aids + x) * k + q] = top[j][q].y;
                out_w[((size_t)j * n_aids + x) * k + q] = top[j][q].w;
            }
        }
    }
    free(rec_off); free(recs); free(sorted); free(xoff);
    return P;
}
