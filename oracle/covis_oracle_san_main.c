/*
 * Sanitizer driver of the C covisitation oracle -- TEST INFRASTRUCTURE (CPU-side sanitizers).
 * Built by `make -C oracle asan` with -fsanitize=address,undefined (never on the GPU box: the sanitizers run on the
 * CPU build only). Reads one binary case file written by tests/test_covis_oracle.py, runs covis_topk_c on it with 1
 * and with 4 threads and writes the outputs back; any ASan / UBSan report makes the process exit non-zero.
 *
 * file: int64 n_sess, E, n_aids, window, max_gap, t0, t1, nk, k | int32 group[nk] | int32 param[3 nk]
 *       | int64 sess_off[n_sess + 1] | uint32 aid[E] | int32 ts[E] | uint8 type[E]
 * out : int64 P | uint32 oy[nk n_aids k] | uint64 ow[nk n_aids k] | int32 on[nk n_aids]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int64_t covis_topk_c(const uint32_t* aid, const int32_t* ts, const uint8_t* type, const int64_t* sess_off, int64_t n_sess,
                     uint32_t n_aids, int window, int max_gap, int64_t t0, int64_t t1, int nk, const int32_t* group,
                     const int32_t* param, int k, uint32_t* oy, uint64_t* ow, int32_t* on, int threads);

static void* xread(FILE* f, size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p || (n && fread(p, 1, n, f) != n)) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t h[9];
    if (fread(h, 8, 9, f) != 9) return 3;
    const int64_t n_sess = h[0], E = h[1], n_aids = h[2];
    const int nk = (int)h[7], k = (int)h[8];
    int32_t* group = xread(f, (size_t)nk * 4);
    int32_t* param = xread(f, (size_t)nk * 12);
    int64_t* off = xread(f, (size_t)(n_sess + 1) * 8);
    uint32_t* aid = xread(f, (size_t)E * 4);
    int32_t* ts = xread(f, (size_t)E * 4);
    uint8_t* type = xread(f, (size_t)E);
    fclose(f);
    const size_t cells = (size_t)nk * (size_t)n_aids * (size_t)k;
    uint32_t* oy[2];
    uint64_t* ow[2];
    int32_t* on[2];
    int64_t P[2];
    for (int r = 0; r < 2; ++r) {
        oy[r] = calloc(cells ? cells : 1, 4);
        ow[r] = calloc(cells ? cells : 1, 8);
        on[r] = calloc((size_t)nk * n_aids ? (size_t)nk * n_aids : 1, 4);
        P[r] = covis_topk_c(aid, ts, type, off, n_sess, (uint32_t)n_aids, (int)h[3], (int)h[4], h[5], h[6], nk, group, param, k,
                            oy[r], ow[r], on[r], r == 0 ? 1 : 4);
        if (P[r] < 0) { fprintf(stderr, "covis_topk_c failed\n"); return 4; }
    }
    if (P[0] != P[1] || memcmp(oy[0], oy[1], cells * 4) || memcmp(ow[0], ow[1], cells * 8) || memcmp(on[0], on[1], (size_t)nk * n_aids * 4)) {
        fprintf(stderr, "1-thread and 4-thread results differ\n");
        return 5;
    }
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(&P[0], 8, 1, f);
    fwrite(oy[0], 4, cells, f);
    fwrite(ow[0], 8, cells, f);
    fwrite(on[0], 4, (size_t)nk * n_aids, f);
    fclose(f);
    for (int r = 0; r < 2; ++r) { free(oy[r]); free(ow[r]); free(on[r]); }
    free(group); free(param); free(off); free(aid); free(ts); free(type);
    return 0;
}
