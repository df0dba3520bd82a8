// Aid-pair dataset builders of the collaborative-filtering trainer on the device (include/otto_pairs.h, SURVEY.md section
// 8 a6; reference: src/matrix_factorization/torch_trainer.py:190-255). Records (x1 << 32 | x2, label) are emitted per
// session, sorted by pair with the LSD radix sort of otto_events.hip and aggregated per run of equal pairs.
#include "common.h"
#include "scan.h"
#include "../../include/otto_pairs.h"
#include "../../include/otto_events.h"

int otto_sort_pairs_in_ws(uint64_t* d_keys, int64_t n, void* d_ws, uint64_t** d_keys_sorted, uint32_t** d_vals_sorted, hipStream_t s);
void otto_sort_ws_buffers(int64_t n, void* d_ws, uint64_t** key0, uint32_t** val0, uint64_t** scan_out, uint64_t** scan_partial);

namespace otto {

constexpr uint64_t PAIR_NONE = ~0ull;      // slot without a record: sorts behind every pair

struct SelfJoinPairs {                     // n (n - 1) slots of session i
    const int64_t* off;
    __device__ uint64_t operator()(int64_t i) const {
        const int64_t n = off[i + 1] - off[i];
        return (uint64_t)(n * (n - 1));
    }
};

// one thread per event i of session s: the row of its n - 1 partners j != i
__global__ void k_time_emit(const uint32_t* aid, const int32_t* ts, const int64_t* off, const uint64_t* pair_off, int64_t n_sess,
                            int64_t n_events, int64_t max_dt, uint64_t* key, uint32_t* val) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_events; e += (int64_t)gridDim.x * blockDim.x) {
        // session of event e: binary search in the CSR offsets
        int64_t lo = 0, hi = n_sess;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= e) lo = mid; else hi = mid;
        }
        const int64_t s0 = off[lo], n = off[lo + 1] - s0, i = e - s0;
        uint64_t o = pair_off[lo] + (uint64_t)(i * (n - 1));
        const uint32_t ax = aid[e];
        const int64_t tx = ts[e];
        for (int64_t j = 0; j < n; ++j) {
            if (j == i) continue;
            const uint32_t ay = aid[s0 + j];
            const int64_t dt = (int64_t)ts[s0 + j] - tx;
            key[o] = ax != ay ? ((uint64_t)ax << 32) | ay : PAIR_NONE;
            val[o] = (dt > 0 && dt <= max_dt) ? 1u : 0u;
            ++o;
        }
    }
}

// slots 2 e (positive) and 2 e + 1 (negative) of event e
__global__ void k_diff_emit(const uint32_t* aid, const uint32_t* shuf, const int64_t* off, int64_t n_sess, int64_t n_events,
                            uint64_t* key, uint32_t* val) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_events; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_sess;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= e) lo = mid; else hi = mid;
        }
        const bool has_next = e + 1 < off[lo + 1];                      // shift(-1): the last event of a session has no x2
        const uint32_t x1 = aid[e], x3 = shuf[e];
        const uint32_t x2 = has_next ? aid[e + 1] : 0u;
        const bool pos = has_next && x2 != x3 && x1 != x2 && x1 != x3;
        const bool neg = has_next && x2 != x3 && x1 != x3;
        key[2 * e] = pos ? ((uint64_t)x1 << 32) | x2 : PAIR_NONE;
        val[2 * e] = 1u;
        key[2 * e + 1] = neg ? ((uint64_t)x1 << 32) | x3 : PAIR_NONE;
        val[2 * e + 1] = 0u;
    }
}

struct PairHead {      // 1 where a new pair starts in the sorted records (the PAIR_NONE tail is one group, dropped later)
    const uint64_t* key;
    __device__ uint64_t operator()(int64_t i) const { return (key[i] != PAIR_NONE && (i == 0 || key[i] != key[i - 1])) ? 1ull : 0ull; }
};

__global__ void k_pairs_aggregate(const uint64_t* key, const uint32_t* val, int64_t n, const uint64_t* head_pos, int32_t agg,
                                  int64_t* x1, int64_t* x2, int64_t* target) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = key[i];
        if (k == PAIR_NONE || (i > 0 && key[i - 1] == k)) continue;
        uint64_t cnt = 0, sum = 0;
        for (int64_t j = i; j < n && key[j] == k; ++j) { ++cnt; sum += val[j]; }
        const uint64_t p = head_pos[i];
        x1[p] = (int64_t)(k >> 32);
        x2[p] = (int64_t)(k & 0xFFFFFFFFull);
        target[p] = agg == OTTO_PAIRS_AGG_MAX ? (sum > 0 ? 1 : 0) : (2 * sum >= cnt ? 1 : 0);
    }
}

}  // namespace otto

using namespace otto;

extern "C" int otto_pairs_raw_count(const int64_t* d_sess_off, int64_t n_sess, int32_t strategy_time, int64_t* h_raw, void* stream) {
    OTTO_REQUIRE(d_sess_off && h_raw && n_sess >= 0, "otto_pairs_raw_count: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (n_sess == 0) { *h_raw = 0; return 0; }
    if (!strategy_time) {
        int64_t e = 0;
        OTTO_HIP(hipMemcpyAsync(&e, d_sess_off + n_sess, 8, hipMemcpyDeviceToHost, s));
        OTTO_HIP(hipStreamSynchronize(s));
        *h_raw = 2 * e;
        return 0;
    }
    uint64_t *out = nullptr, *partial = nullptr;
    OTTO_TRY(device_scratch(SCRATCH_PAIRS_A, (size_t)(n_sess + 1) * 8, (void**)&out, s));
    OTTO_TRY(device_scratch(SCRATCH_PAIRS_B, scan_partial_bytes(n_sess), (void**)&partial, s));
    hipError_t e = hipSuccess;
    int rc = device_scan(SelfJoinPairs{d_sess_off}, n_sess, out, partial, s);
    uint64_t tot = 0;
    if (rc == 0) {
        e = hipMemcpyAsync(&tot, out + n_sess, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = -5;
    }
    OTTO_REQUIRE(rc == 0, "otto_pairs_raw_count failed on the device");
    *h_raw = (int64_t)tot;
    return 0;
}

extern "C" int64_t otto_pairs_workspace(int64_t raw) {
    // sort workspace of `raw` records + the per-session slot bases (their count is bounded by the records + 1)
    return otto_events_sort_workspace(raw > 0 ? raw : 1);
}

static int finish_pairs(int64_t raw, void* ws, int32_t agg, int64_t* x1, int64_t* x2, int64_t* target, int64_t* h_n_rows, hipStream_t s) {
    uint64_t *key0, *scan_out, *scan_partial, *ks;
    uint32_t *val0, *vs;
    otto_sort_ws_buffers(raw, ws, &key0, &val0, &scan_out, &scan_partial);
    OTTO_TRY(otto_sort_pairs_in_ws(key0, raw, ws, &ks, &vs, s));
    OTTO_TRY(device_scan(PairHead{ks}, raw, scan_out, scan_partial, s));
    const int grid = (int)((raw + 255) / 256 < 256 * 16 ? (raw + 255) / 256 : 256 * 16);
    k_pairs_aggregate<<<grid, 256, 0, s>>>(ks, vs, raw, scan_out, agg, x1, x2, target);
    OTTO_HIP(hipGetLastError());
    uint64_t rows = 0;
    OTTO_HIP(hipMemcpyAsync(&rows, scan_out + raw, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    *h_n_rows = (int64_t)rows;
    return 0;
}

extern "C" int otto_pairs_time(const uint32_t* d_aid, const int32_t* d_ts, const int64_t* d_sess_off, int64_t n_sess, int64_t raw,
                               int64_t max_dt_seconds, int32_t aggregation, int64_t* d_out_x1, int64_t* d_out_x2, int64_t* d_out_target,
                               int64_t* h_n_rows, void* d_ws, int64_t ws_bytes, void* stream) {
    OTTO_REQUIRE(h_n_rows, "null argument");
    OTTO_REQUIRE(aggregation == OTTO_PAIRS_AGG_MEAN || aggregation == OTTO_PAIRS_AGG_MAX, "Invalid target aggregation");
    *h_n_rows = 0;
    if (raw == 0 || n_sess == 0) return 0;
    OTTO_REQUIRE(d_aid && d_ts && d_sess_off && d_out_x1 && d_out_x2 && d_out_target && d_ws, "otto_pairs_time: null argument");
    OTTO_REQUIRE(raw > 0 && raw < (1ll << 32), "raw record count must be in (0, 2^32): build the dataset in session chunks");
    OTTO_REQUIRE(ws_bytes >= otto_pairs_workspace(raw), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    uint64_t *key0, *scan_out, *scan_partial;
    uint32_t* val0;
    otto_sort_ws_buffers(raw, d_ws, &key0, &val0, &scan_out, &scan_partial);
    // slot bases of the sessions: scan_out holds raw + 1 >= n_sess + 1 entries only if raw >= n_sess; otherwise a private buffer
    uint64_t* pair_off = nullptr;
    OTTO_TRY(device_scratch(SCRATCH_PAIRS_A, (size_t)(n_sess + 1) * 8, (void**)&pair_off, s));
    uint64_t* part = nullptr;
    OTTO_TRY(device_scratch(SCRATCH_PAIRS_B, scan_partial_bytes(n_sess), (void**)&part, s));
    int rc = device_scan(SelfJoinPairs{d_sess_off}, n_sess, pair_off, part, s);
    int64_t n_events = 0;
    if (rc == 0 && (hipMemcpyAsync(&n_events, d_sess_off + n_sess, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess)) rc = -5;
    if (rc == 0) {
        const int grid = (int)((n_events + 255) / 256 < 256 * 32 ? (n_events + 255) / 256 : 256 * 32);
        k_time_emit<<<grid > 0 ? grid : 1, 256, 0, s>>>(d_aid, d_ts, d_sess_off, pair_off, n_sess, n_events, max_dt_seconds, key0, val0);
        if (hipGetLastError() != hipSuccess) rc = -5;
    }
    if (rc == 0) rc = finish_pairs(raw, d_ws, aggregation, d_out_x1, d_out_x2, d_out_target, h_n_rows, s);
    (void)hipStreamSynchronize(s);
    OTTO_REQUIRE(rc == 0, "otto_pairs_time failed on the device (%d)", rc);
    return 0;
}

extern "C" int otto_pairs_diff(const uint32_t* d_aid, const uint32_t* d_shuffled_aid, const int64_t* d_sess_off, int64_t n_sess,
                               int64_t raw, int64_t* d_out_x1, int64_t* d_out_x2, int64_t* d_out_target, int64_t* h_n_rows, void* d_ws,
                               int64_t ws_bytes, void* stream) {
    OTTO_REQUIRE(h_n_rows, "null argument");
    *h_n_rows = 0;
    if (raw == 0 || n_sess == 0) return 0;
    OTTO_REQUIRE(d_aid && d_shuffled_aid && d_sess_off && d_out_x1 && d_out_x2 && d_out_target && d_ws, "otto_pairs_diff: null argument");
    OTTO_REQUIRE(raw > 0 && raw < (1ll << 32) && (raw & 1) == 0, "raw must be 2 * events, below 2^32");
    OTTO_REQUIRE(ws_bytes >= otto_pairs_workspace(raw), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    uint64_t *key0, *scan_out, *scan_partial;
    uint32_t* val0;
    otto_sort_ws_buffers(raw, d_ws, &key0, &val0, &scan_out, &scan_partial);
    const int64_t n_events = raw / 2;
    const int grid = (int)((n_events + 255) / 256 < 256 * 32 ? (n_events + 255) / 256 : 256 * 32);
    k_diff_emit<<<grid, 256, 0, s>>>(d_aid, d_shuffled_aid, d_sess_off, n_sess, n_events, key0, val0);
    OTTO_HIP(hipGetLastError());
    return finish_pairs(raw, d_ws, OTTO_PAIRS_AGG_MAX, d_out_x1, d_out_x2, d_out_target, h_n_rows, s);
}
