// Generic device-wide exclusive scan of a functor (three launches), shared by the covisitation index and the event sort.
#pragma once
#include "common.h"

namespace otto {

// ---------------------------------------------------------------------------
// generic 3-phase exclusive scan of f(i), i in [0, n): out[i] = sum_{j<i} f(j), out[n] = total
// ---------------------------------------------------------------------------
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename F>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sum(F f, int64_t n, uint64_t* partial) {
    __shared__ uint64_t sm[SCAN_THREADS / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) v += f(base + k);
    uint64_t tot;
    (void)block_excl_scan<uint64_t, SCAN_THREADS>(v, sm, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(SCAN_THREADS) void k_scan_partials(uint64_t* partial, int nb) {
    __shared__ uint64_t sm[SCAN_THREADS / 64 + 1];
    uint64_t run = 0;
    for (int b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        int i = b0 + threadIdx.x;
        uint64_t v = i < nb ? partial[i] : 0;
        uint64_t tot;
        uint64_t ex = block_excl_scan<uint64_t, SCAN_THREADS>(v, sm, &tot);
        if (i < nb) partial[i] = run + ex;
        run += tot;
    }
    if (threadIdx.x == 0) partial[nb] = run;
}

template <typename F>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_write(F f, int64_t n, const uint64_t* partial, int nb,
                                                              uint64_t* out) {
    __shared__ uint64_t sm[SCAN_THREADS / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t x[SCAN_ITEMS];
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        x[k] = (base + k < n) ? f(base + k) : 0;
        v += x[k];
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan<uint64_t, SCAN_THREADS>(v, sm, &tot) + partial[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += x[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = partial[nb];
}

// out must hold n+1 entries, partial ceil(n/TILE)+1 entries.
template <typename F>
static int device_scan(F f, int64_t n, uint64_t* out, uint64_t* partial, hipStream_t s) {
    int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nb == 0) {
        OTTO_HIP(hipMemsetAsync(out, 0, sizeof(uint64_t), s));
        return 0;
    }
    k_scan_sum<F><<<nb, SCAN_THREADS, 0, s>>>(f, n, partial);
    k_scan_partials<<<1, SCAN_THREADS, 0, s>>>(partial, nb);
    k_scan_write<F><<<nb, SCAN_THREADS, 0, s>>>(f, n, partial, nb, out);
    OTTO_HIP(hipGetLastError());
    return 0;
}
static inline size_t scan_partial_bytes(int64_t n) { return ((size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1) * sizeof(uint64_t); }


}  // namespace otto
