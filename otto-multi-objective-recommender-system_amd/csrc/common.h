// Shared host/device helpers for libotto_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace otto {

void set_error(const char* fmt, ...);

#define OTTO_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ::otto::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -5;                                                                      \
        }                                                                                   \
    } while (0)

#define OTTO_TRY(expr)          \
    do {                        \
        int _r = (expr);        \
        if (_r != 0) return _r; \
    } while (0)

#define OTTO_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            ::otto::set_error(__VA_ARGS__); \
            return -22;                  \
        }                                \
    } while (0)

// Growable device buffer owned by a context.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // Grow to at least `bytes`. keep != 0 preserves the first `keep` bytes.
    int ensure(size_t bytes, size_t keep, hipStream_t s) {
        if (bytes <= cap) return 0;
        size_t ncap = bytes;
        if (keep) ncap = bytes + bytes / 4;   // appended-to buffers grow geometrically
        void* np = nullptr;
        hipError_t e = hipMalloc(&np, ncap);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu bytes) failed: %s", ncap, hipGetErrorString(e));
            return -12;
        }
        if (keep && p) {
            e = hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(np);
                set_error("DevBuf grow copy failed: %s", hipGetErrorString(e));
                return -5;
            }
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = ncap;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
constexpr int WAVE = 64;

// Scratch that outlives a call of the context-free entry points (error words, scan partials, per-session offsets): one
// growing buffer per (device, slot), kept until the process exits. A hipMalloc / hipFree pair per call would synchronise the
// device and the caller's stream on every call. Not thread-safe, like the contexts: one slot per entry point family, calls
// of one family on one device must not overlap.
enum { SCRATCH_CAND = 0, SCRATCH_RECENCY, SCRATCH_RECENCY_PRED, SCRATCH_EVENTS, SCRATCH_PAIRS_A, SCRATCH_PAIRS_B, SCRATCH_SLOTS };
inline int device_scratch(int slot, size_t bytes, void** out, hipStream_t s) {
    static DevBuf bufs[16][SCRATCH_SLOTS];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { set_error("device_scratch: no current device"); return -5; }
    const int rc = bufs[dev][slot].ensure(bytes < 256 ? 256 : bytes, 0, s);
    *out = bufs[dev][slot].p;
    return rc;
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// LDS traffic between the lanes of ONE wave: the LDS queue of a wave is in order, so only the compiler needs a fence
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
    const unsigned l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (l >= (unsigned)d) v += o;
    }
    return v;
}
// 32-bit values: six DPP adds (row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3) instead of six dependent ds_bpermute round trips through the LDS crossbar
template <>
__device__ __forceinline__ uint32_t wave_incl_scan<uint32_t>(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

// Block-wide exclusive scan of one value per thread (THREADS multiple of 64, <= 1024).
// `smem` must hold THREADS/64 + 1 elements of T. Returns the exclusive prefix; *total = block sum.
template <typename T, int THREADS>
__device__ __forceinline__ T block_excl_scan(T v, T* smem, T* total) {
    constexpr int NW = THREADS / 64;
    const unsigned l = lane_id(), w = threadIdx.x >> 6;
    T inc = wave_incl_scan(v);
    if (l == 63) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (int i = 0; i < NW; ++i) {
            T t = smem[i];
            smem[i] = run;
            run += t;
        }
        smem[NW] = run;
    }
    __syncthreads();
    T res = inc - v + smem[w];
    *total = smem[NW];
    __syncthreads();
    return res;
}

}  // namespace otto
