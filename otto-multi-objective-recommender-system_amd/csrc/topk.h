// Wave-level top-k primitives shared by the covisitation reduce (otto_covis.hip) and the candidate lookup
// (otto_cand.hip): 64-bit composite keys (bigger = better, 0 = empty), a wide (u64, u32) key for weights that need
// more than 36 bits, a sorted list held one entry per lane, a DPP-assisted 64-lane bitonic sort and a
// threshold-filtered selection.
#pragma once
#include "common.h"

namespace otto {

constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;

// ---- candidate keys: "better" = larger weight, then smaller aid_y
// KeyN: one 64-bit word  unit_weight << 26 | (2^26-1 - aid_y)  (unit weight < 2^36: < 2^28 sessions x weight < 256)
// KeyW: (Q16 weight u64, aid_y) for the time-weighted kind whose weight needs up to 46 bits
struct KeyN { uint64_t c; };
struct KeyW { uint64_t w; uint32_t y; };

__device__ __forceinline__ bool kbetter(KeyN a, KeyN b) { return a.c > b.c; }
__device__ __forceinline__ bool kbetter(KeyW a, KeyW b) { return a.w > b.w || (a.w == b.w && a.y < b.y); }
__device__ __forceinline__ bool kvalid(KeyN a) { return a.c != 0; }
__device__ __forceinline__ bool kvalid(KeyW a) { return a.w != 0; }
__device__ __forceinline__ void kclear(KeyN& a) { a.c = 0; }
__device__ __forceinline__ void kclear(KeyW& a) { a.w = 0; a.y = KEY_EMPTY; }
__device__ __forceinline__ KeyN kshfl(KeyN a, int src) { return {(uint64_t)__shfl((unsigned long long)a.c, src, 64)}; }
__device__ __forceinline__ KeyW kshfl(KeyW a, int src) { return {(uint64_t)__shfl((unsigned long long)a.w, src, 64), (uint32_t)__shfl(a.y, src, 64)}; }
// lane ^ m exchange. m < 16 stays inside a 16-lane row and is done with DPP moves (quad_perm, row_half_mirror,
// row_mirror and their compositions: xor4 = quad-reverse o half-mirror, xor8 = half-mirror o row-mirror) instead
// of an LDS-crossbar ds_bpermute; m = 16, 32 use the shuffle.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t xor_lane32(uint32_t v, int m) {
    switch (m) {
        case 1: return dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
        case 2: return dpp_mov<0x4E>(v);                       // quad_perm [2,3,0,1]
        case 4: return dpp_mov<0x1B>(dpp_mov<0x141>(v));       // quad reverse after row_half_mirror
        case 8: return dpp_mov<0x141>(dpp_mov<0x140>(v));      // row_half_mirror after row_mirror
        default: return (uint32_t)__shfl_xor((int)v, m, 64);
    }
}
__device__ __forceinline__ uint64_t xor_lane64(uint64_t v, int m) {
    return ((uint64_t)xor_lane32((uint32_t)(v >> 32), m) << 32) | xor_lane32((uint32_t)v, m);
}
__device__ __forceinline__ KeyN kshfl_xor(KeyN a, int m) { return {xor_lane64(a.c, m)}; }
__device__ __forceinline__ KeyW kshfl_xor(KeyW a, int m) { return {xor_lane64(a.w, m), xor_lane32(a.y, m)}; }
__device__ __forceinline__ KeyN kshfl_up(KeyN a) { return {(uint64_t)__shfl_up((unsigned long long)a.c, 1, 64)}; }
__device__ __forceinline__ KeyW kshfl_up(KeyW a) { return {(uint64_t)__shfl_up((unsigned long long)a.w, 1, 64), (uint32_t)__shfl_up(a.y, 1, 64)}; }
// storage of partial lists: (u64, u32)
__device__ __forceinline__ void kstore(KeyN k, uint64_t* w, uint32_t* y) { *w = k.c; *y = 0; }
__device__ __forceinline__ void kstore(KeyW k, uint64_t* w, uint32_t* y) { *w = k.w; *y = k.y; }
__device__ __forceinline__ void kload(KeyN& k, uint64_t w, uint32_t) { k.c = w; }
__device__ __forceinline__ void kload(KeyW& k, uint64_t w, uint32_t y) { k.w = w; k.y = w ? y : KEY_EMPTY; }

// rank += (o is better than key). (An inline-asm compare + add-with-carry form -- two vector instructions per candidate -- was
// measured: the M bin's two-pass path went from 7.3 to 8.2 ms; the volatile statements keep the compiler from overlapping the
// LDS broadcasts of the following candidates with the compares.)
template <typename K>
__device__ __forceinline__ void kcount_better(uint32_t& rank, K o, K key) { rank += kbetter(o, key) ? 1u : 0u; }

// Wave-wide sorted top list: lane i holds the i-th best. Insert the per-lane candidates that beat the k-th.
template <typename K>
__device__ __forceinline__ void wave_topk_push(K& best, K cand, int k) {
    const unsigned l = lane_id();
    K thr = kshfl(best, k - 1);
    uint64_t m = __ballot(kvalid(cand) && kbetter(cand, thr));
    while (m) {
        const int src = __ffsll((unsigned long long)m) - 1;
        const K c = kshfl(cand, src);
        const K up = kshfl_up(best);
        if (kbetter(c, best)) best = (l > 0 && kbetter(c, up)) ? up : c;
        thr = kshfl(best, k - 1);
        m &= m - 1;
        m &= __ballot(kvalid(cand) && kbetter(cand, thr));
    }
}

// 64-lane bitonic sort, best first
template <typename K>
__device__ __forceinline__ void wave_bitonic_sort_desc(K& v) {
    const unsigned l = lane_id();
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const K o = kshfl_xor(v, j);
            const bool keep_better = ((l & j) == 0) == ((l & k) == 0);
            if (keep_better ? kbetter(o, v) : kbetter(v, o)) v = o;
        }
    }
}

// Top-k of MPL register-resident candidates per lane: lane-local best -> 64-wide bitonic sort -> only the
// few candidates that still beat the k-th entry are inserted one by one. Result: lane i = i-th best.
template <int MPL, typename K>
__device__ __forceinline__ void wave_topk_select(K (&c)[MPL], int k, K& best) {
    K lb = c[0];
    int bi = 0;
#pragma unroll
    for (int i = 1; i < MPL; ++i)
        if (kbetter(c[i], lb)) { lb = c[i]; bi = i; }
#pragma unroll
    for (int i = 0; i < MPL; ++i)
        if (i == bi) kclear(c[i]);
    best = lb;
    wave_bitonic_sort_desc(best);
    if (MPL > 1) {
        for (;;) {
            const K thr = kshfl(best, k - 1);
            kclear(lb);
            bi = -1;
#pragma unroll
            for (int i = 0; i < MPL; ++i)
                if (kvalid(c[i]) && (bi < 0 || kbetter(c[i], lb))) { lb = c[i]; bi = i; }
            const bool q = bi >= 0 && kbetter(lb, thr);
            if (__ballot(q) == 0) break;
#pragma unroll
            for (int i = 0; i < MPL; ++i)
                if (i == bi) kclear(c[i]);
            if (!q) kclear(lb);
            wave_topk_push(best, lb, k);
        }
    }
}


}  // namespace otto
