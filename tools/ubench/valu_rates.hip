// Microbenchmark (diagnostic, not product): issue cost of the integer VALU instructions the reduce kernels lean on, gfx950.
// Four waves per SIMD (16 waves per workgroup, one workgroup per CU), 8 independent chains per lane; reports shader-clock
// cycles per wave-instruction per SIMD (elapsed cycles of a wave / the wave-instructions its SIMD issued meanwhile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, unsigned long long* cyc, uint32_t seed) {
    uint32_t a[8];
    uint64_t b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = threadIdx.x * 2654435761u + u * 40503u + seed; b[u] = ((uint64_t)a[u] << 20) ^ (a[u] * 97u); }
    const uint32_t m = seed | 1u;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) a[u] = a[u] * m;                                              // v_mul_lo_u32
            if (OP == 1) a[u] = __umul24(a[u], m);                                     // v_mul_u32_u24
            if (OP == 2) a[u] = __umul24(a[u], m) + a[(u + 1) & 7];                    // v_mad_u32_u24
            if (OP == 3) a[u] = a[u] + m;                                              // v_add_u32 (reference)
            if (OP == 4) b[u] = b[u] + ((uint64_t)m << 3);                             // 64-bit add (add_co + addc)
            if (OP == 5) b[u] = (uint64_t)a[u] * m + b[u];                             // v_mad_u64_u32
            if (OP == 6) a[u] += (b[u] > b[(u + 1) & 7]) ? 1u : 0u;                    // v_cmp_gt_u64 + cndmask/addc
            if (OP == 7) b[u] = b[u] << (m & 31);                                      // v_lshlrev_b64
            if (OP == 8) a[u] = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x * 4 + u * 8) & 255), (int)a[u]);   // ds_bpermute_b32
            if (OP == 9) a[u] = __popcll(b[u] ^ a[u]);                                 // 2 x v_bcnt
            if (OP == 10) a[u] = __shfl_xor((int)a[u], 1, 64);                         // DPP / swizzle
        }
    }
    const unsigned long long t1 = clock64();
    uint32_t r = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) r ^= a[u] ^ (uint32_t)b[u] ^ (uint32_t)(b[u] >> 32);
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name) {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 4096;
    k<OP><<<256, 1024>>>(out, iters, cyc, 12345u);
    k<OP><<<256, 1024>>>(out, iters, cyc, 12345u);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-28s %7.2f cycles per wave-instruction per SIMD\n", name, s / 256 / ((double)iters * 8 * 4));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<3>("v_add_u32");
    run<0>("v_mul_lo_u32");
    run<1>("v_mul_u32_u24");
    run<2>("v_mad_u32_u24");
    run<4>("64-bit add");
    run<5>("v_mad_u64_u32");
    run<6>("v_cmp_gt_u64 + select");
    run<7>("v_lshlrev_b64");
    run<8>("ds_bpermute_b32");
    run<9>("popcll (2 x v_bcnt)");
    run<10>("shfl_xor 1");
    return 0;
}
