// Microbenchmark (diagnostic, not product): throughput of LDS atomics with random addresses on gfx950, per CU.
// Each wave issues ITER x UNROLL independent atomics per lane on a table of T slots; reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int OP, int UNROLL>
__global__ void k(uint64_t* out, int T, int iters, unsigned long long* cyc) {
    extern __shared__ unsigned long long tab[];
    uint32_t* tab32 = reinterpret_cast<uint32_t*>(tab);
    for (int i = threadIdx.x; i < T; i += blockDim.x) tab[i] = ~0ull;
    __syncthreads();
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B1u + 12345u;
    unsigned long long acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t slot[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { h = h * 1664525u + 1013904223u; slot[u] = (h >> 8) & (T - 1); }
        if (OP == 0) {            // 64-bit CAS with return, results consumed after the batch
            unsigned long long r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = atomicCAS(&tab[slot[u]], ~0ull, (unsigned long long)slot[u] << 36 | 1);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += r[u];
        } else if (OP == 1) {     // 64-bit CAS, each consumed before the next (serial chain)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { unsigned long long r = atomicCAS(&tab[(slot[u] + (uint32_t)(acc & 1)) & (T - 1)], ~0ull, (unsigned long long)slot[u] << 36 | 1); acc += r; }
        } else if (OP == 2) {     // 64-bit add, no return
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) atomicAdd(&tab[slot[u]], 1ull);
        } else if (OP == 3) {     // 32-bit CAS with return, batched
            uint32_t r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = atomicCAS(&tab32[slot[u]], 0xFFFFFFFFu, slot[u]);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += r[u];
        } else if (OP == 4) {     // 32-bit add, no return
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) atomicAdd(&tab32[slot[u]], 1u);
        } else if (OP == 5) {     // 64-bit plain read (reference)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += tab[slot[u]];
        } else if (OP == 6) {     // 64-bit plain write (reference)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) tab[slot[u]] = h;
        } else if (OP == 7) {     // 32-bit or, no return
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) atomicOr(&tab32[slot[u] >> 5], 1u << (slot[u] & 31));
        } else if (OP == 8) {     // 64-bit add WITH return, batched
            unsigned long long r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = atomicAdd(&tab[slot[u]], 1ull);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += r[u];
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x1234567ull) out[0] = acc + tab[threadIdx.x & (T - 1)];
}

template <int OP>
void run(const char* name, int threads, int T, int nblk_per_cu) {
    constexpr int UNROLL = 8;
    const int iters = 200;
    uint64_t* out; unsigned long long* cyc;
    const int nblk = 256 * nblk_per_cu;
    hipMalloc(&out, 8); hipMalloc(&cyc, nblk * 8);
    hipFuncSetAttribute((const void*)k<OP, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, T * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, UNROLL><<<nblk, threads, T * 8>>>(out, T, 10, cyc);
    hipEventRecord(e0);
    k<OP, UNROLL><<<nblk, threads, T * 8>>>(out, T, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nblk;
    const double winstr = (double)iters * UNROLL * (threads / 64) * nblk_per_cu;   // wave-instructions per CU
    printf("%-34s threads %4d x %d WG/CU  T %6d : %.3f ms, %8.0f clk/block, %6.1f CU-clk per wave-instr (64 lanes), %.2f lanes/clk/CU\n",
           name, threads, nblk_per_cu, T, ms, avg, avg / winstr, 64.0 * winstr / avg);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int threads = cfg == 0 ? 1024 : (cfg == 1 ? 256 : 64);
        const int T = cfg == 0 ? 16384 : (cfg == 1 ? 4096 : 512);
        const int per = cfg == 0 ? 1 : (cfg == 1 ? 4 : 20);
        run<0>("cas64 rtn batched", threads, T, per);
        run<1>("cas64 rtn serial chain", threads, T, per);
        run<2>("add64 no-rtn", threads, T, per);
        run<8>("add64 rtn batched", threads, T, per);
        run<3>("cas32 rtn batched", threads, T, per);
        run<4>("add32 no-rtn", threads, T, per);
        run<7>("or32 no-rtn (bitmap)", threads, T, per);
        run<5>("read64", threads, T, per);
        run<6>("write64", threads, T, per);
    }
    return 0;
}
