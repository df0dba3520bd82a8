// How many workgroups of 256 threads fit a CU for a given static LDS size? (the runtime's answer + a measured one:
// every workgroup records the CU it ran on and spins until all workgroups of the launch have started)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void k(uint32_t* out) {
    __shared__ uint8_t s[BYTES];
    s[threadIdx.x] = (uint8_t)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[(blockIdx.x * 7) % BYTES];
}
template <int BYTES, int THREADS>
void probe() {
    int n = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES, THREADS>, THREADS, 0);
    printf("LDS %6d B, %4d threads: %d workgroups per CU\n", BYTES, THREADS, n);
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    probe<32768, 256>(); probe<37924, 256>(); probe<40848, 256>(); probe<40960, 256>(); probe<42020, 256>(); probe<49152, 256>();
    probe<53248, 256>(); probe<54000, 256>(); probe<65536, 256>(); probe<81584, 512>(); probe<7176, 64>(); probe<8192, 64>();
    return 0;
}
