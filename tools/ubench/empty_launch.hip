// Microbenchmark: cost of an "empty" persistent launch (every wave does one atomic fetch and exits)
// as a function of block shape and static LDS size.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES, bool ATOMIC>
__global__ void k(unsigned *cursor, unsigned n, unsigned *sink) {
    __shared__ unsigned lds[LDS_BYTES / 4 + 1];
    unsigned v = 0;
    if (ATOMIC) {
        if ((threadIdx.x & 63) == 0) v = atomicAdd(cursor, 64u);
        v = __shfl(v, 0, 64);
    } else {
        v = *cursor;
    }
    if (v < n) { lds[threadIdx.x] = v; sink[threadIdx.x] = lds[(threadIdx.x + 1) % 64]; }
}
template <int LDS_BYTES, bool ATOMIC>
void run(const char *name, int blocks, int threads, unsigned *cursor, unsigned *sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LDS_BYTES, ATOMIC>), dim3(blocks), dim3(threads), 0, 0, cursor, 0u, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<LDS_BYTES, ATOMIC>), dim3(blocks), dim3(threads), 0, 0, cursor, 0u, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %5d x %4d : %8.1f us per launch\n", name, blocks, threads, ms * 1000 / 20);
}
int main() {
    unsigned *cursor, *sink; hipMalloc(&cursor, 256); hipMalloc(&sink, 4096); hipMemset(cursor, 0, 256);
    run<0, true>("atomic, no LDS", 4096, 64, cursor, sink);
    run<8192, true>("atomic, 8K LDS", 4096, 64, cursor, sink);
    run<8192, false>("no atomic, 8K LDS", 4096, 64, cursor, sink);
    run<0, false>("no atomic, no LDS", 4096, 64, cursor, sink);
    run<32768, true>("atomic, 32K LDS", 1024, 256, cursor, sink);
    run<32768, false>("no atomic, 32K LDS", 1024, 256, cursor, sink);
    run<8192, true>("atomic, 8K LDS", 1024, 64, cursor, sink);
    run<8192, true>("atomic, 8K LDS", 256, 64, cursor, sink);
    run<0, true>("atomic, no LDS", 32768, 64, cursor, sink);
    return 0;
}
