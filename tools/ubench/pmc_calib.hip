// Calibration of the L2 memory-side byte counters for the access shapes the traversal kernels
// use, on known byte counts:
//   gather64  : every lane reads one random 64-byte record (4 x 16 B) of a table far larger than
//               the on-die caches  -> N * 64 bytes must be fetched
//   stream32  : every lane reads one 32-byte record of a contiguous array (2 x 16 B) and writes
//               one 32-byte record                                        -> N * 32 read, N * 32 written
// Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and compare with the printed counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void gather64(const float4 *table, unsigned records, float4 *sink, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float4 *r = table + size_t(h % records) * 4;
    float4 a = r[0], b = r[1], c = r[2], d = r[3];
    float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) sink[0] = a;  // never true: keeps the loads alive
}

__global__ void stream32(const float4 *in, float4 *out, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = in[size_t(i) * 2], b = in[size_t(i) * 2 + 1];
    out[size_t(i) * 2] = b;
    out[size_t(i) * 2 + 1] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const unsigned records = 1u << 25;  // 2 GiB table of 64-byte records
    const unsigned n = 1u << 24;
    float4 *table, *sink, *in, *out;
    CK(hipMalloc(&table, size_t(records) * 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&in, size_t(n) * 32));
    CK(hipMalloc(&out, size_t(n) * 32));
    CK(hipMemset(table, 0, size_t(records) * 64));
    CK(hipMemset(in, 0, size_t(n) * 32));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(gather64, dim3(n / 256), dim3(256), 0, 0, table, records, sink, n);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(stream32, dim3(n / 256), dim3(256), 0, 0, in, out, n);
    CK(hipDeviceSynchronize());
    printf("gather64: expected read %.1f MB\n", double(n) * 64 / 1e6);
    printf("stream32: expected read %.1f MB, written %.1f MB\n", double(n) * 32 / 1e6, double(n) * 32 / 1e6);
    return 0;
}
