// Microbenchmark: throughput of dependent random 64-byte record gathers from an L2-resident
// table (the BVH node fetch pattern), for three ways of issuing the loads:
//   A  every lane loads its own record with 4 x 16-byte loads (64 distinct lines per instruction)
//   B  4 adjacent lanes load one record together (16 distinct lines per instruction), the
//      records are transposed to their owners through LDS with ds_write_b128 / ds_read_b128
//   C  like B, but the loads land in LDS directly (global_load_lds_dwordx4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int kIters = 256;

__device__ __forceinline__ unsigned mix(unsigned h) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}

template <int VARIANT>
__global__ __launch_bounds__(64) void gather(const float4 *table, unsigned records, unsigned *out, int active_lanes) {
    __shared__ float4 stage[256];
    const unsigned lane = threadIdx.x;
    unsigned idx = mix(blockIdx.x * 64u + lane) % records;
    unsigned acc = 0;
    const bool on = int(lane) < active_lanes;
    for (int it = 0; it < kIters; ++it) {
        float4 a, b, c, d;
        if (VARIANT == 0) {
            if (on) {
                const float4 *r = table + size_t(idx) * 4;
                a = r[0]; b = r[1]; c = r[2]; d = r[3];
            }
        } else {
            for (int k = 0; k < 4; ++k) {
                const int src = 16 * k + int(lane >> 2);
                const unsigned his = __builtin_amdgcn_ds_bpermute(src << 2, idx);
                const float4 *p = table + size_t(his) * 4 + (lane & 3u);
                if (VARIANT == 1) {
                    stage[k * 64 + lane] = *p;
                } else {
                    __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(stage + k * 64), 16, 0, 0);
                }
            }
            if (VARIANT == 2) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the LDS-DMA loads have landed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            a = stage[lane * 4 + 0]; b = stage[lane * 4 + 1]; c = stage[lane * 4 + 2]; d = stage[lane * 4 + 3];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (on) {
            const unsigned v = __float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z) ^ __float_as_uint(d.w);
            acc += v;
            idx = mix(idx ^ v) % records;
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
int run(const char *name, const float4 *table, unsigned records, unsigned *out, int blocks, int active, unsigned *check) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(gather<V>, dim3(blocks), dim3(64), 0, 0, table, records, out, active);
    CK(hipDeviceSynchronize());
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gather<V>, dim3(blocks), dim3(64), 0, 0, table, records, out, active);
    hipEventRecord(b); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    std::vector<unsigned> h(size_t(blocks) * 64);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned sum = 0; for (unsigned v : h) sum += v;
    const double recs = double(blocks) * active * kIters;
    printf("%-26s waves %5d lanes %2d : %7.3f ms  %7.1f Grecords/s  checksum %08x%s\n", name, blocks, active, ms, recs / ms / 1e6, sum,
           (*check && *check != sum) ? "  MISMATCH" : "");
    if (!*check) *check = sum;
    return 0;
}

int main(int argc, char **argv) {
    // default 9.4 MB: L2 / MALL resident like the helmet's BVH; argv[1] = table size in KiB
    const unsigned records = argc > 1 ? unsigned(atoi(argv[1])) * 16u : 147456u;
    printf("table %.2f MB\n", records * 64.0 / 1e6);
    std::vector<float> h(size_t(records) * 16);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; unsigned u = s; memcpy(&v, &u, 4); }
    float4 *table; unsigned *out;
    CK(hipMalloc(&table, h.size() * 4));
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 8192 * 64 * 4));
    for (int blocks : {1024, 4096, 8192}) {
        for (int active : {64}) {
            unsigned check = 0;
            if (run<0>("A per-lane 4x16B", table, records, out, blocks, active, &check)) return 1;
            if (run<1>("B 4-lane coop + LDS", table, records, out, blocks, active, &check)) return 1;
            if (run<2>("C 4-lane coop LDS-DMA", table, records, out, blocks, active, &check)) return 1;
        }
    }
    return 0;
}
