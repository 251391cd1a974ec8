// Microbenchmark: does a wave64 VALU instruction cost less when one 32-lane half of the wave is masked off?
// The same dependent fp32 / min-max loop runs with all 64 lanes, with lanes 0..31 only, with the even lanes only
// (32 active lanes spread over both halves) and with 16 lanes of one half.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 8192;

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float seed) {
    const int lane = threadIdx.x;
    const bool on = KIND == 0 ? true : KIND == 1 ? lane < 32 : KIND == 2 ? (lane & 1) == 0 : lane < 16;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + float(lane + i);
    if (on) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                a[i] = __builtin_fminf(a[i], a[(i + 1) & 7]);
                a[i] = __builtin_fmaxf(a[i], 0.25f);
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + lane] = s;
}

template <int KIND>
void run(const char *name, float *out, int cus) {
    for (int wps : {1, 4}) {
        const int blocks = cus * 4 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / (double(kIters) * 24 * wps);
        printf("%-28s waves/SIMD %d : %7.3f ms  %5.2f cycles per wave-instruction at 2.4 GHz\n", name, wps, ms, ns * 2.4);
    }
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    (void)hipMalloc(&out, size_t(cus) * 4 * 8 * 64 * sizeof(float));
    run<0>("all 64 lanes", out, cus);
    run<1>("lanes 0..31", out, cus);
    run<2>("even lanes", out, cus);
    run<3>("lanes 0..15", out, cus);
    return 0;
}
