// Microbenchmark: is a packed fp32 instruction (v_pk_add_f32 / v_pk_mul_f32, two results per lane) cheaper
// than the two scalar instructions it replaces?  Every wave runs rounds of six independent
// subtract -> multiply chains (the shape of two slab tests), written either as 12 + 12 scalar
// instructions or as 6 + 6 packed ones, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 8192;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void pk(float *out, float seed) {
    v2f a[6], o[6], m[6];
    for (int k = 0; k < 6; ++k) {
        a[k] = v2f{seed + float(threadIdx.x + k), seed - float(k)};
        o[k] = v2f{0.25f * float(k), 0.25f * float(k)};
        m[k] = v2f{1.0001f, 0.9999f};
    }
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (KIND == 0) {
                asm volatile("v_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %3\n\tv_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5"
                             : "+v"(a[k].x), "+v"(a[k].y)
                             : "v"(o[k].x), "v"(o[k].y), "v"(m[k].x), "v"(m[k].y));
            } else {
                asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %2" : "+v"(a[k]) : "v"(o[k]), "v"(m[k]));
            }
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 6; ++k) s += a[k].x + a[k].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, float *out, int cus) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(pk<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(pk<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // cost of one round (two slab tests' worth of differences and products) per wave, in SIMD time
        const double ns_per_round = ms * 1e6 / (double(kIters) * wps);
        printf("%-34s waves/SIMD %d : %8.3f ms  %7.2f ns per round per wave (%.1f cycles at 2.4 GHz)\n", name, wps, ms, ns_per_round, ns_per_round * 2.4);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    hipMalloc(&out, size_t(cus) * 4 * 8 * 64 * sizeof(float));
    run<0>("12 v_sub_f32 + 12 v_mul_f32", out, cus);
    run<1>("6 v_pk_add_f32 + 6 v_pk_mul_f32", out, cus);
    return 0;
}
