// Microbenchmark: cycles per wave64 VALU instruction on one SIMD as a function of the waves that
// share it, for the instruction kinds the traversal loop is made of (fp32 multiply-add, min/max,
// compare + select, byte-to-float conversion, integer min/max).  Every wave runs kIters rounds of
// kUnroll independent instructions; the launch puts `waves_per_simd` waves on every SIMD of the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kIters = 4096;

template <int KIND>
__global__ __launch_bounds__(64) void valu(float *out, float seed) {
    float a[8];
    unsigned u[8];
    for (int k = 0; k < 8; ++k) {
        a[k] = seed + float(threadIdx.x + k);
        u[k] = __float_as_uint(a[k]);
    }
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);
                if (KIND == 1) a[k] = __builtin_fmaxf(__builtin_fminf(a[k], a[(k + 1) & 7]), 0.25f);  // 2 instructions (or one med3)
                if (KIND == 2) a[k] = a[k] < a[(k + 3) & 7] ? a[(k + 1) & 7] : a[k] + 1.0f;            // cmp + add + cndmask
                if (KIND == 3) a[k] += float((u[k] >> 8) & 0xFFu);                                      // cvt_ubyte1 + add
                if (KIND == 4) u[k] = min(u[k] + 3u, u[(k + 1) & 7] ^ 0x55u);                             // add + xor + min_u32
            }
        }
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k] + __uint_as_float(u[k] & 0x3FFFFFFFu);
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int insts_per_op, float *out, int cus) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(valu<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_insts = double(kIters) * 32 * insts_per_op;  // per wave
        // every SIMD runs wps waves one after / beside the other: time per instruction slot on a SIMD
        const double ns_per_inst = ms * 1e6 / (wave_insts * wps);
        printf("%-28s waves/SIMD %d : %8.3f ms  %6.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms, ns_per_inst, ns_per_inst * 2.4);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float *out;
    hipMalloc(&out, size_t(cus) * 4 * 8 * 64 * sizeof(float));
    run<0>("v_fma_f32", 1, out, cus);
    run<1>("v_min_f32 + v_max_f32", 2, out, cus);
    run<2>("v_cmp + v_add + v_cndmask", 3, out, cus);
    run<3>("v_cvt_f32_ubyte1 + v_add", 2, out, cus);
    run<4>("v_add_u32 + v_xor + v_min_u32", 3, out, cus);
    return 0;
}
