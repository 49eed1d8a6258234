// Matrix-pipe issue rates on the box it runs on: back-to-back independent MFMAs from registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Prints TFLOP/s of v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x16_f16 with 1, 2 and 4 waves per SIMD -- what the
// roofline fractions of bench.py (fp32 mode: 157.3 TFLOP/s nominal; fp16: 2.5 PFLOP/s) should be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(threadIdx.x + i + j);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) acc4[i][j] = (float)(threadIdx.x + i + j);
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    half8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(1.0f + j * 1e-3f); hb[j] = (_Float16)(1.0f - j * 1e-3f); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc4[i][j];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_mfma, int per_iter) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        const int grid = 256 * wps, iters = 20000;
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)grid * 4 * iters * per_iter;
        printf("%-28s %d wave(s)/SIMD: %8.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, wps, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12,
               ms * 1e-3 * 2.4e9 / ((double)wps * iters * per_iter));
    }
    hipFree(out);
}

int main() {
    run<0>("v_mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2, 16);
    run<1>("v_mfma_f32_16x16x4_f32", 16.0 * 16 * 4 * 2, 16);
    run<2>("v_mfma_f32_32x32x16_f16", 32.0 * 32 * 16 * 2, 16);
    return 0;
}
