// Vector-ALU issue rates that bound the depthwise kernels (bsyolo_ops.hip, pmsfa_fused.hip, the DWConv half of dwpw_fused_kernel):
// back-to-back independent instructions from registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Prints lane-operations per second (and cycles per wave instruction at the clock the box reports) for
//   v_fma_f32                         (the plain f32 FMA)
//   v_fma_mix_f32 lo / hi             (f16 operand read straight out of a packed register: what the depthwise taps use)
//   v_cvt_f32_f16 + v_fma_f32         (convert once, FMA in f32)
//   v_exp_f32 / v_rcp_f32             (SiLU's two transcendentals)
// with 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float fma_mix_lo(unsigned hpair, float w, float acc) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(w), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(unsigned hpair, float w, float acc) {
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(w), "v"(acc));
    return r;
}
__device__ __forceinline__ float fma_f32(float a, float w, float acc) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(w), "v"(acc));
    return r;
}
__device__ __forceinline__ float cvt_lo(unsigned hpair) {
    float r;
    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(hpair));
    return r;
}
__device__ __forceinline__ float exp2_f(float a) {
    float r;
    asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ float rcp_f(float a) {
    float r;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(a));
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float w = 1.0f - threadIdx.x * 1e-6f;
    const unsigned hp = 0x3c003c00u + threadIdx.x;  // two f16 values near 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) acc[i] = fma_f32(acc[(i + 1) & 15], w, acc[i]);
                else if (MODE == 1) acc[i] = fma_mix_lo(hp, w, acc[i]);
                else if (MODE == 2) acc[i] = fma_mix_hi(hp, w, acc[i]);
                else if (MODE == 3) acc[i] = fma_f32(cvt_lo(hp + i), w, acc[i]);
                else if (MODE == 4) acc[i] = exp2_f(acc[i]);
                else acc[i] = rcp_f(acc[i]);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, int instr_per_op, double ghz) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        const int grid = 256 * wps, iters = 4000;
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, 100);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops_per_wave = (double)iters * 64;                // operations (= `instr_per_op` instructions each) per wave
        const double lane_ops = ops_per_wave * 64 * 4 * grid;          // 4 waves per workgroup
        const double cyc = ms * 1e-3 * ghz * 1e9 / (ops_per_wave * wps);  // SIMD cycles per operation of one wave
        printf("%-28s %d wave(s)/SIMD: %7.2f T lane-op/s  %5.2f cycles per wave-op (%d instr) at %.2f GHz\n", name, wps, lane_ops / (ms * 1e-3) / 1e12, cyc,
               instr_per_op, ghz);
    }
    hipFree(out);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz\n", prop.name, prop.multiProcessorCount, ghz);
    run<0>("v_fma_f32", 1, ghz);
    run<1>("v_fma_mix_f32 (lo)", 1, ghz);
    run<2>("v_fma_mix_f32 (hi)", 1, ghz);
    run<3>("v_cvt_f32_f16 + v_fma_f32", 2, ghz);
    run<4>("v_exp_f32", 1, ghz);
    run<5>("v_rcp_f32", 1, ghz);
    return 0;
}
