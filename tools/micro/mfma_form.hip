// Microbenchmark: issue rate of v_mfma_f32_32x32x16_f16 with the accumulator in VGPRs vs AGPRs (one wave per SIMD, four
// independent accumulators in rotation, or one dependent chain).   hipcc --offload-arch=gfx950 -O3 mfma_form.hip -o mfma_form
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FORM, int NACC>   // FORM 0: AGPR accumulators, 1: VGPR accumulators
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (FORM == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FORM, int NACC> void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<FORM, NACC><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e0);
    k<FORM, NACC><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double n = (double)iters * 16;
    double tf = 256.0 * 4 * n * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-44s %7.2f s_memtime ticks / MFMA   %8.1f ns / MFMA  (%7.1f TFLOP/s chip-wide)\n", name, h[0] / n, ms * 1e6 / n, tf);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 4>("AGPR accumulators, 4 in rotation");
    run<1, 4>("VGPR accumulators, 4 in rotation");
    run<0, 1>("AGPR accumulator, one dependent chain");
    run<1, 1>("VGPR accumulator, one dependent chain");
    run<0, 2>("AGPR accumulators, 2 in rotation");
    run<1, 2>("VGPR accumulators, 2 in rotation");
    return 0;
}
