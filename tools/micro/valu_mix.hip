// Microbenchmark: issue cost of the softmax instruction mix of the d = 64 attention kernels on one wave per SIMD:
// per iteration 64 v_exp_f32, 62 v_add_f32 (four chains per 32 values) and 32 v_cvt_pk_f16_f32, alone and in parts.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 1 exps only, 2 adds only, 4 cvts only, 7 all (bit mask)
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float x[64];
    for (int i = 0; i < 64; ++i) x[i] = -(float)(threadIdx.x % 7) - i * 0.01f;
    float acc = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        float e[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (MODE & 1) e[i] = __builtin_amdgcn_exp2f(x[i]); else e[i] = x[i];
        }
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float r4[4];
#pragma unroll
                for (int r = 0; r < 32; ++r) r4[(r & 15) >> 2] = ((r >> 4) == 0 && (r & 3) == 0) ? e[32 * j + r] : r4[(r & 15) >> 2] + e[32 * j + r];
                acc += (r4[0] + r4[1]) + (r4[2] + r4[3]);
            }
        }
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < 64; i += 2) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 p = {(_Float16)e[i], (_Float16)e[i + 1]};
                asm volatile("" : "+v"(p));
                x[i] = (float)p[0] * 0.5f - 1.0f;   // (keeps x live and changing; cheap extra VALU, counted in the cvt-only line)
            }
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(e[i]));
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(x[i]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = acc;
    for (int i = 0; i < 64; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    k<MODE><<<256, threads>>>(out, cyc, iters);
    k<MODE><<<256, threads>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s %d waves/SIMD: %8.1f ticks / iteration\n", name, threads / 256, (double)h[0] / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run<1>("64 v_exp_f32", th);
        run<2>("62 v_add_f32 (4 chains x 2)", th);
        run<3>("64 exp + 62 add", th);
        run<4>("32 cvt_pk + 32 cvt + 32 fma", th);
        run<7>("all", th);
    }
    return 0;
}
