// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: (1) the A / B lane -> (row, k) map for e4m3 and e2m3 operands, found with exact
// small-integer data against a host product under candidate maps; (2) the meaning of the E8M0 scale operands; (3) the issue rate beside
// v_mfma_f32_32x32x16_f16 (the mix a "fp16 hi x hi + low-precision cross terms" convolution would issue).
//   hipcc --offload-arch=gfx950 -O3 mfma_f8f6f4.hip -o mfma_f8f6f4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- (1) one MFMA on given per-lane operand registers; D stored [lane][16]
template <int FA, int FB>
__global__ void one(const i32x8* a, const i32x8* b, float* d, int sa, int sb) {
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, FA, FB, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = c[r];
}

static uint8_t e4m3_of_int(int v) {   // exact for |v| <= 8 (and more): sign | exponent bias 7 | 3 mantissa bits
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    int a = abs(v), e = 0;
    while ((a >> (e + 1)) != 0) ++e;              // floor(log2 a)
    int man = ((a << 3) >> e) & 7;                // exact while a < 16
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)man;
}
static uint8_t e2m3_of_halfsteps(int v) {   // value = v / 8 for |v| <= 8 ... grid below 2 has step 0.125: code = sign | e(2) | m(3)
    uint8_t s = v < 0 ? 0x20 : 0;
    int a = abs(v);                              // a in 0..15: subnormal (e = 0) a <= 7 -> a / 8; e = 1: 1 + m/8 (a = 8..15)
    return s | (uint8_t)(a & 0xF) ;              // e = a >> 3 (0 or 1), m = a & 7: bits [4:3] = e, [2:0] = m
}

// operand images: A[row][k], B[k][col] integer-valued; map candidates give (lane, byte j) -> k
static int kmap(int cand, int h, int j) {
    switch (cand) {
        case 0: return 32 * h + j;                                   // lane half h holds k = 32 h .. + 31
        case 1: return 16 * h + (j & 15) + 32 * (j >> 4);            // two 16-wide groups: [16 h ..], [32 + 16 h ..]
        case 2: return 8 * h + (j & 7) + 16 * (j >> 3);              // four 8-wide groups
        default: return 2 * j + h;
    }
}

template <int FMT> static void pack_lane(uint8_t* dst, const int* vals32) {   // 32 values of one lane -> register bytes
    memset(dst, 0, 32);
    if (FMT == 0) { for (int j = 0; j < 32; ++j) dst[j] = e4m3_of_int(vals32[j]); return; }
    // e2m3: 6-bit codes packed little-endian, element j at bits [6 j, 6 j + 6)
    for (int j = 0; j < 32; ++j) {
        const uint32_t code = e2m3_of_halfsteps(vals32[j]);
        const int bit = 6 * j;
        for (int t = 0; t < 6; ++t) if (code >> t & 1) dst[(bit + t) >> 3] |= (uint8_t)(1u << ((bit + t) & 7));
    }
}

template <int FMT> static void probe_layout() {
    static int A[32][64], B[64][32];
    srand(1);
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) { A[r][k] = rand() % 9 - 4; B[k][r] = rand() % 7 - 3; }
    const double unit = FMT == 0 ? 1.0 : 1.0 / 64.0;    // e2m3 values are v / 8 on both sides
    for (int cand = 0; cand < 4; ++cand) {
        static uint8_t ha[64][32], hb[64][32];
        for (int l = 0; l < 64; ++l) {
            int va[32], vb[32];
            for (int j = 0; j < 32; ++j) { const int k = kmap(cand, l >> 5, j); va[j] = A[l & 31][k]; vb[j] = B[k][l & 31]; }
            pack_lane<FMT>(ha[l], va);
            pack_lane<FMT>(hb[l], vb);
        }
        i32x8 *da, *db; float* dd;
        hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 64 * 16 * 4);
        hipMemcpy(da, ha, 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64 * 32, hipMemcpyHostToDevice);
        if (FMT == 0) one<0, 0><<<1, 64>>>(da, db, dd, 0x7f7f7f7f, 0x7f7f7f7f);     // E8M0 127 = 2^0 in every byte
        else one<2, 2><<<1, 64>>>(da, db, dd, 0x7f7f7f7f, 0x7f7f7f7f);
        float hd[64][16];
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);   // the C/D map of every 32x32 MFMA
            double want = 0;
            for (int k = 0; k < 64; ++k) want += (double)A[row][k] * B[k][col];
            if (fabs(hd[l][r] - want * unit) > 1e-3) ++bad;
        }
        printf("  %s operands, k-map candidate %d (A row = lane & 31, B col = lane & 31): %s (%d / 1024 wrong)\n", FMT == 0 ? "e4m3" : "e2m3", cand,
               bad ? "no" : "MATCH", bad);
        hipFree(da); hipFree(db); hipFree(dd);
    }
}

static void probe_scale() {
    // all-ones operands (e4m3 1.0 = 0x38): D = 64 * 2^(sa - 127) * 2^(sb - 127) if the scale bytes are E8M0 exponents applied to both k blocks
    static uint8_t h[64][32];
    memset(h, 0x38, sizeof(h));
    i32x8 *da; float* dd;
    hipMalloc(&da, 64 * 32); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, h, 64 * 32, hipMemcpyHostToDevice);
    const int cases[][2] = {{0x7f, 0x7f}, {0x80, 0x7f}, {0x7f, 0x7d}, {0x7f - 21, 0x7f}, {0, 0}};
    for (auto& c : cases) {
        one<0, 0><<<1, 64>>>(da, da, dd, c[0], c[1]);
        float hd[16];
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        printf("  scale bytes (A 0x%02x, B 0x%02x), all-ones operands: D[0] = %g  (64 * 2^%d = %g expected if E8M0 on byte 0 of each lane's scale register)\n",
               c[0], c[1], hd[0], (c[0] - 127) + (c[1] - 127), 64.0 * exp2((double)(c[0] - 127 + c[1] - 127)));
    }
    hipFree(da); hipFree(dd);
}

// ---- (3) issue rate: MODE 0: f16 32x32x16 only; 1: e4m3 32x32x64 only; 2: e2m3 32x32x64 only; 3: 2 f16 + 1 e4m3 per group; 4: 2 f16 + 1 e2m3
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    i32x8 qa, qb;
    for (int e = 0; e < 8; ++e) { qa[e] = 0x38383838 + (int)threadIdx.x * 0x01010101 % 7; qb[e] = 0x30303030 + e; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int s = 0x7f7f7f7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0 || MODE >= 3) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[u], 0, 0, 0);
            }
            if (MODE == 1 || MODE == 3) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[u], 0, 0, 0, s, 0, s);
            if (MODE == 2 || MODE == 4) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[u], 2, 2, 0, s, 0, s);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE> static void run_rate(const char* name, double k_per_group) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<MODE><<<1024, 256>>>(out, iters);
    hipEventRecord(e0);
    rate<MODE><<<1024, 256>>>(out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = 1024.0 * 4 * iters * 4;                          // per wave: iters x 4 groups
    const double flops = groups * 2.0 * 32 * 32 * k_per_group;
    printf("  %-58s %8.2f ms  %8.1f TFLOP/s (K per group %g)\n", name, ms, flops / (ms * 1e-3) / 1e12, k_per_group);
    hipFree(out);
}

int main() {
    printf("(1) operand lane maps, exact integer data\n");
    probe_layout<0>();
    probe_layout<2>();
    printf("(2) scale operands\n");
    probe_scale();
    printf("(3) issue rate, 1024 workgroups x 4 waves, 4 independent accumulators per wave (2 waves per SIMD)\n");
    run_rate<0>("2 x v_mfma_f32_32x32x16_f16", 32);
    run_rate<1>("1 x v_mfma_scale_f32_32x32x64_f8f6f4 e4m3", 64);
    run_rate<2>("1 x v_mfma_scale_f32_32x32x64_f8f6f4 e2m3", 64);
    run_rate<3>("2 x f16 + 1 x e4m3 (one 32-channel chunk, hi x hi + both cross terms)", 32);
    run_rate<4>("2 x f16 + 1 x e2m3", 32);
    printf("    today's split precision issues 6 x 16-bit MFMAs per such group (3 segments x K = 32)\n");
    return 0;
}
