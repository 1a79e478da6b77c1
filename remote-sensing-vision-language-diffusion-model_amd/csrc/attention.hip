// attention.hip — flash-style attention for gfx950: out = softmax(scale * Q K^T) V with an
// fp32 online softmax; the N x N score matrix never exists in HBM.
//
// Two kernels (below): D = 512, one head (SR3 SelfAttention, VAE mid-block attention) and D = 64, multi-head
// (sgm CrossAttention / MemoryEfficientCrossAttention, ZeroCrossAttn).  Both compute S^T = K Q^T "swapped" (keys on the
// accumulator registers, the query row on the lane), so the online softmax is register-local, and use the converted
// score tile in place as the B operand of O^T += V^T P^T.  Their first-generation forms (head dimension split over
// waves with a cross-wave reduction of partial scores; register-staged K / V^T with two query tiles per wave) were
// measured against these on the same boxes (history of the round) and removed.
#include "rsvld_common.h"
#include <type_traits>
#include <utility>

namespace {

struct AttnArgs {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int B, heads, Nq, Nk, D;
    int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
    float scale_log2e;
    unsigned long long* dbg;   // diagnostic builds only (-DA6B_STAMP=1): per-wave cycle sums of the loop's phases, else unused
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// compile-time loop: f(std::integral_constant<int, i>) for i = 0..N-1 (inline-asm "i" operands need constants)
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr float A5_DEFER_LOG2 = 8.0f;   // rescale O only when a row's running max grows by more than 2^8

// ---------------------------------------------------------------------------------------
// D = 512, second generation: one wave owns 32 query rows over the WHOLE head dimension.
//
// The first-generation kernel split the head dimension over waves, so every 32-key tile paid a
// cross-wave reduction of partial scores through LDS (36 KB written and re-read), a softmax by
// other threads than the ones that hold the scores, P through LDS, and three barriers for
// 16 MFMAs per wave.  Here (workgroup = 4 waves = 128 query rows, one workgroup per CU,
// 512 registers per wave):
//   * Q (32 rows x 512) stays in 128 VGPRs as the B operand of S^T = K Q^T;
//   * S^T is computed swapped (keys on the accumulator registers, the query on the lane), so the
//     online softmax is register-local (one shuffle with lane^32) and P is cast in place into
//     the B operand of O^T += V^T P^T (cdna_hip_programming.md §3 "An accumulator tile as the
//     next MFMA's operand");
//   * O^T (512 x 32 per wave) lives in 256 accumulator registers;
//   * K and V tiles (32 keys x 512, 32 KB each) are double-buffered in LDS and filled by LDS-DMA
//     only (no staging registers, no transpose pass): K is read row-wise (ds_read_b128), V is
//     read through the hardware transpose ds_read_b64_tr_b16 (T10), both images XOR-swizzled on
//     the DMA source side so that every read is bank-conflict-free;
//   * one barrier per tile, 64 MFMAs per wave between barriers.
// Split-KV: when the query tiles alone cannot fill the chip, blockIdx.y walks key ranges and the
// partial (O / l, m, l) triples are merged by attn_combine_kernel.
// ---------------------------------------------------------------------------------------
// A5B_M0_CLOBBER = 1 (round 3): the d = 512 kernel's LDS-DMA statements write M0 and declare it clobbered instead of saving and
// restoring it around every piece (2 scalar instructions fewer per piece, and the restore no longer waits behind the DMA's own
// read of M0): 1 050 -> 1 077 TFLOP/s at 262 144 keys (+2.5 %), unchanged at 65 536 (profiles/r03_attn_d64_ab.txt, run 7).
// hipcc reserves M0 and only warns; the build is sound because the COMPILER never touches M0 in this translation unit, which
// tools/audit_m0.py / tests/test_build_audits.py check on the assembly.  The d = 64 kernel keeps the save / restore form
// (A6B_M0_CLOBBER = 0): with the clobber form it ran 12 % SLOWER (1 012 -> 894 TFLOP/s), same run.
#ifndef A5B_M0_CLOBBER
#define A5B_M0_CLOBBER 1
#endif
#ifndef A6B_M0_CLOBBER
#define A6B_M0_CLOBBER 0
#endif
// A5B_FINE = 1 (experiment, round 3): the softmax of tile t cut into 32 parts, ONE behind every MFMA of the S(t+1) chain (<= 3
// vector instructions in a 24-cycle MFMA shadow), instead of 8 parts of ~10 instructions behind every fourth MFMA.  Same
// operations in the same order (bit-identical output).  Measured (profiles/r03_attn_d64_ab.txt, run 17): +1.3 % at 65 536 keys,
// -0.3 % at 262 144 (94 % of the Stage-1 attention time): the granularity of the vector work is not what the ablation's 10 %
// are made of.  Off.
#ifndef A5B_FINE
#define A5B_FINE 0
#endif
// A5B_STEADY_LOOP = 1 (round 3): the key-tile loop body exists three times (attention_d512_body.inc): two steady-state copies per
// loop trip for tiles whose successors t + 1, t + 2 exist and are full -- no branch around the S chain, the 8 / 16 LDS-DMA requests
// per tile or the ragged-tile mask, and the two score register sets swap roles instead of being copied -- and the general copy for
// the last tiles.
#ifndef A5B_STEADY_LOOP
#define A5B_STEADY_LOOP 1
#endif
// A5B_KEYWRAP = n > 0 (diagnostic build, round 5; results WRONG, timing only): the steady-state LDS-DMA requests of key tile t read tile
// t mod n (n a power of two), i.e. the whole key stream of a launch cycles over n x 32 KiB -- n = 64: 2 MiB, resident in every XCD's
// 4 MiB L2 -- with the same loop trip count, the same LDS traffic and the same MFMAs.  What the L2-MISS traffic of the real launch
// (70 GB per 262 144-key launch by FETCH_SIZE, served by the Infinity Cache) costs in time = this build against the shipped one
// (tools/attn_d512_l2_ab.sh, profiles/r05_attn_d512_l2_resident_ab.txt).
#ifndef A5B_KEYWRAP
#define A5B_KEYWRAP 0
#endif
#ifndef A5B_ABL
#define A5B_ABL 0   // diagnostic builds (tools/ablate_attn.sh): 1 no softmax VALU, 2 no V reads, 4 no DMA, 8 no K reads, 16 no per-tile barrier
#endif
constexpr int A5B_TILE = 32 * 1024;        // one 32-key x 512 tile, 16-bit
constexpr int A5B_SMEM = 4 * A5B_TILE;     // K[2] | V[2]
// SH ("shared tile") instantiation: K and V are THE SAME tensor X (k == v pointers and strides).  Single-head attention
// whose keys and values are both linear maps of one token matrix can always be brought to this form on the host side:
//   softmax(q (x Wk)^T) (x Wv)  =  softmax((q Wk) x^T) x  Wv,   with q Wk folded into the query projection and Wv into
// the output projection (biases: the key bias shifts every score of a row by the same amount and cancels in the softmax,
// the value bias passes through because the probabilities of a row sum to 1).  One LDS image of X then serves BOTH the
// row-wise K reads and the transposed V reads, so the in-loop LDS-DMA traffic -- the largest single cost of this kernel
// at long sequences (ablation: 965 -> 1340 TFLOP/s without it, profiles/r02_attn_ablation.txt) -- is halved, and three
// 32 KiB buffers (PV of tile t, S of tile t+1, landing tile t+2) replace four.
// Dual-use swizzle: LDS chunk position c of tile row r holds source chunk c ^ f(r), f(r) = ((r & 3) << 2) | ((r >> 2) & 3).
//   * row-wise ds_read_b128 (16 lanes = 16 rows of distinct r mod 16 per LDS cycle): f is a bijection on 4 bits, so the 16
//     lanes hit 16 distinct 16-byte positions of a 256-byte bank period: conflict-free;
//   * ds_read_b64_tr_b16 (32 lanes = 4 rows r = 4 lh + q (+8 hf, +16 s) x 64 contiguous bytes per LDS cycle): bits [3:2] of f
//     are q, so the four rows land in four different 64-byte quarters of the bank period: conflict-free; bits [1:0] of f
//     ((lh + 2 hf) & 3) only permute chunks inside a row's own quarter.
constexpr int A5B_SMEM_SH = 3 * A5B_TILE;  // X[3]
__host__ __device__ constexpr int a5b_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
// A5B_SWAPMAX = 1 (experiment, round 3): the maximum over lane and lane ^ 32 by v_permlane32_swap instead of ds_bpermute (what
// __shfl_xor compiles to; its result returns through lgkmcnt, and the wait for it also drains the K-fragment reads in flight in the
// middle of the S chain).  Measured (run 17, 262 144 keys): 1 099 TFLOP/s with ds_bpermute, 1 054 with the swap (-4 %; with the
// 32-part softmax 1 020): on one wave per SIMD the swap's own latency and hazard padding cost more than the drained reads.  Off.
#ifndef A5B_SWAPMAX
#define A5B_SWAPMAX 0
#endif
__device__ __forceinline__ float a5b_halfwave_max(float mx) {
#if A5B_SWAPMAX
    const uint32_t mb = __builtin_bit_cast(uint32_t, mx);
    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
#else
    return fmaxf(mx, __shfl_xor(mx, 32));
#endif
}

template <typename T, bool SH>
__global__ __launch_bounds__(256) void attn_d512b_kernel(AttnArgs p, int keys_per_split, float* part_o, float* part_ml) {
    constexpr int D = 512;
    typedef typename Mfma<T>::v8 v8;
    typedef typename Mfma<T>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 128 + w * 32;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int b = blockIdx.z / p.heads, h = blockIdx.z % p.heads;
    const int k_begin = split * keys_per_split;
    const int k_end = min(p.Nk, k_begin + keys_per_split);
    const int nt = (k_end - k_begin + 31) >> 5;
    const T* Qb = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const T* Kb = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D;
    const T* Vb = (const T*)p.v + (int64_t)b * p.v_bs + (int64_t)h * D;

    // ---- tile DMA: wave w moves key rows 8w .. 8w+7 of the tile, one 1-KiB row per wave-instruction.
    // LDS chunk position `lane` of row r holds source chunk lane ^ (r & 15) (K) / lane ^ ((r & 3) << 2) (V).
    // Row addresses are wave-uniform (SGPR base), the per-lane part is a 32-bit byte offset.  The DMAs are inline asm:
    // (1) an LDS-DMA costs the issuing wave ~60-180 cycles, so the 16 of a tile are spread between the MFMAs of the
    // S chain instead of standing in front of it; (2) hipcc puts s_waitcnt vmcnt(0) in front of the first LDS read that
    // follows an LDS-DMA it knows of, which would drain the prefetch at the start of every PV phase.  M0 (the LDS
    // destination) is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5.7).
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    bool abl_dma_on = true;   // ablation 4: the prologue fills BOTH buffers of K and V, the loop issues no DMA (operands stay
                              // real data: zero-filled operands would raise the clock and overstate the saving)
    auto dma_one = [&](const char* base, uint32_t voff, uint32_t dst) {
        if ((A5B_ABL & 4) && !abl_dma_on) return;
#if A5B_M0_CLOBBER   // M0 declared clobbered instead of saved / restored around every piece (2 SALU fewer per piece)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(dst), "s"(base) : "memory", "m0");
#else
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(dst), "s"(base)
                     : "memory");
#endif
    };
    // Full tiles: ONE scalar base per tensor (row 8w of the tile, advanced by 32 rows per tile with two SALU adds);
    // the row i of the piece and its source-side swizzle sit in a precomputed 32-bit lane offset.  The tail tile (rows
    // past Nk re-read the last key and are masked in the softmax) computes clamped row addresses the long way.
    const int64_t k_rowb = p.k_ts * (int64_t)sizeof(T), v_rowb = p.v_ts * (int64_t)sizeof(T);
    uint32_t kvo[8], vvo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        kvo[i] = (uint32_t)(i * k_rowb) + (uint32_t)((lane ^ (SH ? a5b_f((wu * 8 + i) & 15) : ((wu * 8 + i) & 15))) << 4);
        vvo[i] = (uint32_t)(i * v_rowb) + (uint32_t)((lane ^ ((i & 3) << 2)) << 4);
    }
    const char* k_tile0 = (const char*)(Kb + (int64_t)(k_begin + wu * 8) * p.k_ts);   // + t * 32 rows
    const char* v_tile0 = (const char*)(Vb + (int64_t)(k_begin + wu * 8) * p.v_ts);
    auto dma_kb = [&](int t, int buf, int i) {   // key row 8w+i of tile t -> tile buffer `buf`
        const int r = wu * 8 + i;
        const uint32_t dst = lds0 + buf * A5B_TILE + r * 1024;
        if (k_begin + t * 32 + 32 <= p.Nk) {
            dma_one(k_tile0 + (int64_t)t * 32 * k_rowb, kvo[i], dst);
        } else {
            const int key = min(k_begin + t * 32 + r, p.Nk - 1);
            dma_one((const char*)(Kb + (int64_t)key * p.k_ts), (uint32_t)((lane ^ (SH ? a5b_f(r & 15) : (r & 15))) << 4), dst);
        }
    };
    auto dma_k = [&](int t, int i) { dma_kb(t, t & 1, i); };   // K buffer t&1 (two-tensor form)
    auto dma_v = [&](int t, int i) {   // same row of V -> V buffer t&1
        const int r = wu * 8 + i;
        const uint32_t dst = lds0 + (2 + (t & 1)) * A5B_TILE + r * 1024;
        if (k_begin + t * 32 + 32 <= p.Nk) {
            dma_one(v_tile0 + (int64_t)t * 32 * v_rowb, vvo[i], dst);
        } else {
            const int key = min(k_begin + t * 32 + r, p.Nk - 1);
            dma_one((const char*)(Vb + (int64_t)key * p.v_ts), (uint32_t)((lane ^ ((i & 3) << 2)) << 4), dst);
        }
    };
    // the same requests for tiles known to be full (the steady-state copy of the loop body): no bounds test, no branch
    auto tw = [](int t) { return A5B_KEYWRAP > 0 ? (t & (A5B_KEYWRAP - 1)) : t; };   // (diagnostic build: see A5B_KEYWRAP)
    auto dma_kb_fast = [&](int t, int buf, int i) { dma_one(k_tile0 + (int64_t)tw(t) * 32 * k_rowb, kvo[i], lds0 + buf * A5B_TILE + (wu * 8 + i) * 1024); };
    auto dma_k_fast = [&](int t, int i) { dma_kb_fast(t, t & 1, i); };
    auto dma_v_fast = [&](int t, int i) { dma_one(v_tile0 + (int64_t)tw(t) * 32 * v_rowb, vvo[i], lds0 + (2 + (t & 1)) * A5B_TILE + (wu * 8 + i) * 1024); };
    // Software pipeline: iteration t runs S(t+1) beside softmax(t), then PV(t).  K is therefore fetched two tiles ahead
    // of its PV (K(t+2) lands in the buffer S(t) read in iteration t-1), V one tile ahead.
    if constexpr (SH) {   // X(0) -> buffer 0, X(1) -> buffer 1
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_kb(0, 0, i);
        if (nt > 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dma_kb(1, 1, i);
        }
        if (A5B_ABL & 4) {
            if (nt > 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dma_kb(2, 2, i);
            }
            abl_dma_on = false;
        }
    } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) { dma_k(0, i); dma_v(0, i); }
    if (nt > 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_k(1, i);
    }
    }
    if (!SH && (A5B_ABL & 4)) {
        if (nt > 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dma_v(1, i);
        }
        abl_dma_on = false;
    }

    // ---- Q fragments (B operand: col = query row on the lane, k = d)
    const int qrow = q0 + l31;
    v8 qf[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (qrow < p.Nq) v = *(const u32x4*)(Qb + (int64_t)qrow * p.q_ts + ks * 16 + lh * 8);
        qf[ks] = __builtin_bit_cast(v8, v);
    }
    f32x16 oacc[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- per-lane read offsets.  K: row l31, chunk 2ks+lh = 16a + (2c+lh): off = kbase[c] + 256 a.
    int kbase[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kbase[c] = l31 * 1024 + (((2 * c + lh) ^ (SH ? a5b_f(l31 & 15) : (l31 & 15))) << 4);
    // V (transposed read): lane 16g + 4q + p supplies row (4lh + q) + {16s + 8hf}, d = 32dt + 16(g&1) + 4p .. +3;
    // chunk = 4dt + 2(g&1) + (p>>1), swizzled by q<<2: with dt = 4e + f  ->  off = vbase[f] + 256 e + 1024 (16s + 8hf)
    int vbase[4];
    {
        const int qq = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int f = 0; f < 4; ++f)
            vbase[f] = SH ? (4 * lh + qq) * 1024 + ((f ^ qq) << 6) + (((2 * g1 + (pp >> 1)) ^ lh) << 4) + ((pp & 1) << 3)   // f(r): low bits lh (+2 hf)
                          : 2 * A5B_TILE + (4 * lh + qq) * 1024 + ((f ^ qq) << 6) + ((2 * g1 + (pp >> 1)) << 4) + ((pp & 1) << 3);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tiles 0 (K, V) and 1 (K) landed

    // ---- S^T[key][q] over the whole head dimension: one accumulation chain of 32 MFMAs per tile.
    // The chain is issued in its VGPR form by inline asm: all 256 accumulator registers belong to O, and hipcc
    // otherwise parks the score tile in a0..a15 and shuttles one O tile through VGPRs every iteration.  Back-to-back
    // MFMAs that take the previous D whole as C need no wait states.  The K fragment reads are asm as well: hipcc
    // waits lgkmcnt(0) in front of an asm consumer, which stalls the chain on the read it has just issued; here KD
    // reads are in flight and MFMA ks waits only for its own (lgkmcnt(min(KD-1, 31-ks)); LDS returns in order, and any
    // SMEM operation the compiler may have in flight only makes the count more conservative).
    // HOOK(i), i = 0..7, runs after MFMAs 3, 7, .. 31: in the steady state it carries two LDS-DMA rows and one eighth
    // of the PREVIOUS tile's softmax, pinned there with sched_barrier, so the VALU work issues beside the MFMAs.
    constexpr int KD = 6;
    static_assert(KD == 6, "A5B_CHAIN's prologue issues 6 reads");
    uint32_t ka[8];
    v8 kfr[KD];
    // (macros, not lambdas: clang rejects captured arrays as asm operands inside generic lambdas)
#define A5B_KREAD(slot, ks) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kfr[slot]) : "v"(ka[(ks) & 7]), "i"(((ks) >> 3) * 256))
#define A5B_STEP(NAME, SACC, ks)                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(%3)\n\t" NAME " %0, %1, %2, %0"                                                      \
                 : "+v"(SACC)                                                                                           \
                 : "v"(kfr[(ks) % KD]), "v"(qf[ks]), "i"(KD - 1 < 31 - (ks) ? KD - 1 : 31 - (ks)));                      \
    if constexpr ((ks) + KD < 32 && !(A5B_ABL & 8)) A5B_KREAD((ks) % KD, (ks) + KD)
#define A5B_STEP4(NAME, SACC, k) \
    A5B_STEP(NAME, SACC, k); A5B_STEP(NAME, SACC, (k) + 1); A5B_STEP(NAME, SACC, (k) + 2); A5B_STEP(NAME, SACC, (k) + 3)
#if A5B_FINE   // HOOK(k), k = 0..31, behind MFMA k
#define A5B_STEPH(NAME, SACC, k, HOOK) A5B_STEP(NAME, SACC, k); HOOK(k)
#define A5B_STEPH4(NAME, SACC, k, HOOK) \
    A5B_STEPH(NAME, SACC, k, HOOK); A5B_STEPH(NAME, SACC, (k) + 1, HOOK); A5B_STEPH(NAME, SACC, (k) + 2, HOOK); A5B_STEPH(NAME, SACC, (k) + 3, HOOK)
#define A5B_CHAIN(NAME, SACC, KBUF, HOOK)                                                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) ka[c] = lds0 + (uint32_t)((KBUF) * A5B_TILE + kbase[c]);              \
    A5B_KREAD(0, 0); A5B_KREAD(1, 1); A5B_KREAD(2, 2); A5B_KREAD(3, 3); A5B_KREAD(4, 4); A5B_KREAD(5, 5);               \
    asm volatile("s_waitcnt lgkmcnt(%3)\n\t" NAME " %0, %1, %2, 0" : "=&v"(SACC) : "v"(kfr[0]), "v"(qf[0]), "i"(KD - 1)); \
    A5B_KREAD(0, KD); HOOK(0);                                                                                          \
    A5B_STEPH(NAME, SACC, 1, HOOK); A5B_STEPH(NAME, SACC, 2, HOOK); A5B_STEPH(NAME, SACC, 3, HOOK);                     \
    A5B_STEPH4(NAME, SACC, 4, HOOK); A5B_STEPH4(NAME, SACC, 8, HOOK); A5B_STEPH4(NAME, SACC, 12, HOOK);                 \
    A5B_STEPH4(NAME, SACC, 16, HOOK); A5B_STEPH4(NAME, SACC, 20, HOOK); A5B_STEPH4(NAME, SACC, 24, HOOK);               \
    A5B_STEPH4(NAME, SACC, 28, HOOK)
#else
#define A5B_CHAIN(NAME, SACC, KBUF, HOOK)                                                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) ka[c] = lds0 + (uint32_t)((KBUF) * A5B_TILE + kbase[c]);              \
    A5B_KREAD(0, 0); A5B_KREAD(1, 1); A5B_KREAD(2, 2); A5B_KREAD(3, 3); A5B_KREAD(4, 4); A5B_KREAD(5, 5);               \
    asm volatile("s_waitcnt lgkmcnt(%3)\n\t" NAME " %0, %1, %2, 0" : "=&v"(SACC) : "v"(kfr[0]), "v"(qf[0]), "i"(KD - 1)); \
    A5B_KREAD(0, KD);                                                                                                   \
    A5B_STEP(NAME, SACC, 1); A5B_STEP(NAME, SACC, 2); A5B_STEP(NAME, SACC, 3); HOOK(0);                                 \
    A5B_STEP4(NAME, SACC, 4); HOOK(1); A5B_STEP4(NAME, SACC, 8); HOOK(2); A5B_STEP4(NAME, SACC, 12); HOOK(3);           \
    A5B_STEP4(NAME, SACC, 16); HOOK(4); A5B_STEP4(NAME, SACC, 20); HOOK(5); A5B_STEP4(NAME, SACC, 24); HOOK(6);         \
    A5B_STEP4(NAME, SACC, 28); HOOK(7)
#endif
#define A5B_MFMA_NAME(T_) (__is_same(T_, f16) ? "f16" : "bf16")
#define A5B_NOHOOK(i)

    f32x16 sacc;   // scores of the tile whose softmax is due (S runs one tile ahead of PV)
    if constexpr (__is_same(T, f16)) {
        A5B_CHAIN("v_mfma_f32_32x32x16_f16", sacc, 0, A5B_NOHOOK);
    } else {
        A5B_CHAIN("v_mfma_f32_32x32x16_bf16", sacc, 0, A5B_NOHOOK);
    }
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc));   // MFMA D -> VALU reader (cdna_hip_programming.md §5.7 item 2)
    __syncthreads();   // every wave has read K(0): iteration 0 overwrites that buffer with K(2)

    int b_pv = 0;   // SH: ring position of tile t (t % 3)
    int t = 0;
#if A5B_STEADY_LOOP
    f32x16 sacc2;
#define A5B_STEADY 1
#define A5B_PAIR 1
#define dma_kb_s dma_kb_fast
#define dma_k_s dma_k_fast
#define dma_v_s dma_v_fast
    for (; t + 4 < nt;) {   // two tiles per trip: tile t scores in sacc, tile t + 1 in sacc2, tile t + 2 in sacc again
        {
#define S_CUR sacc
#define S_NXT sacc2
#include "attention_d512_body.inc"
#undef S_CUR
#undef S_NXT
        }
        ++t;
        {
#define S_CUR sacc2
#define S_NXT sacc
#include "attention_d512_body.inc"
#undef S_CUR
#undef S_NXT
        }
        ++t;
    }
#undef dma_kb_s
#undef dma_k_s
#undef dma_v_s
#undef A5B_PAIR
#undef A5B_STEADY
#endif
    for (; t < nt; ++t) {
#define A5B_PAIR 0
#define S_CUR sacc
#define S_NXT snext
#define A5B_STEADY 0
#define dma_kb_s dma_kb
#define dma_k_s dma_k
#define dma_v_s dma_v
#include "attention_d512_body.inc"
#undef dma_kb_s
#undef dma_k_s
#undef dma_v_s
#undef A5B_STEADY
#undef S_CUR
#undef S_NXT
#undef A5B_PAIR
    }
#undef A5B_NOHOOK
#undef A5B_MFMA_NAME
#undef A5B_CHAIN
#undef A5B_STEP4
#if A5B_FINE
#undef A5B_STEPH4
#undef A5B_STEPH
#endif
#undef A5B_STEP
#undef A5B_KREAD

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (qrow >= p.Nq) return;
    const float inv = 1.0f / l_tot;
    if (nsplit == 1) {
        T* Ob = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)h * D + (int64_t)qrow * p.o_ts;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)(oacc[dt][4 * g + e] * inv);
                *(v4*)(Ob + dt * 32 + 8 * g + 4 * lh) = o;
            }
    } else {
        const int64_t row = ((int64_t)split * gridDim.z + blockIdx.z) * p.Nq + qrow;
        float* Po = part_o + row * D;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = oacc[dt][4 * g + e] * inv;
                *(f32x4*)(Po + dt * 32 + 8 * g + 4 * lh) = o;
            }
        if (lh == 0) {
            part_ml[row * 2] = m_run;
            part_ml[row * 2 + 1] = l_tot;
        }
    }
}

// merge of split-KV partials: out = sum_i w_i O_i / sum_i w_i,  w_i = l_i 2^(m_i - max m)
template <typename T>
__global__ __launch_bounds__(128) void attn_combine_kernel(AttnArgs p, int nsplit, const float* part_o, const float* part_ml) {
    constexpr int D = 512;
    typedef typename Mfma<T>::v4 v4;
    const int qrow = blockIdx.x, bh = blockIdx.y;
    const int b = bh / p.heads, h = bh % p.heads;
    const int64_t nrows = (int64_t)gridDim.y * p.Nq;
    float mmax = -INFINITY;
    for (int i = 0; i < nsplit; ++i) mmax = fmaxf(mmax, part_ml[((int64_t)i * nrows + (int64_t)bh * p.Nq + qrow) * 2]);
    float wsum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nsplit; ++i) {
        const int64_t row = (int64_t)i * nrows + (int64_t)bh * p.Nq + qrow;
        const float wi = part_ml[row * 2 + 1] * __builtin_amdgcn_exp2f(part_ml[row * 2] - mmax);
        const f32x4 o = *(const f32x4*)(part_o + row * D + threadIdx.x * 4);
        acc += o * wi;
        wsum += wi;
    }
    const float inv = 1.0f / wsum;
    v4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (T)(acc[e] * inv);
    *(v4*)((T*)p.out + (int64_t)b * p.o_bs + (int64_t)h * D + (int64_t)qrow * p.o_ts + threadIdx.x * 4) = o;
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------
// D = 64, second generation.  The first-generation kernel was VALU-bound: per 64-key tile a wave issued 16 MFMAs
// (512 cycles) beside ~270 vector instructions, of which 32 moved the score tile out of the accumulator file,
// 64 were scale + subtract, ~45 address arithmetic of register-staged K / V^T loads, and its two-query-tile form
// needed 354 registers (one wave per SIMD: nothing overlapped the softmax).  Here:
//   * MFMAs are issued in their VGPR form (inline asm), so scores and O are plain VGPRs: no v_accvgpr traffic;
//   * exp2(scale * s - m) is ONE fma + exp2 per element (the max is taken on raw scores, scale > 0);
//   * K and V tiles are filled by LDS-DMA (asm, one scalar base per tensor per tile, no staging registers), V is
//     read through ds_read_b64_tr_b16, so no transpose pass exists;
//   * 4 waves x 32 query rows, 128 registers -> four workgroups per CU: the other waves' softmax runs beside one
//     wave's MFMAs on every SIMD.
// ---------------------------------------------------------------------------------------
constexpr int A6B_TILE = 64 * 128;            // 64 keys x 64 d, 16-bit
#ifndef A6B_ABL
#define A6B_ABL 0   // diagnostic builds (tools/ablate_attn.sh; results WRONG, timing only): 1 no exp, 2 no in-loop DMA / barrier,
                    // 8 no running max / row sum, 16 no V fragment reads, 32 no K fragment reads
#endif
// K[NB] | V[NB] 64-key sub-tiles, NB = 2 KPB.  KPB = sub-tiles per barrier: 1 (double buffer, one barrier per 64 keys) or, for
// the 8-wave workgroups (two per CU: 2 x 64 KiB of LDS), 2 (four buffers, two sub-tiles requested and one barrier per 128 keys:
// the in-loop wait + barrier was worth 9-13 % in the ablation of round 2, profiles/r02_attn_ablation.txt).
#ifndef A6B_SUMCHK
#define A6B_SUMCHK 1   // sum-checked softmax (see the loop): the row maximum is taken only on tiles whose row sum says the bias was stale
#endif
#ifndef A6B_KPB8
#define A6B_KPB8 1   // measured (one box): 2 sub-tiles per barrier 957 / 1006 TFLOP/s vs 958 / 1015 with 1: no gain, kept at 1
#endif
#ifndef A6B_NW16_MIN
#define A6B_NW16_MIN (1 << 30)   // experiment: 16-wave workgroups (512 query rows per K / V tile, one LDS-DMA piece per wave and tile)
#endif
// -DA6B_STAMP=1: diagnostic build (cdna_hip_programming.md section 7, In-kernel stamps).  s_memtime stamps split every loop
// iteration of the d = 64 kernel into  request | K reads + S chain | softmax | V reads + PV chain | wait + barrier;  the sums
// go to the workspace pointer (memory nothing else reads).  Read the SHARES, not the run time: the stamps' waits forbid overlaps.
#ifndef A6B_STAMP
#define A6B_STAMP 0
#endif
__device__ __forceinline__ unsigned long long a6b_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// A6B_OCC4 = 1 (default): 128 registers, FOUR workgroups per CU.  The K fragments are read one 32-key half at a time and
// the V fragments after the softmax instead of behind it, so that no 32-register fragment set is live across another
// phase; the exposed LDS latency is covered by the fourth wave of the SIMD (measured +2.3-2.9 % at 16k-32k tokens against
// the 153-register, three-workgroup form, which -DA6B_OCC4=0 still builds for A/B runs).  Reading the V fragments (or half
// of them) behind the exponentials at four waves per SIMD spills.
#ifndef A6B_OCC4
#define A6B_OCC4 1
#endif
// A6B_BIAS = 1 (default, round 3): the kernel is vector-issue bound (10.8 VALU per MFMA: profiles/r02_attn_pmc.json), so the
// per-score ``fma(s, scale, -m)`` in front of every exponential moves to the matrix pipe:
//   * Q is multiplied by scale * log2(e) once, when its fragments are loaded (one extra 16-bit rounding of Q);
//   * the running maximum enters S as a rank-1 term: one extra 16-deep MFMA step per 32-key half whose A operand is a ones
//     column (k-slot 0) and whose B operand holds -m of the lane's query in that slot, so the accumulator comes out as
//     s' = scale * log2(e) * q.k - m and the softmax is a bare exp2.  m is therefore kept representable in the operand
//     type (it only has to be the SAME number in P, in the row sum and in the rescale of O, not the exact maximum);
//   * the maximum the bias carries is the one known BEFORE the tile (deferred maximum, T13): only when a row's scores exceed
//     it by more than 2^8 (and on the first tile) the tile pays a subtract per score and O / l are rescaled.
// Per 64-key tile and wave: -32 v_fma, +2 MFMAs, +8 moves.  The half-wave exchange of the row maximum is a
// v_permlane32_swap instead of a ds_bpermute round trip.  -DA6B_BIAS=0 builds the round-2 form for A/B runs.
#ifndef A6B_BIAS
#define A6B_BIAS 1
#endif
// With the fma gone the exponentials' results are adjacent, and plain -O3 SLP-packs the row-sum adds into v_pk_add_f32, which
// issues in two passes beside MFMAs (MI355X_MICROARCH.md "price of one filler"; measured here: 841 vs 880 TFLOP/s at 16 384
// tokens).  The file is therefore compiled with -fno-slp-vectorize (csrc/Makefile).  An inline-asm v_add_f32 helper is NOT the
// way: hipcc pads no hazard for an asm statement, and a v_add that reads a v_exp_f32 result one instruction later (the
// "trans use" hazard of gfx940+) returned stale values -- every d = 64 test failed with errors of order 1.
__device__ __forceinline__ float a6b_add(float a, float b) { return a + b; }

// O of one wave's 32 query rows (d = 64): normalised, through a wave-private 4 KiB LDS image (16-byte chunks XOR-swizzled by the
// row) to whole 128-byte rows -- 16 bytes per lane, eight lanes per row -- instead of 8 bytes per lane at a row stride (32 rows x
// 2 pieces per store instruction).  LDS operations of one wave complete in order: no barrier.  A6_STAGE_O = 0: the direct stores.
#ifndef A6_STAGE_O
#define A6_STAGE_O 1
#endif
template <typename T>
__device__ __forceinline__ void a6_store_o(char* wbuf, const f32x16 (&oacc)[2], float inv, int q0, int lane, const AttnArgs& p, int b, int h) {
    typedef typename Mfma<T>::v4 v4;
    const int l31 = lane & 31, lh = lane >> 5;
    T* Ob = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)h * 64;
    if (!A6_STAGE_O) {
        if (q0 + l31 >= p.Nq) return;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)(oacc[dt][4 * g + e] * inv);
                *(v4*)(Ob + (int64_t)(q0 + l31) * p.o_ts + dt * 32 + 8 * g + 4 * lh) = o;
            }
        return;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)(oacc[dt][4 * g + e] * inv);
            *(v4*)(wbuf + l31 * 128 + (((dt * 4 + g) ^ (l31 & 7)) << 4) + lh * 8) = o;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (lane >> 3) + 8 * j, chunk = lane & 7;
        const u32x4 v = *(const u32x4*)(wbuf + row * 128 + ((chunk ^ (row & 7)) << 4));
        if (q0 + row < p.Nq) *(u32x4*)(Ob + (int64_t)(q0 + row) * p.o_ts + chunk * 8) = v;
    }
}

// NW = waves per workgroup: 4 (128 query rows) or 8 (256 query rows).  The 64-key K / V tile is filled once per workgroup,
// so with 8 waves every wave issues ONE LDS-DMA piece per tensor per tile instead of two and there is one barrier per 256
// query rows: the in-loop DMA + barrier cost (ablation: +13 % without it at 65 536 tokens) is halved per MFMA.  Two
// 8-wave workgroups per CU keep the four waves per SIMD.  Chosen for long query sequences (A6B_NW8_MIN); the gain is small: the kernel is not bound by that term alone.
#ifndef A6B_NW8_MIN
#define A6B_NW8_MIN 16384   // round 2 (one box, 2 x 10 heads): +1.5 % at 65 536 tokens, -0.5 % (noise) at 16 384; with the bias step
                            // (round 3, 2 x 20 heads at 16 384 tokens): +1.3 % (880 -> 891 TFLOP/s), so 16 384 tokens take it too
#endif
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW, A6B_OCC4 ? 4 : 2) void attn_d64b_kernel(AttnArgs p) {
    constexpr int D = 64;
    constexpr int KPB = NW == 8 ? A6B_KPB8 : 1;   // sub-tiles per barrier
    constexpr int NB = 2 * KPB;                   // sub-tile buffers per tensor
    constexpr bool SPLIT = NW == 16;            // 16 waves: waves 0..7 move the K pieces, waves 8..15 the V pieces, ONE piece each
    constexpr int RPW = SPLIT ? 8 : 64 / NW;    // tile rows each wave moves: 16 or 8
    constexpr int NP = RPW / 8;                 // LDS-DMA pieces (8 rows x 128 B) per tensor per wave: 2 or 1
    typedef typename Mfma<T>::v8 v8;
    typedef typename Mfma<T>::v4 v4;
    __shared__ __attribute__((aligned(16))) char smem[2 * NB * A6B_TILE];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = (blockIdx.x * NW + w) * 32, h = blockIdx.y, b = blockIdx.z;
    const T* Qb = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const T* Kb = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D;
    const T* Vb = (const T*)p.v + (int64_t)b * p.v_bs + (int64_t)h * D;
    const int nt = (p.Nk + 63) >> 6;

    // ---- tile DMA.  Wave w moves key rows RPW w .. RPW w + RPW-1 of the tile: NP wave-instructions of 8 rows x 128 B per
    // tensor.  Lane (row = lane>>3, pos = lane&7) of piece i fills LDS chunk `pos` of tile row r = RPW w + 8i + row with
    // source chunk pos ^ ((r'>>1)&7) for K (r' = r & 15) and pos ^ (((r'>>1)&1)<<2) for V.
    const int wave_u = __builtin_amdgcn_readfirstlane(w);
    const int wu = SPLIT ? (wave_u & 7) : wave_u;          // row group of the tile this wave moves
    const bool moves_k = !SPLIT || wave_u < 8, moves_v = !SPLIT || wave_u >= 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int64_t k_rowb = p.k_ts * (int64_t)sizeof(T), v_rowb = p.v_ts * (int64_t)sizeof(T);
    uint32_t kvo[NP], vvo[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int rr = 8 * i + (lane >> 3), r16 = (wu * RPW + rr) & 15;
        kvo[i] = (uint32_t)(rr * k_rowb) + (uint32_t)(((lane & 7) ^ ((r16 >> 1) & 7)) << 4);
        vvo[i] = (uint32_t)(rr * v_rowb) + (uint32_t)(((lane & 7) ^ (((r16 >> 1) & 1) << 2)) << 4);
    }
    auto dma_fast = [&](const char* base, uint32_t voff, uint32_t dst) {
#if A6B_M0_CLOBBER   // M0 declared clobbered instead of saved / restored (measured 12 % slower here: off, see A5B_M0_CLOBBER)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(dst), "s"(base) : "memory", "m0");
#else
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
#endif
    };
    auto dma_slow = [&](const char* ptr, uint32_t dst) {   // per-lane 64-bit address (tail tile: clamped rows)
#if A6B_M0_CLOBBER
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(ptr), "s"(dst) : "memory", "m0");
#else
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(ptr), "s"(dst) : "memory");
#endif
    };
    // Full tiles are requested in order t = 0, 1, 2, ...: the two scalar row pointers advance by 64 rows per request (two
    // 64-bit adds) instead of being recomputed from t with 64-bit multiplies (~20 SALU per request and wave, on a CU whose 16
    // waves share one scalar unit: the request phase was 15 % of a wave's iteration in the stamped build).
    const char* k_next = (const char*)(Kb + (int64_t)(wu * RPW) * p.k_ts);
    const char* v_next = (const char*)(Vb + (int64_t)(wu * RPW) * p.v_ts);
    const int64_t k_step = 64 * k_rowb, v_step = 64 * v_rowb;
    auto dma_tile = [&](int t) {
        const int buf = t & (NB - 1);
        const uint32_t kd = lds0 + buf * A6B_TILE + wu * (RPW * 128), vd = kd + NB * A6B_TILE;
        if (t * 64 + 64 <= p.Nk) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (moves_k) dma_fast(k_next, kvo[i], kd + i * 1024);
                if (moves_v) dma_fast(v_next, vvo[i], vd + i * 1024);
            }
            k_next += k_step;
            v_next += v_step;
        } else {   // rows past Nk re-read the last key; their scores are masked
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int rr = 8 * i + (lane >> 3), r16 = (wu * RPW + rr) & 15;
                const int key = min(t * 64 + wu * RPW + rr, p.Nk - 1);
                if (moves_k) dma_slow((const char*)(Kb + (int64_t)key * p.k_ts) + (((lane & 7) ^ ((r16 >> 1) & 7)) << 4), kd + i * 1024);
                if (moves_v) dma_slow((const char*)(Vb + (int64_t)key * p.v_ts) + (((lane & 7) ^ (((r16 >> 1) & 1) << 2)) << 4), vd + i * 1024);
            }
        }
    };
    dma_tile(0);
    if (KPB == 2 && nt > 1) dma_tile(1);

    // ---- Q fragments (B operand: col = query row on the lane, k = d)
    const int qrow = q0 + l31;
    v8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (qrow < p.Nq) v = *(const u32x4*)(Qb + (int64_t)qrow * p.q_ts + ks * 16 + lh * 8);
        qf[ks] = __builtin_bit_cast(v8, v);
        if (A6B_BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (T)((float)qf[ks][e] * p.scale_log2e);
        }
    }
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = A6B_BIAS ? 0.f : -INFINITY, l_run = 0.f;   // A6B_BIAS: the maximum the bias step subtracts (0 before the first tile)
    uint32_t one_lo;   // A operand of the bias step, register 0: 1.0 in element 0 of lanes 0..31 (k-slot 0), zero elsewhere
    {
        T pair[2] = {lh == 0 ? (T)1.f : (T)0.f, (T)0.f};
        __builtin_memcpy(&one_lo, pair, 4);
    }

    // per-lane read offsets.  K (ds_read_b128): row kt*32 + l31, chunk 2ks + lh, swizzled by ((row>>1)&7) [row & 15 = l31 & 15]
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4);
    // V (transposed read): lane 16g + 4q + p supplies row 16 s4 + 8 hf + 4 lh + q, d = 32 dt + 16 (g&1) + 4p .. +3
    int voff[2];
    {
        const int qq = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
        const int row = 4 * lh + qq;   // + 16 s4 + 8 hf: multiples of 8, (row>>1)&1 unchanged
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            voff[dt] = NB * A6B_TILE + row * 128 + (((4 * dt + 2 * g1 + (pp >> 1)) ^ (((row >> 1) & 1) << 2)) << 4) + ((pp & 1) << 3);
    }

    // The Q fragments are consumed here, before the loop: hipcc then puts its own vmcnt wait for their loads here and
    // not in front of the loop's first MFMA, where it would also wait for the tile requested at the top of every iteration.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile 0 landed

    if (A6B_ABL & 2) {   // both buffers hold real tiles; the loop then neither requests nor waits
        if (nt > 1 && KPB == 1) dma_tile(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    v8 abl_frag = *(const v8*)(smem + koff[0]);   // diagnostic builds only (A6B_ABL & 48): one constant fragment instead of LDS reads
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_t = 0;
    if (A6B_STAMP) st_t = a6b_stamp();
#define A6B_MARK(i) do { if (A6B_STAMP) { const unsigned long long now_ = a6b_stamp(); st_sum[i] += now_ - st_t; st_t = now_; } } while (0)
    // Outer loop = one barrier period (KPB sub-tiles), inner loop = its sub-tiles.  (A single loop with the barrier under
    // ``if (t & 1)`` made hipcc spill 26 registers into the loop, unrolled by the period or not.)
    for (int t0 = 0; t0 < nt; t0 += KPB) {
    const int t1 = min(t0 + KPB, nt);
#pragma nounroll
    for (int t = t0; t < t1; ++t) {
        const int buf = t & (NB - 1);
        if (A6B_ABL & 48) asm volatile("" : "+v"(abl_frag));
        // sub-tile t + KPB goes to the buffer sub-tile t - KPB was read from, i.e. before the last barrier (barriers close every
        // KPB-th sub-tile); it is read KPB sub-tiles from now, behind the next vmcnt(0) + barrier
        if (!(A6B_ABL & 2) && t + KPB < nt) dma_tile(t + KPB);
        const char* Ks = smem + buf * A6B_TILE;
        const int vb = buf * A6B_TILE;
        A6B_MARK(0);

        // ---- S^T[key][q]: two 32-key halves x 4 k-steps, VGPR-form MFMAs (a lambda: the sum-checked softmax below runs it a
        // second time on the rare tile whose scores outgrew the bias)
        f32x16 sacc[2];
        auto s_phase = [&]() {
        v8 kf[2][4];
        if (!A6B_OCC4) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *(const v8*)(Ks + kt * 4096 + koff[ks]);
        }
        // bias step operands (transient: built per tile so that no register is held across the PV phase, where the kernel
        // sits at its 128-register limit).  A: 1 in k-slot 0 (lanes 0..31, element 0), B: -m of the lane's query there.
        v8 onesf, biasf;
        if (A6B_BIAS) {
            uint32_t zr;                                  // produced HERE by an asm move: a C++ zero would be hoisted out of
            asm volatile("v_mov_b32 %0, 0" : "=v"(zr));   // the loop and held in registers for the whole kernel
            T pair[2] = {(T)(-m_run), (T)0.f};            // exact: m_run is kept representable in T
            uint32_t b0;
            __builtin_memcpy(&b0, pair, 4);
            const u32x4 av = {one_lo, zr, zr, zr}, bv = {b0, zr, zr, zr};
            onesf = __builtin_bit_cast(v8, av);
            biasf = __builtin_bit_cast(v8, bv);
        }
        asm volatile("s_setprio 1");   // the wave that feeds the matrix pipe issues ahead of its SIMD's softmax waves (+0.6 %)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (A6B_OCC4) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = (A6B_ABL & 32) ? abl_frag : *(const v8*)(Ks + kt * 4096 + koff[ks]);
            }
            if constexpr (A6B_BIAS != 0) {
                if constexpr (__is_same(T, f16)) {
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(sacc[kt]) : "v"(onesf), "v"(biasf));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][0]), "v"(qf[0]));
                } else {
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sacc[kt]) : "v"(onesf), "v"(biasf));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][0]), "v"(qf[0]));
                }
                if constexpr (__is_same(T, f16)) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][1]), "v"(qf[1]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][2]), "v"(qf[2]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][3]), "v"(qf[3]));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][1]), "v"(qf[1]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][2]), "v"(qf[2]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][3]), "v"(qf[3]));
                }
            } else if constexpr (__is_same(T, f16)) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(sacc[kt]) : "v"(kf[kt][0]), "v"(qf[0]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][1]), "v"(qf[1]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][2]), "v"(qf[2]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][3]), "v"(qf[3]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sacc[kt]) : "v"(kf[kt][0]), "v"(qf[0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][1]), "v"(qf[1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][2]), "v"(qf[2]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kt]) : "v"(kf[kt][3]), "v"(qf[3]));
            }
        }
        asm volatile("s_setprio 0");
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0]), "+v"(sacc[1]));   // MFMA D -> VALU reader (§5.7 item 2)
        if ((t + 1) * 64 > p.Nk) {   // ragged last tile only (uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (kv >= p.Nk) sacc[kt][r] = -INFINITY;
                }
        }
        };
        s_phase();
        v8 vf[2][4];
        auto read_v = [&]() {
            // V fragments of this tile: in flight behind the softmax.  A operand (row = d, k = key in P's register order):
            // elements 0..3 = keys 16 s4 + 4 lh + 0..3, elements 4..7 = keys 16 s4 + 8 + 4 lh + 0..3
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (A6B_ABL & 16) { vf[dt][s4] = abl_frag; continue; }
                    const int off = vb + voff[dt] + s4 * 2048;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off + 1024));
                    typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
                    vf[dt][s4] = __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
        };
        if (!A6B_OCC4) read_v();
        A6B_MARK(1);

        // ---- online softmax, register-local (this lane: 32 of its query's 64 scores, lane^32 the rest)
        // The running max and the row sum are reductions over the lane's 32 scores.  As ONE serial chain each (16 dependent
        // v_max3, 32 dependent v_add) they are bound by instruction latency, not issue (ablation: removing them bought 11 %,
        // removing the 32 exponentials nothing); four independent chains each, merged at the end.
        float alpha = 1.0f;
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        bool need;
        auto tile_max = [&]() {   // maximum of the lane's 32 scores, four independent v_max3 chains
            if (A6B_ABL & 8) return sacc[0][0];
            float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) m4[r >> 2] = fmaxf(m4[r >> 2], sacc[kt][r]);
            return fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        };
        auto exp_and_sum = [&]() {   // P = exp2(s') in place, -> the lane's row-sum share
#pragma unroll
            for (int j = 0; j < 4; ++j) r4[j] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv = sacc[kt][r];
                    if (!(A6B_ABL & 1)) pv = __builtin_amdgcn_exp2f(pv);
                    sacc[kt][r] = pv;
                    if (!(A6B_ABL & 8) || r == 0) r4[r >> 2] = a6b_add(r4[r >> 2], pv);
                }
            return a6b_add(a6b_add(r4[0], r4[1]), a6b_add(r4[2], r4[3]));
        };
        if (A6B_BIAS && A6B_SUMCHK) {
            // Sum-checked softmax: no row maximum on the common path.  The scores come out of the matrix pipe relative to the
            // maximum m_run known before the tile; P = exp2(s') and its row sum are needed anyway, and the SUM tells whether
            // the bias was stale: as long as a lane's 32 exponentials add up to <= 2^14 none of them exceeds 2^14 -- finite
            // in fp16 / bf16 -- and O, l are fp32.  Only when the sum is larger (or inf / NaN: s' > 128 overflows exp2), and on
            // the first tile, the tile is REDONE: the scores are recomputed (the exponentials overwrote them; the K
            // sub-tile is still in LDS), the exact maximum is taken, m_run moves up, O and l are rescaled.  -22 vector
            // instructions per tile (17 v_max3, the half-wave exchange, the compare) of ~108.
            float rs = exp_and_sum();
            need = (t == 0) || !(rs <= 16384.0f);
            if (__builtin_expect(__any(need), 0)) {
                s_phase();
                float mx = tile_max();
                {
                    const uint32_t mb = __builtin_bit_cast(uint32_t, mx);
                    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                    mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
                }
                // both half-waves of a query take the same decision from the same mx: m_run stays equal in lane and lane ^ 32
                const bool up = (t == 0) || (mx > 0.0f);                     // the maximum only ever moves UP (alpha <= 1)
                const float m_new = up ? (float)(T)(m_run + mx) : m_run;     // representable in T: next tile's bias operand
                const float delta = m_new - m_run;                           // exact; 0 where the maximum stays
                m_run = m_new;
                alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);      // first tile: nothing to rescale (delta may be << 0)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kt][r] -= delta;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                rs = exp_and_sum();
            }
            l_run = l_run * alpha + rs;
        } else if (A6B_BIAS) {
            float mx = tile_max();
            // the other half-wave's maximum of the same query: one v_permlane32_swap (lanes 32..63 of a <-> lanes 0..31 of b)
            {
                const uint32_t mb = __builtin_bit_cast(uint32_t, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
            }
            // mx = (maximum of the tile's scores) - m_run, in log2 units.  Deferred maximum (T13): the bias keeps m_run unless
            // the scores outgrow it by 2^8; the first tile always takes the branch (m_run = 0 there is not a maximum).
            need = (t == 0) || (mx > A5_DEFER_LOG2);
            if (__any(need)) {
                const float m_new = need ? (float)(T)(m_run + mx) : m_run;   // representable in T: next tile's bias operand
                const float delta = m_new - m_run;                            // exact (both operands are T values); 0 where !need
                m_run = m_new;
                // first tile: O and l are still zero and delta may be hugely NEGATIVE (scores far below the initial 0):
                // exp2(-delta) would be +inf and 0 * inf a NaN -- there is nothing to rescale
                alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kt][r] -= delta;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv = sacc[kt][r];
                    if (!(A6B_ABL & 1)) pv = __builtin_amdgcn_exp2f(pv);
                    sacc[kt][r] = pv;
                    if (!(A6B_ABL & 8) || r == 0) r4[r >> 2] = a6b_add(r4[r >> 2], pv);
                }
            l_run = l_run * alpha + a6b_add(a6b_add(r4[0], r4[1]), a6b_add(r4[2], r4[3]));
        } else {
            float mx = tile_max();
            mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2e;   // scale > 0: max commutes with it
            need = mx > m_run + 8.0f;                              // deferred max (T13); true on the first tile
            if (need) {
                alpha = __builtin_amdgcn_exp2f(m_run - mx);
                m_run = mx;
            }
            const float nm = -m_run;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv = __builtin_fmaf(sacc[kt][r], p.scale_log2e, nm);
                    if (!(A6B_ABL & 1)) pv = __builtin_amdgcn_exp2f(pv);
                    sacc[kt][r] = pv;
                    if (!(A6B_ABL & 8) || r == 0) r4[r >> 2] += pv;
                }
            const float rs = (r4[0] + r4[1]) + (r4[2] + r4[3]);
            l_run = l_run * alpha + rs;
            if (__any(need)) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
        }
        v8 pf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s4][j] = (T)sacc[s4 >> 1][8 * (s4 & 1) + j];
        if (A6B_STAMP) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(pf[s4]));
            A6B_MARK(2);
        }
        if (A6B_OCC4) read_v();   // four waves per SIMD: the fragments may not be live across the softmax

        // ---- O^T[d][q] += V^T P^T, VGPR form
        asm volatile("s_setprio 1");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s4 = i >> 1, dt = i & 1;
            if constexpr (__is_same(T, f16))
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(oacc[dt]) : "v"(vf[dt][s4]), "v"(pf[s4]));
            else
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(oacc[dt]) : "v"(vf[dt][s4]), "v"(pf[s4]));
        }
        asm volatile("s_setprio 0");
        if (A6B_STAMP) {
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(oacc[0]), "+v"(oacc[1]));
            A6B_MARK(3);
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(oacc[0]), "+v"(oacc[1]));   // O readable by VALU
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next KPB sub-tiles have landed
    if (!(A6B_ABL & 2)) __syncthreads();                // everyone done with the buffers of this barrier period
    A6B_MARK(4);
    }
    if (A6B_STAMP && p.dbg != nullptr && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + w) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) d[i] = st_sum[i];
        d[5] = (unsigned long long)nt;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (NW <= 8) {   // (the K / V buffers are dead behind the loop's last barrier: 4 KiB of them per wave; NW = 16 has no room)
        a6_store_o<T>(smem + w * 4096, oacc, inv, q0, lane, p, b, h);
        return;
    }
    if (qrow >= p.Nq) return;
    T* Ob = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)h * D + (int64_t)qrow * p.o_ts;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)(oacc[dt][4 * g + e] * inv);
            *(v4*)(Ob + dt * 32 + 8 * g + 4 * lh) = o;
        }
}

#include "attention_d64c.inc"
#include "attention_d64p.inc"

}  // namespace

// split-KV plan of the D = 512 kernel: key ranges per workgroup
//   (a) so that the grid fills the 256 CUs when the query tiles alone cannot;
//   (b) so that ONE key range stays resident in the 256 MiB Infinity Cache while every query tile streams it.  At
//       N = 262 144 the key/value tensor is 268 MB (shared tile) or 537 MB: every workgroup re-streamed it from HBM
//       (rocprofv3 FETCH_SIZE: ~400 GB per launch, ~3 TB/s for the whole 131 ms).  Workgroups are dispatched range-major
//       (blockIdx.y), so with ranges of A5B_MALL_KEYS keys (64 MiB shared / 128 MiB as two tensors) all the query tiles in
//       flight read the same range out of the cache; the price is one fp32 partial (O, m, l) per range and the merge pass.
//       Measured on one box (N = 262 144, shared tile): 1036 TFLOP/s with 64 Ki-key ranges, 1041 without, 1026 with 32 Ki:
//       the kernel is NOT bound by that HBM stream (its LDS-DMA requests are two tiles ahead), so this buys no time; it is
//       kept because it takes ~3 TB/s of needless HBM traffic (and its power) out of the dominant Stage-1 kernel.
#ifndef A5B_MALL_KEYS
#define A5B_MALL_KEYS 65536
#endif
static void attn512_plan(int B, int heads, int Nq, int Nk, int* nsplit, int* keys_per_split) {
    const int64_t base = (int64_t)((Nq + 127) / 128) * B * heads;
    int ns = 1;
    if (base < 192) {
        ns = (int)((256 + base - 1) / base);
        const int cap = Nk / 256 > 1 ? Nk / 256 : 1;   // at least 8 key tiles per range
        if (ns > cap) ns = cap;
        if (ns > 16) ns = 16;
    }
    if (A5B_MALL_KEYS > 0 && Nk > A5B_MALL_KEYS + A5B_MALL_KEYS / 2) {
        const int nm = (Nk + A5B_MALL_KEYS - 1) / A5B_MALL_KEYS;
        if (nm > ns) ns = nm;
    }
    int kps = ((Nk + ns - 1) / ns + 31) / 32 * 32;
    *keys_per_split = kps;
    *nsplit = (Nk + kps - 1) / kps;
}

extern "C" int64_t rsvld_attention_ws_bytes(int B, int heads, int Nq, int Nk, int D, int plan_div) {
    if (D != 512 || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return 0;
    int ns, kps;
    attn512_plan(plan_div > 1 ? (B + plan_div - 1) / plan_div : B, heads, Nq, Nk, &ns, &kps);
    if (ns == 1) return 0;
    return (int64_t)ns * B * heads * Nq * (512 + 2) * 4;
}

extern "C" int rsvld_attention_tuned(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq, int Nk,
                                     int D, int64_t q_batch_stride, int64_t q_tok_stride, int64_t k_batch_stride,
                                     int64_t k_tok_stride, int64_t v_batch_stride, int64_t v_tok_stride,
                                     int64_t o_batch_stride, int64_t o_tok_stride, float scale, int dtype, int plan_div,
                                     void* ws, void* stream, int tune) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return RSVLD_EINVAL;
    if (dtype != RSVLD_F16 && dtype != RSVLD_BF16) return RSVLD_EINVAL;
    // 16-byte vector access along d: strides must keep rows 8-element aligned
    if ((q_tok_stride | k_tok_stride | v_tok_stride | o_tok_stride | q_batch_stride | k_batch_stride | v_batch_stride |
         o_batch_stride) & 7)
        return RSVLD_EINVAL;
    if (heads > 65535 || B > 65535) return RSVLD_EINVAL;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.D = D;
    a.q_bs = q_batch_stride; a.q_ts = q_tok_stride; a.k_bs = k_batch_stride; a.k_ts = k_tok_stride;
    a.v_bs = v_batch_stride; a.v_ts = v_tok_stride; a.o_bs = o_batch_stride; a.o_ts = o_tok_stride;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.dbg = (A6B_STAMP && D == 64) ? (unsigned long long*)ws : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (D == 512) {
        int ns, kps;   // the key split (an accumulation-order choice) is planned on ONE of the plan_div stacked units
        attn512_plan(plan_div > 1 ? (B + plan_div - 1) / plan_div : B, heads, Nq, Nk, &ns, &kps);
        if (ns > 1 && ws == nullptr) return RSVLD_EINVAL;
        float* part_o = (float*)ws;
        float* part_ml = ns > 1 ? part_o + (int64_t)ns * B * heads * Nq * 512 : nullptr;
        dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)ns, (unsigned)(B * heads));
        // keys and values are one tensor (see the SH note above): the shared-tile instantiation
        const bool shared = k == v && k_batch_stride == v_batch_stride && k_tok_stride == v_tok_stride;
        auto go = [&](auto kern, auto comb, int smem) -> int {
            static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (attr != hipSuccess) return RSVLD_ELAUNCH;
            hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a, kps, part_o, part_ml);
            if (ns > 1) hipLaunchKernelGGL(comb, dim3((unsigned)Nq, (unsigned)(B * heads)), dim3(128), 0, s, a, ns, part_o, part_ml);
            return rsvld_check_launch();
        };
        if (shared)
            return dtype == RSVLD_F16 ? go(attn_d512b_kernel<f16, true>, attn_combine_kernel<f16>, A5B_SMEM_SH)
                                      : go(attn_d512b_kernel<bf16, true>, attn_combine_kernel<bf16>, A5B_SMEM_SH);
        return dtype == RSVLD_F16 ? go(attn_d512b_kernel<f16, false>, attn_combine_kernel<f16>, A5B_SMEM)
                                  : go(attn_d512b_kernel<bf16, false>, attn_combine_kernel<bf16>, A5B_SMEM);
    }
    if (D == 64) {
        // tune (RSVLD_ATTN_D64_*: tests and A/B runs) overrides the choice between the three d = 64 kernels, which agree bit for
        // bit on every shape; 0 = by grid size
        const char force = tune == RSVLD_ATTN_D64_FOUR_WAVE ? 'b' : tune == RSVLD_ATTN_D64_PINGPONG ? 'c' : tune == RSVLD_ATTN_D64_PIPELINED ? 'p' : 0;
        if (force == 'p' && Nk > 64 && k_tok_stride < (1 << 23) && v_tok_stride < (1 << 23)) {   // the pipelined kernel (diagnostic selection only so far; 24-bit row stride in bytes)
            dim3 grid((unsigned)((Nq + 255) / 256), (unsigned)heads, (unsigned)B);
            static const hipError_t a16 = hipFuncSetAttribute((const void*)attn_d64p_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, A6P_SMEM);
            static const hipError_t abf = hipFuncSetAttribute((const void*)attn_d64p_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, A6P_SMEM);
            if (a16 != hipSuccess || abf != hipSuccess) return RSVLD_ELAUNCH;
            if (dtype == RSVLD_F16) hipLaunchKernelGGL((attn_d64p_kernel<f16>), grid, dim3(512), A6P_SMEM, s, a);
            else hipLaunchKernelGGL((attn_d64p_kernel<bf16>), grid, dim3(512), A6P_SMEM, s, a);
            return rsvld_check_launch();
        }
        const int64_t wg_c = (int64_t)((Nq + 511) / 512) * heads * B;
        // attn_d64c forms 32-bit lane offsets of up to 64 key rows: a token stride of ~16 M elements or more would wrap
        const bool c_strides_ok = 64 * k_tok_stride * 2 < ((int64_t)1 << 32) && 64 * v_tok_stride * 2 < ((int64_t)1 << 32);
        if (force == 'c' && !c_strides_ok) return RSVLD_EUNSUPPORTED;
        if (c_strides_ok && (a.dbg == nullptr || force == 'c') && force != 'b' && Nk > 64 && ((wg_c >= A6C_MIN_WG && Nk >= A6C_MIN_KEYS) || force == 'c')) {   // long query sequences: the ping-pong kernel
            dim3 grid((unsigned)((Nq + 511) / 512), (unsigned)heads, (unsigned)B);
            static const hipError_t attr16 = hipFuncSetAttribute((const void*)attn_d64c_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, A6C_SMEM);
            static const hipError_t attrbf = hipFuncSetAttribute((const void*)attn_d64c_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, A6C_SMEM);
            if (attr16 != hipSuccess || attrbf != hipSuccess) return RSVLD_ELAUNCH;
            if (dtype == RSVLD_F16) hipLaunchKernelGGL((attn_d64c_kernel<f16>), grid, dim3(512), A6C_SMEM, s, a);
            else hipLaunchKernelGGL((attn_d64c_kernel<bf16>), grid, dim3(512), A6C_SMEM, s, a);
        } else if (Nq >= A6B_NW16_MIN) {
            dim3 grid((unsigned)((Nq + 511) / 512), (unsigned)heads, (unsigned)B);
            if (dtype == RSVLD_F16) hipLaunchKernelGGL((attn_d64b_kernel<f16, 16>), grid, dim3(1024), 0, s, a);
            else hipLaunchKernelGGL((attn_d64b_kernel<bf16, 16>), grid, dim3(1024), 0, s, a);
        } else if (Nq >= A6B_NW8_MIN) {   // long query sequences: 8 waves (256 query rows) share each K / V tile
            dim3 grid((unsigned)((Nq + 255) / 256), (unsigned)heads, (unsigned)B);
            if (dtype == RSVLD_F16) hipLaunchKernelGGL((attn_d64b_kernel<f16, 8>), grid, dim3(512), 0, s, a);
            else hipLaunchKernelGGL((attn_d64b_kernel<bf16, 8>), grid, dim3(512), 0, s, a);
        } else {
            dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)heads, (unsigned)B);
            if (dtype == RSVLD_F16) hipLaunchKernelGGL((attn_d64b_kernel<f16, 4>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_d64b_kernel<bf16, 4>), grid, dim3(256), 0, s, a);
        }
        return rsvld_check_launch();
    }
    return RSVLD_EUNSUPPORTED;
}

extern "C" int rsvld_attention(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq, int Nk,
                               int D, int64_t q_batch_stride, int64_t q_tok_stride, int64_t k_batch_stride,
                               int64_t k_tok_stride, int64_t v_batch_stride, int64_t v_tok_stride,
                               int64_t o_batch_stride, int64_t o_tok_stride, float scale, int dtype, int plan_div,
                               void* ws, void* stream) {
    return rsvld_attention_tuned(q, k, v, out, B, heads, Nq, Nk, D, q_batch_stride, q_tok_stride, k_batch_stride, k_tok_stride,
                                 v_batch_stride, v_tok_stride, o_batch_stride, o_tok_stride, scale, dtype, plan_div, ws, stream, 0);
}
