// gemm.hip — out[m][n] = act(alpha * (sum_k X[m][k] W[n][k] + bias[n])) + beta * residual[m][n] for the 1x1 / Linear
// layers of the Stage-2 transformer blocks (sgm/modules/attention.py:84-110,196-285,533-635): large M (tokens),
// K = 320 .. 5120, N = 320 .. 10240.  rocprofv3 showed these layers at 420-630 TFLOP/s on conv_igemm's 128x128 tile
// (64x64 per wave: 1 KiB of LDS fragment reads + 0.5 KiB of LDS-DMA writes per MFMA, 16 MFMAs per barrier), where the
// library GEMM of the box reaches 750-1200 on the same shapes; they are 58 % of the Stage-2 time at latent 256.
//
// Structure (one workgroup per CU, 8 waves = 2 per SIMD, 128 KiB LDS):
//   * 256 (m) x 256 (n) tile; wave (g, wn) owns rows 128 g .. +128 and columns 64 wn .. +64: 8 accumulators of
//     32x32, 0.75 KiB of fragment reads and 0.25 KiB of DMA writes per MFMA;
//   * K runs in 32-deep tiles through a 4-stage LDS ring (32 KiB per stage: X rows | W rows, 64-B rows, XOR swizzle on
//     the DMA source side), three tiles in flight, counted vmcnt, raw barriers;
//   * the two waves of a SIMD belong to different groups g and alternate roles every slot: while group 0 issues the
//     16 MFMAs of tile t, group 1 reads its fragments of tile t and issues the DMAs of tile t+3, then they swap
//     (cdna_hip_programming.md §5 "8-phase template" reduced to two phases per tile).  Every slot ends with a barrier
//     that all 8 waves execute; a reader drains its own LDS reads before that barrier, so the ring is race-free by
//     construction: tile t+3 is written one barrier after the last read of tile t-1 has RETURNED.
// The MFMA A operand is the weight tile, the B operand the activation tile: a lane owns one output row (token) and,
// per accumulator quad, four consecutive output channels (8-byte stores; GEGLU pairs are adjacent channels).
//
// PERSIST instantiations (round 4; the 16-bit layers with K >= 128, the library's choice unless RSVLD_TUNE_GEMM_ONE_TILE): one workgroup
// per CU walks its tiles (2-D tile order per XCD); the next tile's first K tiles and bias row are requested before the epilogue, the
// epilogue transposes per wave through a private 4 KiB LDS buffer without a workgroup barrier, the bias is the C operand of the
// tile's first MFMAs; one instantiation per epilogue variant.  Where a tile's time went before and after: tools/gemm_stamps.py
// (-DG_STAMP=1), profiles/r04_gemm_stamps*.txt; the comment at the head of the PERSIST block below.
//
// SPLIT instantiation (round 4; dtype RSVLD_SPLIT): the split-operand precision on THIS tiling.  An fp32 activation x is stored as two
// bf16 planes per row, [lo(K) | hi(K)] with hi = bf16(x), lo = bf16(x - hi); a weight row as the triple [W_hi(K) | W_lo(K) | W_hi(K)].
// The product x W^T = x_lo W_hi + x_hi W_lo + x_hi W_hi (+ the dropped 2^-16 term) is then ONE bf16 GEMM over the concatenated
// K' = 3 K: the K loop below runs 3 K / 32 tiles, the weight pointer walks its row linearly and the activation pointer walks
// lo, hi and re-reads hi (one scalar select per LDS-DMA piece).  Small terms first.  Same ring, same ping-pong, same MFMA count
// per product as any 3-MFMA scheme, and the epilogue is amortised over three times the K loop.  Output: fp32 [M][N_out]
// (+ beta * fp32 residual) or bf16 planes [M][lo(N_out) | hi(N_out)] for a tensor that only feeds another matrix product.
#include <stdlib.h>

#include "rsvld_common.h"
#include <type_traits>

namespace {

struct GemmArgs {
    const void* x;         // [M][K]
    const void* w;         // [N][K]
    const float* bias;     // [N] or nullptr
    const void* residual;  // [M][N_out] or nullptr
    void* out;             // [M][N_out]
    int M, N, K, N_out;
    int act;
    float alpha, beta;
    int out_kind;          // SEG > 1 only: 0 = 16-bit out (fp16), 1 = fp32 out (+ fp32 residual), 2 = bf16 planes out
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int G_STAGE = 32 * 1024;   // X: 256 rows x 64 B | W: 256 rows x 64 B
#ifndef G_ABL
#define G_ABL 0   // diagnostic builds (results WRONG, timing only): 1 no global stores / residual loads, 2 no epilogue, 4 three K steps,
                  // 8 the LDS-DMA source cycles over four K tiles (operands L2-resident: the K loop without memory latency),
                  // 16 no LDS-DMA pieces in the steady-state multiply slots beyond K tile 5 (the K loop without their issue cost; waits pass at once)
#endif
// G_ONEBAR = 1 (experiment, round 3): ONE barrier per K tile instead of two (see the loop).  Correct (kernel tests green) and within
// +-1 % of the two-barrier form on all nine headline shapes at 20 repetitions each (profiles/r03_gemm_onebar_ab.txt): the mid-tile
// barrier is not what the K loop loses.  On the K-dominated shape (32 768 x 5 120 -> 1 280) the kernel is level with the vendor GEMM
// (862-891 vs 876-912 TFLOP/s); the gap is the unhidden epilogue at K <= 2 048 and the 20 % padding of N = 640 to three 256-wide tiles.  Off.
#ifndef G_ONEBAR
#define G_ONEBAR 0
#endif
#ifndef G_EPI_SPECIALISED
#define G_EPI_SPECIALISED 1
#endif
#ifndef G_ASMDMA
#define G_ASMDMA 1   // 1: LDS-DMA pieces as inline asm in the scalar-base form, steady-state loop without the "is there a tile to
#endif               // request" branches (see dma_piece / the slot loops); 0: the round-2 form (builtin, per-lane 64-bit addresses)
#ifndef G_STAGGER
#define G_STAGGER 0   // experiment: first-round workgroups on every other CU start half a K loop late (s_sleep units per K tile),
#endif                // so that the CUs' output bursts stop coinciding; 0 = off
#ifndef G_ORDER2D
#define G_ORDER2D 1   // tile order inside an XCD's run: 1 = blocks of 8 x 4 tiles (see the index computation), 0 = m fastest (rounds 2-3)
#endif
#ifndef G_DMA_SPLIT
#define G_DMA_SPLIT 0     // experiment (round 4), persistent form: activation pieces of tile kt + 3 in the read slot of tile kt, weight pieces between
                          // the MFMAs of its multiply slot (two and two instead of four in one slot).  Correct, and level with 0: see G_DMA_IN_READ.
#endif
#ifndef G_DMA_IN_READ
#define G_DMA_IN_READ 0   // experiment (round 4), persistent form: the LDS-DMA pieces of tile kt + 3 in the READ slot of tile kt (1) instead of between the
                          // MFMAs of its multiply slot (0).  Correct (kernel tests green) and level with 0 on all nine headline shapes, and so is
                          // G_DMA_SPLIT (two and two): WHERE the pieces are issued does not matter.  Without them the K loop runs at the matrix pipe's
                          // rate (-DG_ABL=16: 31.3 -> 26.6 us per tile at K = 1 280 = 1 026 cycles per K tile); a half tile's slots with 8 MFMAs and 3
                          // pieces take as long as a whole tile's with 16 and 4.  What the pieces cost is LDS time: per K tile the CU reads 96 KiB of
                          // fragments (<= 256 B/clk) and the pieces write 32 KiB (~64 B/clk): ~900 of the 1 024 cycles its 128 MFMAs take.
                          // profiles/r04_gemm_dma_issue.txt.  Off.
#endif
#ifndef G_PAIR_XREAD
#define G_PAIR_XREAD 1    // PAIRED K loop (round 5): the two ACTIVATION pieces of an even K' tile are issued in the read slot of the odd tile three
                          // tiles before it (four fragment reads instead of twelve: the one slot with issue time to spare) and only its two weight
                          // pieces between the MFMAs, so that EVERY multiply slot carries two pieces (0: four / two alternating).  An LDS-DMA piece
                          // costs its wave 60-180 issue cycles; 16 MFMAs + four pieces overran the 512 cycles the partner's slot takes.
#endif
#ifndef G_RES_AHEAD
#define G_RES_AHEAD 1     // persistent fp32 + residual epilogue: the residual pieces of this many 32 x 32 half-blocks are in flight ahead of the one being
                          // stored.  2 and 3 measured (round 5, tools/bench_linear_w2.py, A-B-A-B): no change on any of the four to_out / ff.net.2 shapes
                          // (157.9 / 157.7 / 157.8 us at 32 768 x 1 280 -> 1 280) -- with fp16 weights these GEMMs move 8 bytes per output element beside
                          // 2 K FLOP and run at 3.8-4.3 TB/s of HBM traffic: bound by the bytes, not by the latency of the request rounds
#endif
#ifndef G_HALF_TILES
#define G_HALF_TILES 1    // persistent form: the last partial round of an XCD's run as 128-row half tiles (see the tile enumeration); 0 = whole tiles
#endif
#ifndef G_GELU_PACKED
#define G_GELU_PACKED 1   // persistent GEGLU epilogue: the pairwise GELU in packed fp32 arithmetic (gelu_erf2_f); 0 = the scalar form
#endif
#ifndef G_STAMP
#define G_STAMP 0     // diagnostic build: thread 0 of every workgroup records s_memrealtime (100 MHz) at its phase boundaries + HW_ID into
#endif                // g_stamp_buf (read back by rsvld_debug_gemm_stamps; tools/gemm_stamps.py) -- where a tile's time goes, and the gap
                      // between two workgroups on one CU
constexpr int G_NST = 4;
constexpr int G_RING = G_NST * G_STAGE;
constexpr int G_SMEM = G_RING + 1024;   // + the tile's 256 bias values (fp32), fetched once while the ring fills
constexpr int G_SMEM_P = G_RING + 8 * 4096;   // PERSIST: + a private 4 KiB transposition buffer per wave = all 160 KiB of the CU

// 64-B rows: 16-B slot s of row r sits at s ^ ((r>>2)&3): conflict-free ds_read_b128 (brute-forced, see conv_halo.hip)
__device__ __forceinline__ int g_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

#if G_STAMP
__device__ unsigned long long g_stamp_buf[16384 * 8];
#define G_STAMP_AT(i) do { if (threadIdx.x == 0 && stamp_lid < 16384) g_stamp_buf[stamp_lid * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G_STAMP_AT(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void g_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PV (PERSIST only): the epilogue variant, chosen on the host: activation (0 none, 1 SiLU, 2 GEGLU) | 4 x (alpha == 1) | 8 x residual.
// One variant per instantiation: with the seven variants behind a dispatch inside the tile loop hipcc spilled 490-680 dwords around
// the epilogue (none with one variant), and a scratch reload is a vector-memory load that queues behind the ring's LDS-DMA pieces.
// SEG = K segments of one logical product: 1 = a plain 16-bit GEMM; 3 = RSVLD_SPLIT (bf16 planes lo | hi x triples W_hi | W_lo | W_hi);
// 2 = RSVLD_F16W2 (round 5): fp16 activations x fp16 weight PAIRS [W_lo | W_hi] -- the weights keep ~22 significant bits, the
// activation its 11: x W = x W_lo + x W_hi is ONE fp16 GEMM over K' = 2 K whose second segment re-reads the activation row.
// TO = the 16-bit OUTPUT type: T for a plain GEMM, fp16 for both multi-segment forms (a tensor that goes on to an fp16-operand
// consumer: the 16-bit attention kernels, another F16W2 layer); their fp32 / planes outputs leave through the one-tile SEG epilogue.
// WIDE (one-tile form of the multi-segment GEMMs): 0 = the 16-bit epilogue, 1 = the fp32 / planes epilogue without a residual (any
// activation), 2 = fp32 out + fp32 residual (no GEGLU) -- a template parameter, because a kernel that holds several epilogues spills
// (220 scratch instructions with two, 119 with the residual registers beside the GEGLU / planes staging code; a scratch reload
// shares vmcnt with the LDS-DMA ring).
template <typename T, int SEG = 1, bool PERSIST = false, int PV = 0, int WIDE = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
    static_assert(WIDE == 0 || SEG > 1, "the fp32 / planes epilogues belong to the multi-segment kernels");
    static_assert(!(PERSIST && WIDE == 1), "persistent form: the 16-bit epilogue (WIDE 0) or fp32 out + fp32 residual (WIDE 2, PV 4 = none)");
    static_assert(!(PERSIST && WIDE == 2) || PV == 0, "the persistent fp32 epilogue has one variant: no activation, alpha, + beta * residual");
    static_assert(SEG >= 1 && SEG <= 4, "1 plain, 2 fp16 x weight pairs, 3 bf16 planes x weight triples, 4 fp16 x fp16 with the multi-segment outputs");
    // SEG 4 (RSVLD_F16W1): ONE K segment -- a plain fp16 GEMM -- with the output side of the multi-segment family (fp32 out + fp32 residual)
    constexpr int NSEG = SEG == 4 ? 1 : SEG;
    static_assert(SEG == 1 || !(PV & 8), "the persistent residual variants are 16-bit residuals (plain GEMM only)");
    static_assert(PERSIST || PV == 0, "PV is the persistent epilogue's variant");
    constexpr bool SPLIT = SEG == 3;
    constexpr bool PAIRED = SEG == 2 && PERSIST;   // (the persistent weight-pair GEMMs: K' tiles in (W_lo, W_hi) pairs per activation tile, see xk / wk)
    typedef typename Mfma<T>::v8 v8;
    typedef typename std::conditional<SEG == 1, T, f16>::type TO;
    typedef typename Mfma<TO>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- tiles.  XCD-aware order: each XCD (= linear workgroup id mod 8) walks a contiguous run of the tile sequence.  One tile per
    // workgroup (grid = the tile grid), or PERSIST: one workgroup per CU, workgroup c of an XCD takes tiles c, c + step, ... of the run.
    const int nmt = (p.M + 255) >> 8, nnt = (p.N + 255) >> 8;
    // PERSIST, the last round of an XCD's run: when it holds rem <= half as many tiles as the XCD has workgroups, each of them is cut
    // into two 128-row HALF tiles for two workgroups (640 tiles on 256 CUs: 2.5 rounds instead of 3; 1 920 tiles: 7.5 instead of 8).
    int t_cur, t_end, t_step;
    int n_full = 1, t_half = -1, h_half = -1;   // this workgroup: n_full whole tiles t_cur, t_cur + t_step, ..., then half h_half of tile t_half
    {
        const int ntiles = nmt * nnt;
        const int lid = PERSIST ? (int)blockIdx.x : (int)(blockIdx.x + blockIdx.y * gridDim.x);
        const int q = ntiles >> 3, r = ntiles & 7, xcd = lid & 7;
        const int ts = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_cur = ts + (lid >> 3);
        t_end = PERSIST ? ts + q + (xcd < r ? 1 : 0) : t_cur + 1;
        t_step = PERSIST ? ((int)gridDim.x - xcd + 7) >> 3 : 1;
        if constexpr (PERSIST) {
            const int c = lid >> 3, qx = t_end - ts;
            const int fr = qx / t_step, rem = qx - fr * t_step;
            const bool halves = G_HALF_TILES && !PAIRED && rem > 0 && 2 * rem <= t_step;   // (the paired K loop exists for whole tiles)
            n_full = fr + ((!halves && c < rem) ? 1 : 0);
            if (halves && c < 2 * rem) {
                t_half = ts + fr * t_step + (c >> 1);
                h_half = c & 1;
            }
            if (n_full == 0 && t_half < 0) return;
        }
    }
    auto tile_of = [&](int t, int& tile_m, int& tile_n) {
#if G_ORDER2D
        // the 32 workgroups an XCD runs at a time form a block of 8 (m) x up to 4 (n) tiles: an activation tile is fetched into that
        // L2 once per <= 4 workgroups and a weight tile once per 8, instead of 32 different activation tiles beside ONE weight tile
        // (1-D order: every activation byte crossed the fabric once per 256 output columns).  Column blocks of balanced width.
        const int NB = (nnt + 3) >> 2, wb = nnt / NB, ex = nnt - wb * NB;     // the first `ex` column blocks are wb + 1 tiles wide
        const int big = nmt * (wb + 1);
        int nb, rem, w, nfirst;
        if (t < ex * big) { nb = t / big; rem = t - nb * big; w = wb + 1; nfirst = nb * (wb + 1); }
        else { const int u = t - ex * big; nb = u / (nmt * wb); rem = u - nb * (nmt * wb); w = wb; nfirst = ex * (wb + 1) + nb * wb; }
        const int mb = rem / (8 * w), rr = rem - mb * (8 * w);
        const int h = min(8, nmt - 8 * mb);
        const int dn = rr / h;
        tile_n = nfirst + dn;
        tile_m = 8 * mb + (rr - dn * h);
#else
        tile_n = t / nmt;
        tile_m = t - tile_n * nmt;
#endif
    };
    if (PERSIST && t_half >= 0) {   // a lower half that starts beyond M (the last row of tiles holds <= 128 rows) does not exist: its row offsets
        int tm, tn;                 // relative to the tile's first row would be negative, i.e. huge as the 32-bit unsigned offsets of the LDS-DMA pieces
        tile_of(t_half, tm, tn);
        if (tm * 256 + 128 * h_half >= p.M) {
            t_half = -1;
            if (n_full == 0) return;
        }
    }
    const int nk0 = p.K >> 5;
    const int nk = (G_ABL & 4) ? 3 : NSEG * nk0;   // diagnostic build 4: three K steps only (workgroup turnover + ring fill)
    const int64_t rowb = (int64_t)p.K * (int64_t)sizeof(T) * (SPLIT ? 2 : 1);    // activation row: K values, or the planes lo | hi
    const int64_t rowb_w = (int64_t)p.K * (int64_t)sizeof(T) * NSEG;             // weight row: K values, the pair lo | hi, or the triple hi | lo | hi
    // K tile kt of the concatenated K' -> the activation's K tile: SPLIT reads the planes lo, hi, hi (tiles >= 2 nk0 alias the hi
    // plane), the pair form reads the one row twice; the weight row is linear
    // PAIRED (round 5; the persistent weight-pair GEMMs): the two K' tiles of one activation K tile are ADJACENT -- tile 2 p = [X(p) | W_lo(p)],
    // tile 2 p + 1 = [ -- | W_hi(p)] -- so that the second needs neither an LDS-DMA of X nor a fragment read of it: the activation fragments of
    // tile 2 p stay in registers for its partner.  Per 32 MFMAs a CU then moves 48 KiB into LDS instead of 64 and reads 16 KiB of fragments per
    // wave pair instead of 24: this K loop is bound by LDS time (7 of the 8 cycles an MFMA takes), not by the matrix pipe.
    auto xk = [&](int kt) { return PAIRED ? (kt >> 1) : SEG == 3 ? (kt >= 2 * nk0 ? kt - nk0 : kt) : SEG == 2 ? (kt >= nk0 ? kt - nk0 : kt) : kt; };
    auto wk = [&](int kt) { return PAIRED ? (kt & 1) * nk0 + (kt >> 1) : kt; };   // weight row = [W_lo(K) | W_hi(K)]
    if (G_STAGGER > 0 && !PERSIST) {
        const int lid = blockIdx.x + blockIdx.y * gridDim.x;
        if (lid < 256 && ((lid >> 3) & 1))
            for (int i = 0; i < nk; ++i) __builtin_amdgcn_s_sleep(G_STAGGER);
    }

    // ---- per tile: DMA roles.  Wave w fills rows 32w .. 32w+31 of X and of W: two wave-instructions of 16 rows x 64 B each.
    // lane -> (row = lane>>2, position = lane&3) holds source chunk position ^ ((row>>2)&3); rows past the end of the
    // tensor re-read its last row (their outputs are never stored)
    int m0 = 0, n0 = 0;
    uint32_t xvo[2], wvo[2];
    const char *Xb = nullptr, *Wb = nullptr;
    // PERSIST: the tile's bias enters as the C operand of the first k-step (bC[ni][4 g + e] = bias of channel n0 + 64 wn + 32 ni + 8 g
    // + 4 lh + e, the accumulator layout): no bias pass in the epilogue and no accumulator clearing.  (The one-tile form keeps the
    // tile's bias row in LDS and adds it in the epilogue: sum + bias instead of bias + sum, one fp32 rounding apart.)
    // Its way there touches no register: every wave requests the 64 bias values of its columns by ONE LDS-DMA piece (4 bytes per
    // lane) into ring stage 3 -- free from the end of a tile's K loop until the first multiply slot of the next tile requests K tile
    // 3 -- IN FRONT of that tile's K tiles 0..2, so the counted wait before the first fragment read covers it; both groups fetch
    // their eight quads from there before group 0's first multiply slot opens.  (A register-parked value would be an asynchronous
    // load the compiler may copy before it has arrived; a load hipcc can see gets a compiler-placed vmcnt(0) inside the epilogue.)
    f32x16 bC[2];
    auto dma_bias = [&](int n_first) {
        if (p.bias != nullptr) {
            const uint32_t dst = (uint32_t)(uintptr_t)(lptr_t)smem + (uint32_t)(3 * G_STAGE + wave * 256);
            int tq = tid;
            asm volatile("" : "+v"(tq));   // (nothing derived from it is hoisted out of the tile loop and spilled)
            const uint32_t voff = (uint32_t)min(n_first + wn * 64 + (tq & 63), p.N - 1) * 4u;   // channels past N are never stored
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2" : : "v"(voff), "s"(dst), "s"(p.bias) : "memory", "m0");
        }
    };
    auto read_bias = [&]() {
        const char* const Bw = smem + 3 * G_STAGE + wave * 256;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias != nullptr) bv = *(const f32x4*)(Bw + (ni * 32 + 8 * g + 4 * lh) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) bC[ni][4 * g + e] = bv[e];
            }
    };
    // hsel < 0: a whole tile; 0 / 1: its upper / lower 128 rows (a HALF tile: wave w fills X rows 16w .. 16w+15 with ONE piece, xvo[0])
    auto setup_tile = [&](int t, int hsel = -1) {
        int tile_m, tile_n;
        tile_of(t, tile_m, tile_n);
        m0 = tile_m * 256 + (hsel > 0 ? 128 : 0);
        n0 = tile_n * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int tq = tid;
            if (PERSIST) asm volatile("" : "+v"(tq));   // (nothing derived from it is hoisted out of the tile loop: see load_bias)
            const int rw = 32 * wave + 16 * i + ((tq & 63) >> 2);
            const int rx = hsel < 0 ? rw : 16 * wave + ((tq & 63) >> 2);
            const uint32_t ch = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
            // offsets relative to the TILE's first row (the scalar bases below carry the 64-bit part): 256 rows x row bytes < 2^32
            xvo[i] = (uint32_t)((int64_t)(min(m0 + rx, p.M - 1) - m0) * rowb) + ch;
            wvo[i] = (uint32_t)((int64_t)(min(n0 + rw, p.N - 1) - n0) * rowb_w) + ch;
        }
        Xb = (const char*)p.x + (int64_t)m0 * rowb;
        Wb = (const char*)p.w + (int64_t)n0 * rowb_w;
    };
    if (PERSIST && n_full == 0) setup_tile(t_half, h_half);
    else setup_tile(t_cur);
#if G_STAMP
    int stamp_lid = PERSIST ? t_cur : (int)(blockIdx.x + blockIdx.y * gridDim.x);
    unsigned long long stamp_hw;
    {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamp_hw = ((unsigned long long)xcc << 32) | hw;
    }
    G_STAMP_AT(0);
#endif
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // j = 0..3: (X, W) x (rows 0..15, 16..31) of this wave's share of tile kt.
    // G_ASMDMA: the builtin made hipcc form a per-lane 64-bit address for every piece -- a v_lshl_add_u64 INTO the fragment
    // registers the four MFMAs in front of it had just read (a write-after-read wait on the matrix pipe in the middle of the
    // MFMA slot) -- and a branch around every piece.  Here the tile's row base is a scalar (SGPR pair), the lane's row / swizzle
    // offset a 32-bit VGPR that lives across the loop, and M0 (the LDS destination) is written inside the statement and
    // declared clobbered: nothing else in this file uses M0 (tools/audit_m0.py checks the assembly).
    // (the tile's row bases are wave-uniform by construction; behind the per-lane store guards of the persistent fp32 epilogue hipcc's
    //  uniformity analysis gave up on them and handed the asm statements VGPR pairs: readfirstlane states the fact -- in THAT instantiation
    //  only: it does not fold away where the value already lives in SGPRs (+59 instructions per kernel when applied everywhere))
    auto uni = [](const char* q) __attribute__((always_inline)) {
        if constexpr (!(PERSIST && WIDE == 2)) return q;
        const uint64_t v = (uint64_t)(uintptr_t)q;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    auto dma_piece = [&](int kt, int j) {
#if G_ASMDMA
        const uint32_t dst = lds0 + (uint32_t)((kt & (G_NST - 1)) * G_STAGE + wave * 2048 + (j & 1) * 16384 + (j >> 1) * 1024);
        const int ktx = (G_ABL & 8) ? (kt & 3) : xk(kt);
        const char* base = uni((j & 1) ? Wb + ((G_ABL & 8) ? (kt & 3) : wk(kt)) * 64 : Xb + ktx * 64);
        const uint32_t voff = (j & 1) ? wvo[j >> 1] : xvo[j >> 1];
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(dst), "s"(base) : "memory", "m0");
#else
        char* st = smem + (kt & (G_NST - 1)) * G_STAGE + wave * 2048 + (j & 1) * 16384 + (j >> 1) * 1024;
        const int ktx = xk(kt);
        const char* src = (j & 1) ? Wb + kt * 64 : Xb + ktx * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + ((j & 1) ? wvo[j >> 1] : xvo[j >> 1])), (lptr_t)st, 16, 0, 0);
#endif
    };
    auto dma_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!PAIRED || !(kt & 1) || (j & 1)) dma_piece(kt, j);      // PAIRED: an odd tile is its two weight pieces
    };

    f32x16 acc[2][4];   // [n tile][m tile]
    if constexpr (!PERSIST) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    }

    int fa_off[2], fb_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fa_off[ks] = 16384 + g_off(wn * 64 + l31, 2 * ks + lh);     // weights: + ni * 2048
        fb_off[ks] = g_off(grp * 128 + l31, 2 * ks + lh);           // activations: + mi * 2048
    }

    v8 fa[2][2], fb[2][4];   // [k-step][tile]
    auto read_tile = [&](int kt) {
        const char* st = smem + (kt & (G_NST - 1)) * G_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fa[ks][ni] = *(const v8*)(st + fa_off[ks] + ni * 2048);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fb[ks][mi] = *(const v8*)(st + fb_off[ks] + mi * 2048);
        }
    };
    // 16 MFMAs of the fragments in registers; the 4 LDS-DMAs of tile `next` (if >= 0) are issued between them: an
    // LDS-DMA costs its wave 60-180 issue cycles, which the matrix pipe covers here and which would lengthen the
    // partner group's critical read slot otherwise (measured: all four in the read slot -4 %, two and two -5 % at
    // K >= 2560 against this placement)
    auto mma_tile = [&](int next) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], acc[ni][mi]);
                if (next >= 0) dma_piece(next, 2 * ks + ni);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    // FIRST (PERSIST, K tile 0): the first k-step takes the bias registers as its C operand
    auto mma_tile_dma = [&](int next, auto FIRST) {   // steady state: the tile to request always exists, no branch around the pieces
        constexpr bool first = decltype(FIRST)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], (first && ks == 0) ? bC[ni] : acc[ni][mi]);
                __builtin_amdgcn_sched_barrier(0);   // one piece after every fourth MFMA (hipcc otherwise moves the asm
                if (!(G_ABL & 16) || next < 6) dma_piece(next, 2 * ks + ni);   // statements to the head of the slot, three of them behind the first MFMA)
                __builtin_amdgcn_sched_barrier(0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    // own DMAs of tile kt+1 have landed; `ahead` later tiles of this wave may stay in flight (4 instructions per tile)
    auto wait_ahead = [&](int ahead) {
        if (ahead >= 2) g_wait_vm<8>();
        else if (ahead == 1) g_wait_vm<4>();
        else g_wait_vm<0>();
    };
    auto slot_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads have returned
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- HALF tiles (PERSIST): 128 rows x 256 columns on the same eight waves: group g owns rows 64 g .. + 64 (two 32-row accumulators
    // per column block: acc[ni][0..1]), 8 MFMAs per slot and wave, three LDS-DMA pieces per K tile and wave (X: 16 rows, W: 2 x 16)
    int fbh_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fbh_off[ks] = g_off(grp * 64 + l31, 2 * ks + lh);
    auto dma_piece_h = [&](int kt, int j) {   // j = 0: the wave's X rows; 1, 3: its W rows as in a whole tile
        const uint32_t dst = lds0 + (uint32_t)((kt & (G_NST - 1)) * G_STAGE + ((j & 1) ? wave * 2048 + 16384 + (j >> 1) * 1024 : wave * 1024));
        const char* base = uni((j & 1) ? Wb + kt * 64 : Xb + xk(kt) * 64);
        const uint32_t voff = (j & 1) ? wvo[j >> 1] : xvo[0];
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(dst), "s"(base) : "memory", "m0");
    };
    auto dma_tile_h = [&](int kt) {
        dma_piece_h(kt, 0);
        dma_piece_h(kt, 1);
        dma_piece_h(kt, 3);
    };
    auto read_tile_h = [&](int kt) {
        const char* st = smem + (kt & (G_NST - 1)) * G_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fa[ks][ni] = *(const v8*)(st + fa_off[ks] + ni * 2048);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fb[ks][mi] = *(const v8*)(st + fbh_off[ks] + mi * 2048);
        }
    };
    auto mma_tile_h = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], acc[ni][mi]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto mma_tile_dma_h = [&](int next, auto FIRST) {
        constexpr bool first = decltype(FIRST)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], (first && ks == 0) ? bC[ni] : acc[ni][mi]);
                if (2 * ks + ni != 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma_piece_h(next, 2 * ks + ni);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    auto wait_ahead_h = [&](int ahead) {   // three instructions per K tile
        if (ahead >= 2) g_wait_vm<6>();
        else if (ahead == 1) g_wait_vm<3>();
        else g_wait_vm<0>();
    };

    // ---- prologue: three tiles in flight; the one-tile form keeps the tile's bias row in LDS beside the ring (the
    // epilogue read it from global memory once per accumulator quad: a dependent L2 round trip at the head of every output burst)
    if (!PERSIST && tid < 64) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        const int n = n0 + tid * 4;
        if (p.bias != nullptr && n < p.N) bv = *(const f32x4*)(p.bias + n);   // N % 8 == 0: a quad is inside or outside as a whole
        *(f32x4*)(smem + G_RING + tid * 16) = bv;
    }
    if constexpr (PERSIST) dma_bias(n0);
    if (PERSIST && n_full == 0) {
        dma_tile_h(0);
        dma_tile_h(1);
        dma_tile_h(2);
    } else {
        dma_tile(0);
        if (nk > 1) dma_tile(1);
        if (nk > 2) dma_tile(2);
    }

    if constexpr (PERSIST) {
        // ================= PERSIST: one workgroup per CU walks its tiles (host: nk >= 4, 16-bit output) =================
        // What the one-tile form leaves exposed per tile (tools/gemm_stamps.py, profiles/r04_gemm_stamps.txt: 7-11 us beside a K loop of
        // 16-33 us at K = 640 / 1 280): the ring fill of the next workgroup (2-3 us), the gap between two workgroups on a CU (1.3 us),
        // the staging barrier and the store phase (stores issued only after the WHOLE tile is staged, acknowledged before the
        // workgroup may end).  Here: (1) the next tile's first three K tiles are requested BEFORE the epilogue, into ring stages nobody
        // reads any more; (2) the epilogue needs no workgroup barrier and no ring space: every wave transposes its own 32 x 64 blocks
        // through a private 4 KiB LDS buffer (32 rows x 128 B, 16-byte chunks XOR-swizzled by the row) and stores 8 rows x 128 B per
        // instruction as soon as a block is read back; (3) nobody waits for a store: vmcnt counts in order on gfx9, the next
        // counted wait of the K loop covers them; (4) the accumulators are never cleared (FIRST above).
        // Barriers per tile: one in front of the first fragment read + two per K tile; group 1's last multiply slot ends without one
        // (group 0 is in its epilogue by then), which keeps the two groups' barrier counts equal: 2 nk + 1 each.
        const int nk_main = nk - 3;
        char* const P = smem + G_RING + wave * 4096;
        const int wbase = l31 * 128 + lh * 8, x7 = (l31 & 7) << 4;                          // block write: + ((16-byte chunk << 4) ^ x7)
        const int rbase = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);            // block read: + i * 1024 (rows lane/8 + 8 i)
        const int gwbase = l31 * 64 + lh * 4, gx3 = (l31 & 3) << 4;                         // GEGLU: 32 rows x 64 B
        const int grbase = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 2) & 3)) << 4);      // GEGLU read: + i * 1024 (rows lane/4 + 16 i)
        // One tile: HALF = false: a whole tile; true: a half tile (the helpers with the _h suffix, MI = 2 row blocks per group).
        // t_next / h_next (< 0: whole) = the tile whose first pieces this tile's epilogue requests; has_next false: none.
        auto run_tile = [&](auto HALF, const bool has_next, const int t_next, const int h_next) {
            constexpr bool half = decltype(HALF)::value;
            constexpr int MI = half ? 2 : 4;
            // own pieces of K tile 0 (tiles 1, 2 and, from the second tile on, the last stores may stay in flight)
            if constexpr (half || PAIRED) g_wait_vm<6>();      // (PAIRED: K tile 1 is two pieces per wave, K tile 2 four)
            else g_wait_vm<8>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#if G_STAMP
            stamp_lid = t_cur;
            if (threadIdx.x == 0 && stamp_lid < 16384) g_stamp_buf[stamp_lid * 8 + 7] = stamp_hw;
            G_STAMP_AT(0);
            G_STAMP_AT(1);
            const unsigned long long stamp_c0 = __builtin_readcyclecounter();
#endif
            auto rd = [&](int kt) { if constexpr (half) read_tile_h(kt); else read_tile(kt); };
            auto mm = [&]() { if constexpr (half) mma_tile_h(); else mma_tile(-1); };
            auto mmd = [&](int next, auto FIRST) { if constexpr (half) mma_tile_dma_h(next, FIRST); else mma_tile_dma(next, FIRST); };
            auto wa = [&](int ahead) { if constexpr (half) wait_ahead_h(ahead); else wait_ahead(ahead); };
#if G_DMA_SPLIT
            // The pieces of K tile kt + 3 in BOTH slots of tile kt: the activation pieces behind the fragment reads of the read slot, the
            // weight pieces between the MFMAs of the multiply slot.  A piece costs its wave ~130 issue cycles: four of them beside 16 MFMAs
            // (128 issue cycles) or beside 12 fragment reads overrun the 512 cycles the partner's MFMAs take; two and two stay inside.
            // Exception: group 0's first read slot of a tile issues nothing (group 1 still fetches its bias rows from stage 3): all four
            // pieces of K tile 3 in its first multiply slot.
            auto dmx = [&](int kt) {
                if constexpr (half) dma_piece_h(kt, 0);
                else { dma_piece(kt, 0); dma_piece(kt, 2); }
            };
            auto mmw = [&](int next, auto FIRST, auto ALL) {   // the MFMAs of a tile with the weight pieces (ALL: every piece) of tile `next` between them
                constexpr bool first = decltype(FIRST)::value, all = decltype(ALL)::value;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], (first && ks == 0) ? bC[ni] : acc[ni][mi]);
                        const int j = 2 * ks + ni;
                        if ((j & 1) || (all && !(half && j == 2))) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (half) dma_piece_h(next, j); else dma_piece(next, j);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                __builtin_amdgcn_s_setprio(0);
            };
            constexpr int PX = half ? 1 : 2, PT = half ? 3 : 4;   // activation pieces / all pieces per wave and K tile
            if (grp == 0) {
                rd(0);
                read_bias();
                slot_end();
                mmw(3, std::true_type{}, std::true_type{});
                wa(2);
                slot_end();
                for (int kt = 1; kt < nk_main; ++kt) {
                    rd(kt);
                    dmx(kt + 3);
                    slot_end();
                    mmw(kt + 3, std::false_type{}, std::false_type{});
                    wa(2);
                    slot_end();
                }
                for (int kt = nk_main; kt < nk; ++kt) {
                    rd(kt);
                    slot_end();
                    mm();
                    wa(max(nk - 2 - kt, 0));
                    slot_end();
                }
            } else {
                read_bias();
                slot_end();
                rd(0);
                dmx(3);
                g_wait_vm<PT + PX>();      // K tile 1 has landed; tile 2 and the activation pieces of tile 3 may stay in flight
                slot_end();
                mmw(3, std::true_type{}, std::false_type{});
                slot_end();
                for (int kt = 1; kt < nk_main; ++kt) {
                    rd(kt);
                    dmx(kt + 3);
                    g_wait_vm<PT + PX>();
                    slot_end();
                    mmw(kt + 3, std::false_type{}, std::false_type{});
                    slot_end();
                }
                for (int kt = nk_main; kt < nk; ++kt) {
                    rd(kt);
                    wa(max(nk - 2 - kt, 0));
                    slot_end();
                    mm();
                    if (kt != nk - 1) slot_end();
                }
            }
#elif G_DMA_IN_READ
            // The LDS-DMA pieces of K tile kt + 3 are issued in the READ slot of tile kt, behind the fragment reads (their issue -- 60-180
            // cycles each -- then runs under the LDS latency the reading wave waits for anyway, and the multiply slot is 16 bare MFMAs):
            // with the pieces between the MFMAs the K loop ran at 83 % of the matrix pipe's rate and at 100 % without them
            // (-DG_ABL=16, tools/gemm_stamps.py).  The stage of tile kt + 3 held tile kt - 1, whose last fragment reads (group 1's,
            // in the slot before) returned before the barrier that opened this slot.  Exception: group 0 requests K tile 3 at the head
            // of its first MULTIPLY slot -- in its first read slot group 1 still fetches its bias rows from that stage.
            auto dm = [&](int kt) { if constexpr (half) dma_tile_h(kt); else dma_tile(kt); };
            auto mmf = [&]() {   // K tile 0: the bias rows as C operand of the first k-step
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], ks == 0 ? bC[ni] : acc[ni][mi]);
                __builtin_amdgcn_s_setprio(0);
            };
            if (grp == 0) {
                rd(0);
                read_bias();
                slot_end();
                dm(3);
                mmf();
                wa(2);
                slot_end();
                for (int kt = 1; kt < nk_main; ++kt) {
                    rd(kt);
                    dm(kt + 3);
                    slot_end();
                    mm();
                    wa(2);
                    slot_end();
                }
                for (int kt = nk_main; kt < nk; ++kt) {
                    rd(kt);
                    slot_end();
                    mm();
                    wa(max(nk - 2 - kt, 0));
                    slot_end();
                }
            } else {
                read_bias();
                slot_end();
                rd(0);
                dm(3);
                wa(2);
                slot_end();
                mmf();
                slot_end();
                for (int kt = 1; kt < nk_main; ++kt) {
                    rd(kt);
                    dm(kt + 3);
                    wa(2);
                    slot_end();
                    mm();
                    slot_end();
                }
                for (int kt = nk_main; kt < nk; ++kt) {
                    rd(kt);
                    wa(max(nk - 2 - kt, 0));
                    slot_end();
                    mm();
                    if (kt != nk - 1) slot_end();
                }
            }
#else
            if constexpr (PAIRED) {
                // ---- PAIRED K loop (see PAIRED above).  Same slots, same barriers, same ring as the loop below; K' tile kt = 2 p + o:
                //   o = 0: four pieces (X(p) | W_lo(p)), 12 fragment reads;   o = 1: two pieces (W_hi(p)), 4 fragment reads, fb kept from tile 2 p.
                // Pieces in flight behind tile kt + 1: tiles kt + 2, kt + 3 = 6 pieces whatever the parity (group 0: vmcnt(6)); tile kt + 2
                // alone = 4 behind an even kt, 2 behind an odd one (group 1, compile-time by a loop unrolled over the pair).  nk is even and
                // >= 4, nk_main = nk - 3 is odd: the steady-state trips kt = 1 .. nk - 4 are whole (odd, even) pairs.
                auto rd_w = [&](int kt) {           // an odd tile: its weight fragments only
                    const char* st = smem + (kt & (G_NST - 1)) * G_STAGE;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) fa[ks][ni] = *(const v8*)(st + fa_off[ks] + ni * 2048);
                };
                // the 16 MFMAs of the tile in registers with the pieces of tile `next` between them: all four (NEXT_W = false: an even
                // tile) or its two weight pieces (an odd one)
                auto mmp = [&](int next, auto FIRST, auto NEXT_W) {
                    constexpr bool first = decltype(FIRST)::value, next_w = decltype(NEXT_W)::value;
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ks][ni], fb[ks][mi], (first && ks == 0) ? bC[ni] : acc[ni][mi]);
                            if (!next_w || ((2 * ks + ni) & 1)) {
                                __builtin_amdgcn_sched_barrier(0);
                                dma_piece(next, 2 * ks + ni);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    __builtin_amdgcn_s_setprio(0);
                };
                auto wait_pieces = [&](int n) {     // tail only: n in {0, 2, 4, 6}
                    if (n >= 6) g_wait_vm<6>();
                    else if (n >= 4) g_wait_vm<4>();
                    else if (n >= 2) g_wait_vm<2>();
                    else g_wait_vm<0>();
                };
                auto pcs = [&](int kt) { return kt >= nk ? 0 : ((kt & 1) ? 2 : 4); };   // pieces of tile kt per wave
                if (grp == 0) {
                    read_tile(0);
                    read_bias();
                    slot_end();
                    mmp(3, std::true_type{}, std::true_type{});
                    g_wait_vm<6>();
                    slot_end();
                    for (int kt = 1; kt < nk_main; kt += 2) {
                        rd_w(kt);
                        if (G_PAIR_XREAD) {
                            dma_piece(kt + 3, 0);
                            dma_piece(kt + 3, 2);
                        }
                        slot_end();
                        mmp(kt + 3, std::false_type{}, std::integral_constant<bool, G_PAIR_XREAD != 0>{});
                        g_wait_vm<6>();
                        slot_end();
                        read_tile(kt + 1);
                        slot_end();
                        mmp(kt + 4, std::false_type{}, std::true_type{});
                        g_wait_vm<6>();
                        slot_end();
                    }
                    for (int kt = nk_main; kt < nk; ++kt) {
                        if (kt & 1) rd_w(kt); else read_tile(kt);
                        slot_end();
                        mma_tile(-1);
                        wait_pieces(pcs(kt + 2) + pcs(kt + 3));     // tile kt + 1 landed (nothing beyond nk - 1 was requested)
                        slot_end();
                    }
                } else {
                    read_bias();
                    slot_end();
                    read_tile(0);
                    g_wait_vm<4>();                                  // tile 1 landed; tile 2 (four pieces) may stay in flight
                    slot_end();
                    mmp(3, std::true_type{}, std::true_type{});
                    slot_end();
                    for (int kt = 1; kt < nk_main; kt += 2) {
                        rd_w(kt);
                        if (G_PAIR_XREAD) {
                            dma_piece(kt + 3, 0);
                            dma_piece(kt + 3, 2);
                            g_wait_vm<4>();                          // tile kt + 1 landed; tile kt + 2 (odd: two pieces) and these two in flight
                        } else {
                            g_wait_vm<2>();                          // tile kt + 1 landed; tile kt + 2 (odd: two pieces) in flight
                        }
                        slot_end();
                        mmp(kt + 3, std::false_type{}, std::integral_constant<bool, G_PAIR_XREAD != 0>{});
                        slot_end();
                        read_tile(kt + 1);
                        g_wait_vm<4>();                              // tile kt + 2 landed; tile kt + 3 (even: four pieces) in flight
                        slot_end();
                        mmp(kt + 4, std::false_type{}, std::true_type{});
                        slot_end();
                    }
                    for (int kt = nk_main; kt < nk; ++kt) {
                        if (kt & 1) rd_w(kt); else read_tile(kt);
                        wait_pieces(pcs(kt + 2));                    // this group has requested up to tile min(kt + 2, nk - 1)
                        slot_end();
                        mma_tile(-1);
                        if (kt != nk - 1) slot_end();
                    }
                }
            } else
            if (grp == 0) {
                rd(0);
                read_bias();
                slot_end();
                mmd(3, std::true_type{});
                wa(2);
                slot_end();
                for (int kt = 1; kt < nk_main; ++kt) {
                    rd(kt);
                    slot_end();
                    mmd(kt + 3, std::false_type{});
                    wa(2);
                    slot_end();
                }
                for (int kt = nk_main; kt < nk; ++kt) {
                    rd(kt);
                    slot_end();
                    mm();
                    wa(min(nk - 1, kt + 3) - (kt + 1));
                    slot_end();
                }
            } else {
                read_bias();     // (before group 0's first multiply slot, which requests K tile 3 into the stage that holds the bias rows)
                slot_end();
                rd(0);
                wa(1);
                slot_end();
                mmd(3, std::true_type{});
                slot_end();
                for (int kt = 1; kt < nk_main; ++kt) {
                    rd(kt);
                    wa(1);
                    slot_end();
                    mmd(kt + 3, std::false_type{});
                    slot_end();
                }
                for (int kt = nk_main; kt < nk; ++kt) {
                    rd(kt);
                    wa(min(nk - 1, kt + 2) - (kt + 1));
                    slot_end();
                    mm();
                    if (kt != nk - 1) slot_end();
                }
            }
#endif
            G_STAMP_AT(2);
#if G_STAMP
            if (threadIdx.x == 0 && stamp_lid < 16384) g_stamp_buf[stamp_lid * 8 + 6] = __builtin_readcyclecounter() - stamp_c0;
#endif
            // ---- tile end.  Order of the vector-memory instructions of a wave: residual pieces (16) | next tile's K tiles 0..2 (12) |
            // [block stores | next tile's bias quads (8) | block stores].  The residual loads are asm statements with ONE counted wait
            // behind the transposition of all four blocks (vmcnt(12): the ring pieces stay in flight; no store has been issued by
            // then): hipcc's own wait insertion does not see the LDS-DMA statements and would count them short, i.e. wait for the
            // ring in front of every residual use.  vmcnt counts in order on gfx9, stores included.
            const int m0e = m0, n0e = n0;
            if constexpr (WIDE == 2) {
                // ---- fp32 out + fp32 residual (round 5: ff.net.2, to_out, proj_out of the tolerance-compliant composition; the one-tile
                // form left 25-30 % of such a tile outside its K loop).  The wave's eight 32 x 32 fp32 half-blocks (4 KiB: the private
                // buffer's size) go accumulators -> P -> registers as whole 128-byte rows, 8 rows per store instruction; the residual
                // pieces of half-block hb + 1 are requested before half-block hb is read back.  The next tile's ring pieces are requested
                // FIRST: every later wait for a residual piece (a load the compiler sees and counts) then covers them -- vmcnt retires
                // in order -- and they have the whole epilogue to land in.
                if (has_next) {
                    setup_tile(t_next, h_next);
                    dma_bias(n0);
                    if (h_next < 0) { dma_tile(0); dma_tile(1); dma_tile(2); }
                    else { dma_tile_h(0); dma_tile_h(1); dma_tile_h(2); }
                }
                const int rrow = lane >> 3, rchunk = lane & 7;
                const int wb32 = l31 * 128, x7w = l31 & 7;
                const int rb32 = rrow * 128 + ((rchunk ^ (rrow & 7)) << 4);        // + i * 1024: rows rrow + 8 i (same row & 7)
                const int ncol = n0e + wn * 64 + 4 * rchunk;                        // + 32 ni
                const int mrow = m0e + grp * (32 * MI) + rrow;                      // + 32 mi + 8 i
                f32x4 rvp[G_RES_AHEAD + 1][4];
                auto load_res = [&](int hb, f32x4 (&r)[4]) {
                    const int n = ncol + (hb & 1) * 32;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {   // rows / columns past the end re-read a valid address (never stored)
                        const int m = mrow + (hb >> 1) * 32 + 8 * i;
                        r[i] = *(const f32x4*)((const float*)p.residual + (int64_t)min(m, p.M - 1) * p.N_out + (n < p.N_out ? n : 0));
                    }
                };
#pragma unroll
                for (int a = 0; a < G_RES_AHEAD; ++a) load_res(a, rvp[a]);
#pragma unroll
                for (int hb = 0; hb < 2 * MI; ++hb) {
                    const int mi = hb >> 1, ni = hb & 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g + e] * p.alpha;
                        *(f32x4*)(P + wb32 + (((2 * g + lh) ^ x7w) << 4)) = v;
                    }
                    if (hb + G_RES_AHEAD < 2 * MI) load_res(hb + G_RES_AHEAD, rvp[(hb + G_RES_AHEAD) % (G_RES_AHEAD + 1)]);
                    const int n = ncol + ni * 32;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 o = *(const f32x4*)(P + rb32 + i * 1024);
                        const int m = mrow + mi * 32 + 8 * i;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += p.beta * rvp[hb % (G_RES_AHEAD + 1)][i][e];
                        if (m < p.M && n < p.N_out && !((G_ABL & 1) && p.M > 0)) *(f32x4*)((float*)p.out + (int64_t)m * p.N_out + n) = o;
                    }
                }
            } else
            // ---- every wave: its four 32-row blocks through the private buffer.  ONE dispatch on (activation, alpha == 1, residual)
            // around everything that follows, so that the 64 residual registers exist in the residual variants only.
            {
                constexpr int act = PV & 3;
                constexpr bool alpha1 = (PV & 4) != 0;
                constexpr bool res = (PV & 8) != 0;
                // Residual variants: the 16 pieces of a lane in two halves of 8 (the second behind block 1, when 64 accumulator
                // registers have been released), the next tile's ring pieces behind them: acc + residual + transposed blocks <= 192.
                u32x4 rv[res ? 4 * MI : 1];
                auto load_res = [&](int part) {
                    const int nr = n0e + wn * 64 + (lane & 7) * 8;
#pragma unroll
                    for (int c = 2 * MI * part; c < 2 * MI * part + 2 * MI; ++c) {
                        const int m = m0e + grp * (32 * MI) + (c >> 2) * 32 + (c & 3) * 8 + (lane >> 3);
                        // rows / columns past the end re-read a valid address (never stored)
                        rv[res ? c : 0] = *(const u32x4*)((const TO*)p.residual + (int64_t)min(m, p.M - 1) * p.N_out + (nr < p.N_out ? nr : 0));
                    }
                };
                auto prefetch_next = [&]() {
                    if (has_next) {
                        setup_tile(t_next, h_next);
                        dma_bias(n0);
                        if (h_next < 0) {
                            dma_tile(0);
                            dma_tile(1);
                            dma_tile(2);
                        } else {
                            dma_tile_h(0);
                            dma_tile_h(1);
                            dma_tile_h(2);
                        }
                    }
                };
                if constexpr (res) load_res(0);
                else prefetch_next();
                constexpr int NCH = act == 2 ? 2 : 4;          // 16-byte chunks per lane and block
                constexpr int RSTEP = act == 2 ? 16 : 8;       // rows between two chunks of a lane
                // stores: non-GEGLU lane -> rows lane/8 + 8 i (+ 32 mi), channels 8 (lane % 8) .. + 8 of the wave's 64 (8 rows x 128 B per
                // instruction); GEGLU lane -> rows lane/4 + 16 i, output channels 8 (lane % 4) .. + 8 of the wave's 32 (16 rows x 64 B)
                const int nn = act == 2 ? (n0e >> 1) + wn * 32 + (lane & 3) * 8 : n0e + wn * 64 + (lane & 7) * 8;
                const int r0 = m0e + grp * (32 * MI) + (act == 2 ? (lane >> 2) : (lane >> 3));
                TO* const op = (TO*)p.out + (int64_t)r0 * p.N_out + nn;
                const int64_t rstep = (int64_t)RSTEP * p.N_out;
                u32x4 o[res ? 4 * MI : NCH];
                auto store_block = [&](int mi, const u32x4* ob) {
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int m = r0 + mi * 32 + i * RSTEP;
                        if (nn < p.N_out && m < p.M && !((G_ABL & 1) && p.M > 0)) {
                            u32x4 val = ob[i];
                            if constexpr (res) {
                                float f[8], rf[8];
                                unpack8<TO>(val, f);
                                unpack8<TO>(rv[mi * 4 + i], rf);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += p.beta * rf[e];
                                val = pack8<TO>(f);
                            }
                            *(u32x4*)(op + (mi * (32 / RSTEP) + i) * rstep) = val;
                        }
                    }
                };
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g + e];
                            if constexpr (act == 2) {   // channels are (value, gate) interleaved: 2 outputs per quad
#if G_GELU_PACKED
                                const f32x2_t gl = gelu_erf2_f((f32x2_t){v[1], v[3]});
                                const f32x2_t pr = ((f32x2_t){v[0], v[2]} * p.alpha) * gl;
                                TO o2[2] = {(TO)pr[0], (TO)pr[1]};
#else
                                TO o2[2] = {(TO)(p.alpha * v[0] * gelu_erf_f(v[1])), (TO)(p.alpha * v[2] * gelu_erf_f(v[3]))};
#endif
                                uint32_t packed;
                                __builtin_memcpy(&packed, o2, 4);
                                *(uint32_t*)(P + gwbase + (((ni * 2 + (g >> 1)) << 4) ^ gx3) + (g & 1) * 8) = packed;
                            } else {
                                if constexpr (act == 1) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                                }
                                v4 ov;
#pragma unroll
                                for (int e = 0; e < 4; ++e) ov[e] = alpha1 ? (TO)v[e] : (TO)(v[e] * p.alpha);
                                *(v4*)(P + wbase + (((ni * 4 + g) << 4) ^ x7)) = ov;
                            }
                        }
                    u32x4* const ob = res ? o + mi * 4 : o;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) ob[i] = *(const u32x4*)(P + (act == 2 ? grbase : rbase) + i * 1024);
                    if constexpr (!res) {
                        store_block(mi, ob);
                    } else if (mi == MI / 2 - 1) {
                        load_res(1);
                    }
                }
                if constexpr (res) {
                    // Residual variants: the 16 pieces are loads hipcc sees (an asm load is a register the compiler may copy before the
                    // data has arrived: it did, in front of a tied wait statement), so its own wait sits in front of their first use
                    // and counts only what it sees.  The next tile's pieces are therefore requested BEHIND an explicit wait for
                    // the residual (nothing else is in flight at that point) and before the add-and-store pass, whose length
                    // they have to land in; the compiler's later waits then only ask for ring pieces the K loop is about to need.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    prefetch_next();
                    G_STAMP_AT(3);
                    // a full tile takes a branch-free path: in straight-line code hipcc's vmcnt bookkeeping is exact (the residual piece
                    // a store needs, with the younger residual pieces, the 12 ring pieces and the stores issued so far left in flight);
                    // behind the per-row guards of the edge path it falls back to vmcnt(0), i.e. waits for the ring pieces
                    if (m0e + 64 * MI <= p.M && n0e + 256 <= p.N_out && !(G_ABL & 1)) {
#pragma unroll
                        for (int c = 0; c < 4 * MI; ++c) {
                            float f[8], rf[8];
                            unpack8<TO>(o[c], f);
                            unpack8<TO>(rv[c], rf);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += p.beta * rf[e];
                            *(u32x4*)(op + ((c >> 2) * 4 + (c & 3)) * rstep) = pack8<TO>(f);
                        }
                    } else {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) store_block(mi, o + mi * 4);
                    }
                }
            }
            G_STAMP_AT(4);
            G_STAMP_AT(5);
        };
        // this workgroup's list: n_full whole tiles, then (last round of the XCD's run) possibly one half tile
        for (int i = 0; i < n_full; ++i) {
            const bool more = i + 1 < n_full;
            run_tile(std::false_type{}, more || t_half >= 0, more ? t_cur + t_step : t_half, more ? -1 : h_half);
            t_cur += t_step;
        }
        if (t_half >= 0) {
            t_cur = t_half;
            run_tile(std::true_type{}, false, 0, -1);
        }
        return;
    }

    if (nk > 2) g_wait_vm<8>();
    else if (nk > 1) g_wait_vm<4>();
    else g_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    G_STAMP_AT(1);
#if G_STAMP
    if (threadIdx.x == 0 && stamp_lid < 16384) g_stamp_buf[stamp_lid * 8 + 7] = stamp_hw;
    const unsigned long long stamp_c0 = __builtin_readcyclecounter();
#endif

    // ---- 2 nk + 1 slots; group 0 reads in even slots and multiplies in odd ones, group 1 runs one slot behind.
    // Tile kt+3 is requested in the multiply slot of tile kt (its stage held tile kt-1, whose last reads returned
    // before the barrier that opened the slot).  Before the barrier that precedes anybody's read of tile kt+1 every
    // wave has waited for its own pieces of it: group 0 at the end of its multiply slot (tiles kt+2, kt+3 behind it),
    // group 1 at the end of its read slot (only tile kt+2 behind it: it requests kt+3 one slot later).
    const int nk_main = G_ASMDMA ? max(nk - 3, 0) : 0;   // tiles whose multiply slot requests tile kt + 3
#if G_ONEBAR
    // ONE barrier per K tile (experiment; the scheme of attn_d64c): between two barriers a wave reads the fragments of one tile (R)
    // and multiplies one tile (M); group 0 runs  M(kt) R(kt+1) | barrier,  group 1  R(kt) M(kt) | barrier,  so the two waves of a
    // SIMD still alternate on the matrix pipe, but without the mid-tile barrier whose release latency was a bubble of the pipe
    // twice per tile, and a wave stalled on the issue of its LDS-DMA pieces no longer idles the pipe when its partner is
    // multiplying.  Both groups run the same loop body  R(kt) [barrier if group 0] M(kt) [barrier if group 1]  (group 0's loop
    // is rotated by one segment).  Barrier #kt: tile kt + 1 has landed in every wave's view (group 0 then has requested up to
    // tile kt + 2, group 1 up to tile kt + 3); the stage tile kt + 3 overwrites held tile kt - 1, whose reads returned before
    // barrier #(kt - 1).
    for (int kt = 0; kt < nk_main; ++kt) {
        read_tile(kt);
        if (grp == 0) {
            g_wait_vm<4>();
            slot_end();
        }
        mma_tile_dma(kt + 3, std::false_type{});
        if (grp == 1) {
            g_wait_vm<8>();
            slot_end();
        }
    }
    for (int kt = nk_main; kt < nk; ++kt) {
        read_tile(kt);
        if (grp == 0) {
            wait_ahead(min(nk - 1, kt + 2) - (kt + 1));
            slot_end();
        }
        mma_tile(kt + 3 < nk ? kt + 3 : -1);
        if (grp == 1) {
            wait_ahead(min(nk - 1, kt + 3) - (kt + 1));
            slot_end();
        }
    }
#else
    if (grp == 0) {
        for (int kt = 0; kt < nk_main; ++kt) {              // steady state: two tiles stay in flight behind tile kt + 1
            read_tile(kt);
            slot_end();
            mma_tile_dma(kt + 3, std::false_type{});
            g_wait_vm<8>();
            slot_end();
        }
        for (int kt = nk_main; kt < nk; ++kt) {
            read_tile(kt);
            slot_end();
            mma_tile(kt + 3 < nk ? kt + 3 : -1);
            wait_ahead(min(nk - 1, kt + 3) - (kt + 1));
            slot_end();
        }
        slot_end();
    } else {
        slot_end();
        for (int kt = 0; kt < nk_main; ++kt) {              // steady state: one tile stays in flight behind tile kt + 1
            read_tile(kt);
            g_wait_vm<4>();
            slot_end();
            mma_tile_dma(kt + 3, std::false_type{});
            slot_end();
        }
        for (int kt = nk_main; kt < nk; ++kt) {
            read_tile(kt);
            wait_ahead(min(nk - 1, kt + 2) - (kt + 1));
            slot_end();
            mma_tile(kt + 3 < nk ? kt + 3 : -1);
            slot_end();
        }
    }

#endif

    G_STAMP_AT(2);
#if G_STAMP
    if (threadIdx.x == 0 && stamp_lid < 16384) g_stamp_buf[stamp_lid * 8 + 6] = __builtin_readcyclecounter() - stamp_c0;
#endif
#if G_ABL & 2
    if (p.M > 0) return;   // diagnostic build: no epilogue at all (the accumulators stay live for the compiler)
#endif

    if constexpr (WIDE != 0) {
        // ---- SEG epilogue (fp32 / planes out; a 16-bit output takes the plain epilogue below): the 256 x 256 fp32 tile (256 KiB; as planes lo | hi the same bytes) does not fit the 128 KiB ring, so it
        // leaves in two halves of 128 rows = the rows of group 0, then of group 1: the owning group stages bias / activation / alpha
        // in registers -> LDS (16-byte chunks XOR-swizzled by the row), all 512 threads move whole 16-byte chunks out as contiguous
        // runs and add the fp32 residual on that side.  Twice the bytes of the 16-bit epilogue behind three times its K loop.
        const bool geglu = WIDE == 1 && p.act == RSVLD_ACT_GEGLU;
        const int ncol = geglu ? 128 : 256;                       // columns of the stored tile
        const int n_out0 = geglu ? (n0 >> 1) : n0;
        const bool planes = WIDE == 1 && p.out_kind == 2;
        // fp32: row = ncol * 4 bytes = ncol / 4 chunks; planes: row = lo (ncol * 2 bytes) | hi (ncol * 2 bytes) = ncol / 4 chunks as well
        const int rchunks = ncol >> 2;                            // 64 or 32 16-byte chunks per staged row
        const int rbytes = rchunks << 4;
        auto s_off = [&](int row, int chunk) { return row * rbytes + ((chunk ^ (row & 15)) << 4); };
        // The fp32 residual pieces a thread adds on the way out (rows tid / 64 + 8 i of the half, its 16-byte chunk) are requested BEFORE
        // the half is staged, so that their HBM round trip runs under the staging pass and its barrier (round 4 loaded each piece inside
        // the store loop, behind a row test: 16 dependent round trips per half -- ~50 us of a 150 us tile at K = 1 280).  The residual
        // never comes with GEGLU or a planes output: rchunks = 64, 16 rows per thread.
        constexpr bool has_res = WIDE == 2;
        f32x4 rvw[has_res ? 16 : 1];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            if constexpr (has_res) {
                const int oc = n_out0 + (tid & 63) * 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = m0 + half * 128 + (tid >> 6) + 8 * i;
                    rvw[has_res ? i : 0] = (m < p.M && oc < p.N_out) ? *(const f32x4*)((const float*)p.residual + (int64_t)m * p.N_out + oc)
                                                                     : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            if (grp == half) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = wn * 64 + ni * 32 + 8 * g + 4 * lh;                    // tile column of the quad
                        const f32x4 bq = *(const f32x4*)(smem + G_RING + col * 4);             // zeros without a bias / past N
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            const int row = mi * 32 + l31;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g + e] + bq[e];
                            if (geglu) {   // (value, gate) interleaved: 2 outputs per quad, output column col / 2
                                const float o0 = p.alpha * v[0] * gelu_erf_f(v[1]), o1 = p.alpha * v[2] * gelu_erf_f(v[3]);
                                const int oc = col >> 1;
                                if (planes) {
                                    const bf16 h0 = (bf16)o0, h1 = (bf16)o1;
                                    const bf16 lo2[2] = {(bf16)(o0 - (float)h0), (bf16)(o1 - (float)h1)}, hi2[2] = {h0, h1};
                                    uint32_t pl, ph;
                                    __builtin_memcpy(&pl, lo2, 4);
                                    __builtin_memcpy(&ph, hi2, 4);
                                    *(uint32_t*)(smem + s_off(row, oc >> 3) + (oc & 7) * 2) = pl;
                                    *(uint32_t*)(smem + s_off(row, (rchunks >> 1) + (oc >> 3)) + (oc & 7) * 2) = ph;
                                } else {
                                    *(u32x2*)(smem + s_off(row, oc >> 2) + (oc & 3) * 4) =
                                        (u32x2){__builtin_bit_cast(uint32_t, o0), __builtin_bit_cast(uint32_t, o1)};
                                }
                            } else {
                                if (p.act == RSVLD_ACT_SILU) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                                if (planes) {
                                    bf16x4 h, l;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        h[e] = (bf16)v[e];
                                        l[e] = (bf16)(v[e] - (float)h[e]);
                                    }
                                    *(bf16x4*)(smem + s_off(row, col >> 3) + (col & 7) * 2) = l;
                                    *(bf16x4*)(smem + s_off(row, (rchunks >> 1) + (col >> 3)) + (col & 7) * 2) = h;
                                } else {
                                    *(f32x4*)(smem + s_off(row, col >> 2)) = (f32x4){v[0], v[1], v[2], v[3]};
                                }
                            }
                        }
                    }
            }
            __syncthreads();
            // 128 rows x rchunks chunks; thread -> chunk tid % rchunks of rows tid / rchunks + i * (512 / rchunks)
            const int chunk = tid & (rchunks - 1);
            const int rstep = 512 / rchunks;                      // 8 or 16 rows per pass
            const bool hi_sec = planes && chunk >= (rchunks >> 1);
            // global column (elements) of this chunk: fp32 = 4 columns per chunk; planes = 8 columns per chunk inside its section
            const int ocol = planes ? n_out0 + (chunk - (hi_sec ? (rchunks >> 1) : 0)) * 8 : n_out0 + chunk * 4;
            if constexpr (has_res) {            // rchunks = 64, rstep = 8: compile-time indices into rvw[]
                if (ocol < p.N_out) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = (tid >> 6) + 8 * i;
                        const int m = m0 + half * 128 + row;
                        if (m >= p.M) continue;
                        f32x4 f = *(const f32x4*)(smem + s_off(row, chunk));
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] += p.beta * rvw[has_res ? i : 0][e];
                        *(f32x4*)((float*)p.out + (int64_t)m * p.N_out + ocol) = f;
                    }
                }
            } else if (ocol < p.N_out) {
                for (int row = tid / rchunks; row < 128; row += rstep) {
                    const int m = m0 + half * 128 + row;
                    if (m >= p.M) break;
                    u32x4 v = *(const u32x4*)(smem + s_off(row, chunk));
                    if (planes) {
                        bf16* o = (bf16*)p.out + (int64_t)m * (2 * p.N_out) + (hi_sec ? p.N_out : 0) + ocol;
                        *(u32x4*)o = v;
                    } else {
                        *(f32x4*)((float*)p.out + (int64_t)m * p.N_out + ocol) = __builtin_bit_cast(f32x4, v);
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    // ---- epilogue through LDS (the ring is dead after the last barrier): a lane owns one output row and, per register
    // quad, four consecutive channels, i.e. 8-byte pieces 512 B apart -- stored like that the 128 KiB tile leaves the
    // CU in 16-byte fragments (measured: 17 us per tile, more than the K loop of a K = 640 layer).  So: bias / SiLU /
    // GEGLU / alpha in registers -> 16-bit tile [256 rows][256 (GEGLU: 128) channels] in LDS (16-byte chunks XOR-swizzled
    // by the row) -> every thread moves whole 16-byte chunks, rows leave as contiguous 512-byte runs; the residual is
    // added on that side with equally coalesced loads.
    const bool geglu = p.act == RSVLD_ACT_GEGLU;
    const int row_chunks = geglu ? 16 : 32;                 // 16-byte chunks per tile row
    auto c_off = [&](int row, int chunk) { return row * (row_chunks * 16) + ((chunk ^ (row & (row_chunks - 1))) << 4); };
    // The residual pieces this thread will add on the way out (rows tid/32 + 16 i, its 16-byte chunk) are requested NOW, so
    // that their HBM round trip runs under the staging pass and its barrier instead of after them (the fragment registers
    // are dead here).  Never together with GEGLU (checked on the host).
    const bool has_res = p.residual != nullptr;
    u32x4 rv[16];
    if (has_res) {
        const int nn = n0 + (tid & 31) * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + (tid >> 5) + 16 * i;
            rv[i] = (m < p.M && nn < p.N_out && !(G_ABL & 1)) ? *(const u32x4*)((const TO*)p.residual + (int64_t)m * p.N_out + nn)
                                                              : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // Staging pass, specialised OUTSIDE the unrolled loops on (activation, alpha == 1): with the dispatch inside, every one of
    // the 32 quads of a lane carried two uniform branches, re-read its bias quad from LDS (the staging stores may alias it for
    // the compiler) and recomputed its swizzled address.  The bias quad and the staging offset of a quad depend on (ni, g) only
    // (the row enters the swizzle as row & 31 = l31 for every mi): loops reordered, mi innermost.  Same operations on the same
    // values: bit-identical.
#if G_EPI_SPECIALISED
    {
        auto pass = [&](auto ACT, auto ALPHA1) {
            constexpr int act = decltype(ACT)::value;
            constexpr bool alpha1 = decltype(ALPHA1)::value;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wn * 64 + ni * 32 + 8 * g + 4 * lh;                    // tile column of the quad
                    const f32x4 bq = *(const f32x4*)(smem + G_RING + col * 4);             // zeros without a bias / past N
                    const int oc = col >> 1;                                               // GEGLU: output column inside the 128-wide tile
                    const int so = act == 2 ? (grp * 128 + l31) * 256 + (((oc >> 3) ^ (l31 & 15)) << 4) + (oc & 7) * 2
                                            : (grp * 128 + l31) * 512 + (((col >> 3) ^ l31) << 4) + (col & 7) * 2;
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g + e] + bq[e];
                        if constexpr (act == 2) {   // channels are (value, gate) interleaved: 2 outputs per quad
                            TO o2[2] = {(TO)(p.alpha * v[0] * gelu_erf_f(v[1])), (TO)(p.alpha * v[2] * gelu_erf_f(v[3]))};
                            uint32_t packed;
                            __builtin_memcpy(&packed, o2, 4);
                            *(uint32_t*)(smem + so + mi * (32 * 256)) = packed;
                        } else {
                            if constexpr (act == 1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                            }
                            v4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = alpha1 ? (TO)v[e] : (TO)(v[e] * p.alpha);
                            *(v4*)(smem + so + mi * (32 * 512)) = o;
                        }
                    }
                }
        };
        typedef std::integral_constant<int, 0> A0;
        typedef std::integral_constant<int, 1> A1;
        typedef std::integral_constant<int, 2> A2;
        if (geglu) pass(A2{}, std::false_type{});
        else if (p.act == RSVLD_ACT_SILU) pass(A1{}, std::false_type{});
        else if (p.alpha == 1.0f) pass(A0{}, std::true_type{});
        else pass(A0{}, std::false_type{});
    }
#else
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int row = grp * 128 + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wn * 64 + ni * 32 + 8 * g + 4 * lh;   // tile column of the quad
                float v[4] = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                {
                    const f32x4 bv = *(const f32x4*)(smem + G_RING + col * 4);   // zeros without a bias / past N
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
                if (geglu) {   // channels are (value, gate) interleaved: 2 outputs per quad
                    TO o2[2] = {(TO)(p.alpha * v[0] * gelu_erf_f(v[1])), (TO)(p.alpha * v[2] * gelu_erf_f(v[3]))};
                    uint32_t packed;
                    __builtin_memcpy(&packed, o2, 4);
                    const int oc = col >> 1;   // output column inside the 128-wide tile
                    *(uint32_t*)(smem + c_off(row, oc >> 3) + (oc & 7) * 2) = packed;
                } else {
                    if (p.act == RSVLD_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
                    v4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (TO)(v[e] * p.alpha);
                    *(v4*)(smem + c_off(row, col >> 3) + (col & 7) * 2) = o;
                }
            }
    }
#endif
    __syncthreads();
    G_STAMP_AT(3);
    {
        const int n_tile_out = geglu ? 128 : 256;                         // channels of the stored tile
        const int n_out0 = geglu ? (n0 >> 1) : n0;
        const int chunk = tid & (row_chunks - 1);
        const int rows_per_pass = 512 / row_chunks;
        const int nn = n_out0 + chunk * 8;
        if (has_res) {            // row_chunks = 32, 16 rows per thread, compile-time indices into rv[]
            if (nn < p.N_out) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (tid >> 5) + 16 * i;
                    if (m0 + row >= p.M || ((G_ABL & 1) && p.M > 0)) continue;
                    float f[8], rf[8];
                    unpack8<TO>(*(const u32x4*)(smem + c_off(row, chunk)), f);
                    unpack8<TO>(rv[i], rf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += p.beta * rf[e];
                    *(u32x4*)((TO*)p.out + (int64_t)(m0 + row) * p.N_out + nn) = pack8<TO>(f);
                }
            }
        } else if (nn < p.N_out) {
            if (G_EPI_SPECIALISED && !geglu && m0 + 256 <= p.M && !(G_ABL & 1)) {
                // full tile, 32 chunks per row: thread -> (row r0 + 16 i, chunk): one output pointer advanced by a constant, the
                // staging offset alternates between two precomputed values (row & 31 = (r0 & 15) | 16 (i & 1)): 16 x (read, store,
                // pointer add) instead of 16 x (bounds test, 64-bit multiply-add, swizzle)
                const int r0 = tid >> 5;
                char* o = (char*)((TO*)p.out + (int64_t)(m0 + r0) * p.N_out + nn);
                const int64_t ostep = (int64_t)16 * p.N_out * (int64_t)sizeof(TO);
                const int so0 = r0 * 512 + ((chunk ^ r0) << 4), so1 = (r0 + 16) * 512 + ((chunk ^ (r0 + 16)) << 4);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    *(u32x4*)o = *(const u32x4*)(smem + ((i & 1) ? so1 : so0) + (i >> 1) * (32 * 512));
                    o += ostep;
                }
            } else {
                for (int row = tid / row_chunks; row < 256; row += rows_per_pass) {
                    if (m0 + row >= p.M || ((G_ABL & 1) && p.M > 0)) break;
                    *(u32x4*)((TO*)p.out + (int64_t)(m0 + row) * p.N_out + nn) = *(const u32x4*)(smem + c_off(row, chunk));
                }
            }
        }
        (void)n_tile_out;
    }
#if G_STAMP
    G_STAMP_AT(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G_STAMP_AT(5);
#endif
}

}  // namespace

#if G_STAMP
extern "C" int rsvld_debug_gemm_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamp_buf), bytes < sizeof(g_stamp_buf) ? bytes : sizeof(g_stamp_buf)) == hipSuccess ? 0 : -1;
}
#endif

// Eligibility + launch, called from rsvld_conv2d_nhwc for 1x1 / stride-1 / single-source layers.
// Returns RSVLD_EUNSUPPORTED when the shape should stay on the implicit-GEMM kernel.
namespace {
// one launch helper per KERNEL (a non-type template parameter): the one-time registration of the dynamic LDS size is a
// function-local static of THIS instantiation (a generic lambda's static would be shared by every kernel of one function type)
template <auto KERN, int SMEM>
int gemm_go(dim3 grid, hipStream_t s, const GemmArgs& a) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (attr != hipSuccess) return RSVLD_ELAUNCH;
    hipLaunchKernelGGL(KERN, grid, dim3(512), SMEM, s, a);
    return rsvld_check_launch();
}
}  // namespace

int rsvld_gemm256_try(const rsvld_conv_desc* d, void* stream) {
    const bool split = d->dtype == RSVLD_SPLIT, w2 = d->dtype == RSVLD_F16W2, w1 = d->dtype == RSVLD_F16W1;
    const int seg = split ? 3 : w2 ? 2 : 1;          // K segments (RSVLD_F16W1: one, in the SEG = 4 kernels: the multi-segment family's outputs)
    if (d->tune & RSVLD_TUNE_NO_GEMM256) return RSVLD_EUNSUPPORTED;   // A/B switch
    if (w1 && d->out_f32 != 1) return RSVLD_EINVAL;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad_t != 0 || d->pad_l != 0 || d->upsample) return RSVLD_EUNSUPPORTED;
    if (d->x2 != nullptr || d->Cin2 != 0 || d->rowvec != nullptr || (d->out_f32 && seg == 1 && !w1)) return RSVLD_EUNSUPPORTED;
    if (d->Cin % 32 != 0 || d->Cout % 8 != 0) return RSVLD_EUNSUPPORTED;
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    if (d->H != d->Ho || d->W != d->Wo) return RSVLD_EUNSUPPORTED;
    // eligibility is decided on the rows of ONE of the plan_div stacked units (batch-invariant plan)
    const int64_t Mp = d->plan_div > 1 ? (M + d->plan_div - 1) / d->plan_div : M;
    if (d->Cout < 256 || Mp < 4096) return RSVLD_EUNSUPPORTED;
    const int64_t tiles = ((Mp + 255) / 256) * ((d->Cout + 255) / 256);
    if (tiles < 128) return RSVLD_EUNSUPPORTED;   // one workgroup per CU; measured: from half the chip up it beats the 128x128 kernel
    // 32-bit lane offsets inside a tile: 256 rows of the activation tensor (planes: 2 K per row) / of the weights (pair 2 K, triple 3 K)
    if ((int64_t)256 * d->Cin * 2 * seg >= ((int64_t)1 << 32) || M >= ((int64_t)1 << 31)) return RSVLD_EUNSUPPORTED;
    if (d->act == RSVLD_ACT_GEGLU && (d->Cout % 16 != 0 || d->residual != nullptr)) return RSVLD_EINVAL;
    // RSVLD_SPLIT: the residual is the fp32 stream and goes with the fp32 output only (RSVLD_F16W2: the residual has the output's type)
    if (split && d->out_f32 != 1 && d->residual != nullptr) return RSVLD_EINVAL;
    if ((d->out_f32 == 2 && !split) || d->out_f32 < 0 || d->out_f32 > 2) return RSVLD_EINVAL;
    GemmArgs a;
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.residual = d->residual; a.out = d->out;
    a.M = (int)M; a.N = d->Cout; a.K = d->Cin;
    a.N_out = d->act == RSVLD_ACT_GEGLU ? d->Cout / 2 : d->Cout;
    a.act = d->act; a.alpha = d->alpha; a.beta = d->beta;
    // out_kind: 0 = 16-bit (T, or fp16 from the multi-segment forms), 1 = fp32, 2 = bf16 planes (RSVLD_SPLIT's native output)
    a.out_kind = w1 ? 1 : seg == 1 ? 0 : d->out_f32 == 1 ? 1 : (split && d->out_f32 == 0) ? 2 : 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned nmt = (unsigned)((M + 255) / 256), nnt = (unsigned)((d->Cout + 255) / 256);
    const dim3 grid1(nmt, nnt);
    // the persistent form: 16-bit output, at least four K tiles (its K loop peels tile 0 and requests three tiles ahead)
    const bool wide_res = (seg > 1 || w1) && a.out_kind == 1 && a.residual != nullptr && a.act == RSVLD_ACT_NONE;   // fp32 out + fp32 residual
    if (!(d->tune & RSVLD_TUNE_GEMM_ONE_TILE) && (a.out_kind == 0 || wide_res) && seg * d->Cin >= 128) {
        static const int n_cu = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
            return n;
        }();
        if (n_cu > 0) {
            const dim3 pgrid((unsigned)min((int64_t)nmt * nnt, (int64_t)n_cu));
            const int pv = (a.act == RSVLD_ACT_GEGLU ? 2 : a.act == RSVLD_ACT_SILU ? 1 : 0) | (a.alpha == 1.0f ? 4 : 0) | (a.residual != nullptr ? 8 : 0);
            const bool h = d->dtype == RSVLD_F16;
            if (wide_res) {
                return split ? gemm_go<gemm256_kernel<bf16, 3, true, 0, 2>, G_SMEM_P>(pgrid, s, a)
                       : w1  ? gemm_go<gemm256_kernel<f16, 4, true, 0, 2>, G_SMEM_P>(pgrid, s, a)
                             : gemm_go<gemm256_kernel<f16, 2, true, 0, 2>, G_SMEM_P>(pgrid, s, a);
            } else if (w1) {
                // (fp32 out without a residual, or with an activation: the one-tile form below)
            } else if (seg == 1) {
#define G_PV_CASE(V) case V: return h ? gemm_go<gemm256_kernel<f16, 1, true, V>, G_SMEM_P>(pgrid, s, a) : gemm_go<gemm256_kernel<bf16, 1, true, V>, G_SMEM_P>(pgrid, s, a);
                switch (pv) {
                    G_PV_CASE(0) G_PV_CASE(4) G_PV_CASE(8) G_PV_CASE(12)      // no activation: alpha, alpha == 1, + residual
                    G_PV_CASE(1) G_PV_CASE(5) G_PV_CASE(9) G_PV_CASE(13)      // SiLU
                    G_PV_CASE(2) G_PV_CASE(6)                                 // GEGLU (never with a residual: checked above)
                    default: break;
                }
#undef G_PV_CASE
            } else {   // fp16 out of the multi-segment forms (the transformer blocks' q | k | v and GEGLU layers): no SiLU variants
#define G_PV_CASE(V) case V: return split ? gemm_go<gemm256_kernel<bf16, 3, true, V>, G_SMEM_P>(pgrid, s, a) : gemm_go<gemm256_kernel<f16, 2, true, V>, G_SMEM_P>(pgrid, s, a);
                switch (pv) {
                    G_PV_CASE(0) G_PV_CASE(4) G_PV_CASE(2) G_PV_CASE(6)
                    default: break;
                }
#undef G_PV_CASE
            }
        }
    }
    const int wide = a.out_kind == 0 ? 0 : (a.out_kind == 1 && a.residual != nullptr) ? 2 : 1;
    if (split) return wide == 2 ? gemm_go<gemm256_kernel<bf16, 3, false, 0, 2>, G_SMEM>(grid1, s, a)
                    : wide == 1 ? gemm_go<gemm256_kernel<bf16, 3, false, 0, 1>, G_SMEM>(grid1, s, a)
                                : gemm_go<gemm256_kernel<bf16, 3, false, 0, 0>, G_SMEM>(grid1, s, a);
    if (w1) return wide == 2 ? gemm_go<gemm256_kernel<f16, 4, false, 0, 2>, G_SMEM>(grid1, s, a)
                             : gemm_go<gemm256_kernel<f16, 4, false, 0, 1>, G_SMEM>(grid1, s, a);
    if (w2) return wide == 2 ? gemm_go<gemm256_kernel<f16, 2, false, 0, 2>, G_SMEM>(grid1, s, a)
                 : wide == 1 ? gemm_go<gemm256_kernel<f16, 2, false, 0, 1>, G_SMEM>(grid1, s, a)
                             : gemm_go<gemm256_kernel<f16, 2, false, 0, 0>, G_SMEM>(grid1, s, a);
    return d->dtype == RSVLD_F16 ? gemm_go<gemm256_kernel<f16, 1, false>, G_SMEM>(grid1, s, a)
                                 : gemm_go<gemm256_kernel<bf16, 1, false>, G_SMEM>(grid1, s, a);
}
