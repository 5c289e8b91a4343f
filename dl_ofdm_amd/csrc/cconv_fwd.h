// The N = 64 C-Conv forward (dev/py/complex.py:140-196 as ONE real GEMM, see gemm_f32_mfma.h OP_CCONV_W) as a STAGED
// whole-k tile.
//
// out[rows, 2F] = x[rows, 2kin] . Weff[2kin, 2F] has a short k (160 with the cyclic prefix, 128 without) and 256 output
// tiles of 64x64 at the C2 batch: one tile per CU, four waves, one 32x32 accumulator each -- 80 dependent
// v_mfma_f32_32x32x2_f32 per wave = 2.1 us at 2.4 GHz.  The whole-k form of gemm_f32_mfma.h (NBUF = 1) issues all of a
// block's global loads at once -- one exposed memory latency instead of five -- but then WAITS for all 50 KB of them
// before the first MFMA: at ~11 B/clk/CU of prologue bandwidth that is ~1.9 us during which the matrix pipe idles, and the
// 2.1 us chain starts only afterwards (7.0 us in situ, 24 % of the MFMA peak).
//
// Here the k range is cut into KS stages of 32: every load of the block is still issued up front, in stage order, but a
// stage's registers go to LDS -- and its 16 MFMAs start -- as soon as THAT stage has landed (counted vmcnt waits: the
// compiler sees one straight-line body).  The LDS writes of stage s+1 sit inside the MFMA chain of stage s, so in steady
// state the matrix pipe only ever waits for data that has not arrived yet.  No LDS region is reused: one barrier per stage
// orders "written by all" before "read by all", nothing else.
//
// Same values at the same [column][k] positions of the same LDS layout (row stride K + 4) and the same MFMA order as the
// kernel it replaces => bit-identical output (tests/test_gpu_ops.py::test_cconv_fwd_staged_is_bitwise_the_whole_k_tile).
//
// LDS bank conflicts of the weight tile's stores (20 % of the old kernel's LDS cycles, profiles/r04_pmc_counters.txt): a
// ds_write_b128 is serviced in groups of 8 consecutive lanes; lanes running over 8 consecutive filters write rows 2(K+4)
// floats apart = 8 banks apart (mod 32): lanes 4-7 collide with lanes 0-3.  Lanes 4-7 of every group now take the OTHER
// row pair of the wave (k offset +4 floats: +4 banks), so a group covers 8 distinct 4-bank slots; the two halves of the
// wave swap roles, so each global load instruction still reads the same two 128-byte row segments of w.
#pragma once
#include "gemm_f32_mfma.h"

namespace dccn {

template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}

// WS: MFMA step (0..15) of a stage's chain behind which the LDS stores of the next stage are issued, one per step
// BM x BN: the block's output tile, 64 x 64 (waves 2 x 2) or 32 x 128 (waves 1 x 4: every x row is read by ONE block --
// half the HBM-side traffic of the launch -- and the whole [Wa|Wb], 40 KB that live in the L2, by every block)
template <int KS, int WS, int BM = 64, int BN = 64>
__global__ __launch_bounds__(kGemmThreads) void cconv_fwd_staged_kernel(const GemmParams p0, const ChainOffs co) {
    const GemmParams p = p0.at_chain(co.off[blockIdx.z]);           // chain groups (common.h)
    constexpr int K = 32 * KS, LD = K + 4;
    static_assert((BM == 64 && BN == 64) || (BM == 32 && BN == 128), "tile shapes");
    constexpr int NA = BM * 8 / 256;              // float4 pieces of x per thread and stage
    constexpr int NBU = (BN / 2) * 8 / 256;       // (row pair, filter) units of the weight tile per thread and stage
    constexpr int NP = NA + 2 * NBU;              // LDS stores per thread and stage
    static_assert(WS >= 0 && WS + NP <= 14, "store slots must lie in front of the stage barrier (behind step 13)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // [BM][LD]  x rows, k contiguous
    float* sB = smem + BM * LD;       // [BN][LD]  Weff columns, k contiguous
    stamp_mark(p.stamp, 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = BM == 64 ? (wid >> 1) * 32 : 0, wn0 = BM == 64 ? (wid & 1) * 32 : wid * 32;
    const int ntn = (p.N + BN - 1) / BN;
    const int tile = xcd_tile((int)blockIdx.x, (int)gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    // x tile: piece j of a stage = row (tid + 256 j) / 8, float4 column tid % 8 (8 lanes = 128 contiguous bytes of a row)
    const int k4 = tid & 7;
    const float* a_src[NA];
    unsigned a_lds[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = (tid + 256 * j) >> 3;
        a_src[j] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + 4 * k4;
        a_lds[j] = (unsigned)(row * LD + 4 * k4);
    }
    // weight tile: (row pair, filter) units = Weff rows 4np..4np+3 of the stage x columns 2f', 2f'+1; unit u of a thread
    // takes filter f' = 32 u + (lane & 31)
    const int fl = l31;
    const int np = 2 * wid + (h ^ ((fl >> 2) & 1));
    const float* b_src[NBU];
    unsigned b_lds[NBU];
#pragma unroll
    for (int u = 0; u < NBU; ++u) {
        const int f = min((n0 >> 1) + 32 * u + fl, (p.N >> 1) - 1);
        b_src[u] = p.B + (size_t)(2 * np) * p.ldb + f;                   // Wa[n][f]; Wb at + cF; row n + 1 at + ldb
        b_lds[u] = (unsigned)((2 * (32 * u + fl)) * LD + 4 * np);        // column 2f'; column 2f' + 1 at + LD
    }
    const size_t b_stage = (size_t)16 * p.ldb;                           // 16 rows of w per stage

    // bias of this lane's output column (epilogue), requested first
    const int col = n0 + wn0 + l31;
    float bj = 0.f;
    if (p.bias != nullptr && col < p.N) {
        const float d = p.bias[col >> 1] - p.bias[p.cF + (col >> 1)];
        bj = (col & 1) ? -d : d;
    }
    // (compile-time indices everywhere: a register array indexed by a loop variable that only becomes constant after
    // unrolling ends up in scratch here)
    typedef float ccf_f32x4 __attribute__((ext_vector_type(4)));
    ccf_f32x4 ra[NA * KS];
    float rb[4 * NBU * KS];
    static_for<0, KS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        static_for<0, NA>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            ra[NA * s + j] = *reinterpret_cast<const ccf_f32x4*>(a_src[j] + 32 * s);
        });
        static_for<0, NBU>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const float* q = b_src[u] + s * b_stage;
            rb[4 * (NBU * s + u) + 0] = q[0];
            rb[4 * (NBU * s + u) + 1] = q[p.cF];
            rb[4 * (NBU * s + u) + 2] = q[p.ldb];
            rb[4 * (NBU * s + u) + 3] = q[p.ldb + p.cF];
        });
        __builtin_amdgcn_sched_barrier(0);          // issue order = stage order (the counted waits below rely on it)
    });

    // pieces of a stage: 0..NA-1 = x rows; then per unit: column 2f' = (Wa, -Wb, Wa', -Wb'), column 2f'+1 = (Wb, -Wa, Wb', -Wa')
    // (complex.py:185-188)
    auto store_piece = [&](auto s_, auto piece_) {
        constexpr int s = decltype(s_)::value, piece = decltype(piece_)::value;
        if constexpr (piece < NA) {
            *reinterpret_cast<ccf_f32x4*>(sA + a_lds[piece] + 32 * s) = ra[NA * s + piece];
        } else {
            constexpr int u = (piece - NA) / 2, o = 4 * (NBU * s + u);
            if constexpr (((piece - NA) & 1) == 0)
                *reinterpret_cast<float4*>(sB + b_lds[u] + 32 * s) = make_float4(rb[o], -rb[o + 1], rb[o + 2], -rb[o + 3]);
            else
                *reinterpret_cast<float4*>(sB + b_lds[u] + LD + 32 * s) = make_float4(rb[o + 1], -rb[o], rb[o + 3], -rb[o + 2]);
        }
    };
    static_for<0, NP>([&](auto q_) { store_piece(std::integral_constant<int, 0>{}, q_); });
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* As = sA + (wm0 + l31) * LD + 4 * h;
    const float* Bs = sB + (wn0 + l31) * LD + 4 * h;
    // Per stage: 16 dependent MFMAs (64 cycles each).  Inside the chain: the fragments of the next group of 8 k (one
    // ds_read_b128 per operand) at the head of each group; the next stage's LDS stores behind steps WS.. (they wait,
    // counted, for exactly that stage's loads); the stage barrier behind step 13; the next stage's first fragments
    // behind step 14 -- so a stage boundary exposes neither the LDS round trip nor the barrier.
    float4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const float4*>(As);
    fb[0] = *reinterpret_cast<const float4*>(Bs);
    static_for<0, KS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        static_for<0, 16>([&](auto st_) {
            constexpr int step = decltype(st_)::value, g = step / 4, j = step % 4;
            if constexpr (j == 0 && g + 1 < 4) {
                fa[(g + 1) & 1] = *reinterpret_cast<const float4*>(As + 32 * s + 8 * (g + 1));
                fb[(g + 1) & 1] = *reinterpret_cast<const float4*>(Bs + 32 * s + 8 * (g + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(fa[g & 1], j), f4c(fb[g & 1], j), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s + 1 < KS) {
                if constexpr (step >= WS && step < WS + NP)
                    store_piece(std::integral_constant<int, s + 1>{}, std::integral_constant<int, step - WS>{});
                if constexpr (step == 13) __syncthreads();
                if constexpr (step == 14) {
                    fa[0] = *reinterpret_cast<const float4*>(As + 32 * (s + 1));
                    fb[0] = *reinterpret_cast<const float4*>(Bs + 32 * (s + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    });

    // store: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the bias pair
    // (ba - bb, bb - ba) of complex.py:187-188 was requested before the first stage
    float* Cp = p.C + (size_t)(m0 + wm0 + 4 * h) * p.ldc + col;
    if (m0 + BM <= p.M && n0 + BN <= p.N) {                       // interior tile (block-uniform): stores without exec masks
#pragma unroll
        for (int r = 0; r < 16; ++r) out_store<0>(Cp + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc, acc[r] + bj);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < p.M && col < p.N) out_store<0>(Cp + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc, acc[r] + bj);
        }
    }
    stamp_mark(p.stamp, 1);
}

static inline bool cconv_fwd_staged_ok(const GemmParams& p) {
    return (p.K == 160 || p.K == 128) && p.vecA && p.vecB && (p.N % 2) == 0 && p.ldb == p.N && p.cF * 2 == p.N;
}
template <int WS = 8, int BM = 64, int BN = 64>
static int launch_cconv_fwd_staged(const GemmParams& p, hipStream_t s) {
    const dim3 grid(ceil_div(p.N, BN) * ceil_div(p.M, BM));
    constexpr size_t smem5 = (size_t)(BM + BN) * (32 * 5 + 4) * sizeof(float), smem4 = (size_t)(BM + BN) * (32 * 4 + 4) * sizeof(float);
    if (p.K == 160) {
        auto kern = cconv_fwd_staged_kernel<5, WS, BM, BN>;
        DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem5));
        DCCN_LAUNCH_CHAINS_Z(kern, grid, dim3(kGemmThreads), smem5, s, p);
    } else {
        auto kern = cconv_fwd_staged_kernel<4, WS, BM, BN>;
        DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem4));
        DCCN_LAUNCH_CHAINS_Z(kern, grid, dim3(kGemmThreads), smem4, s, p);
    }
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

}  // namespace dccn
