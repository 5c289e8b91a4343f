// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32): exact f32 products, f32 accumulation
// in a fixed order => every launch is bitwise deterministic.
//
// C[M,N] = A[M,K] . B[K,N].  Both operands are staged in LDS as [i][k] tiles (k contiguous, row
// stride BK+4 floats): one ds_read_b128 hands a lane FOUR consecutive k of its row, i.e. the operand
// of four consecutive MFMAs.  (The MFMA pairs the k supplied by lanes 0-31 with the k supplied by
// lanes 32-63; inside every group of 8 k the order is permuted to {0,4},{1,5},{2,6},{3,7} -- the
// same permutation for A and B, so the sum is unchanged.)  With the +4 pad every LDS access of the
// kernel is bank-conflict free: 16-lane read groups land on 16 distinct 16-byte slots, writes are
// 128 contiguous bytes per 8-lane group.
//
// How a tile gets from HBM into that layout is the operand "kind":
//   OP_KCONTIG   element (k,i) at p[i*ld+k]  -> float4 along k, ds_write_b128
//   OP_ICONTIG   element (k,i) at p[k*ld+i]  -> a thread loads a 4(k) x 4(i) block as four float4
//                rows, transposes it in registers and writes four ds_write_b128 (lane -> block map
//                chosen so that both the global rows and the LDS writes stay conflict free)
//   OP_CCONV_W   virtual expanded complex-conv weights Weff[2kin,2F] built on the fly from
//                w[kin,2F] = [Wa|Wb] (dev/py/complex.py:185-188):
//                    Weff[2n  ,2f] =  Wa[n,f]   Weff[2n  ,2f+1] =  Wb[n,f]
//                    Weff[2n+1,2f] = -Wb[n,f]   Weff[2n+1,2f+1] = -Wa[n,f]
//                so that x[rows,(n,iq)] . Weff = out[rows,(f,re/im)] : the four real
//                sub-convolutions of the C-Conv are ONE GEMM with interleaved IQ in/out.
//   OP_CCONV_WT  Weff transposed (for dX = dOut . Weff^T)
//   OP_KPATCH    the im2col patch matrix of a general-k complex convolution (dev/py/complex.py:51-92, 140-196), never
//                materialised: row i = output position (b, lo, wo), column k = (tap ti, tap tj, channel, iq) is
//                x[b, lo*sL + l0 + ti, wo*sW + w0 + tj, c, iq] or 0 where TensorFlow's padding lies (PatchGeom).  For a
//                fixed (row, ti) the (tj, c, iq) run is contiguous in x, so the loader still fetches float4s; what it
//                adds is address arithmetic and a validity mask per piece.  The k-inflated patch tensor of the im2col
//                route (k = 5: five times the input, written once and read once) never touches HBM.
//
// Block = 256 threads = 4 waves in a 2x2 grid; wave tile (BM/2)x(BN/2) made of 32x32 MFMA tiles.
#pragma once
#include <type_traits>
#include "common.h"

namespace dccn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum OperandKind : int { OP_ICONTIG = 0, OP_KCONTIG = 1, OP_CCONV_W = 2, OP_CCONV_WT = 3, OP_KPATCH = 4 };
enum GemmTag : int { TAG_DENSE_FWD = 0, TAG_DENSE_BWD_X = 1, TAG_DENSE_BWD_W = 2, TAG_CCONV_FWD = 3, TAG_CCONV_BWD_X = 4,
                     TAG_CCONV_BWD_W = 5 };

// geometry of an OP_KPATCH operand: x [B, L, Wd, C, 2]; live taps ntl x ntw; l = lo*sL + l0 + ti, w = wo*sW + w0 + tj
// (l0 = first live tap - padding before, may be negative); c2 = 2C floats per cell
struct PatchGeom {
    int L, Wd, c2, Lo, Wo, ntl, ntw, sL, sW, l0, w0;
    unsigned per_mul, wo_mul;        // row -> (b, lo, wo) without integer division (gemm_kmajor.h patch_div): / (Lo*Wo), / Wo
    int per_shift, wo_shift;
    // inverse strides (input gradient of a STRIDED convolution, dccn_cconv_patch_bwd_x): the gathered tensor is the
    // convolution's output; a row's tap meets data only where its fine position n = row position + l0 + ti is a multiple of
    // the forward stride: source l = n / isL (multiply-shift, patch_div_magic), zeros elsewhere.  isL, isW <= 1: plain gather.
    int isL, isW;
    unsigned il_mul, iw_mul;
    int il_shift, iw_shift;
};
// floor(n / d) = (n * mul) >> shift for every 0 <= n < 2^31:  s = ceil(log2 d), mul = ceil(2^(31+s) / d) <= 2^32 - 1
// (mul d = 2^(31+s) + e with e < d <= 2^s, so the error term n e / (d 2^(31+s)) stays below 1/d)
static inline void patch_div_magic(int d, unsigned& mul, int& shift) {
    int s = 0;
    while ((1LL << s) < (long long)d) ++s;
    shift = 31 + s;
    mul = (unsigned)(((1ULL << shift) + (unsigned long long)d - 1) / (unsigned long long)d);
}

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N], epilogue add (only when splits == 1), nullable
    float* colsum;       // [splits][N] partial column sums of the B rows, nullable
    int M, N, K;
    int lda, ldb, ldc;
    int klen;            // K range per split (multiple of 64)
    long long slab;      // elements between the split slabs of C
    int cF;              // F of the cconv operand kinds
    int vecA, vecB;      // vector (16 B) global loads legal for the operand
    int cbias;           // 1: bias is the C-Conv [ba|bb] pair -> col 2f: ba-bb, col 2f+1: bb-ba
    // k-major weight gradients only: nranges > 0 = split z covers rows [koff[z], koff[z+1]) instead of z*klen ...
    // (graded ranges: long items first, short ones last, so that the grid's tail is made of short items)
    int nranges;
    int koff[9];
    // gemm16.h store epilogue with an element-wise stage (template parameter NB of EPI_STORE: 2 tanh, 3 v (1 - aux^2),
    // 4 v + aux): the other operand of stages 3 and 4, same shape and row stride as C
    const float* aux;
    float* out2; float* out3;   // gemm16.h EPI_STORE<5> (equalise stage): eq and corr next to C = h, aux = y
    long long gC;        // CMAP_SPLIT_PAIRS stores (gemm_store): elements between the two destination buffers
    PatchGeom pg;        // OP_KPATCH operand A
    unsigned long long* stamp;   // step timeline stamps of this launch (common.h stamp_mark), nullptr = none
    // chain groups (common.h): the same launch for the chain whose arena lies `off` bytes behind chain 0's
    __device__ __forceinline__ GemmParams at_chain(const long long off) const {
        GemmParams q = *this;
        q.A = chain_at(A, off); q.B = chain_at(B, off); q.C = chain_at(C, off);
        q.bias = chain_at(bias, off); q.colsum = chain_at(colsum, off);
        q.aux = chain_at(aux, off); q.out2 = chain_at(out2, off); q.out3 = chain_at(out3, off);
        return q;
    }
};

constexpr int kGemmThreads = 256;

__device__ __forceinline__ float cconv_weff(const float* __restrict__ w, int F, int row, int col) {
    const int n = row >> 1, q = row & 1, f = col >> 1, c = col & 1;
    const float v = w[(size_t)n * 2 * F + ((q ^ c) ? F : 0) + f];
    return q ? -v : v;
}

__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// One operand tile: BI rows (the M or N index) x BK columns (k).  A "piece" is one 16-byte unit of
// LDS traffic per thread (4 consecutive k of one row); NV pieces per thread per k-tile.
//
// Two load paths:
//   fast   (VEC builds, k-tiles that lie completely inside the K range): every per-thread address is
//          `uniform tile base + constant 32-bit byte offset`; the offsets (with the row clamp folded in) are
//          computed once before the k-loop, the tile base advances on the scalar unit.  No masks, no
//          selects: fp32 MFMAs execute on the SIMD's own FMA lanes, so every VALU instruction in the loop
//          is time taken from the matrix chain -- the steady-state body is kept down to loads, LDS ops, MFMAs.
//   masked (ragged last k-tile, unaligned / odd-sized operands): clamped addresses + zero-fill selects.
template <int KIND, int BI, int BK>
struct Tile {
    static constexpr int NV = BK * BI / 4 / kGemmThreads;
    static constexpr int LD = BK + 4;
    static constexpr bool IC = (KIND == OP_ICONTIG);
    static constexpr bool CC = (KIND == OP_CCONV_W || KIND == OP_CCONV_WT);
    static constexpr bool PATCH = (KIND == OP_KPATCH);
    static constexpr int NOFF = CC ? 2 * NV : (PATCH ? 3 * NV : NV);
    static_assert(NV >= 1 && (!IC || NV % 4 == 0) && (KIND != OP_CCONV_W || NV % 2 == 0), "tile too small for the thread block");
    float4 r[NV];
    unsigned okmask;        // masked path: bit v = piece v lies inside the k range (applied at LDS-write time)
    unsigned voff[NOFF];    // fast path: byte offsets from the uniform tile base
    unsigned negmask;       // CCONV_WT fast path: bit v = piece v's Weff row is odd (negated)

    // ICONTIG: piece v = row e = v%4 of 4x4 block v/4.  Lanes of an 8-lane group cover 2 adjacent i4
    // x 4 consecutive k4 (conflict-free transposed ds_write_b128); the 8 groups of a wave cover 16
    // adjacent i4, so each global row segment a wave touches is 256 contiguous bytes.
    static __device__ __forceinline__ void block_coords(int bidx, int& i4, int& k4) {
        const int sub = bidx & 7, rest = bidx >> 3;
        constexpr int pairs = BI / 8;
        i4 = 2 * (rest % pairs) + (sub & 1);
        k4 = 4 * (rest / pairs) + (sub >> 1);
    }

    // ---- fast path ------------------------------------------------------------------------
    __device__ __forceinline__ void init_fast(int ld, int i0, int I, int cF, int tid) {
        negmask = 0u;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if constexpr (IC) {
                int i4, k4;
                block_coords(tid + (v >> 2) * kGemmThreads, i4, k4);
                voff[v] = (unsigned)(((4 * k4 + (v & 3)) * ld + min(i0 + 4 * i4, I - 4)) * 4);
            } else {
                const int idx = tid + v * kGemmThreads;
                const int ic = min(i0 + idx / (BK / 4), I - 1);
                const int k = (idx % (BK / 4)) * 4;                 // k offset inside the tile (multiple of 4)
                if constexpr (KIND == OP_KCONTIG) {
                    voff[v] = (unsigned)((ic * ld + k) * 4);
                } else if constexpr (KIND == OP_CCONV_W) {
                    // pieces 2u, 2u+1 = the Weff columns 2f', 2f'+1 of one (row pair n0, n0+1 ; filter f') unit: lanes
                    // run over f' first, so the four dword loads of a unit read w rows contiguously (a wave touches
                    // 2-4 cache lines per load instead of 64)
                    if ((v & 1) == 0) {
                        const int uidx = tid + (v >> 1) * kGemmThreads;
                        const int fl = uidx % (BI / 2), np = uidx / (BI / 2);
                        const int f = min(i0 / 2 + fl, I / 2 - 1);
                        voff[2 * v] = (unsigned)(((2 * np) * 2 * cF + f) * 4);            // Wa[n0][f]; Wb at + cF*4
                        voff[2 * v + 1] = (unsigned)((fl * 2) * LD + 4 * np);             // LDS float offset of column 2f'
                    }
                } else {                                          // element (k,i) = Weff[i][k], row i = ic fixed
                    const int n = ic >> 1, qb = ic & 1, f0 = k >> 1;
                    voff[2 * v] = (unsigned)((n * 2 * cF + (qb ? cF : 0) + f0) * 4);      // columns c = 0
                    voff[2 * v + 1] = (unsigned)((n * 2 * cF + (qb ? 0 : cF) + f0) * 4);  // columns c = 1
                    negmask |= (unsigned)qb << v;
                }
            }
        }
    }
    // wave-uniform base of the k-tile starting at k0
    static __device__ __forceinline__ const char* tile_base(const float* p, int ld, int k0, int cF) {
        if constexpr (KIND == OP_ICONTIG) return reinterpret_cast<const char*>(p + (size_t)k0 * ld);
        else if constexpr (KIND == OP_KCONTIG) return reinterpret_cast<const char*>(p + k0);
        else if constexpr (KIND == OP_CCONV_W) return reinterpret_cast<const char*>(p + (size_t)(k0 >> 1) * 2 * cF);
        else return reinterpret_cast<const char*>(p + (k0 >> 1));
    }
    __device__ __forceinline__ void load_fast(int v, const char* __restrict__ base, int cF) {
        if constexpr (KIND == OP_ICONTIG || KIND == OP_KCONTIG) {
            r[v] = *reinterpret_cast<const float4*>(base + voff[v]);
        } else if constexpr (KIND == OP_CCONV_W) {
            if ((v & 1) == 0) {
                const unsigned row = (unsigned)(2 * cF * 4), half = (unsigned)(cF * 4);
                const float a0 = *reinterpret_cast<const float*>(base + voff[2 * v]);
                const float b0 = *reinterpret_cast<const float*>(base + voff[2 * v] + half);
                const float a1 = *reinterpret_cast<const float*>(base + voff[2 * v] + row);
                const float b1 = *reinterpret_cast<const float*>(base + voff[2 * v] + row + half);
                r[v] = make_float4(a0, -b0, a1, -b1);             // column 2f'  : Weff[2n][2f'] = Wa, [2n+1][2f'] = -Wb
                r[v + 1] = make_float4(b0, -a0, b1, -a1);         // column 2f'+1: Weff[2n][.] = Wb,   [2n+1][.]   = -Wa
            }
        } else {
            const float2 a = *reinterpret_cast<const float2*>(base + voff[2 * v]);
            const float2 b = *reinterpret_cast<const float2*>(base + voff[2 * v + 1]);
            const float sgn = ((negmask >> v) & 1u) ? -1.f : 1.f;
            r[v] = make_float4(sgn * a.x, sgn * b.x, sgn * a.y, sgn * b.y);
        }
    }

    // ---- OP_KPATCH ------------------------------------------------------------------------------------
    // Every piece of a thread sits in the same float4 column of the k-tile (idx % (BK/4) = tid % (BK/4)), so the tap
    // decomposition of that column -- two integer divisions -- is done once per thread and k-tile (piece 0) and shared;
    // a piece adds its row's (l, w) origin, tests the padding and forms one address.
    // voff[3v] = element offset of x[b, 0, 0, 0, 0]; voff[3v+1] / [3v+2] = the row's l / w at tap (0,0) (may be < 0)
    int p_ti, p_tj, p_cc;
    bool p_kok;
    __device__ __forceinline__ void init_patch(const PatchGeom& g, int i0, int I, int tid) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * kGemmThreads;
            const int i = min(i0 + idx / (BK / 4), I - 1);
            const int per = g.Lo * g.Wo;
            const int b = i / per, rem = i - b * per, lo = rem / g.Wo, wo = rem - lo * g.Wo;
            voff[3 * v] = (unsigned)(b * g.L * g.Wd * g.c2);
            voff[3 * v + 1] = (unsigned)(lo * g.sL + g.l0);
            voff[3 * v + 2] = (unsigned)(wo * g.sW + g.w0);
        }
    }
    __device__ __forceinline__ void load_patch(int v, const float* __restrict__ p, const PatchGeom& g, int k0, int kend,
                                               int K, int tid) {
        if (v == 0) {
            const int k = k0 + (tid % (BK / 4)) * 4;
            const int kc = min(k, K - 4);
            const int seg = g.ntw * g.c2;
            p_ti = kc / seg;
            const int r2 = kc - p_ti * seg;
            p_tj = r2 / g.c2;
            p_cc = r2 - p_tj * g.c2;
            p_kok = k < kend;
        }
        int l = (int)voff[3 * v + 1] + p_ti, w = (int)voff[3 * v + 2] + p_tj;
        bool ok = p_kok;
        if (g.isL > 1) {             // (uniform branch: the plain gather pays one scalar compare)
            const int q = (int)(((unsigned long long)(unsigned)max(l, 0) * g.il_mul) >> g.il_shift);
            ok = ok && l >= 0 && q * g.isL == l;
            l = q;
        }
        if (g.isW > 1) {
            const int q = (int)(((unsigned long long)(unsigned)max(w, 0) * g.iw_mul) >> g.iw_shift);
            ok = ok && w >= 0 && q * g.isW == w;
            w = q;
        }
        ok = ok && (unsigned)l < (unsigned)g.L && (unsigned)w < (unsigned)g.Wd;
        const unsigned off = ok ? (unsigned)((l * g.Wd + w) * g.c2 + p_cc) : 0u;       // (padding: any legal address)
        r[v] = *reinterpret_cast<const float4*>(p + (size_t)voff[3 * v] + off);
        okmask = (okmask & ~(1u << v)) | ((ok ? 1u : 0u) << v);
    }

    // ---- masked path ------------------------------------------------------------------------
    // Piece v of the (k0..kend) x (i0..I) window.  Out-of-range k must read as 0 (it enters the sums);
    // out-of-range i only feeds outputs that are never stored.  Every access is issued unconditionally
    // from a clamped in-range address (no divergent control flow).
    template <bool VEC>
    __device__ __forceinline__ void load_masked(int v, const float* __restrict__ p, int ld, int k0, int kend, int K,
                                                int i0, int I, int cF, int tid) {
        if constexpr (IC) {
            int i4, k4;
            block_coords(tid + (v >> 2) * kGemmThreads, i4, k4);
            const int k = k0 + 4 * k4 + (v & 3);
            const int i = i0 + 4 * i4;
            const int kc = min(k, K - 1);
            if constexpr (VEC) {
                r[v] = *reinterpret_cast<const float4*>(p + (size_t)kc * ld + min(i, I - 4));
                okmask = (okmask & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
            } else {
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = p[(size_t)kc * ld + min(i + q, I - 1)];
                    e[q] = (k < kend && i + q < I) ? t : 0.f;
                }
                r[v] = make_float4(e[0], e[1], e[2], e[3]);
                okmask |= 1u << v;
            }
        } else {
            const int idx = tid + v * kGemmThreads;
            const int i = i0 + idx / (BK / 4);
            const int k = k0 + (idx % (BK / 4)) * 4;
            const int ic = min(i, I - 1);
            if constexpr (KIND == OP_KCONTIG && VEC) {
                r[v] = *reinterpret_cast<const float4*>(p + (size_t)ic * ld + min(k, K - 4));
                okmask = (okmask & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
            } else {
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kc = min(k + q, K - 1);
                    float t;
                    if constexpr (KIND == OP_KCONTIG) t = p[(size_t)ic * ld + kc];
                    else if constexpr (KIND == OP_CCONV_W) t = cconv_weff(p, cF, kc, ic);      // element (k,i) = Weff[k][i]
                    else t = cconv_weff(p, cF, ic, kc);                                          // element (k,i) = Weff[i][k]
                    e[q] = (k + q < kend && i < I) ? t : 0.f;
                }
                r[v] = make_float4(e[0], e[1], e[2], e[3]);
                okmask |= 1u << v;
            }
        }
    }

    // v must be a compile-time constant after unrolling (static register / component selection)
    template <bool MASKED>
    __device__ __forceinline__ void store_piece(int v, float* __restrict__ lds, int tid) const {
        if constexpr (IC) {
            int i4, k4;
            block_coords(tid + (v >> 2) * kGemmThreads, i4, k4);
            const int b = v & ~3, q = v & 3;               // row i = 4*i4 + q of the transposed block
            float4 val = make_float4(f4c(r[b + 0], q), f4c(r[b + 1], q), f4c(r[b + 2], q), f4c(r[b + 3], q));
            if constexpr (MASKED) {
                val.x = ((okmask >> (b + 0)) & 1u) ? val.x : 0.f;
                val.y = ((okmask >> (b + 1)) & 1u) ? val.y : 0.f;
                val.z = ((okmask >> (b + 2)) & 1u) ? val.z : 0.f;
                val.w = ((okmask >> (b + 3)) & 1u) ? val.w : 0.f;
            }
            *reinterpret_cast<float4*>(lds + (4 * i4 + q) * LD + 4 * k4) = val;
        } else {
            const int idx = tid + v * kGemmThreads;
            float4 val = r[v];
            if constexpr (MASKED) {
                const bool ok = (okmask >> v) & 1u;
                val = make_float4(ok ? val.x : 0.f, ok ? val.y : 0.f, ok ? val.z : 0.f, ok ? val.w : 0.f);
            }
            if constexpr (KIND == OP_CCONV_W && !MASKED) {
                *reinterpret_cast<float4*>(lds + voff[2 * (v & ~1) + 1] + (v & 1) * LD) = val;     // fast-path unit mapping
            } else {
                *reinterpret_cast<float4*>(lds + (idx / (BK / 4)) * LD + (idx % (BK / 4)) * 4) = val;
            }
        }
    }
};

// NBUF = 2: double-buffered k-tiles; NBUF = 1: the whole k range of a block is ONE tile (short-k GEMMs: every global
// load of the block is in flight at once and no per-k-tile barrier remains)
template <int KA, int KB, int BM, int BN, int BK, int NBUF = 2>
constexpr size_t gemm_smem_bytes() {
    return (size_t)(NBUF * BM * Tile<KA, BM, BK>::LD + NBUF * BN * Tile<KB, BN, BK>::LD) * sizeof(float);
}

enum PrefetchMode : int { PF_NONE = 0, PF_FAST = 1, PF_MASKED = 2 };

// XCD-aware tile order: the dispatcher places block L on XCD L % 8 (speed only, never correctness); each XCD gets a
// contiguous run of row-major tiles so its private L2 holds one slice of A rows plus the B panel instead of everything
__device__ __forceinline__ int xcd_tile(const int L, const int T) {
    const int xcd = L & 7, j = L >> 3, q = T >> 3, r = T & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
// tuning knob (dccn_set_tuning key 7): single-tile launches for short k ranges.  A call that runs under a tuning snapshot
// (dccn_abi.hip TuneScope: the plan's own table, or the globals as they stood when the call began) reads the snapshot's value
extern std::atomic<int> g_whole_k_global;      // (dccn_abi.hip)
extern thread_local int tl_whole_k;           // -1: no snapshot in force
struct WholeK {
    operator int() const { return tl_whole_k >= 0 ? tl_whole_k : g_whole_k_global.load(std::memory_order_relaxed); }
};
static const WholeK g_whole_k{};

// The k-loop of one 64x64 / 128x128 output tile (block `L` of `T` tiles, split `z`) of C = A.B: leaves the tile in
// `acc` (32x32 MFMA C layout per wave), its origin in (m0, n0) and the COLSUM partial in `cs`.  Ends with a block
// barrier: the dynamic LDS is free for the caller's epilogue.
template <int KA, int KB, int BM, int BN, int BK, int COLSUM, bool VEC, int NBUF = 2>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, const int L, const int T, const int z,
                                              f32x16 (&acc)[BM / 64][BN / 64], int& m0_out, int& n0_out, float& cs_out) {
    using TA = Tile<KA, BM, BK>;
    using TB = Tile<KB, BN, BK>;
    constexpr int LDA = TA::LD, LDB = TB::LD;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [NBUF][BM][LDA]
    float* sB = smem + NBUF * BM * LDA;     // [NBUF][BN][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);
    const int ntn = (p.N + BN - 1) / BN;
    const int tile = xcd_tile(L, T);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int kbeg = z * p.klen;
    const int kend = min(p.K, kbeg + p.klen);
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    const int nfull = VEC ? (kend - kbeg) / BK : 0;     // k-tiles the fast loader may fetch

#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float cs = 0.f;   // column sum of the B rows (COLSUM)
    const bool do_cs = COLSUM && p.colsum != nullptr && m0 == 0 && tid < BN;

    TA ta;
    TB tb;
    ta.okmask = 0u;
    tb.okmask = 0u;
    // (a patch operand has one loader for every k-tile: address + padding mask per piece; its LDS writes apply the mask)
    constexpr bool APATCH = KA == OP_KPATCH;
    if constexpr (APATCH) ta.init_patch(p.pg, m0, p.M, tid);
    if constexpr (VEC) {
        if constexpr (!APATCH) ta.init_fast(p.lda, m0, p.M, p.cF, tid);
        tb.init_fast(p.ldb, n0, p.N, p.cF, tid);
    }
    constexpr int PA = TA::NV, PB = TB::NV;
    constexpr int NG = BK / 8;                           // groups of 8 k per k-tile
    constexpr int NSTEP = NG * 4;                        // MFMA steps (of TM*TN MFMAs) per k-tile
    static_assert(NSTEP >= 2 * (PA + PB), "k-tile too shallow for the load/store slots");

    auto load_a = [&](auto mode_tag, int v, int k0) {
        constexpr int MODE = decltype(mode_tag)::value;
        if constexpr (KA == OP_KPATCH) ta.load_patch(v, p.A, p.pg, k0, kend, p.K, tid);
        else if constexpr (MODE == PF_FAST) ta.load_fast(v, TA::tile_base(p.A, p.lda, k0, p.cF), p.cF);
        else ta.template load_masked<VEC>(v, p.A, p.lda, k0, kend, p.K, m0, p.M, p.cF, tid);
    };
    auto load_b = [&](auto mode_tag, int v, int k0) {
        constexpr int MODE = decltype(mode_tag)::value;
        if constexpr (MODE == PF_FAST) tb.load_fast(v, TB::tile_base(p.B, p.ldb, k0, p.cF), p.cF);
        else tb.template load_masked<VEC>(v, p.B, p.ldb, k0, kend, p.K, n0, p.N, p.cF, tid);
    };
    auto stage_first = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
        for (int v = 0; v < PA; ++v) load_a(mode_tag, v, kbeg);
#pragma unroll
        for (int v = 0; v < PB; ++v) load_b(mode_tag, v, kbeg);
#pragma unroll
        for (int v = 0; v < PA; ++v) ta.template store_piece<MODE == PF_MASKED || APATCH>(v, sA, tid);
#pragma unroll
        for (int v = 0; v < PB; ++v) tb.template store_piece<MODE == PF_MASKED>(v, sB, tid);
    };
    if (ntiles > 0) {
        if (nfull > 0) stage_first(std::integral_constant<int, PF_FAST>{});
        else stage_first(std::integral_constant<int, PF_MASKED>{});
    }
    __syncthreads();

    // One k-tile = NSTEP dependent MFMA steps (64 cycles each per accumulator).  Everything else the wave
    // has to do for the pipeline is issued INSIDE that chain:
    //   - the operand fragments of the next group of 8 k are read (one ds_read_b128 per 32x32 operand
    //     tile) while the 4 MFMA steps of the current group run,
    //   - the global loads of the next k-tile go out during the first PA+PB steps,
    //   - their LDS writes (other buffer) happen during the last PA+PB steps, ~NSTEP*64 cycles later.
    // The prefetch mode is a template parameter (fast / masked / none), so the steady-state body is
    // straight-line and the compiler keeps counted vmcnt/lgkmcnt waits instead of draining at joins.
    int t = 0;
    auto ktile = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int cur = t & 1;
        const int k0n = kbeg + (t + 1) * BK;
        const float* As = sA + cur * BM * LDA + (wm0 + l31) * LDA + 4 * h;
        const float* Bs = sB + cur * BN * LDB + (wn0 + l31) * LDB + 4 * h;
        float* An = sA + (cur ^ 1) * BM * LDA;
        float* Bn = sB + (cur ^ 1) * BN * LDB;
        float4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[0][a] = *reinterpret_cast<const float4*>(As + a * 32 * LDA);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[0][b] = *reinterpret_cast<const float4*>(Bs + b * 32 * LDB);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    fa[(g + 1) & 1][a] = *reinterpret_cast<const float4*>(As + a * 32 * LDA + 8 * (g + 1));
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    fb[(g + 1) & 1][b] = *reinterpret_cast<const float4*>(Bs + b * 32 * LDB + 8 * (g + 1));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int step = g * 4 + j;
                if constexpr (MODE != PF_NONE) {
                    if (step < PA) load_a(mode_tag, step, k0n);
                    else if (step < PA + PB) load_b(mode_tag, step - PA, k0n);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(fa[g & 1][a], j), f4c(fb[g & 1][b], j),
                                                                         acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE != PF_NONE) {
                    constexpr int S0 = NSTEP - PA - PB;
                    if (step >= S0 && step < S0 + PA) ta.template store_piece<MODE == PF_MASKED || APATCH>(step - S0, An, tid);
                    else if (step >= S0 + PA) tb.template store_piece<MODE == PF_MASKED>(step - S0 - PA, Bn, tid);
                }
            }
        }
        if (COLSUM && do_cs) {
            const float* Bc = sB + cur * BN * LDB + tid * LDB;
#pragma unroll
            for (int k = 0; k < BK; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(Bc + k);
                cs += (v.x + v.y) + (v.z + v.w);
            }
        }
        __syncthreads();
    };
    if constexpr (NBUF > 1) {
        for (; t + 1 < nfull; ++t) ktile(std::integral_constant<int, PF_FAST>{});
        for (; t + 1 < ntiles; ++t) ktile(std::integral_constant<int, PF_MASKED>{});
    }
    if (ntiles > 0) ktile(std::integral_constant<int, PF_NONE>{});
    m0_out = m0;
    n0_out = n0;
    cs_out = cs;
}

// Column maps of the store epilogue (compile-time: the equaliser's concat / split of two IQ-pair streams, model.py:456,
// happens in the stores of the GEMMs on either side of it instead of in two element-wise launches):
//   CMAP_JOIN_PAIRS   this GEMM produces ONE of the two streams of cat[rows, K, (eq re, eq im, corr re, corr im)]:
//                     logical column 2k+j goes to column 4k+j of a row of 2*N floats (C already points at the
//                     stream's first float: +0 / +2)
//   CMAP_SPLIT_PAIRS  this GEMM produces the gradient of such a cat row: logical column 4k+2g+j goes to column 2k+j of
//                     stream g's own [rows, N/2] buffer at C + g*gC
//   CMAP_TANHGRAD / CMAP_ADD  no column map, an element-wise stage instead: v (1 - aux^2) (the gradient through
//                     tanh, aux = its output) / aux + v (gradient accumulation); aux has C's shape and row stride
enum ColumnMap : int { CMAP_NONE = 0, CMAP_JOIN_PAIRS = 1, CMAP_SPLIT_PAIRS = 2, CMAP_TANHGRAD = 3, CMAP_ADD = 4 };
template <int CMAP>
__device__ __forceinline__ float cmap_stage(const GemmParams& p, const size_t i, const float v) {
    if constexpr (CMAP == CMAP_TANHGRAD) { const float y = p.aux[i]; return v * (1.0f - y * y); }
    else if constexpr (CMAP == CMAP_ADD) return p.aux[i] + v;
    else return v;
}
template <int CMAP>
__device__ __forceinline__ size_t cmap_col(const GemmParams& p, const int col) {
    if constexpr (CMAP == CMAP_JOIN_PAIRS) return (size_t)(2 * col - (col & 1));
    else if constexpr (CMAP == CMAP_SPLIT_PAIRS) return (size_t)((col >> 1) & 1) * (size_t)p.gC + (size_t)(2 * (col >> 2) + (col & 1));
    else return (size_t)col;
}

// store epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int BM, int BN, int COLSUM, int CMAP = CMAP_NONE>
__device__ __forceinline__ void gemm_store(const GemmParams& p, const int z, const f32x16 (&acc)[BM / 64][BN / 64],
                                           const int m0, const int n0, const float cs) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);
    const bool do_cs = COLSUM && p.colsum != nullptr && m0 == 0 && tid < BN;
    float* Cz = p.C + (size_t)z * p.slab;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn0 + b * 32 + l31;
            float bj = 0.f;
            if (p.bias != nullptr && col < p.N) {
                if (p.cbias) {
                    const float d = p.bias[col >> 1] - p.bias[p.cF + (col >> 1)];
                    bj = (col & 1) ? -d : d;
                } else {
                    bj = p.bias[col];
                }
            }
            if (m0 + BM <= p.M && n0 + BN <= p.N) {            // interior tile (block-uniform): stores without exec masks
                const size_t c0 = (size_t)(m0 + wm0 + a * 32 + 4 * h) * p.ldc + cmap_col<CMAP>(p, col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const size_t i = c0 + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc;
                    Cz[i] = cmap_stage<CMAP>(p, i, acc[a][b][r] + bj);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < p.M && col < p.N) {
                        const size_t i = (size_t)row * p.ldc + cmap_col<CMAP>(p, col);
                        Cz[i] = cmap_stage<CMAP>(p, i, acc[a][b][r] + bj);
                    }
                }
            }
        }
    }
    if (COLSUM && do_cs) {
        const int col = n0 + tid;
        if (col < p.N) p.colsum[(size_t)z * p.N + col] = cs;
    }
}

// One output tile of C = A.B: k-loop + store
template <int KA, int KB, int BM, int BN, int BK, int COLSUM, bool VEC, int NBUF = 2, int CMAP = CMAP_NONE>
__device__ __forceinline__ void gemm_block(const GemmParams& p, const int L, const int T, const int z) {
    f32x16 acc[BM / 64][BN / 64];
    int m0, n0;
    float cs;
    gemm_mainloop<KA, KB, BM, BN, BK, COLSUM, VEC, NBUF>(p, L, T, z, acc, m0, n0, cs);
    gemm_store<BM, BN, COLSUM, CMAP>(p, z, acc, m0, n0, cs);
}

// Several independent GEMMs of ONE shape in one grid (blockIdx.y = the group): the operands of group g lie at a fixed
// element stride from those of group 0.  The equaliser's corr / eq C-Conv pair (model.py:439-449): two 5-12 us launches
// that never overlapped become one grid of twice the blocks.
struct GroupStride {
    long long a, b, c, bias, colsum;
};
__device__ __forceinline__ GemmParams group_params(const GemmParams& p, const GroupStride& gs, const int g) {
    GemmParams q = p;
    q.A = p.A + g * gs.a;
    q.B = p.B + g * gs.b;
    q.C = p.C + g * gs.c;
    if (p.bias) q.bias = p.bias + g * gs.bias;
    if (p.colsum) q.colsum = p.colsum + g * gs.colsum;
    return q;
}
template <int KA, int KB, int BM, int BN, int BK, int TAG, bool VEC, int NBUF, int CMAP>
__global__ __launch_bounds__(kGemmThreads) void gemm_grouped_kernel(const GemmParams p0, const GroupStride gs, const ChainOffs co) {
    const GemmParams p = p0.at_chain(co.off[blockIdx.z]);           // chain groups (common.h); y = the layer of the pair
    const GemmParams q = group_params(p, gs, (int)blockIdx.y);
    stamp_mark(p.stamp, 0);
    gemm_block<KA, KB, BM, BN, BK, 0, VEC, NBUF, CMAP>(q, (int)blockIdx.x, (int)gridDim.x, 0);
    stamp_mark(p.stamp, 1);
}
template <int KA, int KB, int BM, int BN, int BK, int TAG, int NBUF, int CMAP>
static int launch_gemm_grouped(const GemmParams& p, const GroupStride& gs, int groups, hipStream_t s) {
    auto kern = gemm_grouped_kernel<KA, KB, BM, BN, BK, TAG, true, NBUF, CMAP>;
    constexpr size_t smem = gemm_smem_bytes<KA, KB, BM, BN, BK, NBUF>();
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
    dim3 grid(ceil_div(p.N, BN) * ceil_div(p.M, BM), groups, 1);
    DCCN_LAUNCH_CHAINS_Z(kern, grid, dim3(kGemmThreads), smem, s, p, gs);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// TAG only makes the symbol unique per call site so profiles attribute time to the right operator
template <int KA, int KB, int BM, int BN, int BK, int COLSUM, int TAG, bool VEC, int NBUF = 2>
__global__ __launch_bounds__(kGemmThreads) void gemm_f32_mfma_kernel(const GemmParams p) {
    stamp_mark(p.stamp, 0);
    gemm_block<KA, KB, BM, BN, BK, COLSUM, VEC, NBUF>(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z);
    stamp_mark(p.stamp, 1);
}

// Two independent GEMMs in ONE grid (the dense layer's dX = dY.W^T and dW = X^T.dY share dY but not
// their outputs): blocks [0, nx) run problem X (the longer k-loops first), the rest run problem W's
// tiles x splits.  ~830 blocks of mixed length pack the 256 CUs far better than 266 and 560 blocks in
// two launches (each of which ends with a mostly idle chip).
template <int BM, int BN, int BK, bool VEC>
__global__ __launch_bounds__(kGemmThreads) void dense_bwd_grouped_kernel(const GemmParams px, const GemmParams pw,
                                                                         const int nx, const int tw) {
    const int b = (int)blockIdx.x;
#ifdef GROUPED_ABL          // timing experiments only: 1 = the dX blocks return at once, 2 = the dW blocks do
    if ((GROUPED_ABL == 1) == (b < nx)) return;
#endif
    stamp_mark(px.stamp, 0);
    if (b < nx) {
        gemm_block<OP_KCONTIG, OP_KCONTIG, BM, BN, BK, 0, VEC>(px, b, nx, 0);
    } else {
        const int c = b - nx;
        gemm_block<OP_ICONTIG, OP_ICONTIG, BM, BN, BK, 1, VEC>(pw, c % tw, tw, c / tw);
    }
    stamp_mark(px.stamp, 1);
}

template <int KA, int KB, int BM, int BN, int BK, int COLSUM, int TAG, bool VEC, int NBUF = 2>
static int launch_gemm_cfg2(const GemmParams& p, int splits, hipStream_t s) {
    auto kern = gemm_f32_mfma_kernel<KA, KB, BM, BN, BK, COLSUM, TAG, VEC, NBUF>;
    constexpr size_t smem = gemm_smem_bytes<KA, KB, BM, BN, BK, NBUF>();
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
    dim3 grid(ceil_div(p.N, BN) * ceil_div(p.M, BM), 1, splits);
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, grid, dim3(kGemmThreads), smem, s, p);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// the fast (unmasked, 32-bit-offset) loaders need both operands vector-legal and < 2 GiB
template <int KA, int KB, int BM, int BN, int BK, int COLSUM, int TAG>
static int launch_gemm_cfg(const GemmParams& p, int splits, hipStream_t s) {
    const bool vec = p.vecA && p.vecB;
    if (vec) return launch_gemm_cfg2<KA, KB, BM, BN, BK, COLSUM, TAG, true>(p, splits, s);
    return launch_gemm_cfg2<KA, KB, BM, BN, BK, COLSUM, TAG, false>(p, splits, s);
}

template <bool VEC, int BM = 64, int BN = 64, int BK = 64>
static int launch_dense_bwd_grouped(const GemmParams& px, const GemmParams& pw, int splits_w, hipStream_t s) {
    auto kern = dense_bwd_grouped_kernel<BM, BN, BK, VEC>;
    constexpr size_t smem = gemm_smem_bytes<OP_KCONTIG, OP_KCONTIG, BM, BN, BK>();
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
    const int nx = ceil_div(px.N, BN) * ceil_div(px.M, BM);
    const int tw = ceil_div(pw.N, BN) * ceil_div(pw.M, BM);
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, dim3(nx + tw * splits_w), dim3(kGemmThreads), smem, s, px, pw, nx, tw);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
static inline bool grouped_ok(const GemmParams& px, const GemmParams& pw, int splits_w) {
    // only worth it (and only instantiated) for the 64x64x64 configuration
    const long long bigx = (long long)ceil_div(px.M, 128) * ceil_div(px.N, 128);
    const long long bigw = (long long)ceil_div(pw.M, 128) * ceil_div(pw.N, 128) * splits_w;
    return bigx < 2 * kCUs && bigw < 2 * kCUs;
}
// large layers (N = 1024: 128x128x32 tiles on both GEMMs): the few hundred long dX tiles do not fill a whole number of
// rounds of the chip (560 tiles on 512 slots at C4), the thousands of short dW tiles behind them in the same grid do
static inline bool grouped_big_ok(const GemmParams& px, const GemmParams& pw, int splits_w) {
    const long long bigx = (long long)ceil_div(px.M, 128) * ceil_div(px.N, 128);
    const long long bigw = (long long)ceil_div(pw.M, 128) * ceil_div(pw.N, 128) * splits_w;
    return bigx >= 2 * kCUs && bigw >= 2 * kCUs && bigx + bigw < (1LL << 30);
}

// tile choice: 128x128x32 (four accumulators per wave) only when it still yields >= 2 blocks per CU,
// else 64x64x64
template <int KA, int KB, int COLSUM, int TAG>
static int launch_gemm(const GemmParams& p, int splits, hipStream_t s) {
    const long long big = (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128) * splits;
    // (an output of <= 64 columns or rows would leave half of every 128-wide tile empty)
    if (big >= 2 * kCUs && p.N > 64 && p.M > 64) return launch_gemm_cfg<KA, KB, 128, 128, 32, COLSUM, TAG>(p, splits, s);
    // the N=64 C-Conv forward (K = 2*(N+CP) = 160, or 128 without the cyclic prefix): the whole k range as ONE tile --
    // all of a block's loads are issued together (one exposed memory latency instead of five) and no k-tile barrier
    if constexpr (KA == OP_KCONTIG && KB == OP_CCONV_W) {
        if (splits == 1 && p.vecA && p.vecB && g_whole_k) {
            if (p.K == 160) return launch_gemm_cfg2<KA, KB, 64, 64, 160, COLSUM, TAG, true, 1>(p, splits, s);
            if (p.K == 128) return launch_gemm_cfg2<KA, KB, 64, 64, 128, COLSUM, TAG, true, 1>(p, splits, s);
        }
    }
    // short k ranges that are a multiple of 32 but not of 64 (the C-Conv: K = 2*(N+CP) = 160): 32-deep k-tiles keep
    // every tile on the fast unmasked loaders and do not multiply zeros in a half-empty last tile
    if constexpr (KA != OP_ICONTIG && KB != OP_ICONTIG) {
        if (splits == 1 && p.K % 64 != 0 && p.K % 32 == 0 && p.K <= 512)
            return launch_gemm_cfg<KA, KB, 64, 64, 32, COLSUM, TAG>(p, splits, s);
    }
    return launch_gemm_cfg<KA, KB, 64, 64, 64, COLSUM, TAG>(p, splits, s);
}

// split-K plan for the weight-gradient GEMMs (K = batch rows is the long axis)
struct SplitPlan {
    int splits, klen;
};
static inline SplitPlan plan_splitk(int M, int N, int K) {
    const long long tiles = (long long)ceil_div(M, 64) * ceil_div(N, 64);
    // ~2 blocks per CU when the output has many tiles; exactly one round of blocks when it has few
    // (then every block is short and a second, partial round would double the kernel time)
    long long want = tiles >= 32 ? (2LL * kCUs + tiles - 1) / tiles : (long long)kCUs / tiles;
    const long long max_splits = (K + 127) / 128;               // >= 4 k-tiles per split
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int klen = (int)((K + want - 1) / want);
    klen = (klen + 63) / 64 * 64;
    SplitPlan sp;
    sp.klen = klen;
    sp.splits = (K + klen - 1) / klen;
    if (sp.splits < 1) sp.splits = 1;
    return sp;
}

// ---- split-K reductions (fixed summation order => deterministic) -------------------------
// Block = 64 element-lanes x 4 z-groups: a thread sums the slabs z = g, g+4, g+8, ... with all of its
// loads independent (no latency chain), the 4 group sums are combined through LDS in a fixed order.
constexpr int kRedLanes = 64, kRedGroups = 4;

// out[i] = sum_z partial[z*slab + i],  i in float4 units when vec4
template <bool VEC4>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ partial, int splits, long long slab,
                                                   float* __restrict__ out, long long n, const int block) {
    __shared__ float4 red[kRedGroups][kRedLanes];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = ((long long)block * kRedLanes + lane) * (VEC4 ? 4 : 1);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        // batches of 8 slabs: loads first (clamped slab index, weight by validity), then a fixed-order sum
        for (int zb = grp; zb < splits; zb += 8 * kRedGroups) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = min(zb + u * kRedGroups, splits - 1);
                const float* q = partial + (size_t)z * slab + i;
                if constexpr (VEC4) v[u] = *reinterpret_cast<const float4*>(q);
                else v[u] = make_float4(q[0], 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (zb + u * kRedGroups < splits) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
        }
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        float4 t = red[0][lane];
#pragma unroll
        for (int g = 1; g < kRedGroups; ++g) {
            t.x += red[g][lane].x; t.y += red[g][lane].y; t.z += red[g][lane].z; t.w += red[g][lane].w;
        }
        if constexpr (VEC4) *reinterpret_cast<float4*>(out + i) = t;
        else out[i] = t.x;
    }
}
template <bool VEC4>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                            long long slab, float* __restrict__ out, long long n) {
    splitk_reduce_body<VEC4>(partial, splits, slab, out, n, (int)blockIdx.x);
}
// two reductions of the same split count in one launch (a weight gradient and its bias gradient): blocks [0, blocks_a)
// take segment a, the rest segment b (scalar path)
template <bool VEC4A>
__global__ __launch_bounds__(256) void splitk_reduce2_kernel(const float* __restrict__ pa, long long slab_a,
                                                             float* __restrict__ out_a, long long na, int blocks_a,
                                                             const float* __restrict__ pb, long long slab_b,
                                                             float* __restrict__ out_b, long long nb, int splits) {
    if ((int)blockIdx.x < blocks_a) splitk_reduce_body<VEC4A>(pa, splits, slab_a, out_a, na, (int)blockIdx.x);
    else splitk_reduce_body<false>(pb, splits, slab_b, out_b, nb, (int)blockIdx.x - blocks_a);
}

static int launch_splitk_reduce(const float* partial, int splits, long long slab, float* out, long long n,
                                hipStream_t s) {
    const bool vec4 = (n % 4 == 0) && (slab % 4 == 0) && aligned16(partial) && aligned16(out);
    const long long units = vec4 ? n / 4 : n;
    const unsigned blocks = (unsigned)ceil_div_ll(units, kRedLanes);
    DCCN_NO_CHAINS();
    if (vec4) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, partial, splits, slab, out, n);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, partial, splits, slab, out, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

static int launch_splitk_reduce2(const float* pa, int splits, long long slab_a, float* out_a, long long na,
                                 const float* pb, long long slab_b, float* out_b, long long nb, hipStream_t s) {
    const bool vec4 = (na % 4 == 0) && (slab_a % 4 == 0) && aligned16(pa) && aligned16(out_a);
    const int blocks_a = (int)ceil_div_ll(vec4 ? na / 4 : na, kRedLanes), blocks_b = (int)ceil_div_ll(nb, kRedLanes);
    DCCN_NO_CHAINS();
    if (vec4)
        hipLaunchKernelGGL(splitk_reduce2_kernel<true>, dim3(blocks_a + blocks_b), dim3(256), 0, s, pa, slab_a, out_a, na,
                           blocks_a, pb, slab_b, out_b, nb, splits);
    else
        hipLaunchKernelGGL(splitk_reduce2_kernel<false>, dim3(blocks_a + blocks_b), dim3(256), 0, s, pa, slab_a, out_a, na,
                           blocks_a, pb, slab_b, out_b, nb, splits);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// C-Conv weight gradient fold: partial slabs hold dWeff[2kin,2F]; colsum [splits][2F]
//   dWa[n,f] = dWeff[2n,2f]   - dWeff[2n+1,2f+1]
//   dWb[n,f] = dWeff[2n,2f+1] - dWeff[2n+1,2f]
//   dba[f] = sum_r(dRe - dIm) = cs[2f] - cs[2f+1],  dbb = -dba     
// element index e in [0, kin*F) -> (n,f); e in [kin*F, kin*F+F) -> bias f
// out2/gout (optional, per thread): the two element offsets (relative to dw; the bias follows the kernel) and
// gradient values this thread produced, -1 when none -- lets a caller apply the optimizer update in the same pass
// tilew > 0: the dX-epilogue partials of rx_bwd.h: term z, column tile h (tilew columns = tilew/2 filters) at
// partial + (z*nh + h)*kin*tilew, ALREADY FOLDED per tile: [kin][tilew/2][{a, b}] with a = dWeff[2n,2f] - dWeff[2n+1,2f+1],
// b = dWeff[2n,2f+1] - dWeff[2n+1,2f] (one float2 per (n, f): half the bytes of the raw tile); colsum [z*nh + h][tilew];
// `slab` is the distance between consecutive terms (nh tiles).  tilew == 0: full-width [2kin, 2F] slabs (split-K GEMM output).
// LANES element-lanes x 256/LANES term groups per block: 64 x 4 for the split-K slabs (<= 128 terms of a few MB each:
// few, long rows), 16 x 16 for the dX-epilogue partials (133 short terms per element: with four groups a thread walked
// five dependent batches of loads and the fold was the long pole of the optimizer launch, 10 us; with sixteen it is one
// batch + one short one)
constexpr int kFoldLanesTiled = 16;
template <int LANES = kRedLanes>
__device__ __forceinline__ void cconv_fold_body(const float* __restrict__ partial, int splits, long long slab,
                                                const float* __restrict__ colsum, float* __restrict__ dw,
                                                float* __restrict__ dbias, int kin, int F, int block,
                                                long long* out2 = nullptr, float* gout = nullptr, int tilew = 0) {
    if (out2) { out2[0] = -1; out2[1] = -1; }
    constexpr int GROUPS = 256 / LANES;
    __shared__ float2 red[GROUPS][LANES];
    const int lane = threadIdx.x % LANES, grp = threadIdx.x / LANES;
    const int e = block * LANES + lane;
    const int total = kin * F, N2 = 2 * F;
    const bool is_w = e < total, is_b = (!is_w) && (e < total + F) && (dbias != nullptr);
    const int n = is_w ? e / F : 0, f = is_w ? e % F : (e - total);
    float a = 0.f, b = 0.f;
    // element (row, col = 2f) of term z: partial + z*slab + eoff + row*ld
    const int ld = tilew > 0 ? tilew : N2;
    const size_t eoff = tilew > 0 ? (size_t)((2 * f) / tilew) * (size_t)kin * tilew + (size_t)n * tilew + (2 * f) % tilew
                                  : (size_t)(2 * f);
    const size_t coff = tilew > 0 ? (size_t)((2 * f) / tilew) * tilew + (2 * f) % tilew : (size_t)(2 * f);
    const size_t cstride = tilew > 0 ? (size_t)(N2 / tilew) * tilew : (size_t)N2;
    // terms per batch of independent loads: 133 terms over 16 groups = 9 per thread -> ONE batch (with 8 the five groups that
    // own a ninth term walked a second, dependent batch: one more memory latency on the optimizer launch's long pole)
    constexpr int UB = LANES == kFoldLanesTiled ? 9 : 8;
    if (is_w && tilew > 0) {
        // folded tiles: one float2 {a, b} per term; the sums below add the same values in the same order as the raw form
        // (there: a += top.x - bot.y with the difference rounded first)
        for (int zb = grp; zb < splits; zb += UB * GROUPS) {
            float2 ab[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u)
                ab[u] = *reinterpret_cast<const float2*>(partial + (size_t)min(zb + u * GROUPS, splits - 1) * slab + eoff);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (zb + u * GROUPS < splits) {
                    a += ab[u].x;
                    b += ab[u].y;
                }
            }
        }
    } else if (is_w) {
        for (int zb = grp; zb < splits; zb += UB * GROUPS) {
            float2 top[UB], bot[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const float* P = partial + (size_t)min(zb + u * GROUPS, splits - 1) * slab + eoff;
                top[u] = *reinterpret_cast<const float2*>(P + (size_t)(2 * n) * ld);
                bot[u] = *reinterpret_cast<const float2*>(P + (size_t)(2 * n + 1) * ld);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (zb + u * GROUPS < splits) {
                    a += top[u].x - bot[u].y;
                    b += top[u].y - bot[u].x;
                }
            }
        }
    } else if (is_b) {
#pragma unroll 8
        for (int z = grp; z < splits; z += GROUPS) {
            const float2 c = *reinterpret_cast<const float2*>(colsum + (size_t)z * cstride + coff);
            a += c.x - c.y;
        }
    }
    red[grp][lane] = make_float2(a, b);
    __syncthreads();
    if (grp == 0) {
        float2 t = red[0][lane];
#pragma unroll
        for (int g = 1; g < GROUPS; ++g) { t.x += red[g][lane].x; t.y += red[g][lane].y; }
        if (is_w) {
            dw[(size_t)n * N2 + f] = t.x;
            dw[(size_t)n * N2 + F + f] = t.y;
            if (out2) { out2[0] = (long long)n * N2 + f; out2[1] = (long long)n * N2 + F + f; gout[0] = t.x; gout[1] = t.y; }
        } else if (is_b) {
            dbias[f] = t.x;
            dbias[F + f] = -t.x;
            if (out2) { out2[0] = (long long)kin * N2 + f; out2[1] = (long long)kin * N2 + F + f; gout[0] = t.x; gout[1] = -t.x; }
        }
    }
}

static __global__ __launch_bounds__(256) void cconv_fold_kernel(const float* __restrict__ partial, int splits,
                                                         long long slab, const float* __restrict__ colsum,
                                                         float* __restrict__ dw, float* __restrict__ dbias,
                                                         int kin, int F, int tilew = 0) {
    if (tilew > 0) cconv_fold_body<kFoldLanesTiled>(partial, splits, slab, colsum, dw, dbias, kin, F, blockIdx.x, nullptr, nullptr, tilew);
    else cconv_fold_body<kRedLanes>(partial, splits, slab, colsum, dw, dbias, kin, F, blockIdx.x, nullptr, nullptr, 0);
}

}  // namespace dccn
