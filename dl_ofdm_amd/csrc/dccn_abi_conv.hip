// libdccn.so -- general-k complex convolutions, CRC32C (see abi_impl.h for how the library is cut into units)
#include "abi_impl.h"

using namespace dccn;

extern "C" {

// ---- patch gather of the general-k complex convolutions -----------------------------------------------------
static bool im2col_geom_ok(const Im2colGeom& g) {
    return g.B > 0 && g.L > 0 && g.Wd > 0 && g.C > 0 && g.Lo > 0 && g.Wo > 0 && g.ntl > 0 && g.ntw > 0 && g.sL > 0 && g.sW > 0 &&
           g.tl0 >= 0 && g.tw0 >= 0 && g.pl0 >= 0 && g.pw0 >= 0;
}
int dccn_cconv_im2col(const float* x, float* rows, int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int tl0,
                      int tw0, int sL, int sW, int pl0, int pw0, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!x || !rows || !im2col_geom_ok(g)) return DCCN_ERR_INVALID_ARG;
    const long long n = (long long)B * Lo * Wo * ntl * ntw * C;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (float2*)rows, g, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
// the same convolution without the patch tensor: the GEMM's A-loader gathers the taps (gemm_f32_mfma.h OP_KPATCH)
int dccn_cconv_patch_supported(int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int F) {
    // float4 pieces must not straddle cells: 2C multiple of 4; 32-bit element offsets into x; weights on the vector loaders
    if (B <= 0 || L <= 0 || Wd <= 0 || C <= 0 || Lo <= 0 || Wo <= 0 || ntl <= 0 || ntw <= 0 || F <= 0) return 0;
    if ((C % 2) != 0 || (F % 2) != 0) return 0;
    if ((long long)B * L * Wd * C * 2 >= (1LL << 31) || (long long)B * Lo * Wo >= (1LL << 31)) return 0;
    if ((long long)ntl * ntw * C * 2 * 2 * F * 4 >= (1LL << 31)) return 0;
    return 1;
}
int dccn_cconv_patch_fwd(const float* x, const float* w, const float* bias, float* out, int B, int L, int Wd, int C, int Lo,
                         int Wo, int ntl, int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F,
                         dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!x || !w || !out || !im2col_geom_ok(g) || !dccn_cconv_patch_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, F) ||
        !aligned16(x) || !aligned16(w))
        return DCCN_ERR_INVALID_ARG;
    const int kin = ntl * ntw * C;
    GemmParams p = gp_zero();                 // out[rows,2F] = patches[rows,2kin] . Weff[2kin,2F]
    p.A = x; p.B = w; p.C = out; p.bias = bias; p.cbias = 1;
    p.M = B * Lo * Wo; p.N = 2 * F; p.K = 2 * kin;
    p.lda = 0; p.ldb = 2 * F; p.ldc = 2 * F;
    p.klen = round_k(2 * kin);
    p.cF = F;
    p.vecA = 1; p.vecB = 1;
    p.pg.L = L; p.pg.Wd = Wd; p.pg.c2 = 2 * C; p.pg.Lo = Lo; p.pg.Wo = Wo; p.pg.ntl = ntl; p.pg.ntw = ntw;
    p.pg.sL = sL; p.pg.sW = sW; p.pg.l0 = tl0 - pl0; p.pg.w0 = tw0 - pw0;
    return launch_gemm<OP_KPATCH, OP_CCONV_W, 0, TAG_CCONV_FWD>(p, 1, (hipStream_t)stream);
}
// ---- the backward of the same convolutions without the patch tensor ---------------------------------------------
// weight gradient: dWeff[(ti,tj,c,iq), n] = sum over output positions of patch(x)^T . dout -- the k-major weight-gradient
// GEMM (gemm_kmajor.h) with its A rows gathered from x itself (APATCH); split-K slabs folded like every C-Conv's
int dccn_cconv_patch_bwd_supported(int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int sL, int sW, int F) {
    if (!dccn_cconv_patch_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, F)) return 0;
    if (sL <= 0 || sW <= 0) return 0;
    if ((long long)B * Lo * Wo * F * 2 >= (1LL << 31)) return 0;          // 32-bit element offsets into dout
    int mode = 1 | 2;                  // bit 0: weight gradient; bit 1: input gradient as an implicit GEMM (any stride, round 6)
    // bit 2: ... and it is expected to beat GEMM + col2im.  Cost per INPUT position in units of 16 tile columns x 2F deep rows:
    //   few channels (2C <= 32, cconv_dx_narrow.h): ceil(2C/16) columns x the taps of the position's stride phase (ntl*ntw / (sL*sW));
    //   otherwise (64-wide tiles, inverse-stride gather): 4 ceil(2C/64) columns x ALL ntl*ntw taps (zeros where the stride skips);
    //   GEMM + col2im: 4 ceil(2 kin/64) columns per OUTPUT position (sL*sW times fewer) plus the [rows, kin, 2] round trip and the
    //   scatter launch -- which is why the implicit route may cost up to 3x (narrow) / 2x (wide) the other's tile columns.
    const long long taps = (long long)ntl * ntw, ss = (long long)sL * sW, col = 4LL * ceil_div(2 * ntl * ntw * C, 64);
    if (cconv_dx_narrow_ok(C, F, sL, sW)) {
        if ((long long)ceil_div(2 * C, 16) * taps <= 3 * col) mode |= 4;          // (both sides per input position: x ss cancels)
    } else if (4LL * ceil_div(2 * C, 64) * taps * ss * ss <= 2 * col) {
        mode |= 4;
    }
    return mode;
}
size_t dccn_cconv_patch_bwd_w_workspace_size(int B, int Lo, int Wo, int C, int ntl, int ntw, int F) {
    if (B <= 0 || Lo <= 0 || Wo <= 0 || C <= 0 || ntl <= 0 || ntw <= 0 || F <= 0) return 0;
    return cconv_bw_ws_bytes(B * Lo * Wo, ntl * ntw * C, F);
}
int dccn_cconv_patch_bwd_w(const float* x, const float* dout, float* dw, float* dbias, int B, int L, int Wd, int C, int Lo,
                           int Wo, int ntl, int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F,
                           void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!x || !dout || !dw || !im2col_geom_ok(g) || !aligned16(x) || !aligned16(dout) ||
        !(dccn_cconv_patch_bwd_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, sL, sW, F) & 1))
        return DCCN_ERR_INVALID_ARG;
    const int rows = B * Lo * Wo, kin = ntl * ntw * C;
    if (!workspace || workspace_bytes < cconv_bw_ws_bytes(rows, kin, F)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const SplitPlan sp = plan_splitk(2 * kin, 2 * F, rows);
    Carver c(workspace, workspace_bytes);
    float* slabs = c.take<float>((size_t)sp.splits * 4 * kin * F);
    float* cs = c.take<float>((size_t)sp.splits * 2 * F);
    GemmParams p = gp_zero();                 // dWeff[2kin,2F] = patches(x)[rows,2kin]^T . dout[rows,2F]
    p.A = x; p.B = dout; p.C = slabs; p.colsum = cs;
    p.M = 2 * kin; p.N = 2 * F; p.K = rows;
    p.lda = 0; p.ldb = 2 * F; p.ldc = 2 * F;
    p.klen = sp.klen;
    p.slab = (long long)4 * kin * F;
    p.vecA = 1; p.vecB = 1;
    p.pg.L = L; p.pg.Wd = Wd; p.pg.c2 = 2 * C; p.pg.Lo = Lo; p.pg.Wo = Wo; p.pg.ntl = ntl; p.pg.ntw = ntw;
    p.pg.sL = sL; p.pg.sW = sW; p.pg.l0 = tl0 - pl0; p.pg.w0 = tw0 - pw0;
    patch_div_magic(Lo * Wo, p.pg.per_mul, p.pg.per_shift);
    patch_div_magic(Wo, p.pg.wo_mul, p.pg.wo_shift);
    if (!kmajor_ok(p)) return DCCN_ERR_INVALID_ARG;
    DCCN_TRY((launch_kmajor<1, TAG_CCONV_BWD_W, true>(p, sp.splits, s)));
    const int fold_blocks = ceil_div(kin * F + F, kRedLanes);
    hipLaunchKernelGGL(cconv_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, slabs, sp.splits, p.slab, cs, dw, dbias, kin, F);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
// input gradient: dx[b,l,w,(c,iq)] = sum over taps of dout[b, (l-l0-ti)/sL, (w-w0-tj)/sW, :] . Weff[(ti,tj,c,iq), :] -- a
// convolution of dout with the tap-flipped transposed weights, i.e. the SAME implicit GEMM with dout as the gathered operand
// (geometry: rows = input positions, taps t' = nt-1-t, origin -(l0+ntl-1)) and Bt[(c,iq)][(ti',tj',n)] as a plain k-contiguous
// operand built from w by the little kernel below (4 kin F floats).  No [rows, kin, 2] gradient-of-patches tensor, no col2im.
// Strides (round 6): a tap contributes where its fine position is a multiple of the forward stride -- the loader's
// inverse-stride gather (PatchGeom::isL / isW) reads dout there and zeros elsewhere.
__global__ __launch_bounds__(256) void cconv_flip_wt_kernel(const float* __restrict__ w, float* __restrict__ bt, int C, int ntl,
                                                            int ntw, int F) {
    const long long K = (long long)ntl * ntw * 2 * F, total = 2LL * C * K;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i / K);
    const int k = (int)(i - (long long)j * K);
    const int tp = k / (2 * F), nn = k - tp * 2 * F;
    const int tip = tp / ntw, tjp = tp - tip * ntw;
    const int n = (((ntl - 1 - tip) * ntw) + (ntw - 1 - tjp)) * C + (j >> 1);
    const int iq = j & 1, f = nn >> 1, oq = nn & 1;
    const float wa = w[(size_t)n * 2 * F + f], wb = w[(size_t)n * 2 * F + F + f];
    // Weff[2n, 2f] = Wa, [2n, 2f+1] = Wb, [2n+1, 2f] = -Wb, [2n+1, 2f+1] = -Wa (gemm_f32_mfma.h OP_CCONV_W)
    bt[i] = iq == 0 ? (oq == 0 ? wa : wb) : (oq == 0 ? -wb : -wa);
}
size_t dccn_cconv_patch_bwd_x_workspace_size(int C, int ntl, int ntw, int F) {
    if (C <= 0 || ntl <= 0 || ntw <= 0 || F <= 0) return 0;
    return align_up((size_t)4 * C * ntl * ntw * F * sizeof(float), 256);
}
int dccn_cconv_patch_bwd_x(const float* dout, const float* w, float* dx, int B, int L, int Wd, int C, int Lo, int Wo, int ntl,
                           int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F, void* workspace,
                           size_t workspace_bytes, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!dout || !w || !dx || !im2col_geom_ok(g) || !aligned16(dout) || !aligned16(dx) ||
        !(dccn_cconv_patch_bwd_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, sL, sW, F) & 2))
        return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_cconv_patch_bwd_x_workspace_size(C, ntl, ntw, F) || !aligned16(workspace))
        return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* bt = reinterpret_cast<float*>(workspace);
    const long long total = 4LL * C * ntl * ntw * F;
    hipLaunchKernelGGL(cconv_flip_wt_kernel, dim3((unsigned)ceil_div_ll(total, 256)), dim3(256), 0, s, w, bt, C, ntl, ntw, F);
    DCCN_LAUNCH_CHECK();
    if (cconv_dx_narrow_ok(C, F, sL, sW)) {   // few channels: 16-column tiles, strides by phase (cconv_dx_narrow.h)
        DxNarrowArgs a;
        memset(&a, 0, sizeof(a));
        a.dout = dout; a.bt = bt; a.dx = dx;
        a.B = B; a.L = L; a.Wd = Wd; a.C2 = 2 * C; a.Lo = Lo; a.Wo = Wo; a.ntl = ntl; a.ntw = ntw; a.F2 = 2 * F;
        a.l0 = -(tl0 - pl0) - (ntl - 1); a.w0 = -(tw0 - pw0) - (ntw - 1);
        a.sL = sL; a.sW = sW;
        return launch_cconv_dx_narrow(a, s);
    }
    GemmParams p = gp_zero();                 // dx[B*L*Wd, 2C] = patches'(dout)[., ntl*ntw*2F] . Bt^T
    p.A = dout; p.B = bt; p.C = dx;
    p.M = B * L * Wd; p.N = 2 * C; p.K = ntl * ntw * 2 * F;
    p.lda = 0; p.ldb = p.K; p.ldc = 2 * C;
    p.klen = round_k(p.K);
    p.vecA = 1; p.vecB = 1;
    p.pg.L = Lo; p.pg.Wd = Wo; p.pg.c2 = 2 * F; p.pg.Lo = L; p.pg.Wo = Wd; p.pg.ntl = ntl; p.pg.ntw = ntw;
    p.pg.sL = 1; p.pg.sW = 1; p.pg.l0 = -(tl0 - pl0) - (ntl - 1); p.pg.w0 = -(tw0 - pw0) - (ntw - 1);
    p.pg.isL = sL; p.pg.isW = sW;
    patch_div_magic(sL, p.pg.il_mul, p.pg.il_shift);
    patch_div_magic(sW, p.pg.iw_mul, p.pg.iw_shift);
    return launch_gemm<OP_KPATCH, OP_KCONTIG, 0, TAG_CCONV_BWD_X>(p, 1, s);
}
// ---- few-channel 1-D C-Conv: input, weight and bias gradient in one pass over dout (cconv1d_bwd.h) -------------------------
constexpr int kConv1dMaxBlocks = 1024;
static int conv1d_bwd_pl(int ntl, int sL) { return 62 * sL - ntl + 2; }          // positions per chunk: its rows fit 64 (cconv1d_bwd.h)
int dccn_cconv1d_bwd_supported(int B, int L, int C, int Lo, int ntl, int sL, int F) {
    if (B <= 0 || L <= 0 || C <= 0 || Lo <= 0 || ntl <= 0 || sL <= 0 || F <= 0) return 0;
    if ((C % 2) != 0 || (F != 32 && F != 64) || 2 * ntl * C > 30 || conv1d_bwd_pl(ntl, sL) < 1) return 0;
    if ((long long)B * L * C * 2 >= (1LL << 31) || (long long)B * Lo * F * 2 >= (1LL << 31)) return 0;
    return 1;
}
size_t dccn_cconv1d_bwd_workspace_size(int F) {
    if (F <= 0) return 0;
    size_t o = 0;
    o = carve_size(o, (size_t)kConv1dMaxBlocks * 32 * 2 * F * sizeof(float));
    o = carve_size(o, (size_t)kConv1dMaxBlocks * 2 * F * sizeof(float));
    return align_up(o, 256);
}
int dccn_cconv1d_bwd(const float* x, const float* dout, const float* w, float* dx, float* dw, float* dbias, int B, int L, int C,
                     int Lo, int ntl, int tl0, int sL, int pl0, int F, void* workspace, size_t workspace_bytes,
                     dccn_stream_t stream) {
    if (!x || !dout || !w || !dx || !dw || !dccn_cconv1d_bwd_supported(B, L, C, Lo, ntl, sL, F) || !aligned16(dout) || !aligned16(w))
        return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_cconv1d_bwd_workspace_size(F)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    Conv1dBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dout = dout; a.w = w; a.dx = dx;
    a.slabs = c.take<float>((size_t)kConv1dMaxBlocks * 32 * 2 * F);
    a.colsum = c.take<float>((size_t)kConv1dMaxBlocks * 2 * F);
    a.B = B; a.L = L; a.C2 = 2 * C; a.Lo = Lo; a.nt = ntl; a.F2 = 2 * F; a.NC = 2 * ntl * C;
    a.o = tl0 - pl0; a.s = sL;
    a.PL = conv1d_bwd_pl(ntl, sL);
    a.nch = ceil_div(L, a.PL);
    const long long total = (long long)B * a.nch;
    int grid = 2 * kCUs;
    if (grid > kConv1dMaxBlocks) grid = kConv1dMaxBlocks;
    if (grid > total) grid = (int)total;
    DCCN_NO_CHAINS();
    if (F == 64) {
        auto kern = cconv1d_bwd_fused_kernel<128>;
        DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), conv1d_bwd_smem_bytes<128>()));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), conv1d_bwd_smem_bytes<128>(), s, a);
    } else {
        auto kern = cconv1d_bwd_fused_kernel<64>;
        DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), conv1d_bwd_smem_bytes<64>()));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), conv1d_bwd_smem_bytes<64>(), s, a);
    }
    DCCN_LAUNCH_CHECK();
    const int kin = ntl * C;
    const int fold_blocks = ceil_div(kin * F + F, kRedLanes);
    hipLaunchKernelGGL(cconv_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, a.slabs, grid, (long long)32 * 2 * F, a.colsum, dw, dbias,
                       kin, F);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_cconv_col2im(const float* drows, float* dx, int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int tl0,
                      int tw0, int sL, int sW, int pl0, int pw0, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!drows || !dx || !im2col_geom_ok(g)) return DCCN_ERR_INVALID_ARG;
    const long long n = (long long)B * L * Wd * C;
    hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)drows, (float2*)dx, g, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---- CRC32C (host) --------------------------------------------------------------------------------------
static uint32_t g_crc32c_table[8][256];
static bool g_crc32c_ready = false;
static void crc32c_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc32c_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t)
            g_crc32c_table[t][i] = (g_crc32c_table[t - 1][i] >> 8) ^ g_crc32c_table[0][g_crc32c_table[t - 1][i] & 0xffu];
    g_crc32c_ready = true;
}
uint32_t dccn_crc32c(uint32_t crc, const void* data, size_t n) {
    if (!g_crc32c_ready) crc32c_init();
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xffffffffu;
    while (n >= 8) {                                     // slicing-by-8
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc32c_table[7][lo & 0xffu] ^ g_crc32c_table[6][(lo >> 8) & 0xffu] ^ g_crc32c_table[5][(lo >> 16) & 0xffu] ^
            g_crc32c_table[4][lo >> 24] ^ g_crc32c_table[3][hi & 0xffu] ^ g_crc32c_table[2][(hi >> 8) & 0xffu] ^
            g_crc32c_table[1][(hi >> 16) & 0xffu] ^ g_crc32c_table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc32c_table[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

}  // extern "C"
