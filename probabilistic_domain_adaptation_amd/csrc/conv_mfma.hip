// K1 / K2: im2col-free NHWC direct convolution (3x3 s1 p1, or 1x1) on the fp32 matrix cores of gfx950.
//
// Replaces, on the reference's hot path, every nn.Conv2d(+ReLU) with Cin >= 8:
//   prob_utils/my_models/unet_blocks.py:19-24, probabilistic_unet.py:56-61 (3x3 stacks),
//   probabilistic_unet.py:168-173 (Fcomb 1x1 stack), and their autograd input-gradients
//   (convolution_backward) - the dgrad form is the same contraction with the tap order
//   flipped and the weight slab transposed (prepared once per step by punet_flip_transpose_weights).
//
// Formulation: implicit GEMM  Out[pixel][n] = sum_{tap, k} In[pixel + tap][k] * Wt[tap][k][n]
//   M = 32 consecutive pixels of one image row (one MFMA row block), N = output channels, K = input
//   channels, accumulated with v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 64 cycles / SIMD).
// Workgroup = 256 threads = 4 waves; output tile = 8 rows x 32 cols x 64 channels; wave w owns rows
//   {2w, 2w+1} x both 32-channel blocks (2x2 accumulator tiles = 64 VGPRs).
// Per K chunk of 8 channels the workgroup stages the (8+2)x(32+2) halo tile [pixel][8+1] (pad -> the 32
//   lanes of an A read hit 32 distinct banks) and the weight slab [tap][8][64] in LDS (30.6 KB), with
//   the next chunk's global loads issued before the MFMA loop (register prefetch, one barrier pair
//   per chunk).  3 workgroups are resident per CU so that staging of one overlaps MFMA of another.
// Persistent: the grid is ONE residency round (3 workgroups x 256 CUs); each workgroup strides over the
//   (spatial tile, channel tile) list and the first chunk of its next tile is prefetched before the
//   epilogue of the current one, so neither the prologue load latency nor the store tail idles the
//   matrix pipe (this was worth ~15 % on the 64-channel layers, which have only 8 chunks per tile).
// The channel concat of the U-Net up-blocks (unet_blocks.py:56, [up, bridge]) is never materialised:
//   the K loop walks source 0 then source 1.
// Epilogue (registers): + bias (per channel, or per image and channel: Fcomb's folded z term),
//   ReLU, optional multiply by (mask > 0) (ReLU backward of the *destination* activation when used
//   as dgrad), optional accumulate into the destination, coalesced 128-B stores.
#include "common.h"
#include "punet_hip.h"

namespace {

struct ConvArgs {
    const float* x0;
    const float* x1;
    int c0, c1;
    const float* w;
    long w_tap_stride;
    int w_k_stride, w_n_stride;
    int flip;
    const float* bias;
    int bias_img_stride;
    float* y;
    int n_total;
    const float* mask;
    int accumulate, relu;
    int B, H, W;
    int ntiles;  // spatial tiles x channel tiles
};

constexpr int TH = 8, TW = 32, TN = 64, CK = 8;
constexpr int WG_PER_CU = 3;  // persistent grid = one residency round (30.6 KB of LDS, 148 VGPRs: three workgroups per CU)

struct TileInfo {
    int b, ty0, tx0, n0;
};

template <int KS, bool TRANSPOSED>
__global__ __launch_bounds__(256, 3) void conv_mfma_kernel(ConvArgs a) {
    constexpr int KK = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int HTH = TH + KS - 1, HTW = TW + KS - 1;
    constexpr int PST = CK + 1;
    constexpr int XUNITS = HTH * HTW * (CK / 4);
    constexpr int XITER = (XUNITS + 255) / 256;
    constexpr int WUNITS = KK * CK * (TN / 4);
    constexpr int WITER = (WUNITS + 255) / 256;

    __shared__ float sX[HTH * HTW * PST];
    __shared__ __attribute__((aligned(16))) float sW[KK * CK * TN];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntn = (a.n_total + TN - 1) / TN;
    const int N = a.n_total;
    const int K = a.c0 + a.c1;
    const int nchunks = K / CK;

    auto decode = [&](int tile) {
        TileInfo t;
        t.n0 = (tile % ntn) * TN;
        int sid = tile / ntn;
        t.tx0 = (sid % tiles_x) * TW;
        sid /= tiles_x;
        t.ty0 = (sid % tiles_y) * TH;
        t.b = sid / tiles_y;
        return t;
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;
    };
    zero_acc();

    float4 rx[XITER];
    float4 rw[WITER];

    auto prefetch = [&](const TileInfo& ti, int ch) {
        const int kbase = ch * CK;
        const float* src;
        int cs, coff;
        if (kbase < a.c0) {
            src = a.x0; cs = a.c0; coff = kbase;
        } else {
            src = a.x1; cs = a.c1; coff = kbase - a.c0;
        }
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int u = tid + it * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < XUNITS) {
                const int q = u % (CK / 4), pix = u / (CK / 4);
                const int py = pix / HTW, px = pix % HTW;
                const int gy = ti.ty0 + py - PAD, gx = ti.tx0 + px - PAD;
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *reinterpret_cast<const float4*>(src + ((size_t)(ti.b * a.H + gy) * a.W + gx) * cs + coff + q * 4);
            }
            rx[it] = v;
        }
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int u = tid + it * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < WUNITS) {
                if (!TRANSPOSED) {
                    const int n4 = u % (TN / 4), k = (u / (TN / 4)) % CK, t = u / ((TN / 4) * CK);
                    const int n = ti.n0 + n4 * 4;
                    const int tt = a.flip ? (KK - 1 - t) : t;
                    if (n < N)
                        v = *reinterpret_cast<const float4*>(a.w + (size_t)tt * a.w_tap_stride +
                                                             (size_t)(kbase + k) * a.w_k_stride + n);
                } else {
                    const int k4 = u % (CK / 4), n = (u / (CK / 4)) % TN, t = u / ((CK / 4) * TN);
                    const int tt = a.flip ? (KK - 1 - t) : t;
                    if (ti.n0 + n < N)
                        v = *reinterpret_cast<const float4*>(a.w + (size_t)tt * a.w_tap_stride +
                                                             (size_t)(ti.n0 + n) * a.w_n_stride + kbase + k4 * 4);
                }
            }
            rw[it] = v;
        }
    };

    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int u = tid + it * 256;
            if (u < XUNITS) {
                const int q = u % (CK / 4), pix = u / (CK / 4);
                float* d = &sX[pix * PST + q * 4];
                d[0] = rx[it].x; d[1] = rx[it].y; d[2] = rx[it].z; d[3] = rx[it].w;
            }
        }
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int u = tid + it * 256;
            if (u < WUNITS) {
                if (!TRANSPOSED) {
                    const int n4 = u % (TN / 4), k = (u / (TN / 4)) % CK, t = u / ((TN / 4) * CK);
                    *reinterpret_cast<float4*>(&sW[(t * CK + k) * TN + n4 * 4]) = rw[it];
                } else {
                    const int k4 = u % (CK / 4), n = (u / (CK / 4)) % TN, t = u / ((CK / 4) * TN);
                    float* d = &sW[(t * CK + k4 * 4) * TN + n];
                    d[0] = rw[it].x; d[TN] = rw[it].y; d[2 * TN] = rw[it].z; d[3 * TN] = rw[it].w;
                }
            }
        }
    };

    auto epilogue = [&](const TileInfo& ti) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int n = ti.n0 + nb * 32 + l31;
            if (n >= N) continue;
            float bv = 0.f;
            if (a.bias) bv = a.bias[(size_t)ti.b * a.bias_img_stride + n];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int gy = ti.ty0 + 2 * wave + m;
                if (gy >= a.H) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gx = ti.tx0 + mfma32_row(r, half);
                    if (gx >= a.W) continue;
                    const size_t idx = ((size_t)(ti.b * a.H + gy) * a.W + gx) * N + n;
                    float v = acc[m][nb][r] + bv;
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (a.mask) v = (a.mask[idx] > 0.f) ? v : 0.f;
                    if (a.accumulate) v += a.y[idx];
                    a.y[idx] = v;
                }
            }
        }
    };

    // lane-constant LDS bases: A = sX[((2*wave + m + ky) * HTW + l31 + kx) * PST + 2*s + half]
    //                          B = sW[(t * CK + 2*s + half) * TN + nb * 32 + l31]
    const float* aBase = &sX[((2 * wave) * HTW + l31) * PST + half];
    const float* bBase = &sW[half * TN + l31];

    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    int ch = 0;
    TileInfo ti = decode(tile);
    prefetch(ti, 0);
    while (true) {
        __syncthreads();
        commit();
        __syncthreads();
        int nch = ch + 1, ntile = tile;
        TileInfo nti = ti;
        if (nch == nchunks) {
            nch = 0;
            ntile = tile + (int)gridDim.x;
            if (ntile < a.ntiles) nti = decode(ntile);
        }
        const bool has_next = ntile < a.ntiles;
        if (has_next) prefetch(nti, nch);
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            const int ky = t / KS, kx = t % KS;
#pragma unroll
            for (int s = 0; s < CK / 2; ++s) {
                const float a0 = aBase[((0 + ky) * HTW + kx) * PST + 2 * s];
                const float a1 = aBase[((1 + ky) * HTW + kx) * PST + 2 * s];
                const float b0 = bBase[(t * CK + 2 * s) * TN];
                const float b1 = bBase[(t * CK + 2 * s) * TN + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (ch == nchunks - 1) {
            epilogue(ti);
            zero_acc();
        }
        if (!has_next) break;
        tile = ntile;
        ch = nch;
        ti = nti;
    }
}

template <int KS, bool TR>
int launch_conv(ConvArgs& a, hipStream_t stream) {
    a.ntiles = a.B * ceil_div(a.H, TH) * ceil_div(a.W, TW) * ceil_div(a.n_total, TN);
    const int cap = WG_PER_CU * punet_num_cus();
    int grid = a.ntiles < cap ? a.ntiles : cap;
    conv_mfma_kernel<KS, TR><<<grid, 256, 0, stream>>>(a);
    PUNET_LAUNCH_CHECK("conv_mfma");
    return PUNET_OK;
}

// wt[kk-1-t][co][ci] = w[t][ci][co]: the dgrad's weight slab in forward-kernel orientation (32x32 LDS-tiled transpose,
// coalesced on both sides).  grid = (cout/32, cin/32, k*k), block = 32 x 8.
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin,
                                                             int cout, int kk) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = w + (size_t)t * cin * cout;
    float* dst = wt + (size_t)(kk - 1 - t) * cin * cout;
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (ci0 + r < cin && co0 + tx < cout) tile[r][tx] = src[(size_t)(ci0 + r) * cout + co0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (co0 + r < cout && ci0 + tx < cin) dst[(size_t)(co0 + r) * cin + ci0 + tx] = tile[tx][r];
}

}  // namespace

extern "C" int punet_conv_fwd(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias,
                              int bias_img_stride, float* y, int cout, int B, int H, int W, int ksize, int relu,
                              void* stream) {
    PUNET_CHECK_ARG(ksize == 1 || ksize == 3, "punet_conv_fwd: ksize must be 1 or 3 (got %d)", ksize);
    PUNET_CHECK_ARG(x0 && w && y, "punet_conv_fwd: null pointer");
    PUNET_CHECK_ARG(c0 > 0 && c0 % CK == 0 && c1 >= 0 && c1 % CK == 0 && (c1 == 0 || x1),
                    "punet_conv_fwd: source channels must be multiples of %d (c0=%d c1=%d)", CK, c0, c1);
    PUNET_CHECK_ARG(cout > 0 && cout % 4 == 0, "punet_conv_fwd: cout must be a multiple of 4 (got %d)", cout);
    PUNET_CHECK_ARG(B > 0 && H > 0 && W > 0, "punet_conv_fwd: bad shape");
    ConvArgs a{};
    a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1;
    a.w = w; a.w_tap_stride = (long)(c0 + c1) * cout; a.w_k_stride = cout; a.w_n_stride = 1; a.flip = 0;
    a.bias = bias; a.bias_img_stride = bias_img_stride;
    a.y = y; a.n_total = cout; a.mask = nullptr; a.accumulate = 0; a.relu = relu;
    a.B = B; a.H = H; a.W = W;
    return ksize == 3 ? launch_conv<3, false>(a, (hipStream_t)stream) : launch_conv<1, false>(a, (hipStream_t)stream);
}

extern "C" int punet_flip_transpose_weights(const float* w, float* wt, int cin, int cout, int ksize, void* stream) {
    PUNET_CHECK_ARG(w && wt && cin > 0 && cout > 0 && (ksize == 1 || ksize == 3), "punet_flip_transpose_weights: bad args");
    dim3 grid(ceil_div(cout, 32), ceil_div(cin, 32), ksize * ksize);
    flip_transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w, wt, cin, cout, ksize * ksize);
    PUNET_LAUNCH_CHECK("flip_transpose");
    return PUNET_OK;
}

// dgrad with a weight slab prepared by punet_flip_transpose_weights (wt[k][k][cout][cin_total]): runs the forward kernel.
extern "C" int punet_conv_dgrad_t(const float* dz, int cout, const float* wt, int cin_total, int cin_off, int cin,
                                  float* dx, const float* mask, int accumulate, int B, int H, int W, int ksize,
                                  void* stream) {
    PUNET_CHECK_ARG(ksize == 1 || ksize == 3, "punet_conv_dgrad_t: ksize must be 1 or 3 (got %d)", ksize);
    PUNET_CHECK_ARG(dz && wt && dx, "punet_conv_dgrad_t: null pointer");
    PUNET_CHECK_ARG(cout > 0 && cout % CK == 0, "punet_conv_dgrad_t: cout must be a multiple of %d (got %d)", CK, cout);
    PUNET_CHECK_ARG(cin > 0 && cin % 4 == 0 && cin_off >= 0 && cin_off % 4 == 0 && cin_off + cin <= cin_total &&
                        cin_total % 4 == 0,
                    "punet_conv_dgrad_t: bad channel window off=%d n=%d total=%d (all must be multiples of 4)", cin_off, cin,
                    cin_total);
    ConvArgs a{};
    a.x0 = dz; a.x1 = nullptr; a.c0 = cout; a.c1 = 0;
    a.w = wt + cin_off; a.w_tap_stride = (long)cin_total * cout; a.w_k_stride = cin_total; a.w_n_stride = 1; a.flip = 0;
    a.bias = nullptr; a.bias_img_stride = 0;
    a.y = dx; a.n_total = cin; a.mask = mask; a.accumulate = accumulate; a.relu = 0;
    a.B = B; a.H = H; a.W = W;
    return ksize == 3 ? launch_conv<3, false>(a, (hipStream_t)stream) : launch_conv<1, false>(a, (hipStream_t)stream);
}

extern "C" int punet_conv_dgrad(const float* dz, int cout, const float* w, int cin_total, int cin_off, int cin,
                                float* dx, const float* mask, int accumulate, int B, int H, int W, int ksize,
                                void* stream) {
    PUNET_CHECK_ARG(ksize == 1 || ksize == 3, "punet_conv_dgrad: ksize must be 1 or 3 (got %d)", ksize);
    PUNET_CHECK_ARG(dz && w && dx, "punet_conv_dgrad: null pointer");
    PUNET_CHECK_ARG(cout > 0 && cout % CK == 0, "punet_conv_dgrad: cout must be a multiple of %d (got %d)", CK, cout);
    PUNET_CHECK_ARG(cin > 0 && cin % 4 == 0 && cin_off >= 0 && cin_off + cin <= cin_total,
                    "punet_conv_dgrad: bad channel window off=%d n=%d total=%d", cin_off, cin, cin_total);
    ConvArgs a{};
    a.x0 = dz; a.x1 = nullptr; a.c0 = cout; a.c1 = 0;
    a.w = w + (size_t)cin_off * cout; a.w_tap_stride = (long)cin_total * cout; a.w_k_stride = 1; a.w_n_stride = cout;
    a.flip = 1;
    a.bias = nullptr; a.bias_img_stride = 0;
    a.y = dx; a.n_total = cin; a.mask = mask; a.accumulate = accumulate; a.relu = 0;
    a.B = B; a.H = H; a.W = W;
    return ksize == 3 ? launch_conv<3, true>(a, (hipStream_t)stream) : launch_conv<1, true>(a, (hipStream_t)stream);
}
