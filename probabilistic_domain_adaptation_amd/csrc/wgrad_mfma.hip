// K3: weight gradient of the 3x3 / 1x1 NHWC convolutions on the fp32 matrix cores (+ fused bias gradient).
//
// Replaces autograd's convolution_backward(weight, bias) for every conv of the reference's hot path
// (prob_utils/my_models/unet_blocks.py:19-24, probabilistic_unet.py:56-61,168-173).
//
//   dW[tap][ci][co] = sum_{b,y,x} X[b][y+ky-1][x+kx-1][ci] * dZ[b][y][x][co]        db[co] = sum dZ[..][co]
//
// Per tap this is a GEMM with M = ci, N = co and the reduction over B*H*W pixels as K, so
// A[i=ci][k=pixel] and B[k=pixel][j=co] are both "32 consecutive channels of one pixel": with a
// [pixel][channel] LDS image every operand read is 32 consecutive dwords (conflict-free).
// Workgroup = 4 waves = one 64(ci) x 64(co) tile pair x all 9 taps; wave (cib, cob) keeps nine 32x32
// accumulator tiles (144 VGPRs).  Pixels are walked in 2x32 tiles staged with their halo.
// Split-K: grid.x slices stride over the pixel tiles; every slice writes its partial slab with plain
// stores and a second kernel sums the slabs in slice order -> bit-reproducible (no float atomics),
// which the data-parallel gradient-equality test relies on.
#include "common.h"
#include "punet_hip.h"

namespace {

constexpr int WTH = 2, WTW = 32, WC = 64;

struct WgradArgs {
    const float* x;
    int cs;
    const float* dz;
    int cout;
    float* partial;
    float* bias_partial;
    int B, H, W, nslices;
};

// MFMA work of one staged pixel tile for taps [T0, T0 + NT): k-step s, half h takes pixel q = 2s + h (row q/32, col q%32)
template <int KS, int T0, int NT>
__device__ __forceinline__ void wgrad_tile_mfma(f32x16 (&acc)[5], float& bsum, const float* aBase, const float* bBase) {
    constexpr int HTW = WTW + KS - 1;
#pragma unroll
    for (int s = 0; s < WTH * WTW / 2; ++s) {
        const int row = (2 * s) / WTW, col = (2 * s) % WTW;
        const float bv = bBase[(2 * s) * WC];
        bsum += bv;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = T0 + tt, ky = t / KS, kx = t % KS;
            const float av = aBase[((row + ky) * HTW + col + kx) * WC];
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tt], 0, 0, 0);
        }
    }
}

template <int KS, int T0, int NT>
__device__ __forceinline__ void wgrad_store(const f32x16 (&acc)[5], const WgradArgs& a, int slice, int ci_base, int co,
                                            int half) {
    constexpr int KK = KS * KS;
    if (co >= a.cout) return;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci_base + mfma32_row(r, half);
            if (ci < a.cs) a.partial[(((size_t)slice * KK + T0 + tt) * a.cs + ci) * a.cout + co] = acc[tt][r];
        }
}

// 3x3: 8 waves = 4 (ci block, co block) pairs x 2 tap groups (taps 0-4 | 5-8): each SIMD hosts one wave of each group,
// so the nine taps of a pair are balanced per SIMD, every wave holds <= 5 accumulator tiles (80 VGPRs) and 4 waves per
// SIMD (2 workgroups per CU) hide each other's LDS latencies.  1x1: 4 waves, one tap.
template <int KS>
__global__ __launch_bounds__(KS == 3 ? 512 : 256, KS == 3 ? 4 : 2) void wgrad_mfma_kernel(WgradArgs a) {
    constexpr int NT_THREADS = KS == 3 ? 512 : 256;
    constexpr int KK = KS * KS;
    constexpr int TSPLIT = KS == 3 ? 5 : 1;
    constexpr int PAD = KS / 2;
    constexpr int HTH = WTH + KS - 1, HTW = WTW + KS - 1;
    constexpr int XUNITS = HTH * HTW * (WC / 4);
    constexpr int DUNITS = WTH * WTW * (WC / 4);

    __shared__ __attribute__((aligned(16))) float sX[HTH * HTW * WC];
    __shared__ __attribute__((aligned(16))) float sD[WTH * WTW * WC];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int pair = wave & 3, group = wave >> 2;
    const int cib = pair >> 1, cob = pair & 1;

    const int co_tiles = (a.cout + WC - 1) / WC;
    const int ci_tile = blockIdx.y / co_tiles, co_tile = blockIdx.y % co_tiles;
    const int ci0 = ci_tile * WC, co0 = co_tile * WC;
    const int slice = blockIdx.x;

    const int tiles_x = (a.W + WTW - 1) / WTW, tiles_y = (a.H + WTH - 1) / WTH;
    const int ntiles = a.B * tiles_x * tiles_y;

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const float* aBase = &sX[half * WC + cib * 32 + l31];
    const float* bBase = &sD[half * WC + cob * 32 + l31];

    for (int tile = slice; tile < ntiles; tile += a.nslices) {
        int sid = tile;
        const int txi = sid % tiles_x;
        sid /= tiles_x;
        const int tyi = sid % tiles_y;
        const int b = sid / tiles_y;
        const int ty0 = tyi * WTH, tx0 = txi * WTW;

        __syncthreads();
        for (int u = tid; u < XUNITS; u += NT_THREADS) {
            const int c4 = u % (WC / 4), pix = u / (WC / 4);
            const int py = pix / HTW, px = pix % HTW;
            const int gy = ty0 + py - PAD, gx = tx0 + px - PAD;
            const int c = ci0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.cs)
                v = *reinterpret_cast<const float4*>(a.x + ((size_t)(b * a.H + gy) * a.W + gx) * a.cs + c);
            *reinterpret_cast<float4*>(&sX[pix * WC + c4 * 4]) = v;
        }
        for (int u = tid; u < DUNITS; u += NT_THREADS) {
            const int c4 = u % (WC / 4), pix = u / (WC / 4);
            const int py = pix / WTW, px = pix % WTW;
            const int gy = ty0 + py, gx = tx0 + px;
            const int c = co0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < a.H && gx < a.W && c < a.cout)
                v = *reinterpret_cast<const float4*>(a.dz + ((size_t)(b * a.H + gy) * a.W + gx) * a.cout + c);
            *reinterpret_cast<float4*>(&sD[pix * WC + c4 * 4]) = v;
        }
        __syncthreads();

        if (group == 0)
            wgrad_tile_mfma<KS, 0, TSPLIT>(acc, bsum, aBase, bBase);
        else if (KK > TSPLIT)
            wgrad_tile_mfma<KS, TSPLIT, (KK > TSPLIT ? KK - TSPLIT : 1)>(acc, bsum, aBase, bBase);
    }

    // partial slab [slice][tap][cs][cout]
    const int co = co0 + cob * 32 + l31;
    if (group == 0)
        wgrad_store<KS, 0, TSPLIT>(acc, a, slice, ci0 + cib * 32, co, half);
    else if (KK > TSPLIT)
        wgrad_store<KS, TSPLIT, (KK > TSPLIT ? KK - TSPLIT : 1)>(acc, a, slice, ci0 + cib * 32, co, half);
    if (a.bias_partial && ci_tile == 0 && cib == 0 && group == 0) {
        const float tot = bsum + __shfl_xor(bsum, 32);
        if (half == 0 && co < a.cout) a.bias_partial[(size_t)slice * a.cout + co] = tot;
    }
}

}  // namespace

extern "C" size_t punet_wgrad_workspace_floats(int cs, int cout, int ksize, int nslices) {
    return (size_t)nslices * ((size_t)ksize * ksize * cs * cout + cout);
}

extern "C" int punet_wgrad_slices(int cs, int cout, int B, int H, int W) {
    const int ntiles = B * ceil_div(H, WTH) * ceil_div(W, WTW);
    const int pairs = ceil_div(cs, WC) * ceil_div(cout, WC);
    if (pairs < 1 || ntiles < 1) return 1;  // (a shape the launch would refuse)
    // 2 workgroups are resident per CU (244 VGPRs, 51 KB LDS): size the grid to ONE full residency round
    int ns = 2 * punet_num_cus() / pairs;
    if (ns > ntiles) ns = ntiles;
    if (ns < 1) ns = 1;
    return ns;
}

extern "C" int punet_conv_wgrad(const float* x, int cs, const float* dz, int cout, float* dw, int cin_total,
                                int ci_off, float* dbias, float* workspace, int nslices, int B, int H, int W,
                                int ksize, int accumulate, float scale, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PUNET_CHECK_ARG(ksize == 1 || ksize == 3, "punet_conv_wgrad: ksize must be 1 or 3 (got %d)", ksize);
    PUNET_CHECK_ARG(x && dz && dw && workspace, "punet_conv_wgrad: null pointer");
    PUNET_CHECK_ARG(cs > 0 && cs % 4 == 0 && cout > 0 && cout % 4 == 0,
                    "punet_conv_wgrad: channels must be multiples of 4 (cs=%d cout=%d)", cs, cout);
    PUNET_CHECK_ARG(ci_off >= 0 && ci_off + cs <= cin_total, "punet_conv_wgrad: bad channel window");
    PUNET_CHECK_ARG(nslices >= 1, "punet_conv_wgrad: nslices must be >= 1");
    const int KK = ksize * ksize;
    WgradArgs a{};
    a.x = x; a.cs = cs; a.dz = dz; a.cout = cout;
    a.partial = workspace;
    a.bias_partial = dbias ? workspace + (size_t)nslices * KK * cs * cout : nullptr;
    a.B = B; a.H = H; a.W = W; a.nslices = nslices;
    dim3 grid(nslices, ceil_div(cs, WC) * ceil_div(cout, WC));
    if (ksize == 3)
        wgrad_mfma_kernel<3><<<grid, 512, 0, stream>>>(a);
    else
        wgrad_mfma_kernel<1><<<grid, 256, 0, stream>>>(a);
    PUNET_LAUNCH_CHECK("wgrad_mfma");
    const size_t n = (size_t)KK * cs * cout;
    if (dbias)
        return punet_launch_reduce_slabs_pair(a.partial, n, KK * cs, cout, cs, cin_total, ci_off, dw, a.bias_partial, (size_t)cout,
                                              dbias, nslices, accumulate, scale, stream);
    return punet_launch_reduce_slabs(a.partial, nslices, n, KK * cs, cout, cs, cin_total, ci_off, dw, accumulate, scale, stream);
}
