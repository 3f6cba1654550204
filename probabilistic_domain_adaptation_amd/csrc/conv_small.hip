// K1s: first-layer 3x3 convolutions with 1..8 input channels (image, or [image, mask] of the posterior
// encoder: probabilistic_unet.py:39-42,118) and their weight/bias gradients.  1..4 channels (every reference script) keep the
// weights in registers; 5..8 read them from LDS (forward) / take two passes over channel sub-ranges (weight gradient).
//
// These layers are HBM-bound (a few FLOP per output byte), so they stay on the VALU: every thread owns
// four consecutive output channels of a pixel (Cout/4 threads per pixel -> one coalesced 16 B x Cout/4
// store per pixel), keeps its 9*CIN weight float4s in registers and walks a strip of pixels.
// No input gradient exists for them (the image needs none).
#include <type_traits>
#include "common.h"
#include "punet_hip.h"

namespace {

// grid = (ceil(W / (ppb * XITER)), rows <= 65535): a block walks XITER strips of ppb pixels of ONE image row, so the
// row / batch decomposition is block-uniform and the per-pixel index arithmetic is a handful of 32-bit adds (two 32-bit
// divisions per pixel had made this store-bound kernel ALU-bound: 3.0 TB/s of stores, r02)
template <int CIN, int OUT16 = 0>  // OUT16: y holds 1 = bf16 (the "bf16s" fast mode), 2 = fp16 ("f16s", saturating)
__global__ __launch_bounds__(256) void conv3x3_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int cout, int B, int H, int W, int relu, int xiter,
                                                               float* __restrict__ amax_out) {
    // amax_out (nullable): max |y| is published there (atomic max on the bits; the fp16-operand modes scale the next layer's
    // operand by it, common.h) - every thread reaches the end of the kernel (256 % groups == 0: no thread is surplus)
    __shared__ float sam[4];
    float am = 0.f;
    constexpr bool WLDS = CIN > 4;  // 9 * CIN float4 weights per thread: registers up to 4 channels, LDS beyond
    extern __shared__ __attribute__((aligned(16))) float4 swr[];  // WLDS: [9 * CIN][groups]
    const int groups = cout / 4;
    const int ppb = 256 / groups;
    const int g = threadIdx.x % groups, pl = threadIdx.x / groups;
    if (WLDS) {
        for (int i = threadIdx.x; i < 9 * CIN * groups; i += 256)
            swr[i] = *reinterpret_cast<const float4*>(w + (size_t)(i / groups) * cout + (i % groups) * 4);
        __syncthreads();
    }
    float4 wr[WLDS ? 1 : 9 * CIN];  // (pl < ppb for every thread: the host checks 256 % groups == 0)
    if (!WLDS) {
#pragma unroll
        for (int i = 0; i < 9 * CIN; ++i) wr[WLDS ? 0 : i] = *reinterpret_cast<const float4*>(w + (size_t)i * cout + g * 4);
    }
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned row = blockIdx.y; row < (unsigned)(B * H); row += gridDim.y) {  // b * H + py
        const int py = (int)(row % (unsigned)H);
        const float* xrow = x + (size_t)row * W * CIN;  // 64-bit base per row, 32-bit offsets inside
        typedef typename std::conditional<OUT16 == 2, _Float16, __bf16>::type o16;
        float* yrow = OUT16 ? reinterpret_cast<float*>(reinterpret_cast<o16*>(y) + (size_t)row * W * cout + g * 4)
                            : y + (size_t)row * W * cout + g * 4;
        const bool up = py > 0, down = py < H - 1;
        for (int it = 0; it < xiter; ++it) {
            const int px = ((int)blockIdx.x * xiter + it) * ppb + pl;
            if (px >= W) break;
            float4 o = bv;
            // branch-free taps: all 9 * CIN loads are issued together (clamped address, zero outside the image); a
            // `continue` per tap had serialised them into nine load -> wait -> fma rounds
            float xin[9 * CIN];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool in = (dy < 0 ? up : (dy > 0 ? down : true)) && px + dx >= 0 && px + dx < W;
                const float* xp = xrow + (in ? (dy * W + px + dx) * CIN : px * CIN);
#pragma unroll
                for (int c = 0; c < CIN; ++c) xin[t * CIN + c] = in ? xp[c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 9 * CIN; ++i) {
                const float xv = xin[i];
                const float4 ww = WLDS ? swr[i * groups + g] : wr[WLDS ? 0 : i];
                o.x = fmaf(xv, ww.x, o.x); o.y = fmaf(xv, ww.y, o.y);
                o.z = fmaf(xv, ww.z, o.z); o.w = fmaf(xv, ww.w, o.w);
            }
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            if constexpr (OUT16 != 0) {
                typedef o16 o16x4 __attribute__((ext_vector_type(4)));
                if constexpr (OUT16 == 2) {  // fp16's largest finite value instead of inf
                    o.x = __builtin_amdgcn_fmed3f(o.x, -65504.f, 65504.f); o.y = __builtin_amdgcn_fmed3f(o.y, -65504.f, 65504.f);
                    o.z = __builtin_amdgcn_fmed3f(o.z, -65504.f, 65504.f); o.w = __builtin_amdgcn_fmed3f(o.w, -65504.f, 65504.f);
                }
                o16x4 ob;
                ob[0] = (o16)o.x; ob[1] = (o16)o.y; ob[2] = (o16)o.z; ob[3] = (o16)o.w;
                *reinterpret_cast<o16x4*>(reinterpret_cast<o16*>(yrow) + (size_t)px * cout) = ob;
            } else {
                *reinterpret_cast<float4*>(yrow + (size_t)px * cout) = o;
            }
        }
    }
    if (amax_out) punet_amax_publish_block256(amax_out, am, sam);  // (block-uniform)
}

// partial[block][9*CIN + 1][cout]: rows 0..9*CIN-1 = dW[tap][ci][:], last row = dbias
// x holds `xs` channels per pixel, of which this launch takes [c_off, c_off + CIN) (xs = CIN, c_off = 0 for 1..4 channels)
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 float* __restrict__ partial, int cout, int B, int H, int W,
                                                                 int xs, int c_off) {
    constexpr int ROWS = 9 * CIN + 1;
    extern __shared__ __attribute__((aligned(16))) float4 sred[];  // [4 waves][ROWS][groups]
    const int groups = cout / 4;
    const int ppb = 256 / groups;
    const int g = threadIdx.x % groups, pl = threadIdx.x / groups;
    float4 acc[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < ppb) {
        // a block walks whole image rows (row = blockIdx.x, blockIdx.x + gridDim.x, ...): the row / batch decomposition is
        // block-uniform, per pixel only 32-bit adds remain (two 32-bit divisions per pixel had made this read-bound kernel
        // ALU-bound, as in the forward)
        for (unsigned row = blockIdx.x; row < (unsigned)(B * H); row += gridDim.x) {
            const int py = (int)(row % (unsigned)H);
            const float* xrow = x + (size_t)row * W * xs + c_off;
            const float* drow = dz + (size_t)row * W * cout + g * 4;
            const bool up = py > 0, down = py < H - 1;
            for (int px = pl; px < W; px += ppb) {
                const float4 d = *reinterpret_cast<const float4*>(drow + (size_t)px * cout);
                acc[ROWS - 1].x += d.x; acc[ROWS - 1].y += d.y; acc[ROWS - 1].z += d.z; acc[ROWS - 1].w += d.w;
                float xin[9 * CIN];  // branch-free taps, as in the forward
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    const bool in = (dy < 0 ? up : (dy > 0 ? down : true)) && px + dx >= 0 && px + dx < W;
                    const float* xp = xrow + (in ? (dy * W + px + dx) * xs : px * xs);
#pragma unroll
                    for (int c = 0; c < CIN; ++c) xin[t * CIN + c] = in ? xp[c] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 9 * CIN; ++i) {
                    const float xv = xin[i];
                    float4& a = acc[i];
                    a.x = fmaf(xv, d.x, a.x); a.y = fmaf(xv, d.y, a.y);
                    a.z = fmaf(xv, d.z, a.z); a.w = fmaf(xv, d.w, a.w);
                }
            }
        }
    }
    // reduce over the pixel lanes of a wave (lanes with equal g), then over the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        for (int off = groups; off < 64; off <<= 1) {
            acc[i].x += __shfl_xor(acc[i].x, off); acc[i].y += __shfl_xor(acc[i].y, off);
            acc[i].z += __shfl_xor(acc[i].z, off); acc[i].w += __shfl_xor(acc[i].w, off);
        }
        if (lane < groups) sred[(wave * ROWS + i) * groups + lane] = acc[i];
    }
    __syncthreads();
    for (int u = threadIdx.x; u < ROWS * groups; u += 256) {
        float4 s = sred[u];
        for (int wv = 1; wv < 4; ++wv) {
            const float4 v = sred[wv * ROWS * groups + u];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int row = u / groups, gg = u % groups;
        *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * ROWS + row) * cout + gg * 4) = s;
    }
}

constexpr int SMALL_WGRAD_BLOCKS = 1024;

}  // namespace

static int small_fwd_launch(int out16, const float* x, int cin, const float* w, const float* bias, void* y, int cout, int B, int H,
                            int W, int relu, void* stream_, float* amax_y = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    PUNET_CHECK_ARG(x && w && y, "punet_conv3x3_small_fwd: null pointer");
    PUNET_CHECK_ARG(cin >= 1 && cin <= 8, "punet_conv3x3_small_fwd: cin must be 1..8 (got %d)", cin);
    PUNET_CHECK_ARG(cout >= 4 && cout % 4 == 0 && 256 % (cout / 4) == 0 && cout <= 1024,
                    "punet_conv3x3_small_fwd: cout/4 must divide 256 (got cout=%d)", cout);
    PUNET_CHECK_ARG((unsigned long long)W * (cin > cout ? cin : cout) < (1ull << 31), "punet_conv3x3_small_fwd: image row too long");
    const int ppb = 256 / (cout / 4);
    const int xiter = 8;
    const dim3 grid(ceil_div(W, ppb * xiter), B * H < 65535 ? B * H : 65535);
    float* yf = (float*)y;
    const size_t wlds = cin > 4 ? (size_t)9 * cin * (cout / 4) * sizeof(float4) : 0;  // 5..8 channels: weights in LDS (<= 72 KB)
    PUNET_CHECK_ARG(wlds <= 160 * 1024, "punet_conv3x3_small_fwd: cin %d x cout %d needs more than 160 KB of LDS for the weights", cin, cout);
#define PUNET_SMALL_FWD_ATTR(K)                                                                                                \
    if (wlds > 48 * 1024 && hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds) != hipSuccess) { \
        (void)hipGetLastError();                                                                                               \
        punet_set_error("punet_conv3x3_small_fwd: cannot opt in to %zu bytes of dynamic LDS", wlds);                           \
        return PUNET_ERR_HIP;                                                                                                  \
    }
#define PUNET_SMALL_FWD(CI)                                                                                          \
    do {                                                                                                             \
        if (out16 == 2) {                                                                                            \
            PUNET_SMALL_FWD_ATTR((conv3x3_small_fwd_kernel<CI, 2>))                                                  \
            conv3x3_small_fwd_kernel<CI, 2><<<grid, 256, wlds, stream>>>(x, w, bias, yf, cout, B, H, W, relu, xiter, amax_y); \
        } else if (out16 == 1) {                                                                                     \
            PUNET_SMALL_FWD_ATTR((conv3x3_small_fwd_kernel<CI, 1>))                                                  \
            conv3x3_small_fwd_kernel<CI, 1><<<grid, 256, wlds, stream>>>(x, w, bias, yf, cout, B, H, W, relu, xiter, amax_y); \
        } else {                                                                                                     \
            PUNET_SMALL_FWD_ATTR((conv3x3_small_fwd_kernel<CI, 0>))                                                  \
            conv3x3_small_fwd_kernel<CI, 0><<<grid, 256, wlds, stream>>>(x, w, bias, yf, cout, B, H, W, relu, xiter, amax_y); \
        }                                                                                                            \
    } while (0)
    switch (cin) {
        case 1: PUNET_SMALL_FWD(1); break;
        case 2: PUNET_SMALL_FWD(2); break;
        case 3: PUNET_SMALL_FWD(3); break;
        case 4: PUNET_SMALL_FWD(4); break;
        case 5: PUNET_SMALL_FWD(5); break;
        case 6: PUNET_SMALL_FWD(6); break;
        case 7: PUNET_SMALL_FWD(7); break;
        default: PUNET_SMALL_FWD(8); break;
    }
#undef PUNET_SMALL_FWD
#undef PUNET_SMALL_FWD_ATTR
    PUNET_LAUNCH_CHECK("conv3x3_small_fwd");
    return PUNET_OK;
}

extern "C" int punet_conv3x3_small_fwd(const float* x, int cin, const float* w, const float* bias, float* y, int cout,
                                       int B, int H, int W, int relu, void* stream) {
    return small_fwd_launch(0, x, cin, w, bias, y, cout, B, H, W, relu, stream);
}

// the same, publishing max |y| into the amax slot amax_y (fp16-operand modes: the next layer scales its operand by it)
extern "C" int punet_conv3x3_small_fwd_amax(const float* x, int cin, const float* w, const float* bias, float* y, int cout, int B,
                                            int H, int W, int relu, float* amax_y, void* stream) {
    PUNET_CHECK_ARG(amax_y, "punet_conv3x3_small_fwd_amax: null amax slot");
    return small_fwd_launch(0, x, cin, w, bias, y, cout, B, H, W, relu, stream, amax_y);
}

extern "C" int punet_conv3x3_small_fwd_bf16s(const float* x, int cin, const float* w, const float* bias, void* y, int cout, int B,
                                             int H, int W, int relu, void* stream) {
    return small_fwd_launch(1, x, cin, w, bias, y, cout, B, H, W, relu, stream);
}

extern "C" int punet_conv3x3_small_fwd_f16s(const float* x, int cin, const float* w, const float* bias, void* y, int cout, int B,
                                            int H, int W, int relu, void* stream) {
    return small_fwd_launch(2, x, cin, w, bias, y, cout, B, H, W, relu, stream);
}

extern "C" size_t punet_small_wgrad_workspace_floats(int cin, int cout) {
    return (size_t)SMALL_WGRAD_BLOCKS * (9 * cin + 1) * cout;  // (5..8 channels run as two passes of <= 4: never more than this)
}

extern "C" int punet_conv3x3_small_wgrad(const float* x, int cin, const float* dz, int cout, float* dw, float* dbias,
                                         float* workspace, int B, int H, int W, int accumulate, float scale,
                                         void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PUNET_CHECK_ARG(x && dz && dw && dbias && workspace, "punet_conv3x3_small_wgrad: null pointer");
    PUNET_CHECK_ARG(cin >= 1 && cin <= 8, "punet_conv3x3_small_wgrad: cin must be 1..8 (got %d)", cin);
    PUNET_CHECK_ARG(cout >= 4 && cout % 4 == 0 && 64 % (cout / 4) == 0,
                    "punet_conv3x3_small_wgrad: cout/4 must divide 64 (got cout=%d)", cout);
    const size_t P = (size_t)B * H * W;
    PUNET_CHECK_ARG(P < (1ull << 31), "punet_conv3x3_small_wgrad: more than 2^31 pixels");
    const int groups = cout / 4;
    int grid = B * H;  // one image row per block and round
    if (grid > SMALL_WGRAD_BLOCKS) grid = SMALL_WGRAD_BLOCKS;
    // channel sub-ranges of at most 4 (the accumulators of a range live in registers); the bias gradient comes from the first
    for (int c_off = 0; c_off < cin; c_off += 4) {
        const int cs = cin - c_off < 4 ? cin - c_off : 4;
        const int rows = 9 * cs + 1;
        const size_t shmem = (size_t)4 * rows * groups * sizeof(float4);
        PUNET_CHECK_ARG(shmem <= 160 * 1024, "punet_conv3x3_small_wgrad: cout %d needs %zu bytes of LDS (> 160 KB)", cout, shmem);
        // wide first layers (num_filters[0] = 128 with 4 channels, 256 with >= 2) pass 64 KB of dynamic LDS: opt in
#define PUNET_SMALL_WGRAD(CS)                                                                                                   \
    do {                                                                                                                        \
        if (shmem > 48 * 1024 && hipFuncSetAttribute((const void*)conv3x3_small_wgrad_kernel<CS>,                               \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {   \
            (void)hipGetLastError();                                                                                            \
            punet_set_error("punet_conv3x3_small_wgrad: cannot opt in to %zu bytes of dynamic LDS", shmem);                     \
            return PUNET_ERR_HIP;                                                                                               \
        }                                                                                                                       \
        conv3x3_small_wgrad_kernel<CS><<<grid, 256, shmem, stream>>>(x, dz, workspace, cout, B, H, W, cin, c_off);              \
    } while (0)
        switch (cs) {
            case 1: PUNET_SMALL_WGRAD(1); break;
            case 2: PUNET_SMALL_WGRAD(2); break;
            case 3: PUNET_SMALL_WGRAD(3); break;
            default: PUNET_SMALL_WGRAD(4); break;
        }
#undef PUNET_SMALL_WGRAD
        PUNET_LAUNCH_CHECK("conv3x3_small_wgrad");
        const size_t slab = (size_t)rows * cout;
        const int rc = punet_launch_reduce_slabs_pair(workspace, slab, 9 * cs, cout, cs, cin, c_off, dw, workspace + (size_t)9 * cs * cout,
                                                      slab, c_off == 0 ? dbias : nullptr, grid, accumulate, scale, stream);
        if (rc != PUNET_OK) return rc;
    }
    return PUNET_OK;
}
