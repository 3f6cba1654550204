// K1s: first-layer 3x3 convolutions with 1..8 input channels (image, or [image, mask] of the posterior
// encoder: probabilistic_unet.py:39-42,118) and their weight/bias gradients.  1..4 channels (every reference script) keep the
// weights in registers; 5..8 read them from LDS (forward) / take two passes over channel sub-ranges (weight gradient).
//
// These layers are HBM-bound (a few FLOP per output byte), so they stay on the VALU: every thread owns
// four consecutive output channels of a pixel (Cout/4 threads per pixel -> one coalesced 16 B x Cout/4
// store per pixel), keeps its 9*CIN weight float4s in registers and walks a strip of pixels.
// No input gradient exists for them (the image needs none).
#include <cstdlib>
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

// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the first layer with 2..4 input channels (MitoEM-shaped 3-channel patches, the posterior's [image, label] input) on the
// matrix cores.  The VALU kernel above needs 9 CIN loads + 36 CIN FMAs + their addressing per output quad: with one channel it is
// bound by its stores (4.4 TB/s), with three / four it is bound by its vector instructions (1.25 / 1.13 TB/s: 0.40 / 0.45 ms per
// launch at 704 x 704, batch 4).  As a GEMM the layer is out[px][co] = sum_k P[px][k] W[k][co] with k = tap * CIN + c < 9 CIN <= 36:
// ONE (CIN <= 3) or TWO v_mfma_f32_16x16x32 K-steps per {16 pixels, 16 channels} in the six-product arithmetic of every other
// layer (operands split into three bf16 planes: fp32-accurate) - ~25 us of matrix time per launch, i.e. the
// kernel is back at its stores.  A lane (pixel l16, k-octet g) gathers its 8 patch values itself (offsets per lane, fixed for the
// kernel); the weights are split once per workgroup into LDS planes in the A-operand order.  Output mapping = the 3x3 kernels':
// a lane holds one pixel and four consecutive channels per accumulator quad (16-byte stores).
typedef float fm_f32x4 __attribute__((ext_vector_type(4)));
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int cout,
                                                                int B, int H, int W, int relu, float* __restrict__ amax_out) {
    constexpr int KT = 9 * CIN, KS = (KT + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) uint4 swp[];  // [plane][ks][g][co] units of 8 k-values (bf16)
    __shared__ float sam[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int units = KS * 4 * cout;
    for (int u = tid; u < units; u += 256) {
        const int co = u % cout, kg = u / cout;  // kg = ks * 4 + g: k = 8 kg .. 8 kg + 7
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kg + j;
            v[j] = k < KT ? w[(size_t)k * cout + co] : 0.f;
        }
        uint4 ph, pm, pl;
        split3_pair<3>(v[0], v[1], ph.x, pm.x, pl.x);
        split3_pair<3>(v[2], v[3], ph.y, pm.y, pl.y);
        split3_pair<3>(v[4], v[5], ph.z, pm.z, pl.z);
        split3_pair<3>(v[6], v[7], ph.w, pm.w, pl.w);
        swp[u] = ph;
        swp[units + u] = pm;
        swp[2 * units + u] = pl;
    }
    __syncthreads();
    // this lane's 8 KS patch elements: tap offset (in floats, relative to the pixel) and (dy, dx) for the border test
    int koff[KS][8], kdy[KS][8], kdx[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * ks + 8 * g + j;
            const int t = k / CIN, c = k % CIN;
            const bool real = k < KT;
            kdy[ks][j] = real ? t / 3 - 1 : 0x40000000;  // (a row that is never inside the image)
            kdx[ks][j] = real ? t % 3 - 1 : 0;
            koff[ks][j] = real ? ((t / 3 - 1) * W + (t % 3 - 1)) * CIN + c : 0;
        }
    const int bpr = (W + 15) / 16;  // 16-pixel blocks per image row
    const long nblk = (long)B * H * bpr;
    float am = 0.f;
    // two 16-pixel blocks per iteration: their 16 KS gathers are in flight together, and a weight fragment read serves both
    constexpr int NB = 2;
    for (long blk0 = ((long)blockIdx.x * 4 + wave) * NB; blk0 < nblk; blk0 += (long)gridDim.x * 4 * NB) {
        uint4 pf[NB][KS][3];
        float* yp[NB];
        bool pin[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long blk = blk0 + nb < nblk ? blk0 + nb : nblk - 1;  // (wave-uniform; a duplicate of the last block stores nothing)
            const int row = (int)(blk / bpr), bx = (int)(blk % bpr);   // row = b * H + py
            const int py = row % H, px = bx * 16 + l16;
            pin[nb] = px < W && blk0 + nb < nblk;
            const float* xp = x + ((size_t)row * W + (px < W ? px : W - 1)) * CIN;
            yp[nb] = y + ((size_t)row * W + px) * cout + 4 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = pin[nb] && (unsigned)(py + kdy[ks][j]) < (unsigned)H && (unsigned)(px + kdx[ks][j]) < (unsigned)W;
                    v[j] = in ? xp[koff[ks][j]] : 0.f;
                }
                split3_pair<3>(v[0], v[1], pf[nb][ks][0].x, pf[nb][ks][1].x, pf[nb][ks][2].x);
                split3_pair<3>(v[2], v[3], pf[nb][ks][0].y, pf[nb][ks][1].y, pf[nb][ks][2].y);
                split3_pair<3>(v[4], v[5], pf[nb][ks][0].z, pf[nb][ks][1].z, pf[nb][ks][2].z);
                split3_pair<3>(v[6], v[7], pf[nb][ks][0].w, pf[nb][ks][1].w, pf[nb][ks][2].w);
            }
        }
        for (int co0 = 0; co0 < cout; co0 += 16) {
            fm_f32x4 c[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) c[nb] = fm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int wu = (ks * 4 + g) * cout + co0 + l16;
                const uint4 wh = swp[wu], wm = swp[units + wu], wl = swp[2 * units + wu];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    c[nb] = punet_mfma16<false>(wm, pf[nb][ks][1], c[nb]);  // mm
                    c[nb] = punet_mfma16<false>(wh, pf[nb][ks][2], c[nb]);  // lh
                    c[nb] = punet_mfma16<false>(wl, pf[nb][ks][0], c[nb]);  // hl
                    c[nb] = punet_mfma16<false>(wh, pf[nb][ks][1], c[nb]);  // mh
                    c[nb] = punet_mfma16<false>(wm, pf[nb][ks][0], c[nb]);  // hm
                    c[nb] = punet_mfma16<false>(wh, pf[nb][ks][0], c[nb]);  // hh
                }
            }
            const fm_f32x4 bq = bias ? *reinterpret_cast<const fm_f32x4*>(bias + co0 + 4 * g) : fm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                fm_f32x4 v = c[nb] + bq;
                if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (pin[nb]) {
                    *reinterpret_cast<fm_f32x4*>(yp[nb] + co0) = v;
                    am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                }
            }
        }
    }
    if (amax_out) punet_amax_publish_block256(amax_out, am, sam);  // (block-uniform)
}

// The weight / bias gradient of the same layers on the matrix cores:  dW[k][co] = sum_px P[px][k] dz[px][co]  with the PIXELS as the
// contracted dimension (32 per MFMA): A = P^T - lane (row k = 16 kb + l16, g) gathers patch element k of 8 consecutive pixels -,
// B = dz - lane (column co = 16 cb + l16, g) loads channel co of the same 8 pixels (16 lanes = 64 contiguous bytes of one pixel).
// Row 9 CIN of P is the constant 1 of every valid pixel: its row of the product is the bias gradient.  Six-product arithmetic
// as everywhere; a wave keeps its [KB x 16][cout] accumulators over all its 32-pixel blocks, the four waves of a workgroup are summed
// through LDS, and the workgroup writes ONE partial slab in conv3x3_small_wgrad_kernel's layout (the same slab reduction follows).
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                      float* __restrict__ partial, int cout, int B, int H, int W) {
    constexpr int KT = 9 * CIN, ROWS = KT + 1, KB = (ROWS + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float sredf[];  // [4 waves][KB * 16][cout]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int ncb = cout / 16;  // <= 8 (host)
    fm_f32x4 acc[KB][8];
#pragma unroll
    for (int i = 0; i < KB * 8; ++i) acc[i / 8][i % 8] = fm_f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's patch rows k = 16 kb + l16: tap offset relative to the pixel, (dy, dx) for the border test; row KT = the ones row
    int roff[KB], rdy[KB], rdx[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int k = 16 * kb + l16, t = k / CIN, c = k % CIN;
        const bool real = k < KT;
        rdy[kb] = real ? t / 3 - 1 : (k == KT ? 0 : 0x40000000);
        rdx[kb] = real ? t % 3 - 1 : 0;
        roff[kb] = real ? ((t / 3 - 1) * W + (t % 3 - 1)) * CIN + c : 0;
    }
    const int bpr = (W + 31) / 32;  // 32-pixel blocks per image row
    const long nblk = (long)B * H * bpr;
    for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk; blk += (long)gridDim.x * 4) {
        const int row = (int)(blk / bpr), bx = (int)(blk % bpr);  // row = b * H + py
        const int py = row % H, px0 = bx * 32 + 8 * g;              // this lane's 8 pixels: px0 .. px0 + 7
        const float* xr = x + (size_t)row * W * CIN;
        const float* dr = dz + (size_t)row * W * cout;
        uint4 af[KB][3];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const bool ones = 16 * kb + l16 == KT;
            const bool rowin = (unsigned)(py + rdy[kb]) < (unsigned)H;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = px0 + j;
                const bool in = rowin && px < W && (unsigned)(px + rdx[kb]) < (unsigned)W;
                const float xv = in && !ones ? xr[px * CIN + roff[kb]] : 0.f;
                v[j] = ones ? (px < W ? 1.f : 0.f) : xv;
            }
            split3_pair<3>(v[0], v[1], af[kb][0].x, af[kb][1].x, af[kb][2].x);
            split3_pair<3>(v[2], v[3], af[kb][0].y, af[kb][1].y, af[kb][2].y);
            split3_pair<3>(v[4], v[5], af[kb][0].z, af[kb][1].z, af[kb][2].z);
            split3_pair<3>(v[6], v[7], af[kb][0].w, af[kb][1].w, af[kb][2].w);
        }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            if (cb < ncb) {  // (wave-uniform)
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int px = px0 + j;
                    v[j] = px < W ? dr[(size_t)px * cout + 16 * cb + l16] : 0.f;
                }
                uint4 bh, bm, bl;
                split3_pair<3>(v[0], v[1], bh.x, bm.x, bl.x);
                split3_pair<3>(v[2], v[3], bh.y, bm.y, bl.y);
                split3_pair<3>(v[4], v[5], bh.z, bm.z, bl.z);
                split3_pair<3>(v[6], v[7], bh.w, bm.w, bl.w);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    fm_f32x4 c = acc[kb][cb];
                    c = punet_mfma16<false>(af[kb][1], bm, c);  // mm
                    c = punet_mfma16<false>(af[kb][0], bl, c);  // hl
                    c = punet_mfma16<false>(af[kb][2], bh, c);  // lh
                    c = punet_mfma16<false>(af[kb][0], bm, c);  // hm
                    c = punet_mfma16<false>(af[kb][1], bh, c);  // mh
                    c = punet_mfma16<false>(af[kb][0], bh, c);  // hh
                    acc[kb][cb] = c;
                }
            }
        }
    }
    // D[row = k][col = co]: lane (l16 = co within the block, g) holds rows 4 g .. 4 g + 3
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
            if (cb < ncb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sredf[((size_t)wave * KB * 16 + 16 * kb + 4 * g + r) * cout + 16 * cb + l16] = acc[kb][cb][r];
    __syncthreads();
    for (int u = tid; u < ROWS * cout; u += 256) {
        const int k = u / cout, co = u % cout;
        float sum = sredf[(size_t)k * cout + co];
        for (int wv = 1; wv < 4; ++wv) sum += sredf[((size_t)wv * KB * 16 + k) * cout + co];
        partial[((size_t)blockIdx.x * ROWS + k) * cout + co] = sum;
    }
}

// $PUNET_FIRST_MFMA=0: the VALU kernel for every channel count, as before round 5 (same-box A/B); read once per process
static int punet_first_mfma() {
    static const int on = [] {
        const char* e = getenv("PUNET_FIRST_MFMA");
        return !(e && e[0] == '0');
    }();
    return on;
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
    float* yf = (float*)y;
    if (out16 == 0 && cin >= 2 && cin <= 4 && cout % 16 == 0 && punet_first_mfma()) {  // fp32 output, 2..4 channels: on the matrix cores
        PUNET_CHECK_ARG((unsigned long long)B * H * W < (1ull << 31), "punet_conv3x3_small_fwd: more than 2^31 pixels");
        const int ks = (9 * cin + 31) / 32;
        const size_t lds = (size_t)3 * ks * 4 * cout * sizeof(uint4);
        PUNET_CHECK_ARG(lds <= 64 * 1024, "punet_conv3x3_small_fwd: cout %d needs %zu bytes of LDS for the weight planes", cout, lds);
        const long nblk = (long)B * H * ceil_div(W, 16);
        const int cap = 8 * punet_num_cus();  // 8 workgroups of 4 waves per CU: the kernel lives on memory-level parallelism
        const int grid1 = (int)((nblk + 7) / 8 < cap ? (nblk + 7) / 8 : cap);  // (a wave takes two blocks per iteration)
        switch (cin) {
            case 2: conv3x3_first_mfma_kernel<2><<<grid1, 256, lds, stream>>>(x, w, bias, yf, cout, B, H, W, relu, amax_y); break;
            case 3: conv3x3_first_mfma_kernel<3><<<grid1, 256, lds, stream>>>(x, w, bias, yf, cout, B, H, W, relu, amax_y); break;
            default: conv3x3_first_mfma_kernel<4><<<grid1, 256, lds, stream>>>(x, w, bias, yf, cout, B, H, W, relu, amax_y); break;
        }
        PUNET_LAUNCH_CHECK("conv3x3_first_mfma");
        return PUNET_OK;
    }
    const int ppb = 256 / (cout / 4);
    const int xiter = 8;
    const dim3 grid(ceil_div(W, ppb * xiter), B * H < 65535 ? B * H : 65535);
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
    if (cin >= 2 && cin <= 4 && cout % 16 == 0 && cout <= 128 && punet_first_mfma()) {  // 2..4 channels: on the matrix cores (round 5)
        const int rows = 9 * cin + 1, kb = (rows + 15) / 16;
        const size_t shmem = (size_t)4 * kb * 16 * cout * sizeof(float);
        const long nblk = (long)B * H * ceil_div(W, 32);
        int gridm = (int)((nblk + 3) / 4 < SMALL_WGRAD_BLOCKS ? (nblk + 3) / 4 : SMALL_WGRAD_BLOCKS);
#define PUNET_FIRST_WGRAD(CI)                                                                                                        \
    do {                                                                                                                             \
        if (shmem > 48 * 1024 && hipFuncSetAttribute((const void*)conv3x3_first_wgrad_mfma_kernel<CI>,                               \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {        \
            (void)hipGetLastError();                                                                                                 \
            punet_set_error("punet_conv3x3_small_wgrad: cannot opt in to %zu bytes of dynamic LDS", shmem);                          \
            return PUNET_ERR_HIP;                                                                                                    \
        }                                                                                                                            \
        conv3x3_first_wgrad_mfma_kernel<CI><<<gridm, 256, shmem, stream>>>(x, dz, workspace, cout, B, H, W);                         \
    } while (0)
        switch (cin) {
            case 2: PUNET_FIRST_WGRAD(2); break;
            case 3: PUNET_FIRST_WGRAD(3); break;
            default: PUNET_FIRST_WGRAD(4); break;
        }
#undef PUNET_FIRST_WGRAD
        PUNET_LAUNCH_CHECK("conv3x3_first_wgrad_mfma");
        const size_t slab = (size_t)rows * cout;
        return punet_launch_reduce_slabs_pair(workspace, slab, 9 * cin, cout, cin, cin, 0, dw, workspace + (size_t)9 * cin * cout, slab, dbias,
                                              gridm, accumulate, scale, stream);
    }
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
