// HBM-bound pieces of the Probabilistic U-Net hot path (NHWC fp32, float4-vectorised, deterministic
// two-pass reductions - no float atomics anywhere).  Each kernel cites the reference op it replaces.
#include "common.h"
#include "punet_hip.h"

#include <math.h>
#include <type_traits>

namespace {

constexpr int EW_BLOCK = 256;
inline int ew_grid(size_t n, int cap = 4096) {
    size_t g = (n + EW_BLOCK - 1) / EW_BLOCK;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
inline int pow2_shift(int v) {  // log2(v) when v is a power of two, else -1
    return v > 0 && (v & (v - 1)) == 0 ? __builtin_ctz((unsigned)v) : -1;
}
#define GRID_STRIDE(i, n) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------
// deterministic slab reduction (second pass of every split reduction)
// block = 32 outputs x 8 slice lanes; each lane sums its slices in order, lanes combined in order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ partial, int nslices,
                                                           size_t slab_stride, int rows, int cols, int cs, int cin_total,
                                                           int ci_off, float* __restrict__ out, int accumulate,
                                                           float scale) {
    __shared__ float sh[8][32];
    const size_t n = (size_t)rows * cols;
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + o;
    float s = 0.f;
    if (i < n)
        for (int k = sl; k < nslices; k += 8) s += partial[(size_t)k * slab_stride + i];
    sh[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = sh[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][o];
        const int row = (int)(i / cols), col = (int)(i % cols);
        const size_t oi = ((size_t)(row / cs) * cin_total + ci_off + row % cs) * cols + col;
        const float v = t * scale;
        out[oi] = accumulate ? out[oi] + v : v;
    }
}

// the same for 4 consecutive outputs per thread (cols % 4 == 0, 16-byte aligned slabs): identical summation order per
// output, a quarter of the load instructions (the scalar version reads the split-K slabs at ~3 TB/s)
__global__ __launch_bounds__(256) void reduce_slabs4_kernel(const float4* __restrict__ partial, int nslices,
                                                            size_t slab_stride4, int rows, int cols, int cs, int cin_total,
                                                            int ci_off, float* __restrict__ out, int accumulate, float scale) {
    __shared__ float4 sh[8][32];
    const size_t n4 = (size_t)rows * cols / 4;
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + o;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4)
        for (int k = sl; k < nslices; k += 8) {
            const float4 v = partial[(size_t)k * slab_stride4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    sh[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n4) {
        float4 t = sh[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) { t.x += sh[k][o].x; t.y += sh[k][o].y; t.z += sh[k][o].z; t.w += sh[k][o].w; }
        const size_t e = i * 4;
        const int row = (int)(e / cols), col = (int)(e % cols);
        float4* dst = reinterpret_cast<float4*>(out + ((size_t)(row / cs) * cin_total + ci_off + row % cs) * cols + col);
        float4 v = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
        if (accumulate) { const float4 p = *dst; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        *dst = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K4: AvgPool2d(2, 2, ceil_mode=True)  (unet_blocks.py:17, probabilistic_unet.py:54) and its backward
// ------------------------------------------------------------------------------------------------
__global__ void pool2x2_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, int B, int H, int W, int C4) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const size_t n = (size_t)B * OH * OW * C4;
    GRID_STRIDE(i, n) {
        unsigned p = (unsigned)i / (unsigned)C4;  // 32-bit index arithmetic (n < 2^31 checked on the host)
        const int c = (int)((unsigned)i - p * (unsigned)C4);
        unsigned q = p / (unsigned)OW;
        const int ox = (int)(p - q * (unsigned)OW);
        const unsigned bu = q / (unsigned)OH;
        const int oy = (int)(q - bu * (unsigned)OH);
        const size_t b = bu;
        const int y0 = oy * 2, x0 = ox * 2;
        const int y1 = min(y0 + 2, H), x1 = min(x0 + 2, W);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                const float4 v = x[((b * H + yy) * W + xx) * C4 + c];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        y[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    }
}

// dx[b][iy][ix] (+)= (x > 0) * dy[b][iy/2][ix/2] / window
// grid = (ceil(W * C4 / 256), B * H): the row-wise index arithmetic is block-uniform, a thread does one division (a shift
// when C4 is a power of two) instead of three per element of a grid-stride loop - these kernels were ALU-bound on the
// index arithmetic, not HBM-bound (2.8 TB/s; r02: see profiles/r02/README.md)
__global__ __launch_bounds__(256) void pool2x2_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ xmask,
                                                          const unsigned* __restrict__ xbits, float4* __restrict__ dx, int B,
                                                          int H, int W, int C4, int c4_shift, int accumulate) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    for (unsigned row = blockIdx.y; row < (unsigned)(B * H); row += gridDim.y) {  // b * rows + y
        const unsigned bu = row / (unsigned)H;
        const int iy = (int)(row - bu * (unsigned)H);
        const unsigned e = blockIdx.x * 256u + threadIdx.x;
        if (e >= (unsigned)(W * C4)) return;
        const int ix = c4_shift >= 0 ? (int)(e >> c4_shift) : (int)(e / (unsigned)C4);
        const int c = (int)e - ix * C4;
        const size_t b = bu;
        const size_t i = (size_t)row * W * C4 + e;
        const int oy = iy / 2, ox = ix / 2;
        const int wh = min(oy * 2 + 2, H) - oy * 2, ww = min(ox * 2 + 2, W) - ox * 2;
        const float inv = 1.f / (float)(wh * ww);
        const float4 g = dy[((b * OH + oy) * OW + ox) * C4 + c];
        float4 v = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
        if (xbits) {
            const unsigned g4 = punet_gate_nibble(xbits, b, H, W, 4 * C4, iy, ix, 4 * c);
            v.x = g4 & 1u ? v.x : 0.f; v.y = g4 & 2u ? v.y : 0.f; v.z = g4 & 4u ? v.z : 0.f; v.w = g4 & 8u ? v.w : 0.f;
        } else if (xmask) {
            const float4 m = xmask[i];
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
            v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (accumulate) {
            const float4 o = dx[i];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        dx[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K5: bilinear x2 upsampling, align_corners=True (unet_blocks.py:51) and its backward (gather form)
// source index as ATen computes it: scale = (in - 1) / (out - 1) in fp32, src = scale * dst.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up_src(int o, float scale, int in_size, int& i0, int& ip, float& l1) {
    punet_up_src(o, scale, in_size, i0, ip, l1);  // (common.h: shared with upconv.hip)
}

// grid = (ceil(OW * C4 / 256), B * OH): the source row, its neighbour and the row weight are block-uniform
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, int B, int H,
                                                             int W, int C4, int c4_shift) {
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    for (unsigned row = blockIdx.y; row < (unsigned)(B * OH); row += gridDim.y) {  // b * rows + y
        const unsigned bu = row / (unsigned)OH;
        const int oy = (int)(row - bu * (unsigned)OH);
        const unsigned e = blockIdx.x * 256u + threadIdx.x;
        if (e >= (unsigned)(OW * C4)) return;
        const int ox = c4_shift >= 0 ? (int)(e >> c4_shift) : (int)(e / (unsigned)C4);
        const int c = (int)e - ox * C4;
        const size_t b = bu;
        int h1, hp, w1, wp;
        float hl1, wl1;
        up_src(oy, sh, H, h1, hp, hl1);
        up_src(ox, sw, W, w1, wp, wl1);
        const float hl0 = 1.f - hl1, wl0 = 1.f - wl1;
        const float4* r0 = x + ((b * H + h1) * W) * C4 + c;
        const float4* r1 = x + ((b * H + h1 + hp) * W) * C4 + c;
        const float4 v00 = r0[(size_t)w1 * C4], v01 = r0[(size_t)(w1 + wp) * C4];
        const float4 v10 = r1[(size_t)w1 * C4], v11 = r1[(size_t)(w1 + wp) * C4];
        float4 o;
        o.x = hl0 * (wl0 * v00.x + wl1 * v01.x) + hl1 * (wl0 * v10.x + wl1 * v11.x);
        o.y = hl0 * (wl0 * v00.y + wl1 * v01.y) + hl1 * (wl0 * v10.y + wl1 * v11.y);
        o.z = hl0 * (wl0 * v00.z + wl1 * v01.z) + hl1 * (wl0 * v10.z + wl1 * v11.z);
        o.w = hl0 * (wl0 * v00.w + wl1 * v01.w) + hl1 * (wl0 * v10.w + wl1 * v11.w);
        y[(size_t)row * OW * C4 + e] = o;
    }
}

// the same on 16-bit tensors (T16 = __bf16: the "bf16s" fast mode; _Float16: "f16s"): 8-byte loads / stores of four channels,
// fp32 interpolation
template <typename T16>
__global__ __launch_bounds__(256) void upsample2x_fwd16_kernel(const void* __restrict__ xv, void* __restrict__ yv, int B, int H,
                                                               int W, int C4, int c4_shift) {
    typedef T16 t16x4 __attribute__((ext_vector_type(4)));
    const t16x4* x = reinterpret_cast<const t16x4*>(xv);
    t16x4* y = reinterpret_cast<t16x4*>(yv);
    auto ld4 = [](const t16x4* p) { const t16x4 v = *p; return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); };
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    for (unsigned row = blockIdx.y; row < (unsigned)(B * OH); row += gridDim.y) {
        const unsigned bu = row / (unsigned)OH;
        const int oy = (int)(row - bu * (unsigned)OH);
        const unsigned e = blockIdx.x * 256u + threadIdx.x;
        if (e >= (unsigned)(OW * C4)) return;
        const int ox = c4_shift >= 0 ? (int)(e >> c4_shift) : (int)(e / (unsigned)C4);
        const int c = (int)e - ox * C4;
        const size_t b = bu;
        int h1, hp, w1, wp;
        float hl1, wl1;
        up_src(oy, sh, H, h1, hp, hl1);
        up_src(ox, sw, W, w1, wp, wl1);
        const float hl0 = 1.f - hl1, wl0 = 1.f - wl1;
        const t16x4* r0 = x + ((b * H + h1) * W) * C4 + c;
        const t16x4* r1 = x + ((b * H + h1 + hp) * W) * C4 + c;
        const float4 v00 = ld4(r0 + (size_t)w1 * C4), v01 = ld4(r0 + (size_t)(w1 + wp) * C4);
        const float4 v10 = ld4(r1 + (size_t)w1 * C4), v11 = ld4(r1 + (size_t)(w1 + wp) * C4);
        t16x4 o;
        o[0] = (T16)(hl0 * (wl0 * v00.x + wl1 * v01.x) + hl1 * (wl0 * v10.x + wl1 * v11.x));
        o[1] = (T16)(hl0 * (wl0 * v00.y + wl1 * v01.y) + hl1 * (wl0 * v10.y + wl1 * v11.y));
        o[2] = (T16)(hl0 * (wl0 * v00.z + wl1 * v01.z) + hl1 * (wl0 * v10.z + wl1 * v11.z));
        o[3] = (T16)(hl0 * (wl0 * v00.w + wl1 * v01.w) + hl1 * (wl0 * v10.w + wl1 * v11.w));
        y[(size_t)row * OW * C4 + e] = o;
    }
}

__device__ __forceinline__ float up_weight(int o, int i, float scale, int in_size) { return punet_up_weight(o, i, scale, in_size); }

#ifndef PUNET_UPBWD_BATCH
#define PUNET_UPBWD_BATCH 1  // 0: round 2's form (per-tap divergent loads), for same-box A/B (tools/ab_variants.sh)
#endif
// dx[b][iy][ix] = (x > 0) * sum_{oy, ox} wy(oy, iy) * wx(ox, ix) * dy[b][oy][ox]   (gather form: deterministic)
// grid = (ceil(W * C4 / 256), B * H); the (up to 7) row weights are block-uniform, the column weights are computed once
// per thread instead of once per (oy, ox) pair; summation order (oy outer, ox inner, zero weights skipped) as before
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ xmask,
                                                             const unsigned* __restrict__ xbits, float4* __restrict__ dx, int B,
                                                             int H, int W, int C4, int c4_shift) {
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    for (unsigned row = blockIdx.y; row < (unsigned)(B * H); row += gridDim.y) {  // b * rows + y
        const unsigned bu = row / (unsigned)H;
        const int iy = (int)(row - bu * (unsigned)H);
        const unsigned e = blockIdx.x * 256u + threadIdx.x;
        if (e >= (unsigned)(W * C4)) return;
        const int ix = c4_shift >= 0 ? (int)(e >> c4_shift) : (int)(e / (unsigned)C4);
        const int c = (int)e - ix * C4;
        const size_t b = bu;
        const size_t i = (size_t)row * W * C4 + e;
        const int oy0 = max(0, 2 * iy - 2), oy1 = min(OH - 1, 2 * iy + 4);
        const int ox0 = max(0, 2 * ix - 2), ox1 = min(OW - 1, 2 * ix + 4);
        float wx[7];
    #pragma unroll
        for (int k = 0; k < 7; ++k) wx[k] = ox0 + k <= ox1 ? up_weight(ox0 + k, ix, sw, W) : 0.f;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = oy0; oy <= oy1; ++oy) {
            const float wy = up_weight(oy, iy, sh, H);  // block-uniform: the skip below is a scalar branch
            if (wy == 0.f) continue;
            const float4* drow = dy + ((b * OH + oy) * OW + ox0) * C4 + c;
#if PUNET_UPBWD_BATCH
            // all seven column taps of the row are loaded together (clamped to the row's last valid column; taps of weight zero
            // are selected away, not multiplied by zero) and only then accumulated, in the same order as before: with the loads
            // inside per-tap divergent branches every tap was its own load -> wait -> fma round trip (28 per thread)
            float4 g[7];
    #pragma unroll
            for (int k = 0; k < 7; ++k) g[k] = drow[(size_t)min(k, ox1 - ox0) * C4];
    #pragma unroll
            for (int k = 0; k < 7; ++k) {
                const bool on = wx[k] != 0.f;
                const float wgt = wy * wx[k];
                s.x = on ? s.x + wgt * g[k].x : s.x; s.y = on ? s.y + wgt * g[k].y : s.y;
                s.z = on ? s.z + wgt * g[k].z : s.z; s.w = on ? s.w + wgt * g[k].w : s.w;
            }
#else
    #pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (wx[k] == 0.f) continue;
                const float4 g = drow[(size_t)k * C4];
                const float wgt = wy * wx[k];
                s.x += wgt * g.x; s.y += wgt * g.y; s.z += wgt * g.z; s.w += wgt * g.w;
            }
#endif
        }
        if (xbits) {
            const unsigned g4 = punet_gate_nibble(xbits, b, H, W, 4 * C4, iy, ix, 4 * c);
            s.x = g4 & 1u ? s.x : 0.f; s.y = g4 & 2u ? s.y : 0.f; s.z = g4 & 4u ? s.z : 0.f; s.w = g4 & 8u ? s.w : 0.f;
        } else if (xmask) {
            const float4 m = xmask[i];
            s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f;
            s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
        }
        dx[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// K6: latent head = global average pool + 1x1 conv C -> 2L (probabilistic_unet.py:126-137)
// ------------------------------------------------------------------------------------------------
// partial[b][s][c] = sum over the pixels of slice s of x[b][p][c]
template <int IN16>  // IN16: x holds 1 = bf16 (the "bf16s" fast mode), 2 = fp16 ("f16s"); sums are fp32 either way
__global__ __launch_bounds__(256) void chan_sum_partial_kernel(const void* __restrict__ xv_, float4* __restrict__ partial,
                                                               int P, int C4, int S) {
    const float4* x = reinterpret_cast<const float4*>(xv_);
    // thread = (pixel lane pl, channel quad c); the 256 / C4 pixel lanes of a block are combined in lane order via LDS
    extern __shared__ __attribute__((aligned(16))) float4 sh4[];  // [ppl][C4]
    const int s = blockIdx.x, b = blockIdx.y;
    const int per = (P + S - 1) / S;
    const int p0 = s * per, p1 = min(P, p0 + per);
    const int ppl = C4 >= 256 ? 1 : 256 / C4;
    for (int c0 = 0; c0 < C4; c0 += 256) {
        const int c = c0 + (int)threadIdx.x % min(C4, 256), pl = (int)threadIdx.x / min(C4, 256);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C4 && pl < ppl)
            for (int p = p0 + pl; p < p1; p += ppl) {
                float4 v;
                if constexpr (IN16 != 0) {
                    typedef typename std::conditional<IN16 == 2, _Float16, __bf16>::type cs_t16;
                    typedef cs_t16 cs_bf16x4 __attribute__((ext_vector_type(4)));
                    const cs_bf16x4 h = reinterpret_cast<const cs_bf16x4*>(xv_)[((size_t)b * P + p) * C4 + c];
                    v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                } else {
                    v = x[((size_t)b * P + p) * C4 + c];
                }
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        __syncthreads();
        if (c < C4 && pl < ppl) sh4[pl * min(C4, 256) + (c - c0)] = a;
        __syncthreads();
        if (pl == 0 && c < C4) {
            float4 t = sh4[c - c0];
            for (int k = 1; k < ppl; ++k) {
                const float4 v = sh4[k * min(C4, 256) + (c - c0)];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            partial[((size_t)b * S + s) * C4 + c] = t;
        }
    }
}

// out[b][c] = scale * sum_s partial[b][s][c]
__global__ void chan_sum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int S, int C,
                                         float scale) {
    // block = 64 channels x 4 slice lanes; lane k sums slices k, k+4, ... and the four are added in lane order
    __shared__ float sh[4][64];
    const int b = blockIdx.x, c = blockIdx.y * 64 + ((int)threadIdx.x & 63), k = (int)threadIdx.x >> 6;
    float a = 0.f;
    if (c < C)
        for (int s = k; s < S; s += 4) a += partial[((size_t)b * S + s) * C + c];
    sh[k][threadIdx.x & 63] = a;
    __syncthreads();
    if (k == 0 && c < C) {
        const int l = threadIdx.x & 63;
        out[(size_t)b * C + c] = (((sh[0][l] + sh[1][l]) + sh[2][l]) + sh[3][l]) * scale;
    }
}

// muls[b][j] = bias[j] + sum_c mean[b][c] * w[c][j]
__global__ __launch_bounds__(256) void latent_head_kernel(const float* __restrict__ mean, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ muls, int C,
                                                          int J) {
    __shared__ float sh[256];
    const int b = blockIdx.x;
    for (int j = 0; j < J; ++j) {
        float a = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) a += mean[(size_t)b * C + c] * w[(size_t)c * J + j];
        sh[threadIdx.x] = a;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) muls[(size_t)b * J + j] = sh[0] + bias[j];
        __syncthreads();
    }
}

// z[n][b][d] = mu[b][d] + exp(logsig[b][d]) * eps[n][b][d]       (Normal.rsample, probabilistic_unet.py:302,349)
__global__ void reparam_kernel(const float* __restrict__ muls, const float* __restrict__ eps, float* __restrict__ z, int N,
                               int B, int L) {
    const int n = N * B * L;
    GRID_STRIDE(i, (size_t)n) {
        const int d = (int)(i % L), b = (int)((i / L) % B);
        z[i] = muls[b * 2 * L + d] + expf(muls[b * 2 * L + L + d]) * eps[i];
    }
}

// kl = mean_b sum_d KL(N(mu_q, s_q) || N(mu_p, s_p))   (torch.distributions.kl, probabilistic_unet.py:332,351)
__global__ void latent_kl_kernel(const float* __restrict__ mq, const float* __restrict__ mp, float* __restrict__ kl, int B,
                                 int L) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
        float kb = 0.f;
        for (int d = 0; d < L; ++d) {
            const float muq = mq[b * 2 * L + d], sq = expf(mq[b * 2 * L + L + d]);
            const float mup = mp[b * 2 * L + d], sp = expf(mp[b * 2 * L + L + d]);
            const float r = sq / sp, vr = r * r;
            const float t = (muq - mup) / sp, t1 = t * t;
            kb += 0.5f * (vr + t1 - 1.f - logf(vr));
        }
        tot += kb;
    }
    kl[0] = tot / (float)B;
}

// gradients of  gscale * beta * kl  and of z = mu_q + s_q * eps (upstream dz) w.r.t. the 1x1-conv outputs
__global__ void latent_kl_bwd_kernel(const float* __restrict__ mq, const float* __restrict__ mp, const float* __restrict__ eps,
                                     const float* __restrict__ dz, float coef, float* __restrict__ dq, float* __restrict__ dp,
                                     int B, int L) {
    const int n = B * L;
    GRID_STRIDE(i, (size_t)n) {
        const int d = (int)(i % L), b = (int)(i / L);
        const float muq = mq[b * 2 * L + d], sq = expf(mq[b * 2 * L + L + d]);
        const float mup = mp[b * 2 * L + d], sp = expf(mp[b * 2 * L + L + d]);
        const float ivp = 1.f / (sp * sp);
        const float diff = muq - mup;
        const float g = dz[i];
        dq[b * 2 * L + d] = g + coef * diff * ivp;
        dq[b * 2 * L + L + d] = g * sq * eps[i] + coef * (sq * sq * ivp - 1.f);
        dp[b * 2 * L + d] = -coef * diff * ivp;
        dp[b * 2 * L + L + d] = coef * (1.f - sq * sq * ivp - diff * diff * ivp);
    }
}

// per-image KL (probabilistic_unet.py:324-339): analytic (z == NULL; torch.distributions.kl, :332) or the single-sample
// estimate log q(z) - log p(z) of :334-338 (the -log sqrt(2 pi) terms of the two log_probs cancel); kl_mean = batch mean
// (:351).  One thread per image, images summed in order (B is the per-GPU batch).
__global__ void latent_kl_rows_kernel(const float* __restrict__ mq, const float* __restrict__ mp, const float* __restrict__ z,
                                      float* __restrict__ kl_rows, float* __restrict__ kl_mean, int B, int L) {
    __shared__ float sh[256];
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + threadIdx.x;
        if (b < B) {
            float kb = 0.f;
            if (z == nullptr) {
                for (int d = 0; d < L; ++d) {
                    const float muq = mq[b * 2 * L + d], sq = expf(mq[b * 2 * L + L + d]);
                    const float mup = mp[b * 2 * L + d], sp = expf(mp[b * 2 * L + L + d]);
                    const float r = sq / sp, vr = r * r;
                    const float t = (muq - mup) / sp, t1 = t * t;
                    kb += 0.5f * (vr + t1 - 1.f - logf(vr));
                }
            } else {
                float lq = 0.f, lp = 0.f;  // log_prob sums as Normal.log_prob writes them (minus the common constant)
                for (int d = 0; d < L; ++d) {
                    const float zz = z[b * L + d];
                    const float muq = mq[b * 2 * L + d], lsq = mq[b * 2 * L + L + d], sq = expf(lsq);
                    const float mup = mp[b * 2 * L + d], lsp = mp[b * 2 * L + L + d], sp = expf(lsp);
                    lq += -((zz - muq) * (zz - muq)) / (2.f * sq * sq) - lsq;
                    lp += -((zz - mup) * (zz - mup)) / (2.f * sp * sp) - lsp;
                }
                kb = lq - lp;
            }
            if (kl_rows) kl_rows[b] = kb;
            sh[threadIdx.x] = kb;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = b0 == 0 ? 0.f : kl_mean[0];
            const int nb = B - b0 < 256 ? B - b0 : 256;
            for (int i = 0; i < nb; ++i) tot += sh[i];
            kl_mean[0] = (b0 + 256 >= B) ? tot / (float)B : tot;
        }
        __syncthreads();
    }
}

// latent_kl_bwd with the two options of elbo() (probabilistic_unet.py:341-358): kl_sampled != 0: the KL term is the
// single-sample estimate at z = mu_q + s_q * eps (analytic_kl=False); z_is_mean != 0: Fcomb saw mu_q instead of z
// (reconstruct_posterior_mean=True), so the upstream dz reaches mu_q only.
__global__ void latent_kl_bwd_ex_kernel(const float* __restrict__ mq, const float* __restrict__ mp, const float* __restrict__ eps,
                                        const float* __restrict__ dz, float coef, int kl_sampled, int z_is_mean,
                                        float* __restrict__ dq, float* __restrict__ dp, int B, int L) {
    const int n = B * L;
    GRID_STRIDE(i, (size_t)n) {
        const int d = (int)(i % L), b = (int)(i / L);
        const float muq = mq[b * 2 * L + d], sq = expf(mq[b * 2 * L + L + d]);
        const float mup = mp[b * 2 * L + d], sp = expf(mp[b * 2 * L + L + d]);
        const float ivp = 1.f / (sp * sp);
        const float g = dz[i], e = eps[i];
        float kmq, ksq, kmp, ksp;
        if (kl_sampled) {  // log q(z) = sum -ls_q - eps^2/2 (z - mu_q = s_q eps exactly), log p(z) = sum -ls_p - (z - mu_p)^2 / (2 s_p^2)
            const float u = (muq + sq * e - mup);
            kmq = u * ivp;
            ksq = -1.f + u * sq * e * ivp;
            kmp = -u * ivp;
            ksp = 1.f - u * u * ivp;
        } else {
            const float diff = muq - mup;
            kmq = diff * ivp;
            ksq = sq * sq * ivp - 1.f;
            kmp = -diff * ivp;
            ksp = 1.f - sq * sq * ivp - diff * diff * ivp;
        }
        dq[b * 2 * L + d] = g + coef * kmq;
        dq[b * 2 * L + L + d] = (z_is_mean ? 0.f : g * sq * e) + coef * ksq;
        dp[b * 2 * L + d] = coef * kmp;
        dp[b * 2 * L + L + d] = coef * ksp;
    }
}

// (mu, sigma) of Independent(Normal(mu, exp(log_sigma)), 1) as separate tensors (probabilistic_unet.py:136-141)
__global__ void latent_split_kernel(const float* __restrict__ muls, float* __restrict__ loc, float* __restrict__ scale, int B,
                                    int L) {
    GRID_STRIDE(i, (size_t)(B * L)) {
        const int d = (int)(i % L), b = (int)(i / L);
        loc[i] = muls[b * 2 * L + d];
        scale[i] = expf(muls[b * 2 * L + L + d]);
    }
}

// torch.cat([a, b], dim=1) of two NHWC tensors with few channels (the posterior's [patch, segm] input,
// probabilistic_unet.py:115-118): out[p][0:ca] = a[p], out[p][ca:ca+cb] = b[p]
__global__ void concat_nhwc_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                   float* __restrict__ out, size_t P) {
    const int c = ca + cb;
    GRID_STRIDE(i, P * (size_t)c) {
        const size_t p = i / c;
        const int k = (int)(i % c);
        out[i] = k < ca ? a[p * ca + k] : b[p * cb + (k - ca)];
    }
}

// [B,C,H,W] -> [B,H,W,C] (the API takes the reference's NCHW patches; the kernels run NHWC)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, size_t HW) {
    GRID_STRIDE(i, (size_t)B * C * HW) {
        const int c = (int)(i % C);
        const size_t p = (i / C) % HW, b = i / ((size_t)C * HW);
        y[i] = x[(b * C + c) * HW + p];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, size_t HW) {
    GRID_STRIDE(i, (size_t)B * C * HW) {
        const size_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const size_t b = i / ((size_t)C * HW);
        y[i] = x[(b * HW + p) * C + c];
    }
}

// FixMatch distribution alignment (fixmatch_trainer.py:77-84): count (y >= 0.5) exactly (integer atomics: order
// independent), then y = clip(y * (source[c] / target[c]), 0, 1) with c = (y >= 0.5), target = counts / numel
__global__ __launch_bounds__(256) void count_ge_kernel(const float* __restrict__ y, size_t P, float thr,
                                                       unsigned long long* __restrict__ count) {
    __shared__ unsigned sh[4];
    unsigned c = 0;
    GRID_STRIDE(i, P) c += y[i] >= thr ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, (unsigned long long)(sh[0] + sh[1] + sh[2] + sh[3]));
}
__global__ void align_distribution_kernel(const float* __restrict__ y, const float* __restrict__ src, size_t P,
                                          const unsigned long long* __restrict__ count, float* __restrict__ out,
                                          float* __restrict__ ratio) {
    const float n = (float)P, pos = (float)count[0], neg = (float)(P - count[0]);
    const float r0 = src[0] / (neg / n), r1 = src[1] / (pos / n);
    if (ratio && blockIdx.x == 0 && threadIdx.x == 0) { ratio[0] = r0; ratio[1] = r1; }
    GRID_STRIDE(i, P) {
        const float v = y[i];
        out[i] = fminf(fmaxf(v < 0.5f ? v * r0 : v * r1, 0.f), 1.f);
    }
}

// backward of the latent head's 1x1 conv on the pooled vector
__global__ void latent_head_bwd_kernel(const float* __restrict__ d, const float* __restrict__ mean, const float* __restrict__ w,
                                       float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dmean, int B, int C,
                                       int J, int accumulate) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < C * J; i += nth) {
        const int c = i / J, j = i % J;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += mean[(size_t)b * C + c] * d[b * J + j];
        dw[i] = accumulate ? dw[i] + a : a;
    }
    for (int j = tid; j < J; j += nth) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += d[b * J + j];
        db[j] = accumulate ? db[j] + a : a;
    }
    for (int i = tid; i < B * C; i += nth) {
        const int b = i / C, c = i % C;
        float a = 0.f;
        for (int j = 0; j < J; ++j) a += w[(size_t)c * J + j] * d[b * J + j];
        dmean[i] = a;
    }
}

// denc[b][p][c] = (enc > 0) * dmean[b][c] * inv_p     (backward of the two means, through the last ReLU)
__global__ void bcast_mean_bwd_kernel(const float4* __restrict__ dmean, const float4* __restrict__ enc,
                                      float4* __restrict__ denc, int B, int P, int C4, float inv_p) {
    const size_t n = (size_t)B * P * C4;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C4);
        const size_t b = i / ((size_t)P * C4);
        const float4 g = dmean[b * C4 + c], m = enc[i];
        float4 v;
        v.x = m.x > 0.f ? g.x * inv_p : 0.f; v.y = m.y > 0.f ? g.y * inv_p : 0.f;
        v.z = m.z > 0.f ? g.z * inv_p : 0.f; v.w = m.w > 0.f ? g.w * inv_p : 0.f;
        denc[i] = v;
    }
}

// ReLU gate bits (punet_gate_dwords layout) of a bf16 activation tensor: for producers that do not publish them themselves (the
// first-layer kernel in the "bf16s" mode).  One thread per dword: pixel x = 32 t + (l & 31), bit 16 nb + 4 rg + e <-> channel
// 64 cb + 32 nb + 8 rg + 4 (l >> 5) + e.
template <typename T16>
__global__ void gate_bits16_kernel(const T16* __restrict__ y, unsigned* __restrict__ bits, int B, int H, int W, int C) {
    const int tiles = (W + 31) >> 5, cbs = (C + 63) >> 6;
    const size_t n = (size_t)B * H * tiles * cbs * 64;
    GRID_STRIDE(i, n) {
        const int l = (int)(i & 63);
        size_t r = i >> 6;
        const int cb = (int)(r % cbs); r /= cbs;
        const int t = (int)(r % tiles); r /= tiles;
        const int yy = (int)(r % H);
        const size_t b = r / H;
        const int x = 32 * t + (l & 31), half = l >> 5;
        unsigned g = 0u;
        if (x < W) {
            const T16* p = y + ((b * H + yy) * W + x) * C;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c = 64 * cb + 32 * nb + 8 * rg + 4 * half;
                    if (c < C) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) g |= ((float)p[c + e] > 0.f ? 1u : 0u) << (16 * nb + 4 * rg + e);
                    }
                }
        }
        bits[i] = g;
    }
}

// the same gated by the encoder output's ReLU gate BITS (punet_gate_dwords layout) instead of its values
__global__ void bcast_mean_bwd_bits_kernel(const float4* __restrict__ dmean, const unsigned* __restrict__ bits,
                                           float4* __restrict__ denc, int B, int H, int W, int C4, float inv_p) {
    const size_t P = (size_t)H * W, n = (size_t)B * P * C4;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C4);
        const size_t pix = (i / C4) % P, b = i / (P * C4);
        const int y = (int)(pix / W), x = (int)(pix % W);
        const float4 g = dmean[b * C4 + c];
        const unsigned g4 = punet_gate_nibble(bits, b, H, W, 4 * C4, y, x, 4 * c);
        float4 v;
        v.x = g4 & 1u ? g.x * inv_p : 0.f; v.y = g4 & 2u ? g.y * inv_p : 0.f;
        v.z = g4 & 4u ? g.z * inv_p : 0.f; v.w = g4 & 8u ? g.w * inv_p : 0.f;
        denc[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Fcomb pieces (probabilistic_unet.py:200-214): the z term of the first 1x1 conv folded into a per-image
// bias, and the last 1x1 conv C -> 1 with its backward
// ------------------------------------------------------------------------------------------------
// zb[n][b][co] = b1[co] + sum_d w1[(CH + d)][co] * z[n][b][d]
__global__ void fcomb_zbias_kernel(const float* __restrict__ z, const float* __restrict__ w1, const float* __restrict__ b1,
                                   float* __restrict__ zb, int NB, int L, int CH) {
    const int n = NB * CH;
    GRID_STRIDE(i, (size_t)n) {
        const int co = (int)(i % CH), nb = (int)(i / CH);
        float a = b1[co];
        for (int d = 0; d < L; ++d) a += w1[(size_t)(CH + d) * CH + co] * z[nb * L + d];
        zb[i] = a;
    }
}

// logits[p] = b3 + sum_c h[p][c] * w3[c];   C/4 lanes per pixel
__global__ __launch_bounds__(256) void dot_bias_kernel(const float4* __restrict__ h, const float4* __restrict__ w3,
                                                       const float* __restrict__ b3, float* __restrict__ out, size_t P,
                                                       int C4) {
    const int g = threadIdx.x % C4, pl = threadIdx.x / C4, ppb = 256 / C4;
    const float4 wv = w3[g];
    const float bv = b3[0];
    for (size_t p = (size_t)blockIdx.x * ppb + pl; p < P; p += (size_t)gridDim.x * ppb) {
        const float4 v = h[p * C4 + g];
        float a = v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        for (int off = 1; off < C4; off <<= 1) a += __shfl_xor(a, off);
        if (g == 0) out[p] = a + bv;
    }
}

// dZ[p][c] = (h > 0) * dlogit[p] * w3[c];  partial[block][0..C-1] = sum_p dlogit[p] * h[p][c];  partial[block][C] = sum_p dlogit[p]
__global__ __launch_bounds__(256) void dot_bwd_kernel(const float4* __restrict__ h, const float* __restrict__ dlogit,
                                                      const float4* __restrict__ w3, float4* __restrict__ dz,
                                                      float* __restrict__ partial, size_t P, int C4) {
    __shared__ float4 sw[4][65];
    __shared__ float sb[4];
    const int g = threadIdx.x % C4, pl = threadIdx.x / C4, ppb = 256 / C4;
    const float4 wv = w3[g];
    float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
    float ab = 0.f;
    for (size_t p = (size_t)blockIdx.x * ppb + pl; p < P; p += (size_t)gridDim.x * ppb) {
        const float4 v = h[p * C4 + g];
        const float d = dlogit[p];
        float4 o;
        o.x = v.x > 0.f ? d * wv.x : 0.f; o.y = v.y > 0.f ? d * wv.y : 0.f;
        o.z = v.z > 0.f ? d * wv.z : 0.f; o.w = v.w > 0.f ? d * wv.w : 0.f;
        dz[p * C4 + g] = o;
        aw.x += d * v.x; aw.y += d * v.y; aw.z += d * v.z; aw.w += d * v.w;
        if (g == 0) ab += d;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = C4; off < 64; off <<= 1) {
        aw.x += __shfl_xor(aw.x, off); aw.y += __shfl_xor(aw.y, off);
        aw.z += __shfl_xor(aw.z, off); aw.w += __shfl_xor(aw.w, off);
        ab += __shfl_xor(ab, off);
    }
    if (lane < C4) sw[wave][lane] = aw;
    if (lane == 0) sb[wave] = ab;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 s = sw[0][threadIdx.x];
        for (int k = 1; k < 4; ++k) {
            const float4 v = sw[k][threadIdx.x];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* o = partial + (size_t)blockIdx.x * (C4 * 4 + 1) + threadIdx.x * 4;
        o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * (C4 * 4 + 1) + C4 * 4] = sb[0] + sb[1] + sb[2] + sb[3];
}

// ------------------------------------------------------------------------------------------------
// K9: reconstruction loss (probabilistic_unet.py:347-369) forward + d/dlogit, with the consensus mask
// multiplied into BOTH logits and targets exactly as :363-364 does
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
    return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void bce_fwd_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                          const float* __restrict__ consm, float gscale,
                                                          float* __restrict__ dlogit, float* __restrict__ partial, size_t P) {
    __shared__ float sh[4];
    float acc = 0.f;
    GRID_STRIDE(i, P) {
        const float cm = consm ? consm[i] : 1.f;
        const float x = logits[i] * cm, t = target[i] * cm;
        // BCEWithLogits: (1 - t) * x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
        acc += (1.f - t) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
        if (dlogit) dlogit[i] = gscale * (sigmoidf_(x) - t) * cm;
    }
    const float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// [RECALLED, parity unpinned] torch_em DiceLossWithLogits, one channel: sums of p*t, p*p, t*t
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                        const float* __restrict__ consm, float* __restrict__ partial, size_t P) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f, c = 0.f;
    GRID_STRIDE(i, P) {
        const float cm = consm ? consm[i] : 1.f;
        const float p = sigmoidf_(logits[i] * cm), t = target[i] * cm;
        a += p * t; b += p * p; c += t * t;
    }
    const float ta = block_sum(a, sh), tb = block_sum(b, sh), tc = block_sum(c, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = ta;
        partial[gridDim.x + blockIdx.x] = tb;
        partial[2 * gridDim.x + blockIdx.x] = tc;
    }
}

// sums[3] -> loss = 1 - 2 a / max(b + c, eps);  dlogit = gscale * dloss/dp * p (1 - p) * cm
__global__ void dice_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                const float* __restrict__ consm, const float* __restrict__ sums, float gscale,
                                float* __restrict__ dlogit, float* __restrict__ loss_out, size_t P) {
    const float a = sums[0], den = sums[1] + sums[2];
    const float eps = 1e-7f;
    const bool clamped = den < eps;
    const float D = clamped ? eps : den;
    if (blockIdx.x == 0 && threadIdx.x == 0) loss_out[0] = 1.f - 2.f * a / D;
    GRID_STRIDE(i, P) {
        const float cm = consm ? consm[i] : 1.f;
        const float p = sigmoidf_(logits[i] * cm), t = target[i] * cm;
        float dp = -2.f * t / D;
        if (!clamped) dp += 2.f * a * 2.f * p / (D * D);
        dlogit[i] = gscale * dp * p * (1.f - p) * cm;
    }
}

// out[k] = sum_i partial[k * n + i]  (double accumulation, one block)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, int K,
                                                           float* __restrict__ out) {
    __shared__ double sh[256];
    for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) a += (double)partial[(size_t)k * n + i];
        sh[threadIdx.x] = a;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = (float)sh[0];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K8: consensus / mean over N Monte-Carlo samples (mean_teacher_trainer.py:74-86, punet_predictions.py:114-124)
// ------------------------------------------------------------------------------------------------
__global__ void consensus_kernel(const float* __restrict__ logits, int N, size_t P, float upper, float lower, int masking,
                                 float* __restrict__ y, float* __restrict__ z) {
    GRID_STRIDE(i, P) {
        float s = 0.f, cnt = 0.f;
        for (int n = 0; n < N; ++n) {
            const float v = sigmoidf_(logits[(size_t)n * P + i]);
            s += v;
            cnt += (v >= upper || v <= lower) ? 1.f : 0.f;
        }
        y[i] = s / (float)N;
        const float c = cnt / (float)N;
        z[i] = masking ? (c == 1.f ? 1.f : 0.f) : c;
    }
}

// ------------------------------------------------------------------------------------------------
// E1: dice_score sums (my_utils/util.py:17-44): [sum(gt*seg), sum(gt), sum(seg)] with optional thresholds
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_score_sums_kernel(const float* __restrict__ seg, const float* __restrict__ gt,
                                                              float thr_seg, int use_thr_seg, float thr_gt, int use_thr_gt,
                                                              float* __restrict__ partial, size_t P) {
    __shared__ float sh[4];
    float a = 0.f, b = 0.f, c = 0.f;
    GRID_STRIDE(i, P) {
        const float s = use_thr_seg ? (seg[i] > thr_seg ? 1.f : 0.f) : seg[i];
        const float g = use_thr_gt ? (gt[i] > thr_gt ? 1.f : 0.f) : gt[i];
        a += g * s; b += g; c += s;
    }
    const float ta = block_sum(a, sh), tb = block_sum(b, sh), tc = block_sum(c, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = ta;
        partial[gridDim.x + blockIdx.x] = tb;
        partial[2 * gridDim.x + blockIdx.x] = tc;
    }
}

// ------------------------------------------------------------------------------------------------
// K10: l2_regularisation (utils.py:32-40): sum over tensors of ||W||_2 ; gradient W / ||W||
// chunk table: {offset, length, tensor index}; a tensor's chunks are consecutive in the table
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_sumsq_kernel(const float* __restrict__ w, const int64_t* __restrict__ chunks,
                                                       float* __restrict__ partial) {
    __shared__ float sh[4];
    const int64_t off = chunks[blockIdx.x * 3], len = chunks[blockIdx.x * 3 + 1];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < len; i += 256) {
        const float v = w[off + i];
        a += v * v;
    }
    const float t = block_sum(a, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void l2_finalize_kernel(const float* __restrict__ partial, const int64_t* __restrict__ chunks, int nchunks,
                                   int ntensors, float* __restrict__ norms, float* __restrict__ reg_out) {
    // one thread per tensor: its chunks are consecutive in the table (tensor ids ascending) -> binary search for the first,
    // then a fixed-order sum; thread 0 adds the norms in tensor order.  (A single-thread version of this took 215 us.)
    __shared__ float sn[256];
    for (int t0 = 0; t0 < ntensors; t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        float nrm = 0.f;
        if (t < ntensors) {
            int lo = 0, hi = nchunks;  // first chunk with tensor id >= t
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (chunks[mid * 3 + 2] < t) lo = mid + 1; else hi = mid;
            }
            float s = 0.f;
            for (int c = lo; c < nchunks && chunks[c * 3 + 2] == t; ++c) s += partial[c];
            nrm = sqrtf(s);
            norms[t] = nrm;
        }
        sn[threadIdx.x] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = t0 == 0 ? 0.f : reg_out[0];
            const int n = ntensors - t0 < 256 ? ntensors - t0 : 256;
            for (int k = 0; k < n; ++k) tot += sn[k];
            reg_out[0] = tot;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void l2_add_grad_kernel(const float* __restrict__ w, float* __restrict__ g,
                                                          const int64_t* __restrict__ chunks, const float* __restrict__ norms,
                                                          float coef) {
    const int64_t off = chunks[blockIdx.x * 3], len = chunks[blockIdx.x * 3 + 1];
    const float nrm = norms[chunks[blockIdx.x * 3 + 2]];
    if (nrm == 0.f) return;
    const float k = coef / nrm;
    for (int64_t i = threadIdx.x; i < len; i += 256) g[off + i] += k * w[off + i];
}

// ------------------------------------------------------------------------------------------------
// K11 + K12: torch.optim.Adam step (defaults; LIVECell/livecell_mt.py:83) fused with the EMA teacher update
// (mean_teacher_trainer.py:52-55) over the flat parameter buffer
// ------------------------------------------------------------------------------------------------
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                float* __restrict__ teacher, size_t n, float gscale, float step_size, float beta1, float beta2,
                                float bc2_sqrt, float eps, float ema_m, int do_adam) {
    GRID_STRIDE(i, n) {
        float pv = p[i];
        if (do_adam) {
            const float gv = g[i] * gscale;
            float mv = m[i], vv = v[i];
            mv = mv + (gv - mv) * (1.f - beta1);
            vv = vv * beta2 + (1.f - beta2) * gv * gv;
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            pv = pv - step_size * (mv / denom);
            m[i] = mv; v[i] = vv; p[i] = pv;
        }
        if (teacher) teacher[i] = teacher[i] * ema_m + pv * (1.f - ema_m);
    }
}

// backward of the folded z term of Fcomb's first 1x1 conv: S[b][co] = sum_pixels dZ1[b][.][co]
//   dW1[(CH + d)][co] (+)= sum_b z[b][d] * S[b][co]          dz[b][d] = sum_co W1[(CH + d)][co] * S[b][co]
// The step-dependent scalars of Adam + EMA computed ON THE DEVICE from a small state block, so that a step captured in a
// hipGraph replays with the right bias corrections (host scalars would be frozen into the graph).
// hs (double[8]): [0] Adam steps taken, [1] lr, [2] beta1, [3] beta2, [4] EMA momentum (cap), [5] EMA rule (0 none,
// 1 constant - mean_teacher_trainer.py:52-55, 2 warm-up min(1 - 1/(it + 1), m) - adamt_trainer.py:41), [6] trainer iteration
// `it` (advanced here), [7] gradient scale (1 / world_size).
// hyper (float[4]): gscale, step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t), EMA momentum of this step.
__global__ void adam_hyper_kernel(double* __restrict__ hs, float* __restrict__ hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t = hs[0] + 1.0;
    hs[0] = t;
    const double bc1 = 1.0 - pow(hs[2], t), bc2 = 1.0 - pow(hs[3], t);
    hyper[0] = (float)hs[7];
    hyper[1] = (float)(hs[1] / bc1);
    hyper[2] = (float)sqrt(bc2);
    const double it = hs[6];
    const double mw = 1.0 - 1.0 / (it + 1.0);
    hyper[3] = hs[5] == 2.0 ? (float)(mw < hs[4] ? mw : hs[4]) : (float)hs[4];
    hs[6] = it + 1.0;
}

__global__ void adam_ema_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, float* __restrict__ teacher, size_t n,
                                    const float* __restrict__ hyper, float beta1, float beta2, float eps) {
    const float gscale = hyper[0], step_size = hyper[1], bc2_sqrt = hyper[2], ema_m = hyper[3];
    GRID_STRIDE(i, n) {
        const float gv = g[i] * gscale;
        float pv = p[i], mv = m[i], vv = v[i];
        mv = mv + (gv - mv) * (1.f - beta1);
        vv = vv * beta2 + (1.f - beta2) * gv * gv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pv = pv - step_size * (mv / denom);
        m[i] = mv; v[i] = vv; p[i] = pv;
        if (teacher) teacher[i] = teacher[i] * ema_m + pv * (1.f - ema_m);
    }
}

__global__ void fcomb_z_bwd_kernel(const float* __restrict__ S, const float* __restrict__ z, const float* __restrict__ w1,
                                   float* __restrict__ dw1, float* __restrict__ dz, int B, int L, int CH, int accumulate) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < L * CH; i += nth) {
        const int d = i / CH, co = i % CH;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += z[b * L + d] * S[b * CH + co];
        float* o = dw1 + (size_t)(CH + d) * CH + co;
        *o = accumulate ? *o + a : a;
    }
    for (int i = tid; i < B * L; i += nth) {
        const int b = i / L, d = i % L;
        float a = 0.f;
        for (int co = 0; co < CH; ++co) a += w1[(size_t)(CH + d) * CH + co] * S[b * CH + co];
        dz[i] = a;
    }
}

__global__ void scale_kernel(float* __restrict__ x, size_t n, float s) {
    GRID_STRIDE(i, n) x[i] *= s;
}

}  // namespace

int punet_launch_reduce_slabs(const float* partial, int nslices, size_t slab_stride, int rows, int cols, int cs,
                              int cin_total, int ci_off, float* out, int accumulate, float scale, hipStream_t stream) {
    const size_t n = (size_t)rows * cols;
    if (cols % 4 == 0 && slab_stride % 4 == 0 && ((uintptr_t)partial & 15) == 0 && ((uintptr_t)out & 15) == 0)
        reduce_slabs4_kernel<<<(unsigned)((n / 4 + 31) / 32), 256, 0, stream>>>((const float4*)partial, nslices, slab_stride / 4,
                                                                              rows, cols, cs, cin_total, ci_off, out,
                                                                              accumulate, scale);
    else
        reduce_slabs_kernel<<<(unsigned)((n + 31) / 32), 256, 0, stream>>>(partial, nslices, slab_stride, rows, cols, cs,
                                                                           cin_total, ci_off, out, accumulate, scale);
    PUNET_LAUNCH_CHECK("reduce_slabs");
    return PUNET_OK;
}

// Two reductions in ONE launch (a layer's weight slabs and its bias slabs): the first nb0 blocks work on job 0, the rest
// on job 1.  Per output the summation is exactly reduce_slabs_kernel's (lane sl sums slices sl, sl + 8, ... in order, lanes
// combined in order), so results are bit-identical to two separate launches.
struct SlabJob {
    const float* partial;
    size_t slab_stride;
    int rows, cols, cs, cin_total, ci_off;
    float* out;
};

__global__ __launch_bounds__(256) void reduce_slabs_pair_kernel(SlabJob j0, SlabJob j1, unsigned nb0, int vec0, int nslices,
                                                                int accumulate, float scale) {
    __shared__ float4 sh[8][32];
    const bool first = blockIdx.x < nb0;
    const SlabJob& j = first ? j0 : j1;
    const int vec = first ? vec0 : 1;  // job 0 may run 4 outputs per thread, job 1 (the short one) is scalar
    const unsigned blk = first ? blockIdx.x : blockIdx.x - nb0;
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t n = (size_t)j.rows * j.cols / vec;
    const size_t i = (size_t)blk * 32 + o;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        if (vec == 4) {
            const float4* p4 = reinterpret_cast<const float4*>(j.partial);
            const size_t st4 = j.slab_stride / 4;
            for (int k = sl; k < nslices; k += 8) {
                const float4 v = p4[(size_t)k * st4 + i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        } else {
            for (int k = sl; k < nslices; k += 8) s.x += j.partial[(size_t)k * j.slab_stride + i];
        }
    }
    sh[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float4 t = sh[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) { t.x += sh[k][o].x; t.y += sh[k][o].y; t.z += sh[k][o].z; t.w += sh[k][o].w; }
        const size_t e = i * vec;
        const int row = (int)(e / j.cols), col = (int)(e % j.cols);
        float* dst = j.out + ((size_t)(row / j.cs) * j.cin_total + j.ci_off + row % j.cs) * j.cols + col;
        if (vec == 4) {
            float4 v = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
            if (accumulate) { const float4 q = *reinterpret_cast<float4*>(dst); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            *reinterpret_cast<float4*>(dst) = v;
        } else {
            const float v = t.x * scale;
            *dst = accumulate ? *dst + v : v;
        }
    }
}

int punet_launch_reduce_slabs_pair(const float* partial_w, size_t stride_w, int rows, int cols, int cs, int cin_total, int ci_off,
                                   float* out_w, const float* partial_b, size_t stride_b, float* out_b, int nslices,
                                   int accumulate, float scale, hipStream_t stream) {
    const size_t n = (size_t)rows * cols;
    const int vec0 = (cols % 4 == 0 && stride_w % 4 == 0 && ((uintptr_t)partial_w & 15) == 0 && ((uintptr_t)out_w & 15) == 0) ? 4 : 1;
    const unsigned nb0 = (unsigned)((n / vec0 + 31) / 32), nb1 = out_b ? (unsigned)((cols + 31) / 32) : 0u;  // out_b may be null
    SlabJob j0{partial_w, stride_w, rows, cols, cs, cin_total, ci_off, out_w};
    SlabJob j1{partial_b, stride_b, 1, cols, 1, 1, 0, out_b};
    reduce_slabs_pair_kernel<<<nb0 + nb1, 256, 0, stream>>>(j0, j1, nb0, vec0, nslices, accumulate, scale);
    PUNET_LAUNCH_CHECK("reduce_slabs_pair");
    return PUNET_OK;
}

// ---- amax slots of the fp16-operand modes (common.h): max |x| of a tensor (standalone pass, for producers that do not publish
// it themselves) and an upper BOUND composed from other slots, out = ca * a + cb * b (pool / up-sampling backward: the output's
// magnitude follows from the input's; a bound costs range, not precision - the scaled values keep 39 binades below the bound)
__global__ __launch_bounds__(256) void amax_kernel(const float4* __restrict__ x, size_t n4, const float* __restrict__ tail, int ntail,
                                                   float* __restrict__ slot) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per thread
        const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(b.x), fabsf(b.y))), fmaxf(fabsf(b.z), fabsf(b.w)));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(c.x), fabsf(c.y))), fmaxf(fabsf(c.z), fabsf(c.w)));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
    }
    for (; i < n4; i += stride) {
        const float4 a = x[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) m = fmaxf(m, fabsf(tail[threadIdx.x]));
    // one atomic per block (LDS across the four waves): 2,048 waves on one address would queue at the memory side
    __shared__ float sm[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) punet_amax_atomic(slot, fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])));
}
__global__ void amax_bound_kernel(const float* a, float ca, const float* b, float cb, float* out) {
    out[0] = (a ? ca * a[0] : 0.f) + (b ? cb * b[0] : 0.f);
}

#define STREAM ((hipStream_t)stream)

extern "C" int punet_amax(const float* x, long n, float* slot, int zero_first, void* stream) {
    PUNET_CHECK_ARG(x && slot && n > 0 && ((uintptr_t)x & 15) == 0, "punet_amax: bad args (x must be 16-byte aligned)");
    if (zero_first && punet_zero_fill(slot, sizeof(float), STREAM) != hipSuccess) {
        punet_set_error("punet_amax: memset failed");
        return PUNET_ERR_HIP;
    }
    const size_t n4 = (size_t)n / 4;
    amax_kernel<<<ew_grid(n4 ? (n4 + 3) / 4 : 1, 2048), EW_BLOCK, 0, STREAM>>>((const float4*)x, n4, x + 4 * n4, (int)(n - 4 * (long)n4), slot);
    PUNET_LAUNCH_CHECK("amax");
    return PUNET_OK;
}

extern "C" int punet_amax_bound(const float* a, float ca, const float* b, float cb, float* out, void* stream) {
    PUNET_CHECK_ARG(out && (a || b), "punet_amax_bound: bad args");
    amax_bound_kernel<<<1, 1, 0, STREAM>>>(a, ca, b, cb, out);
    PUNET_LAUNCH_CHECK("amax_bound");
    return PUNET_OK;
}

extern "C" int punet_pool2x2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    PUNET_CHECK_ARG(x && y && C % 4 == 0 && B > 0 && H > 0 && W > 0, "punet_pool2x2_fwd: bad args (C=%d)", C);
    const size_t n = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    PUNET_CHECK_ARG(n < (1ull << 31), "punet_pool2x2_fwd: more than 2^31 float4 elements");
    pool2x2_fwd_kernel<<<ew_grid(n), EW_BLOCK, 0, STREAM>>>((const float4*)x, (float4*)y, B, H, W, C / 4);
    PUNET_LAUNCH_CHECK("pool2x2_fwd");
    return PUNET_OK;
}

extern "C" int punet_pool2x2_bwd(const float* dy, const float* xmask, float* dx, int B, int H, int W, int C,
                                 int accumulate, void* stream) {
    return punet_pool2x2_bwd_gated(dy, xmask, nullptr, dx, B, H, W, C, accumulate, stream);
}

extern "C" int punet_pool2x2_bwd_gated(const float* dy, const float* xmask, const void* xbits, float* dx, int B, int H, int W,
                                       int C, int accumulate, void* stream) {
    PUNET_CHECK_ARG(dy && dx && C % 4 == 0 && B > 0 && H > 0 && W > 0, "punet_pool2x2_bwd: bad args (C=%d)", C);
    const size_t n = (size_t)B * H * W * (C / 4);
    PUNET_CHECK_ARG(n < (1ull << 31), "punet_pool2x2_bwd: more than 2^31 float4 elements");
    pool2x2_bwd_kernel<<<dim3(ceil_div(W * (C / 4), 256), B * H < 65535 ? B * H : 65535), 256, 0, STREAM>>>(
        (const float4*)dy, (const float4*)xmask, (const unsigned*)xbits, (float4*)dx, B, H, W, C / 4, pow2_shift(C / 4), accumulate);
    PUNET_LAUNCH_CHECK("pool2x2_bwd");
    return PUNET_OK;
}

extern "C" int punet_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    PUNET_CHECK_ARG(x && y && C % 4 == 0 && B > 0 && H > 0 && W > 0, "punet_upsample2x_fwd: bad args (C=%d)", C);
    const size_t n = (size_t)B * 4 * H * W * (C / 4);
    PUNET_CHECK_ARG(n < (1ull << 31), "punet_upsample2x_fwd: more than 2^31 float4 elements");
    upsample2x_fwd_kernel<<<dim3(ceil_div(2 * W * (C / 4), 256), B * 2 * H < 65535 ? B * 2 * H : 65535), 256, 0, STREAM>>>((const float4*)x, (float4*)y, B, H, W,
                                                                                              C / 4, pow2_shift(C / 4));
    PUNET_LAUNCH_CHECK("upsample2x_fwd");
    return PUNET_OK;
}

extern "C" int punet_upsample2x_bwd(const float* dy, const float* xmask, float* dx, int B, int H, int W, int C,
                                    void* stream) {
    return punet_upsample2x_bwd_gated(dy, xmask, nullptr, dx, B, H, W, C, stream);
}

extern "C" int punet_upsample2x_bwd_gated(const float* dy, const float* xmask, const void* xbits, float* dx, int B, int H,
                                          int W, int C, void* stream) {
    PUNET_CHECK_ARG(dy && dx && C % 4 == 0 && B > 0 && H > 0 && W > 0, "punet_upsample2x_bwd: bad args (C=%d)", C);
    const size_t n = (size_t)B * H * W * (C / 4);
    PUNET_CHECK_ARG(n < (1ull << 31), "punet_upsample2x_bwd: more than 2^31 float4 elements");
    upsample2x_bwd_kernel<<<dim3(ceil_div(W * (C / 4), 256), B * H < 65535 ? B * H : 65535), 256, 0, STREAM>>>(
        (const float4*)dy, (const float4*)xmask, (const unsigned*)xbits, (float4*)dx, B, H, W, C / 4, pow2_shift(C / 4));
    PUNET_LAUNCH_CHECK("upsample2x_bwd");
    return PUNET_OK;
}

static int chan_slices(int P, int C) {
    const int c4 = C / 4, ppl = c4 >= 256 ? 1 : 256 / c4;
    int s = (P + ppl * 16 - 1) / (ppl * 16);
    return s < 1 ? 1 : (s > 256 ? 256 : s);
}

extern "C" size_t punet_chan_sum_workspace_floats(int B, int P, int C) {
    return B > 0 && P > 0 && C >= 4 ? (size_t)B * chan_slices(P, C) * C : 0;  // (0 for a shape punet_chan_sum would refuse)
}

extern "C" int punet_chan_sum(const float* x, float* out, float* workspace, int B, int P, int C, float scale,
                              void* stream) {
    PUNET_CHECK_ARG(x && out && workspace && C % 4 == 0 && B > 0 && P > 0, "punet_chan_sum: bad args (C=%d)", C);
    const int S = chan_slices(P, C);
    chan_sum_partial_kernel<0><<<dim3(S, B), 256, 256 * sizeof(float4), STREAM>>>(x, (float4*)workspace, P, C / 4, S);
    PUNET_LAUNCH_CHECK("chan_sum_partial");
    chan_sum_finalize_kernel<<<dim3(B, (C + 63) / 64), 256, 0, STREAM>>>(workspace, out, S, C, scale);
    PUNET_LAUNCH_CHECK("chan_sum_finalize");
    return PUNET_OK;
}

template <typename T16>
static int upsample16_launch(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    PUNET_CHECK_ARG(x && y && C % 4 == 0 && B > 0 && H > 0 && W > 0, "punet_upsample2x_fwd (16-bit storage): bad args (C=%d)", C);
    upsample2x_fwd16_kernel<T16><<<dim3(ceil_div(2 * W * (C / 4), 256), B * 2 * H < 65535 ? B * 2 * H : 65535), 256, 0, STREAM>>>(
        x, y, B, H, W, C / 4, pow2_shift(C / 4));
    PUNET_LAUNCH_CHECK("upsample2x_fwd16");
    return PUNET_OK;
}
extern "C" int punet_upsample2x_fwd_bf16s(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    return upsample16_launch<__bf16>(x, y, B, H, W, C, stream);
}
extern "C" int punet_upsample2x_fwd_f16s(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    return upsample16_launch<_Float16>(x, y, B, H, W, C, stream);
}

extern "C" int punet_latent_head_fwd(const float* enc, const float* w, const float* bias, float* mean, float* muls,
                                     float* workspace, int B, int P, int C, int J, void* stream) {
    PUNET_CHECK_ARG(enc && w && bias && mean && muls && workspace, "punet_latent_head_fwd: null pointer");
    int rc = punet_chan_sum(enc, mean, workspace, B, P, C, 1.f / (float)P, stream);
    if (rc != PUNET_OK) return rc;
    latent_head_kernel<<<B, 256, 0, STREAM>>>(mean, w, bias, muls, C, J);
    PUNET_LAUNCH_CHECK("latent_head");
    return PUNET_OK;
}

template <int IN16>
static int latent_head16_launch(const void* enc, const float* w, const float* bias, float* mean, float* muls, float* workspace, int B,
                                int P, int C, int J, void* stream) {
    PUNET_CHECK_ARG(enc && w && bias && mean && muls && workspace && C % 4 == 0, "punet_latent_head_fwd (16-bit storage): bad args");
    const int S = chan_slices(P, C);
    chan_sum_partial_kernel<IN16><<<dim3(S, B), 256, 256 * sizeof(float4), STREAM>>>(enc, (float4*)workspace, P, C / 4, S);
    PUNET_LAUNCH_CHECK("chan_sum_partial16");
    chan_sum_finalize_kernel<<<dim3(B, (C + 63) / 64), 256, 0, STREAM>>>(workspace, mean, S, C, 1.f / (float)P);
    PUNET_LAUNCH_CHECK("chan_sum_finalize");
    latent_head_kernel<<<B, 256, 0, STREAM>>>(mean, w, bias, muls, C, J);
    PUNET_LAUNCH_CHECK("latent_head");
    return PUNET_OK;
}
extern "C" int punet_latent_head_fwd_bf16s(const void* enc, const float* w, const float* bias, float* mean, float* muls,
                                           float* workspace, int B, int P, int C, int J, void* stream) {
    return latent_head16_launch<1>(enc, w, bias, mean, muls, workspace, B, P, C, J, stream);
}
extern "C" int punet_latent_head_fwd_f16s(const void* enc, const float* w, const float* bias, float* mean, float* muls,
                                          float* workspace, int B, int P, int C, int J, void* stream) {
    return latent_head16_launch<2>(enc, w, bias, mean, muls, workspace, B, P, C, J, stream);
}

extern "C" int punet_reparam(const float* muls, const float* eps, float* z, int N, int B, int L, void* stream) {
    PUNET_CHECK_ARG(muls && eps && z && N > 0 && B > 0 && L > 0, "punet_reparam: bad args");
    reparam_kernel<<<ew_grid((size_t)N * B * L), EW_BLOCK, 0, STREAM>>>(muls, eps, z, N, B, L);
    PUNET_LAUNCH_CHECK("reparam");
    return PUNET_OK;
}

extern "C" int punet_latent_kl(const float* muls_q, const float* muls_p, float* kl, int B, int L, void* stream) {
    PUNET_CHECK_ARG(muls_q && muls_p && kl, "punet_latent_kl: null pointer");
    latent_kl_kernel<<<1, 64, 0, STREAM>>>(muls_q, muls_p, kl, B, L);
    PUNET_LAUNCH_CHECK("latent_kl");
    return PUNET_OK;
}

extern "C" int punet_latent_kl_bwd(const float* muls_q, const float* muls_p, const float* eps, const float* dz, float coef,
                                   float* d_muls_q, float* d_muls_p, int B, int L, void* stream) {
    PUNET_CHECK_ARG(muls_q && muls_p && eps && dz && d_muls_q && d_muls_p, "punet_latent_kl_bwd: null pointer");
    latent_kl_bwd_kernel<<<1, 256, 0, STREAM>>>(muls_q, muls_p, eps, dz, coef, d_muls_q, d_muls_p, B, L);
    PUNET_LAUNCH_CHECK("latent_kl_bwd");
    return PUNET_OK;
}

extern "C" int punet_latent_kl_rows(const float* muls_q, const float* muls_p, const float* z, float* kl_rows, float* kl_mean,
                                    int B, int L, void* stream) {
    PUNET_CHECK_ARG(muls_q && muls_p && kl_mean && B > 0 && L > 0, "punet_latent_kl_rows: bad args");
    latent_kl_rows_kernel<<<1, 256, 0, STREAM>>>(muls_q, muls_p, z, kl_rows, kl_mean, B, L);
    PUNET_LAUNCH_CHECK("latent_kl_rows");
    return PUNET_OK;
}

extern "C" int punet_latent_kl_bwd_ex(const float* muls_q, const float* muls_p, const float* eps, const float* dz, float coef,
                                      int kl_sampled, int z_is_mean, float* d_muls_q, float* d_muls_p, int B, int L,
                                      void* stream) {
    PUNET_CHECK_ARG(muls_q && muls_p && eps && dz && d_muls_q && d_muls_p, "punet_latent_kl_bwd_ex: null pointer");
    latent_kl_bwd_ex_kernel<<<1, 256, 0, STREAM>>>(muls_q, muls_p, eps, dz, coef, kl_sampled, z_is_mean, d_muls_q, d_muls_p, B, L);
    PUNET_LAUNCH_CHECK("latent_kl_bwd_ex");
    return PUNET_OK;
}

extern "C" int punet_latent_split(const float* muls, float* loc, float* scale, int B, int L, void* stream) {
    PUNET_CHECK_ARG(muls && loc && scale && B > 0 && L > 0, "punet_latent_split: bad args");
    latent_split_kernel<<<ew_grid((size_t)B * L), EW_BLOCK, 0, STREAM>>>(muls, loc, scale, B, L);
    PUNET_LAUNCH_CHECK("latent_split");
    return PUNET_OK;
}

extern "C" int punet_concat_nhwc(const float* a, int ca, const float* b, int cb, float* out, long P, void* stream) {
    PUNET_CHECK_ARG(a && b && out && ca > 0 && cb > 0 && P > 0, "punet_concat_nhwc: bad args");
    concat_nhwc_kernel<<<ew_grid((size_t)P * (ca + cb), 8192), EW_BLOCK, 0, STREAM>>>(a, ca, b, cb, out, (size_t)P);
    PUNET_LAUNCH_CHECK("concat_nhwc");
    return PUNET_OK;
}

extern "C" int punet_nchw_to_nhwc(const float* x, float* y, int B, int C, long HW, void* stream) {
    PUNET_CHECK_ARG(x && y && x != y && B > 0 && C > 0 && HW > 0, "punet_nchw_to_nhwc: bad args");
    nchw_to_nhwc_kernel<<<ew_grid((size_t)B * C * HW, 8192), EW_BLOCK, 0, STREAM>>>(x, y, B, C, (size_t)HW);
    PUNET_LAUNCH_CHECK("nchw_to_nhwc");
    return PUNET_OK;
}

extern "C" int punet_nhwc_to_nchw(const float* x, float* y, int B, int C, long HW, void* stream) {
    PUNET_CHECK_ARG(x && y && x != y && B > 0 && C > 0 && HW > 0, "punet_nhwc_to_nchw: bad args");
    nhwc_to_nchw_kernel<<<ew_grid((size_t)B * C * HW, 8192), EW_BLOCK, 0, STREAM>>>(x, y, B, C, (size_t)HW);
    PUNET_LAUNCH_CHECK("nhwc_to_nchw");
    return PUNET_OK;
}

extern "C" int punet_align_distribution(const float* y, const float* source_distribution, float* out, float* ratio,
                                        unsigned long long* count_ws, long P, void* stream) {
    PUNET_CHECK_ARG(y && source_distribution && out && count_ws && P > 0, "punet_align_distribution: bad args");
    if (punet_zero_fill(count_ws, sizeof(unsigned long long), STREAM) != hipSuccess) {
        punet_set_error("punet_align_distribution: hipMemsetAsync failed");
        return PUNET_ERR_HIP;
    }
    count_ge_kernel<<<ew_grid((size_t)P, 1024), 256, 0, STREAM>>>(y, (size_t)P, 0.5f, count_ws);
    PUNET_LAUNCH_CHECK("count_ge");
    align_distribution_kernel<<<ew_grid((size_t)P), EW_BLOCK, 0, STREAM>>>(y, source_distribution, (size_t)P, count_ws, out, ratio);
    PUNET_LAUNCH_CHECK("align_distribution");
    return PUNET_OK;
}

extern "C" int punet_latent_head_bwd(const float* d_muls, const float* mean, const float* w, const float* enc, float* dw,
                                     float* db, float* dmean, float* denc, int B, int P, int C, int J, int accumulate,
                                     void* stream) {
    PUNET_CHECK_ARG(d_muls && mean && w && enc && dw && db && dmean && denc && C % 4 == 0,
                    "punet_latent_head_bwd: bad args");
    latent_head_bwd_kernel<<<16, 256, 0, STREAM>>>(d_muls, mean, w, dw, db, dmean, B, C, J, accumulate);
    PUNET_LAUNCH_CHECK("latent_head_bwd");
    const size_t n = (size_t)B * P * (C / 4);
    bcast_mean_bwd_kernel<<<ew_grid(n), EW_BLOCK, 0, STREAM>>>((const float4*)dmean, (const float4*)enc, (float4*)denc, B, P,
                                                               C / 4, 1.f / (float)P);
    PUNET_LAUNCH_CHECK("bcast_mean_bwd");
    return PUNET_OK;
}

template <typename T16>
static int gate_bits16_launch(const void* y, void* bits, int B, int H, int W, int C, void* stream) {
    PUNET_CHECK_ARG(y && bits && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "punet_gate_bits (16-bit storage): bad args");
    const size_t n = (size_t)B * H * ((W + 31) / 32) * ((C + 63) / 64) * 64;
    gate_bits16_kernel<T16><<<ew_grid(n), EW_BLOCK, 0, STREAM>>>((const T16*)y, (unsigned*)bits, B, H, W, C);
    PUNET_LAUNCH_CHECK("gate_bits16");
    return PUNET_OK;
}
extern "C" int punet_gate_bits_bf16(const void* y, void* bits, int B, int H, int W, int C, void* stream) {
    return gate_bits16_launch<__bf16>(y, bits, B, H, W, C, stream);
}
extern "C" int punet_gate_bits_f16(const void* y, void* bits, int B, int H, int W, int C, void* stream) {
    return gate_bits16_launch<_Float16>(y, bits, B, H, W, C, stream);
}

extern "C" int punet_latent_head_bwd_bits(const float* d_muls, const float* mean, const float* w, const void* enc_bits, float* dw,
                                          float* db, float* dmean, float* denc, int B, int H, int W, int C, int J, int accumulate,
                                          void* stream) {
    PUNET_CHECK_ARG(d_muls && mean && w && enc_bits && dw && db && dmean && denc && C % 4 == 0 && B > 0 && H > 0 && W > 0,
                    "punet_latent_head_bwd_bits: bad args");
    latent_head_bwd_kernel<<<16, 256, 0, STREAM>>>(d_muls, mean, w, dw, db, dmean, B, C, J, accumulate);
    PUNET_LAUNCH_CHECK("latent_head_bwd");
    const size_t n = (size_t)B * H * W * (C / 4);
    bcast_mean_bwd_bits_kernel<<<ew_grid(n), EW_BLOCK, 0, STREAM>>>((const float4*)dmean, (const unsigned*)enc_bits, (float4*)denc, B, H,
                                                                   W, C / 4, 1.f / (float)((size_t)H * W));
    PUNET_LAUNCH_CHECK("bcast_mean_bwd_bits");
    return PUNET_OK;
}

extern "C" int punet_fcomb_zbias(const float* z, const float* w1, const float* b1, float* zb, int NB, int L, int CH,
                                 void* stream) {
    PUNET_CHECK_ARG(z && w1 && b1 && zb && NB > 0, "punet_fcomb_zbias: bad args");
    fcomb_zbias_kernel<<<ew_grid((size_t)NB * CH), EW_BLOCK, 0, STREAM>>>(z, w1, b1, zb, NB, L, CH);
    PUNET_LAUNCH_CHECK("fcomb_zbias");
    return PUNET_OK;
}

extern "C" int punet_fcomb_z_bwd(const float* S, const float* z, const float* w1, float* dw1, float* dz, int B, int L, int CH,
                                 int accumulate, void* stream) {
    PUNET_CHECK_ARG(S && z && w1 && dw1 && dz && B > 0 && L > 0 && CH > 0, "punet_fcomb_z_bwd: bad args");
    fcomb_z_bwd_kernel<<<4, 256, 0, STREAM>>>(S, z, w1, dw1, dz, B, L, CH, accumulate);
    PUNET_LAUNCH_CHECK("fcomb_z_bwd");
    return PUNET_OK;
}

static bool dot_c_ok(int C) { return C >= 4 && C % 4 == 0 && 64 % (C / 4) == 0; }

extern "C" int punet_dot_bias(const float* h, const float* w3, const float* b3, float* logits, long P, int C,
                              void* stream) {
    PUNET_CHECK_ARG(h && w3 && b3 && logits && dot_c_ok(C), "punet_dot_bias: C/4 must divide 64 (C=%d)", C);
    const int ppb = 256 / (C / 4);
    dot_bias_kernel<<<ew_grid(((size_t)P + ppb - 1) / ppb * 256, 2048), 256, 0, STREAM>>>((const float4*)h, (const float4*)w3, b3,
                                                                                         logits, (size_t)P, C / 4);
    PUNET_LAUNCH_CHECK("dot_bias");
    return PUNET_OK;
}

constexpr int DOT_BWD_BLOCKS = 1024;
extern "C" size_t punet_dot_bwd_workspace_floats(int C) { return (size_t)DOT_BWD_BLOCKS * (C + 1); }

extern "C" int punet_dot_bwd(const float* h, const float* dlogit, const float* w3, float* dz, float* dw3, float* db3,
                             float* workspace, long P, int C, int accumulate, void* stream) {
    PUNET_CHECK_ARG(h && dlogit && w3 && dz && dw3 && db3 && workspace && dot_c_ok(C),
                    "punet_dot_bwd: C/4 must divide 64 (C=%d)", C);
    const int ppb = 256 / (C / 4);
    int grid = (int)(((size_t)P + ppb - 1) / ppb);
    if (grid > DOT_BWD_BLOCKS) grid = DOT_BWD_BLOCKS;
    dot_bwd_kernel<<<grid, 256, 0, STREAM>>>((const float4*)h, dlogit, (const float4*)w3, (float4*)dz, workspace, (size_t)P,
                                             C / 4);
    PUNET_LAUNCH_CHECK("dot_bwd");
    int rc = punet_launch_reduce_slabs(workspace, grid, (size_t)(C + 1), 1, C, 1, 1, 0, dw3, accumulate, 1.f, STREAM);
    if (rc != PUNET_OK) return rc;
    return punet_launch_reduce_slabs(workspace + C, grid, (size_t)(C + 1), 1, 1, 1, 1, 0, db3, accumulate, 1.f, STREAM);
}

constexpr int LOSS_BLOCKS = 1024;
extern "C" size_t punet_loss_workspace_floats(void) { return (size_t)3 * LOSS_BLOCKS + 8; }

extern "C" int punet_bce_fwd_bwd(const float* logits, const float* target, const float* consm, float gscale, float* dlogit,
                                 float* loss_sum, float* workspace, long P, void* stream) {
    PUNET_CHECK_ARG(logits && target && loss_sum && workspace && P > 0, "punet_bce_fwd_bwd: bad args");
    const int grid = ew_grid((size_t)P, LOSS_BLOCKS);
    bce_fwd_bwd_kernel<<<grid, 256, 0, STREAM>>>(logits, target, consm, gscale, dlogit, workspace, (size_t)P);
    PUNET_LAUNCH_CHECK("bce_fwd_bwd");
    sum_partials_kernel<<<1, 256, 0, STREAM>>>(workspace, grid, 1, loss_sum);
    PUNET_LAUNCH_CHECK("sum_partials");
    return PUNET_OK;
}

extern "C" int punet_dice_fwd_bwd(const float* logits, const float* target, const float* consm, float gscale, float* dlogit,
                                  float* loss_out, float* workspace, long P, void* stream) {
    PUNET_CHECK_ARG(logits && target && dlogit && loss_out && workspace && P > 0, "punet_dice_fwd_bwd: bad args");
    const int grid = ew_grid((size_t)P, LOSS_BLOCKS);
    dice_sums_kernel<<<grid, 256, 0, STREAM>>>(logits, target, consm, workspace, (size_t)P);
    PUNET_LAUNCH_CHECK("dice_sums");
    float* sums = workspace + 3 * LOSS_BLOCKS;
    sum_partials_kernel<<<1, 256, 0, STREAM>>>(workspace, grid, 3, sums);
    PUNET_LAUNCH_CHECK("sum_partials");
    dice_bwd_kernel<<<ew_grid((size_t)P), 256, 0, STREAM>>>(logits, target, consm, sums, gscale, dlogit, loss_out, (size_t)P);
    PUNET_LAUNCH_CHECK("dice_bwd");
    return PUNET_OK;
}

extern "C" int punet_consensus(const float* logits, int N, long P, float upper, float lower, int masking, float* y, float* z,
                               void* stream) {
    PUNET_CHECK_ARG(logits && y && z && N > 0 && P > 0, "punet_consensus: bad args");
    consensus_kernel<<<ew_grid((size_t)P), EW_BLOCK, 0, STREAM>>>(logits, N, (size_t)P, upper, lower, masking, y, z);
    PUNET_LAUNCH_CHECK("consensus");
    return PUNET_OK;
}

extern "C" int punet_dice_score_sums(const float* seg, const float* gt, float thr_seg, int use_thr_seg, float thr_gt,
                                     int use_thr_gt, float* sums3, float* workspace, long P, void* stream) {
    PUNET_CHECK_ARG(seg && gt && sums3 && workspace && P > 0, "punet_dice_score_sums: bad args");
    const int grid = ew_grid((size_t)P, LOSS_BLOCKS);
    dice_score_sums_kernel<<<grid, 256, 0, STREAM>>>(seg, gt, thr_seg, use_thr_seg, thr_gt, use_thr_gt, workspace, (size_t)P);
    PUNET_LAUNCH_CHECK("dice_score_sums");
    sum_partials_kernel<<<1, 256, 0, STREAM>>>(workspace, grid, 3, sums3);
    PUNET_LAUNCH_CHECK("sum_partials");
    return PUNET_OK;
}

extern "C" int punet_l2_reg(const float* w, const int64_t* chunks, int nchunks, int ntensors, float* partial, float* norms,
                            float* reg_out, void* stream) {
    PUNET_CHECK_ARG(w && chunks && partial && norms && reg_out && nchunks > 0 && ntensors > 0, "punet_l2_reg: bad args");
    l2_sumsq_kernel<<<nchunks, 256, 0, STREAM>>>(w, chunks, partial);
    PUNET_LAUNCH_CHECK("l2_sumsq");
    l2_finalize_kernel<<<1, 256, 0, STREAM>>>(partial, chunks, nchunks, ntensors, norms, reg_out);
    PUNET_LAUNCH_CHECK("l2_finalize");
    return PUNET_OK;
}

extern "C" int punet_l2_reg_add_grad(const float* w, float* g, const int64_t* chunks, int nchunks, const float* norms,
                                     float coef, void* stream) {
    PUNET_CHECK_ARG(w && g && chunks && norms && nchunks > 0, "punet_l2_reg_add_grad: bad args");
    l2_add_grad_kernel<<<nchunks, 256, 0, STREAM>>>(w, g, chunks, norms, coef);
    PUNET_LAUNCH_CHECK("l2_add_grad");
    return PUNET_OK;
}

extern "C" int punet_adam_ema(float* p, const float* g, float* m, float* v, float* teacher, long n, float gscale, float lr,
                              float beta1, float beta2, float eps, int step, float ema_momentum, int do_adam,
                              void* stream) {
    PUNET_CHECK_ARG(p && n > 0 && (!do_adam || (g && m && v && step >= 1)), "punet_adam_ema: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = do_adam ? (float)((double)lr / bc1) : 0.f;
    const float bc2_sqrt = do_adam ? (float)sqrt(bc2) : 1.f;
    adam_ema_kernel<<<ew_grid((size_t)n, 8192), EW_BLOCK, 0, STREAM>>>(p, g, m, v, teacher, (size_t)n, gscale, step_size, beta1,
                                                                      beta2, bc2_sqrt, eps, ema_momentum, do_adam);
    PUNET_LAUNCH_CHECK("adam_ema");
    return PUNET_OK;
}

extern "C" int punet_adam_hyper(double* hstate, float* hyper, void* stream) {
    PUNET_CHECK_ARG(hstate && hyper, "punet_adam_hyper: null pointer");
    adam_hyper_kernel<<<1, 64, 0, STREAM>>>(hstate, hyper);
    PUNET_LAUNCH_CHECK("adam_hyper");
    return PUNET_OK;
}

extern "C" int punet_adam_ema_dev(float* p, const float* g, float* m, float* v, float* teacher, long n, const float* hyper,
                                  float beta1, float beta2, float eps, void* stream) {
    PUNET_CHECK_ARG(p && g && m && v && hyper && n > 0, "punet_adam_ema_dev: bad args");
    adam_ema_dev_kernel<<<ew_grid((size_t)n, 8192), EW_BLOCK, 0, STREAM>>>(p, g, m, v, teacher, (size_t)n, hyper, beta1, beta2, eps);
    PUNET_LAUNCH_CHECK("adam_ema_dev");
    return PUNET_OK;
}

extern "C" int punet_scale(float* x, long n, float s, void* stream) {
    PUNET_CHECK_ARG(x && n > 0, "punet_scale: bad args");
    scale_kernel<<<ew_grid((size_t)n, 8192), EW_BLOCK, 0, STREAM>>>(x, (size_t)n, s);
    PUNET_LAUNCH_CHECK("scale");
    return PUNET_OK;
}
