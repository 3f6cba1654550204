// K5x: the first convolution of an UpConvBlock with its `up` source contracted at LOW resolution.
//
// The reference computes  y = ReLU(conv3x3(cat[up(x), bridge]) + b)  with up = bilinear x2, align_corners=True
// (unet_blocks.py:51-57).  Interpolation acts per channel, the convolution's channel contraction per pixel, so they commute:
//     W_t . up(x)[q] = up(W_t . x)[q]      for every tap t and every high-resolution position q,
// hence, with zero padding at the HIGH resolution as in the reference,
//     conv3x3(up(x))[p] = sum_t [q = p + t - 1 inside the image] up(z_t)[q],      z_t = W_t . x .
// The nine z_t are ONE 1x1 convolution of the low-resolution tensor with 9 cout output channels (a quarter of the multiply-adds
// of the `up` half of the layer, which is two thirds of it), and the up-sampled tensor never exists.  What is left at the high
// resolution is a gather: 9 taps x 4 bilinear neighbours of [cout]-vectors per output pixel, added to the bridge half's
// convolution (the ordinary 3x3 kernel on the bridge alone), then bias (already in the bridge half), ReLU and the gate bits.
// The same algebra turns the input gradient w.r.t. x into nine ADJOINT-interpolated maps of dz followed by one 1x1 input gradient
//     dz_t[j] = sum_q up_weight(q, j) . dz[q - t + 1],        dx = gate(x) . sum_t W_t^T . dz_t ,
// and the weight gradient of the `up` rows into  dW_t = x^T . dz_t  (contracted over the low-resolution pixels): with all three in
// this form the up-sampled tensor is needed by nothing, forward or backward.
// Interpolation weights are K5's (punet_up_src in common.h: ATen's rounding of the source index).
//
// Entry points (include/punet_hip.h): punet_upconv_relayout (the weight slab's `up` rows as the [cup][9 cout] matrix of that 1x1
// convolution, its bridge rows as a contiguous 3x3 slab), punet_upconv_gather_fwd, punet_upconv_gather_bwd, punet_upconv_wgrad_x6
// (+ punet_upconv_wgrad_slices / _workspace_floats).  The 1x1 convolutions themselves are punet_conv1x1_x6_gated (conv_x6.hip).
#include "common.h"
#include "punet_hip.h"

namespace {

#define STREAM ((hipStream_t)stream)
#ifndef PUNET_UPCONV_NX
#define PUNET_UPCONV_NX 4  // output columns per thread of the forward gather (the kernel is written for 4)
#endif

// w[t][ci][co] (ci < cup + cb) -> w9[ci][t * cout + co] for ci < cup ; wb[t][ci - cup][co] for the others (float4 units)
__global__ __launch_bounds__(256) void upconv_relayout_kernel(const float4* __restrict__ w, float4* __restrict__ w9, float4* __restrict__ wb,
                                                              int cup, int cb, int co4) {
    const int ctot = cup + cb;
    const size_t n = (size_t)9 * ctot * co4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % co4);
        const int ci = (int)((i / co4) % ctot);
        const int t = (int)(i / ((size_t)co4 * ctot));
        const float4 v = w[i];
        if (ci < cup)
            w9[((size_t)ci * 9 + t) * co4 + c] = v;
        else
            wb[((size_t)t * cb + (ci - cup)) * co4 + c] = v;
    }
}

// y[b][Y][X][c] = act(y[b][Y][X][c] + sum_{ky, kx} up(z_{ky,kx})[Y + ky - 1][X + kx - 1][c]),  z: [B][h][w][9][C], y: [B][2h][2w][C]
// A thread owns NX consecutive output columns of one channel quad: per tap their 2 NX bilinear column neighbours fall into
// NX / 2 + 2 consecutive low-resolution columns (the x2 scale is just below 1/2), so a tap costs 2 (NX / 2 + 2) loads for NX outputs
// instead of 4 NX - the first version (one output per thread, 36 loads each) ran at the vector L1's rate, not at HBM's.  The two
// source rows are interpolated first (per loaded column), then each output picks its two columns by select.
// grid = (ceil(ceil(2w / NX) * C4 / 256), B * 2h rows): the three source-row pairs and their weights are block-uniform.
// BITS: publish (y > 0) in the punet_gate_dwords layout (C a multiple of 64: the 16 float4 of a pixel's 64-channel block sit in 16
// consecutive lanes; a dword's nibbles are joined by an OR over the lanes that differ in bits 1..3).
template <bool BITS, int NX>
__global__ __launch_bounds__(256, 3) void upconv_gather_fwd_kernel(const float4* __restrict__ z, float4* __restrict__ y,
                                                                unsigned* __restrict__ bits, int B, int h, int w, int C4, int c4_shift,
                                                                int relu) {
    static_assert(NX == 4, "four outputs per thread: their neighbours fall into NX / 2 + 2 = 4 consecutive low-resolution columns");
    const int H = 2 * h, W = 2 * w;
    const int WG = (W + NX - 1) / NX;  // column groups per row
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)(WG * C4)) return;  // (C4 is a multiple of 16 when BITS: the lane groups of the reduction leave together)
    const int xg = c4_shift >= 0 ? (int)(e >> c4_shift) : (int)(e / (unsigned)C4);
    const int c4 = (int)e - xg * C4;
    const int X0 = xg * NX;
    // per tap column kx: the first low-resolution column of the run, and per output its interpolation as WEIGHTS on the run's columns
    // (at most two are non-zero; zero padding and a missing second neighbour are weights of zero).  Output i can only reach columns
    // 0 .. NCOL[i] - 1 of the run: the scale is below 1/2, so its first neighbour lies at most (i + 1) / 2 columns behind output 0's.
    // Weights instead of a select of the two columns: the kernel is bound by its vector instructions (the select form needed 590
    // per output against ~210 here), not by its loads.
    constexpr int NCOL[4] = {2, 3, 3, 4};
    int cmin[3];
    float wgt[3][4][4];  // [kx][output][column]: only the first NCOL[output] entries are used (the rest is never read)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int qx = X0 + i + kx - 1;
            const bool ok = qx >= 0 && qx < W && X0 + i < W;
            int c0, cp;
            float l1;
            punet_up_src(min(max(qx, 0), W - 1), sw, w, c0, cp, l1);
            if (i == 0) cmin[kx] = c0;
            const int d = c0 - cmin[kx];  // monotone in qx
            const float w0 = ok ? (cp ? 1.f - l1 : 1.f) : 0.f, w1 = (ok && cp) ? l1 : 0.f;
#pragma unroll
            for (int j = 0; j < NCOL[i]; ++j) wgt[kx][i][j] = j == d ? w0 : (j == d + 1 ? w1 : 0.f);
        }
    }
    const unsigned pix9 = 9u * (unsigned)C4;  // float4 units per low-resolution pixel of z
    // ONE output row per block (grid.y x grid.z), no loop over rows: a grid-stride row loop made every per-thread address part a
    // loop invariant that the optimiser hoisted in front of it - ~70 64-bit values, 250 registers, spills
    {
        const unsigned row = blockIdx.z * gridDim.y + blockIdx.y;
        if (row >= (unsigned)(B * H)) return;
        const unsigned b = row / (unsigned)H;
        const int Y = (int)(row - b * (unsigned)H);
        // the accumulators start from the bridge half's convolution: its four loads are the first in flight, not the last (issued
        // behind the gather they stood exposed in front of the stores - the instantiation without gate bits ran 10-15 % slower)
        float4 a[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const unsigned xo = (unsigned)((min(X0 + i, W - 1) * C4 + c4) * 16);
            a[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(y + (size_t)row * W * C4) + xo);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int qy = Y + ky - 1;
            if (qy < 0 || qy >= H) continue;  // block-uniform
            int r0, rp;
            float hl1;
            punet_up_src(qy, sh, h, r0, rp, hl1);
            const float hl0 = 1.f - hl1;
            // block-uniform 64-bit row bases + 32-bit per-thread offsets inside a low-resolution row
            const float4* z0 = z + ((size_t)b * h + r0) * w * pix9 + (size_t)(3 * ky) * C4;
            const float4* z1 = z0 + (size_t)rp * w * pix9;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // both rows of the NC = 4 columns first, then their interpolation.  Four NAMED values, not an array: a select chain
                // over array elements is turned into a dynamically indexed load by the optimiser - the array then lives in scratch
                // memory (measured: 112 bytes of scratch per lane, 256 registers, 9x slower)
                auto col = [&](int j) __attribute__((always_inline)) -> float4 {
                    // 32-bit BYTE offset from a block-uniform base: the scalar-base + vector-offset addressing mode (a 64-bit vector
                    // address per load cost two more vector instructions each, in a kernel that is bound by them)
                    const unsigned ob = ((unsigned)min(cmin[kx] + j, w - 1) * pix9 + (unsigned)(kx * C4 + c4)) * 16u;
                    const float4 t0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(z0) + ob);
                    const float4 t1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(z1) + ob);
                    return make_float4(hl0 * t0.x + hl1 * t1.x, hl0 * t0.y + hl1 * t1.y, hl0 * t0.z + hl1 * t1.z, hl0 * t0.w + hl1 * t1.w);
                };
                const float4 v0 = col(0), v1 = col(1), v2 = col(2), v3 = col(3);
                const float4 vv[4] = {v0, v1, v2, v3};  // (indexed by compile-time constants only)
#pragma unroll
                for (int i = 0; i < NX; ++i) {
#pragma unroll
                    for (int j = 0; j < NCOL[i]; ++j) {
                        const float g = wgt[kx][i][j];
                        a[i].x += g * vv[j].x; a[i].y += g * vv[j].y; a[i].z += g * vv[j].z; a[i].w += g * vv[j].w;
                    }
                }
                // one tap's loads in flight per wave, not all nine (the scheduler otherwise hoists every load of the row to the top:
                // 256 registers, one wave per SIMD - latency is hidden by the resident waves instead)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int X = X0 + i;
            if (X >= W) continue;  // (W is even; with NX = 4 the last group of a row may be half empty)
            const size_t idx4 = ((size_t)row * W + X) * C4 + c4;
            float4 o = a[i];
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            y[idx4] = o;
            if constexpr (BITS) {
                const int cl = (4 * c4) & 63;  // first channel of this float4 inside its 64-channel block
                unsigned g = ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u))
                             << (16 * (cl >> 5) + 4 * ((cl >> 3) & 3));
                g |= __shfl_xor(g, 2, 64);
                g |= __shfl_xor(g, 4, 64);
                g |= __shfl_xor(g, 8, 64);
                if ((c4 & 14) == 0) {  // lanes c4 % 16 == 0 / 1 write the dwords of half 0 / 1
                    const int tiles = (W + 31) >> 5, cbs = (4 * C4 + 63) >> 6;
                    const size_t idx = (((((size_t)b * H + Y) * tiles + (X >> 5)) * cbs + (c4 >> 4)) << 6) + (X & 31) + 32 * (c4 & 1);
                    bits[idx] = g;
                }
            }
        }
    }
}

// dzt[b][jy][jx][t][c] = sum_{oy, ox} wy(oy, jy) wx(ox, jx) dz[b][oy - ky + 1][ox - kx + 1][c]   (t = 3 ky + kx; positions outside
// the image contribute nothing: the forward's zero padding)  - gather form, deterministic.
// A thread owns ALL NINE taps of one low-resolution pixel and channel quad: their windows into dz overlap almost completely (rows
// oy0 - 1 .. oy1 + 1, columns ox0 - 1 .. ox0 + 7), so the window is loaded once - 9 columns per row for 9 outputs - instead of once
// per tap (the first version: one tap per thread, ~25 loads per output, ran at a third of the forward's rate).
// grid = (ceil(w * C4 / 256), B * h rows): the (up to 7) row weights are block-uniform (as in K5's backward).
__global__ __launch_bounds__(256) void upconv_gather_bwd_kernel(const float4* __restrict__ dz, float4* __restrict__ dzt, int B, int h,
                                                                int w, int C4, int c4_shift) {
    const int H = 2 * h, W = 2 * w;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)(w * C4)) return;
    const int jx = c4_shift >= 0 ? (int)(e >> c4_shift) : (int)(e / (unsigned)C4);
    const int c4 = (int)e - jx * C4;
    const int ox0 = max(0, 2 * jx - 2), ox1 = min(W - 1, 2 * jx + 4);
    float wx[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wx[k] = ox0 + k <= ox1 ? punet_up_weight(ox0 + k, jx, sw, w) : 0.f;
    // the nine window columns ox0 - 1 .. ox0 + 7: clamped address, zero outside the image
    int pcol[9];
    bool pin[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        const int p = ox0 - 1 + m;
        pin[m] = p >= 0 && p < W;
        pcol[m] = min(max(p, 0), W - 1) * C4;
    }
    for (unsigned row = blockIdx.y; row < (unsigned)(B * h); row += gridDim.y) {
        const unsigned b = row / (unsigned)h;
        const int jy = (int)(row - b * (unsigned)h);
        const int oy0 = max(0, 2 * jy - 2), oy1 = min(H - 1, 2 * jy + 4);
        float wy[7];  // block-uniform
#pragma unroll
        for (int k = 0; k < 7; ++k) wy[k] = oy0 + k <= oy1 ? punet_up_weight(oy0 + k, jy, sh, h) : 0.f;
        float4 acc[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t / 3][t % 3] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pi = 0; pi < 9; ++pi) {  // window row p = oy0 - 1 + pi of dz; tap row ky reads it for oy = p + ky - 1 = oy0 + pi + ky - 2
            const int p = oy0 - 1 + pi;
            float wk[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) wk[ky] = (pi + ky - 2 >= 0 && pi + ky - 2 < 7) ? wy[pi + ky - 2] : 0.f;
            if ((wk[0] == 0.f && wk[1] == 0.f && wk[2] == 0.f) || p < 0 || p >= H) continue;  // block-uniform
            const float4* drow = dz + (((size_t)b * H + p) * W) * C4 + c4;
            float4 g[9];
#pragma unroll
            for (int m = 0; m < 9; ++m) g[m] = drow[pcol[m]];
#pragma unroll
            for (int m = 0; m < 9; ++m)
                if (!pin[m]) g[m] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {  // tap column kx reads dz column ox - kx + 1 = window index k - kx + 2 for ox = ox0 + k
                float4 hsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float4 v = g[k - kx + 2];
                    hsum.x += wx[k] * v.x; hsum.y += wx[k] * v.y; hsum.z += wx[k] * v.z; hsum.w += wx[k] * v.w;
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    acc[ky][kx].x += wk[ky] * hsum.x; acc[ky][kx].y += wk[ky] * hsum.y;
                    acc[ky][kx].z += wk[ky] * hsum.z; acc[ky][kx].w += wk[ky] * hsum.w;
                }
            }
        }
        float4* out = dzt + ((size_t)row * w + jx) * 9 * C4 + c4;
#pragma unroll
        for (int t = 0; t < 9; ++t) out[(size_t)t * C4] = acc[t / 3][t % 3];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The weight gradient of the `up` rows in the low-resolution form:   dW[t][ci][co] = sum_j x[j][ci] * dzt[j][t][co]
// (j over ALL low-resolution pixels of the batch; dzt = the nine adjoint-interpolated maps of upconv_gather_bwd_kernel) - a
// quarter of the multiply-adds of the 3x3 weight gradient on the up-sampled tensor, which then no longer has to exist at all.
// Structure of wgrad3x3_x6_kernel (wgrad_x6.hip): six-product bf16 arithmetic, [pixel][32 ch] LDS images read with the
// transposing ds_read_b64_tr_b16, deterministic split-K (slabs summed in fixed order by reduce_slabs) - with the operands' roles
// swapped: there ONE dz image serves nine shifted views of x, here ONE x image serves the nine channel blocks of dzt.  A
// workgroup owns a 64 x 64 (ci, co) tile, THREE taps (their dzt images are 72 KB of LDS; all nine would be 216 KB) and a slice of
// the pixels; its 8 waves = 2 co blocks x 4 quarters of the 64-pixel tile (the quarters are four split-K slabs), each wave both ci blocks.
// Single-buffered (96 KB): the next tile's global loads are in flight during the MFMAs and split in their shadow.
typedef __bf16 up_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 up_bf16x4 __attribute__((ext_vector_type(4)));
typedef short up_s16x4 __attribute__((ext_vector_type(4)));
union UpFrag {
    up_bf16x8 v;
    up_s16x4 half4[2];
};
__device__ __forceinline__ up_s16x4 up_tr_read(const char* lds_base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) up_s16x4*)(lds_base + byte_off));
}
constexpr int UWC = 64, UNPD = 64, UROWB = 64, UIMG = UNPD * UROWB, UTAPS = 3;  // 64 x 64 channel tile, 64 pixels per tile, 4 KB images

struct UpWgradArgs {
    const float* x;    // [M][cs]      low-resolution source, M = B h w pixels
    const float* dzt;  // [M][9][cout]
    float* partial;    // [4 nslices][9][cs][cout]
    int cs, cout, nslices;
    long M;
};

__global__ __launch_bounds__(512, 2) void upconv_wgrad_x6_kernel(UpWgradArgs a) {
    constexpr int NP = 3;
    constexpr int XIT = UNPD * (UWC / 4) / 512, DIT = UTAPS * UNPD * (UWC / 4) / 512;  // 2 and 6 float4 units per thread and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sX = smem;                     // [plane][ci block][64 px][32 ch]
    char* const sD = smem + NP * 2 * UIMG;     // [tap][plane][co block][64 px][32 ch]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    // a wave = one 32-channel co block x BOTH ci blocks x one 16-pixel quarter of the tile: its x fragments serve three taps and its
    // dzt fragments two ci blocks - 30 fragment reads per 36 MFMAs (the first version - one ci block, two quarters - read 24 per 18)
    const int cob = wave & 1, quarter = wave >> 1;
    const int co_tiles = (a.cout + UWC - 1) / UWC;
    const int tg = blockIdx.y % 3;                         // taps 3 tg .. 3 tg + 2
    const int co_tile = (blockIdx.y / 3) % co_tiles, ci_tile = blockIdx.y / (3 * co_tiles);
    const int ci0 = ci_tile * UWC, co0 = co_tile * UWC;
    const int slice = blockIdx.x;
    const long ntiles = (a.M + UNPD - 1) / UNPD;

    f32x16 acc[2][UTAPS];
#pragma unroll
    for (int i = 0; i < 2 * UTAPS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i / UTAPS][i % UTAPS][r] = 0.f;

    const int l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, cb = (lane >> 4) & 1;
    const int lane_off = (8 * half + q) * UROWB + (16 * cb + 4 * p4) * 2;  // (wgrad_x6.hip: the transposing read's lane mapping)

    float4 rx[XIT], rd[DIT];
    unsigned inmask = 0u;
    auto prefetch = [&](long tile) __attribute__((always_inline)) {
        const long p0 = tile * UNPD;
        inmask = 0u;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (UWC / 4), pix = u / (UWC / 4);
            const int c = ci0 + c4 * 4;
            const bool in = p0 + pix < a.M && c < a.cs;
            const long pp = p0 + pix < a.M ? p0 + pix : a.M - 1;
            rx[it] = *reinterpret_cast<const float4*>(a.x + (size_t)pp * a.cs + min(c, a.cs - 4));
            inmask = in ? inmask | (1u << it) : inmask;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (UWC / 4), tap = (u / (UWC / 4)) % UTAPS, pix = u / ((UWC / 4) * UTAPS);
            const int c = co0 + c4 * 4;
            const bool in = p0 + pix < a.M && c < a.cout;
            const long pp = p0 + pix < a.M ? p0 + pix : a.M - 1;
            rd[it] = *reinterpret_cast<const float4*>(a.dzt + ((size_t)pp * 9 + (3 * tg + tap)) * a.cout + min(c, a.cout - 4));
            inmask = in ? inmask | (1u << (XIT + it)) : inmask;
        }
    };
    up_bf16x4 px[XIT + DIT][NP];
    auto split_unit = [&](int i) __attribute__((always_inline)) {  // i < XIT: x unit, else dzt unit
        const float4 v = i < XIT ? rx[i < XIT ? i : 0] : rd[i >= XIT ? i - XIT : 0];
        const float4 r = (inmask >> i) & 1u ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 ph, pm, pl;
        split3_quad<NP>(r, ph, pm, pl);
        px[i][0] = __builtin_bit_cast(up_bf16x4, ph);
        px[i][1] = __builtin_bit_cast(up_bf16x4, pm);
        px[i][2] = __builtin_bit_cast(up_bf16x4, pl);
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (UWC / 4), pix = u / (UWC / 4);
            const int off = (c4 >> 3) * UIMG + pix * UROWB + (c4 & 7) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<up_bf16x4*>(sX + p * 2 * UIMG + off) = px[it][p];
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (UWC / 4), tap = (u / (UWC / 4)) % UTAPS, pix = u / ((UWC / 4) * UTAPS);
            const int off = (c4 >> 3) * UIMG + pix * UROWB + (c4 & 7) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<up_bf16x4*>(sD + (tap * NP + p) * 2 * UIMG + off) = px[XIT + it][p];
        }
    };

    if (slice < ntiles) {
        prefetch(slice);
#pragma unroll
        for (int i = 0; i < XIT + DIT; ++i) split_unit(i);
    }
    for (long tile = slice; tile < ntiles; tile += a.nslices) {
        const long next = tile + a.nslices;
        const bool has_next = next < ntiles;  // block-uniform
        // the next tile's loads FIRST (their staging registers were consumed by the previous iteration's split): issued behind the
        // commit they had only half an MFMA phase to arrive before the split needed them
        prefetch(has_next ? next : tile);     // unconditional (a re-fetch is dropped); in flight during the commit and the MFMAs
        // raw barriers + an LDS-only wait: __syncthreads() fences with s_waitcnt vmcnt(0), i.e. it would wait right here for the
        // loads just issued - a full memory round trip exposed per tile (that form ran at 93-107 TFLOP/s)
        __builtin_amdgcn_s_barrier();         // the previous tile's fragment reads are done (every wave waited for its own)
        commit();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // vmcnt(63) expcnt(7) lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_s_barrier();
        {
            const int s = quarter;  // this wave's pixel group of 16
            UpFrag fa[2][NP];
#pragma unroll
            for (int cib = 0; cib < 2; ++cib)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int r = 0; r < 2; ++r) fa[cib][p].half4[r] = up_tr_read(sX, (p * 2 + cib) * UIMG + (16 * s + 4 * r) * UROWB + lane_off);
#pragma unroll
            for (int t = 0; t < UTAPS; ++t) {
                UpFrag fb[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        fb[p].half4[r] = up_tr_read(sD, ((t * NP + p) * 2 + cob) * UIMG + (16 * s + 4 * r) * UROWB + lane_off);
#pragma unroll
                for (int cib = 0; cib < 2; ++cib) {
                    f32x16 c = acc[cib][t];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cib][1].v, fb[1].v, c, 0, 0, 0);  // mm
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cib][2].v, fb[0].v, c, 0, 0, 0);  // lh
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cib][0].v, fb[2].v, c, 0, 0, 0);  // hl
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cib][1].v, fb[0].v, c, 0, 0, 0);  // mh
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cib][0].v, fb[1].v, c, 0, 0, 0);  // hm
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cib][0].v, fb[0].v, c, 0, 0, 0);  // hh
                    acc[cib][t] = c;
                }
                // the next tile's split in the shadow of these MFMAs (its loads were issued before the commit)
                if (t == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) split_unit(i);
                } else if (t == 1) {
#pragma unroll
                    for (int i = 3; i < 6; ++i) split_unit(i);
                } else {
#pragma unroll
                    for (int i = 6; i < XIT + DIT; ++i) split_unit(i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // slab (4 slice + quarter): rows [tap][ci], columns co - the layout of wgrad3x3_x6_kernel's slabs (reduce_slabs sums them)
    const int co = co0 + cob * 32 + l31;
    if (co < a.cout) {
#pragma unroll
        for (int cib = 0; cib < 2; ++cib)
#pragma unroll
            for (int t = 0; t < UTAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + cib * 32 + mfma32_row(r, half);
                    if (ci < a.cs) a.partial[((((size_t)(4 * slice + quarter)) * 9 + 3 * tg + t) * a.cs + ci) * a.cout + co] = acc[cib][t][r];
                }
    }
}

// ---- round 5: the same weight gradient with a 128 x 64 (ci, co) tile, 32-pixel tiles and TWO operand image sets.
// The first kernel above owns 64 x 64 channels x 3 taps and 64-pixel tiles in one single-buffered 96 KB image set: at the deepest
// up-block (cup = 512, cout = 256) that is 96 channel groups x 2 pixel slices = 192 workgroups on 256 CUs, the nine adjoint maps
// are re-read by 8 ci tiles, x by 12 (co, tap) groups (923 MB fetched per launch against ~430 MB algorithmic), two barriers per
// tile: 0.31 of the peak's matrix cycles.  Here a workgroup owns 128 ci x 64 co x 3 taps: its 8 waves = 2 co blocks x 2 ci
// halves (two 32-channel blocks each) x 2 halves of a 32-pixel tile (the halves are two split-K slabs); per tile it stages
// 16 KB of x + 24 KB of dzt for the same 36 MFMAs per wave (0.625 x the bytes per MFMA), the nine maps are read by HALF as many
// ci tiles, and with 2 x 60 KB of images the next tile is written while the current one is still being read: ONE barrier per
// tile.  48 channel groups x 5 slices = 240 workgroups at the deepest block (12 x 21, 3 x 85 at the other two).
// (images 2 KB + 64 bytes apart: the 16 lanes of a staging ds_write_b64 group are one pixel's channel quads of TWO neighbouring
// channel-block images, and every ds_write maps onto 32 banks = 128 bytes - a multiple of 128 bytes apart the two images shared 16
// banks: SQ_LDS_BANK_CONFLICT was 33 % of this kernel's LDS cycles; wgrad_x6.hip has the same padding)
#ifndef PUNET_UPWGRAD_IMGPAD
#define PUNET_UPWGRAD_IMGPAD 64
#endif
constexpr int U2CI = 128, U2PD = 32, U2IMG = U2PD * UROWB + PUNET_UPWGRAD_IMGPAD;  // 2 KB images: 32 pixels x 32 channels
constexpr int U2X = 3 * 4 * U2IMG, U2D = UTAPS * 3 * 2 * U2IMG, U2SET = U2X + U2D;  // 24 KB + 36 KB per set

__global__ __launch_bounds__(512, 2) void upconv_wgrad_x6w_kernel(UpWgradArgs a) {
    constexpr int NP = 3;
    constexpr int XIT = U2PD * (U2CI / 4) / 512, DIT = UTAPS * U2PD * (UWC / 4) / 512;  // 2 and 3 float4 units per thread and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int cob = wave & 1, cih = (wave >> 1) & 1, ks = wave >> 2;  // co block, ci half (blocks 2 cih, 2 cih + 1), 16-pixel half of the tile
    const int co_tiles = (a.cout + UWC - 1) / UWC;
    const int tg = blockIdx.y % 3;
    const int co_tile = (blockIdx.y / 3) % co_tiles, ci_tile = blockIdx.y / (3 * co_tiles);
    const int ci0 = ci_tile * U2CI, co0 = co_tile * UWC;
    const int slice = blockIdx.x;
    const long ntiles = (a.M + U2PD - 1) / U2PD;

    f32x16 acc[2][UTAPS];
#pragma unroll
    for (int i = 0; i < 2 * UTAPS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i / UTAPS][i % UTAPS][r] = 0.f;

    const int l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, cb = (lane >> 4) & 1;
    const int lane_off = (8 * half + q) * UROWB + (16 * cb + 4 * p4) * 2;  // (wgrad_x6.hip: the transposing read's lane mapping)

    float4 rx[XIT], rd[DIT];
    unsigned inmask = 0u;
    auto prefetch = [&](long tile) __attribute__((always_inline)) {
        const long p0 = tile * U2PD;
        inmask = 0u;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (U2CI / 4), pix = u / (U2CI / 4);
            const int c = ci0 + c4 * 4;
            const bool in = p0 + pix < a.M && c < a.cs;
            const long pp = p0 + pix < a.M ? p0 + pix : a.M - 1;
            rx[it] = *reinterpret_cast<const float4*>(a.x + (size_t)pp * a.cs + min(c, a.cs - 4));
            inmask = in ? inmask | (1u << it) : inmask;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (UWC / 4), tap = (u / (UWC / 4)) % UTAPS, pix = u / ((UWC / 4) * UTAPS);
            const int c = co0 + c4 * 4;
            const bool in = p0 + pix < a.M && c < a.cout;
            const long pp = p0 + pix < a.M ? p0 + pix : a.M - 1;
            rd[it] = *reinterpret_cast<const float4*>(a.dzt + ((size_t)pp * 9 + (3 * tg + tap)) * a.cout + min(c, a.cout - 4));
            inmask = in ? inmask | (1u << (XIT + it)) : inmask;
        }
    };
    up_bf16x4 px[XIT + DIT][NP];
    auto split_unit = [&](int i) __attribute__((always_inline)) {  // i < XIT: x unit, else dzt unit
        const float4 v = i < XIT ? rx[i < XIT ? i : 0] : rd[i >= XIT ? i - XIT : 0];
        const float4 r = (inmask >> i) & 1u ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 ph, pm, pl;
        split3_quad<NP>(r, ph, pm, pl);
        px[i][0] = __builtin_bit_cast(up_bf16x4, ph);
        px[i][1] = __builtin_bit_cast(up_bf16x4, pm);
        px[i][2] = __builtin_bit_cast(up_bf16x4, pl);
    };
    auto commit = [&](int set) __attribute__((always_inline)) {
        char* const sX = smem + set * U2SET;  // [plane][ci block][32 px][32 ch]
        char* const sD = sX + U2X;            // [tap][plane][co block][32 px][32 ch]
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (U2CI / 4), pix = u / (U2CI / 4);
            const int off = (c4 >> 3) * U2IMG + pix * UROWB + (c4 & 7) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<up_bf16x4*>(sX + p * 4 * U2IMG + off) = px[it][p];
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (UWC / 4), tap = (u / (UWC / 4)) % UTAPS, pix = u / ((UWC / 4) * UTAPS);
            const int off = (c4 >> 3) * U2IMG + pix * UROWB + (c4 & 7) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<up_bf16x4*>(sD + (tap * NP + p) * 2 * U2IMG + off) = px[XIT + it][p];
        }
    };

    int set = 0;
    if (slice < ntiles) {
        prefetch(slice);
#pragma unroll
        for (int i = 0; i < XIT + DIT; ++i) split_unit(i);
        commit(0);
    }
    for (long tile = slice; tile < ntiles; tile += a.nslices) {
        const long next = tile + a.nslices;
        const bool has_next = next < ntiles;  // block-uniform
        prefetch(has_next ? next : tile);     // in flight across the barrier and the MFMAs (a re-fetch of the last tile is dropped)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // vmcnt(63) expcnt(7) lgkmcnt(0): this wave's LDS writes of set `set` have landed
        __builtin_amdgcn_s_barrier();         // ... and everybody's; everybody has finished reading the other set (the previous tile)
        {
            const char* const sX = smem + set * U2SET;
            const char* const sD = sX + U2X;
            UpFrag fa[2][NP];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        fa[c2][p].half4[r] = up_tr_read(sX, (p * 4 + 2 * cih + c2) * U2IMG + (16 * ks + 4 * r) * UROWB + lane_off);
#pragma unroll
            for (int t = 0; t < UTAPS; ++t) {
                UpFrag fb[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        fb[p].half4[r] = up_tr_read(sD, ((t * NP + p) * 2 + cob) * U2IMG + (16 * ks + 4 * r) * UROWB + lane_off);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    f32x16 c = acc[c2][t];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c2][1].v, fb[1].v, c, 0, 0, 0);  // mm
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c2][2].v, fb[0].v, c, 0, 0, 0);  // lh
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c2][0].v, fb[2].v, c, 0, 0, 0);  // hl
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c2][1].v, fb[0].v, c, 0, 0, 0);  // mh
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c2][0].v, fb[1].v, c, 0, 0, 0);  // hm
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c2][0].v, fb[0].v, c, 0, 0, 0);  // hh
                    acc[c2][t] = c;
                }
                // the next tile's split in the shadow of these MFMAs (its loads were issued in front of the barrier)
                if (t == 0) {
                    split_unit(0);
                    split_unit(1);
                } else if (t == 1) {
                    split_unit(2);
                    split_unit(3);
                } else {
                    split_unit(4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (has_next) commit(set ^ 1);  // into the set nobody reads any more (everybody passed this tile's barrier)
        set ^= 1;
    }
    // slab (2 slice + ks): rows [tap][ci], columns co - the layout of wgrad3x3_x6_kernel's slabs (reduce_slabs sums them)
    const int co = co0 + cob * 32 + l31;
    if (co < a.cout) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int t = 0; t < UTAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + (2 * cih + c2) * 32 + mfma32_row(r, half);
                    if (ci < a.cs) a.partial[((((size_t)(2 * slice + ks)) * 9 + 3 * tg + t) * a.cs + ci) * a.cout + co] = acc[c2][t][r];
                }
    }
}

// $PUNET_UPCONV_WGRAD=64: the first kernel (64 x 64 tiles, 64-pixel tiles, one image set), for same-box A/B runs; read once
static int punet_upconv_wgrad_wide() {
    static const int on = [] {
        const char* e = getenv("PUNET_UPCONV_WGRAD");
        return !(e && e[0] == '6');
    }();
    return on;
}

int shift_of(int c4) {  // log2 when a power of two, else -1
    for (int s = 0; s < 16; ++s)
        if ((1 << s) == c4) return s;
    return -1;
}

}  // namespace

// ONE statement of where the low-resolution form applies, used by both graph walkers (engine.py::_upconv_ok / _split_up and
// runtime.hip::up_eligible - ADVICE r4: the rule used to be written out three times): the 3x3 entry convolution of an up-block with
// `cup` up-sampled and `cb` bridge channels, both past the first-layer kernels' range (> 7) and multiples of 8, and whole 64-channel
// output blocks (the gather publishes gate bits per 64-channel block).  punet_upconv_min_cup_grad(): below this many `up` channels a
// forward pass that a backward follows keeps the layer as written (the gather's cost is per OUTPUT channel, the matrix work it saves
// per `up` channel: at 128 the form still wins by ~0.1 ms per launch at 512 x 512, below that it is unmeasured).
extern "C" int punet_upconv_eligible(int cup, int cb, int cout, int ksize) {
    return ksize == 3 && cup > 7 && cb > 7 && cup % 8 == 0 && cb % 8 == 0 && cout > 0 && cout % 64 == 0;
}
extern "C" int punet_upconv_min_cup_grad(void) { return 128; }

extern "C" int punet_upconv_relayout(const float* w, int cup, int cb, int cout, float* w9, float* wb, void* stream) {
    PUNET_CHECK_ARG(w && w9 && wb && cup > 0 && cb > 0 && cout > 0 && cout % 4 == 0, "punet_upconv_relayout: bad args (cup=%d cb=%d cout=%d)",
                    cup, cb, cout);
    const size_t n = (size_t)9 * (cup + cb) * (cout / 4);
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    upconv_relayout_kernel<<<grid, 256, 0, STREAM>>>((const float4*)w, (float4*)w9, (float4*)wb, cup, cb, cout / 4);
    PUNET_LAUNCH_CHECK("upconv_relayout");
    return PUNET_OK;
}

extern "C" int punet_upconv_gather_fwd(const float* z, float* y, void* gate_out, int relu, int B, int h, int w, int cout, void* stream) {
    PUNET_CHECK_ARG(z && y && B > 0 && h > 0 && w > 0 && cout > 0 && cout % 4 == 0, "punet_upconv_gather_fwd: bad args (cout=%d)", cout);
    PUNET_CHECK_ARG(!gate_out || cout % 64 == 0, "punet_upconv_gather_fwd: gate bits need cout %% 64 == 0 (got %d)", cout);
    const int C4 = cout / 4, W = 2 * w;
    PUNET_CHECK_ARG((size_t)w * 9 * C4 * 16 < (1ull << 32) && (size_t)W * C4 < (1ull << 31), "punet_upconv_gather_fwd: a row of z exceeds 2^32 bytes");
    constexpr int NX = PUNET_UPCONV_NX;
    const int rows = B * 2 * h, gy = rows < 65535 ? rows : 65535;
    const dim3 grid(ceil_div(ceil_div(W, NX) * C4, 256), gy, ceil_div(rows, gy));
    if (gate_out)
        upconv_gather_fwd_kernel<true, NX><<<grid, 256, 0, STREAM>>>((const float4*)z, (float4*)y, (unsigned*)gate_out, B, h, w, C4, shift_of(C4), relu);
    else
        upconv_gather_fwd_kernel<false, NX><<<grid, 256, 0, STREAM>>>((const float4*)z, (float4*)y, nullptr, B, h, w, C4, shift_of(C4), relu);
    PUNET_LAUNCH_CHECK("upconv_gather_fwd");
    return PUNET_OK;
}

extern "C" int punet_upconv_gather_bwd(const float* dz, float* dzt, int B, int h, int w, int cout, void* stream) {
    PUNET_CHECK_ARG(dz && dzt && B > 0 && h > 0 && w > 0 && cout > 0 && cout % 4 == 0, "punet_upconv_gather_bwd: bad args (cout=%d)", cout);
    const int C4 = cout / 4;
    PUNET_CHECK_ARG((size_t)B * 4 * h * w * C4 < (1ull << 31) && (size_t)B * h * w * 9 * C4 < (1ull << 31),
                    "punet_upconv_gather_bwd: more than 2^31 float4 elements");
    const dim3 grid(ceil_div(w * C4, 256), B * h < 65535 ? B * h : 65535);
    upconv_gather_bwd_kernel<<<grid, 256, 0, STREAM>>>((const float4*)dz, (float4*)dzt, B, h, w, C4, shift_of(C4));
    PUNET_LAUNCH_CHECK("upconv_gather_bwd");
    return PUNET_OK;
}

// split-K slices of punet_upconv_wgrad_x6 (about one resident workgroup per CU) and its workspace (4 slabs per slice)
extern "C" int punet_upconv_wgrad_slices(int cup, int cout, int B, int h, int w) {
    const bool wide = punet_upconv_wgrad_wide();
    const long ntiles = ((long)B * h * w + (wide ? U2PD : UNPD) - 1) / (wide ? U2PD : UNPD);
    const int groups = 3 * ceil_div(cup, wide ? U2CI : UWC) * ceil_div(cout, UWC);
    if (groups < 1 || ntiles < 1) return 1;
    long ns = punet_num_cus() / groups;
    if (ns > ntiles) ns = ntiles;
    return ns < 1 ? 1 : (int)ns;
}
extern "C" size_t punet_upconv_wgrad_workspace_floats(int cup, int cout, int nslices) { return (size_t)4 * nslices * 9 * cup * cout; }

extern "C" int punet_upconv_wgrad_x6(const float* x, int cup, const float* dzt, int cout, float* dw, int cin_total, float* workspace,
                                     int nslices, int B, int h, int w, int accumulate, float scale, void* stream) {
    PUNET_CHECK_ARG(x && dzt && dw && workspace, "punet_upconv_wgrad_x6: null pointer");
    PUNET_CHECK_ARG(cup > 0 && cup % 4 == 0 && cout > 0 && cout % 4 == 0 && cup <= cin_total && nslices >= 1 && B > 0 && h > 0 && w > 0,
                    "punet_upconv_wgrad_x6: bad args (cup=%d cout=%d cin_total=%d nslices=%d)", cup, cout, cin_total, nslices);
    UpWgradArgs a{};
    a.x = x; a.dzt = dzt; a.partial = workspace; a.cs = cup; a.cout = cout; a.nslices = nslices; a.M = (long)B * h * w;
    int nslabs;
    if (punet_upconv_wgrad_wide()) {
        constexpr int SMEM = 2 * U2SET;  // 122,880 B + the images' padding
        if (hipFuncSetAttribute((const void*)upconv_wgrad_x6w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            punet_set_error("punet_upconv_wgrad_x6: cannot reserve %d bytes of LDS", SMEM);
            return PUNET_ERR_HIP;
        }
        const dim3 grid(nslices, 3 * ceil_div(cup, U2CI) * ceil_div(cout, UWC));
        upconv_wgrad_x6w_kernel<<<grid, 512, SMEM, STREAM>>>(a);
        nslabs = 2 * nslices;
    } else {
        constexpr int SMEM = 3 * 2 * UIMG + UTAPS * 3 * 2 * UIMG;  // 98,304 B
        if (hipFuncSetAttribute((const void*)upconv_wgrad_x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            punet_set_error("punet_upconv_wgrad_x6: cannot reserve %d bytes of LDS", SMEM);
            return PUNET_ERR_HIP;
        }
        const dim3 grid(nslices, 3 * ceil_div(cup, UWC) * ceil_div(cout, UWC));
        upconv_wgrad_x6_kernel<<<grid, 512, SMEM, STREAM>>>(a);
        nslabs = 4 * nslices;
    }
    PUNET_LAUNCH_CHECK("upconv_wgrad_x6");
    // rows [tap][ci] of the slabs -> rows [0, cup) of every tap's [cin_total][cout] block of dw (deterministic: slabs in slice order)
    return punet_launch_reduce_slabs(workspace, nslabs, (size_t)9 * cup * cout, 9 * cup, cout, cup, cin_total, 0, dw, accumulate, scale, STREAM);
}
