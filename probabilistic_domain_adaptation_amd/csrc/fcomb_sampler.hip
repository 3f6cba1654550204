// K7: persistent latent-sampler + Fcomb kernel with the consensus / mean epilogue.
//
// Replaces the N-sample Monte-Carlo loop of the reference:
//   [sigmoid(model.sample()) for _ in range(n_samples)]  + thresholds + stack/sum
//   (prob_utils/my_trainer/mean_teacher_trainer.py:72-88, fixmatch_trainer.py:37-54,
//    my_predictions/punet_predictions.py:29-33,108-124) where every sample() runs
//   Fcomb.forward (probabilistic_unet.py:200-214): tile z, cat [features, z], 1x1 convs.
// The reference materialises a [B, C+L, H, W] concat plus two [B, C, H, W] temporaries per sample;
// here nothing per-sample ever reaches HBM.
//
// Algebra (exact up to fp32 summation order): the first 1x1 conv splits into a feature term
//   T1 = W1[:C]^T . F  (computed ONCE per pixel) and a z term folded into a per-(sample, image) bias
//   zb = b1 + W1[C:]^T . z  (punet_fcomb_zbias).  Per sample only relu(T1 + zb), the middle 1x1
//   conv(s), the last C->1 dot, the sigmoid and the consensus test remain.
// Mapping: one wave = 32 consecutive pixels.  All products are computed "channels x pixels"
//   (A = weights [cout][k], B = activations [k][pixel]) so that the 32x32 accumulator tile of one layer,
//   which holds pixel = lane & 31 and 16 channels in its registers, IS the B operand of the next
//   layer's v_mfma_f32_32x32x2_f32 - activations never leave the register file between layers.
// Persistent: the grid is a fixed number of workgroups that stride over the pixel tiles; the weights
//   (<= 64 KB) are staged in LDS once per workgroup.
#include "common.h"
#include "punet_hip.h"

namespace {

struct SamplerArgs {
    const float* feat;   // [P][CH]
    const float* w1;     // [CH + L][CH]   (rows 0..CH-1 used here)
    const float* zb;     // [N][B][CH]
    const float* wmid;   // [NMID][CH][CH]
    const float* bmid;   // [NMID][CH]
    const float* w3;     // [CH]
    const float* b3;     // [1]
    float* y;            // [P] mean of sigmoids (nullable)
    float* z;            // [P] consensus (nullable)
    float* logits;       // [N][P] (nullable)
    long wmid_stride, bmid_stride;  // elements between consecutive middle layers
    int N, B, HW, nmid, masking, ntiles, tpi;  // tpi = tiles per image = ceil(HW / 32)
    float upper, lower;
};

template <int NBLK>
__global__ __launch_bounds__(256) void fcomb_sampler_kernel(SamplerArgs a) {
    constexpr int CH = 32 * NBLK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;                          // [CH][CH]
    float* sWm = sW1 + CH * CH;                 // [nmid][CH][CH]
    float* sBm = sWm + a.nmid * CH * CH;        // [nmid][CH]
    float* sW3 = sBm + a.nmid * CH;             // [CH]

    const int tid = threadIdx.x;
    for (int i = tid; i < CH * CH / 4; i += 256)
        reinterpret_cast<float4*>(sW1)[i] = reinterpret_cast<const float4*>(a.w1)[i];
    for (int l = 0; l < a.nmid; ++l) {
        for (int i = tid; i < CH * CH; i += 256) sWm[l * CH * CH + i] = a.wmid[l * a.wmid_stride + i];
        for (int i = tid; i < CH; i += 256) sBm[l * CH + i] = a.bmid[l * a.bmid_stride + i];
    }
    for (int i = tid; i < CH; i += 256) sW3[i] = a.w3[i];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const float b3 = a.b3[0];
    const size_t P = (size_t)a.B * a.HW;

    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        // tiles never straddle images (z differs per image): image b, 32 pixels from p0; the last tile of an image whose H * W is
        // not a multiple of 32 is ragged - its dead lanes recompute the image's last pixel and store nothing
        const int b = tile / a.tpi;
        const int pl = (tile - b * a.tpi) * 32 + l31;
        const bool live = pl < a.HW;
        const size_t pix = (size_t)b * a.HW + (live ? pl : a.HW - 1);

        // ---- feature term, once per pixel: T1[co][pix] = sum_k W1[k][co] * F[pix][k]
        // k slot of half h in step (g, e) is channel 8g + 4h + e (A and B use the same bijection)
        float4 fr[CH / 8];
#pragma unroll
        for (int g = 0; g < CH / 8; ++g)
            fr[g] = *reinterpret_cast<const float4*>(a.feat + pix * CH + 8 * g + 4 * half);
        f32x16 t1[NBLK];
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[cb][r] = 0.f;
#pragma unroll
        for (int g = 0; g < CH / 8; ++g) {
            const float fv[4] = {fr[g].x, fr[g].y, fr[g].z, fr[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * g + e;  // + 4 * half at run time
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb) {
                    const float av = sW1[(k + 4 * half) * CH + cb * 32 + l31];
                    t1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, fv[e], t1[cb], 0, 0, 0);
                }
            }
        }

        float ysum = 0.f, cnt = 0.f;
        for (int n = 0; n < a.N; ++n) {
            const float* zbp = a.zb + ((size_t)n * a.B + b) * CH;
            f32x16 ha[NBLK], hb[NBLK];
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ha[cb][r] = fmaxf(t1[cb][r] + zbp[cb * 32 + mfma32_row(r, half)], 0.f);

            for (int l = 0; l < a.nmid; ++l) {
                const float* wl = sWm + l * CH * CH;
                const float* bl = sBm + l * CH;
#pragma unroll
                for (int ob = 0; ob < NBLK; ++ob) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) hb[ob][r] = 0.f;
#pragma unroll
                    for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float av = wl[(cb * 32 + mfma32_row(r, half)) * CH + ob * 32 + l31];
                            hb[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ha[cb][r], hb[ob], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int ob = 0; ob < NBLK; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ha[ob][r] = fmaxf(hb[ob][r] + bl[ob * 32 + mfma32_row(r, half)], 0.f);
            }

            float part = 0.f;
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(ha[cb][r], sW3[cb * 32 + mfma32_row(r, half)], part);
            const float logit = part + __shfl_xor(part, 32) + b3;
            if (a.logits && half == 0 && live) a.logits[(size_t)n * P + pix] = logit;
            const float s = 1.f / (1.f + expf(-logit));
            ysum += s;
            cnt += (s >= a.upper || s <= a.lower) ? 1.f : 0.f;
        }
        if (half == 0 && live) {
            if (a.y) a.y[pix] = ysum / (float)a.N;
            if (a.z) {
                const float c = cnt / (float)a.N;
                a.z[pix] = a.masking ? (c == 1.f ? 1.f : 0.f) : c;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same kernel in the split-plane arithmetic of conv_x6.hip (round 2): every fp32 operand = three bf16 planes, every
// product = hh + hm + mh + hl + lh + mm on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (fp32-accurate, dropped terms
// <= 2^-26 relative).  Six bf16 MFMAs of 32 cycles contract K = 16 where the fp32 kernel needs eight MFMAs of 64 cycles:
// 2.67x less matrix-pipe time, and this kernel is matrix-pipe-bound (76 % busy).
// K ordering: an accumulator register block holds, per lane half h, the channels row(r, h) = (r & 3) + 8 (r >> 2) + 4 h; a
// K block of 16 is {registers 8 jj .. 8 jj + 7 of both halves} = channels 32 cbk + 16 jj .. + 15 in a fixed permutation, so
// the B operand of the next layer is just 8 consecutive accumulator registers split into planes - still no data movement
// between layers.  The weights are split once per workgroup into LDS in the matching order: [layer][plane][K block][half][co]
// x 8 bf16 (one conflict-free ds_read_b128 per fragment).  The feature rows (global, 8 g + 4 h + e) follow the same map.
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3s(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

struct Planes {
    bf16x8s p[3];
};

__device__ __forceinline__ Planes split8(const float (&v)[8]) {
    Planes o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h, m, l;
        split3s(v[i], h, m, l);
        o.p[0][i] = h; o.p[1][i] = m; o.p[2][i] = l;
    }
    return o;
}

// acc += W (planes wa, as A) x act (planes xb, as B): the six products, smallest terms first
__device__ __forceinline__ f32x16 mfma6(const bf16x8s (&wa)[3], const Planes& xb, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], xb.p[1], c, 0, 0, 0);  // mm
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], xb.p[2], c, 0, 0, 0);  // hl
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[2], xb.p[0], c, 0, 0, 0);  // lh
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], xb.p[1], c, 0, 0, 0);  // hm
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], xb.p[0], c, 0, 0, 0);  // mh
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], xb.p[0], c, 0, 0, 0);  // hh
    return c;
}

#ifndef PUNET_K7X_WGS
#define PUNET_K7X_WGS 2  // workgroups per CU the register allocation aims at (tools/ab_variants.sh experiment knob)
#endif
template <int NBLK, int F16 = 0>  // F16: a.feat holds 1 = bf16 (the "bf16s" fast mode), 2 = fp16 ("f16s")
__global__ __launch_bounds__(256, PUNET_K7X_WGS) void fcomb_sampler_x6_kernel(SamplerArgs a) {
    constexpr int CH = 32 * NBLK, KB = CH / 16;
    constexpr int LAYER = 3 * KB * 2 * CH;  // uint4 (8 x bf16) units per layer: [plane][K block][half][co]
    extern __shared__ __attribute__((aligned(16))) char smem_x6[];
    uint4* sA = reinterpret_cast<uint4*>(smem_x6);                       // [1 + nmid][LAYER]
    float* sBm = reinterpret_cast<float*>(sA + (1 + a.nmid) * LAYER);    // [nmid][CH]
    float* sW3 = sBm + a.nmid * CH;                                      // [CH]

    const int tid = threadIdx.x;
    // split the weights into planes, K-permuted as described above: unit u = ((j * 2 + h) * CH + co), slot i <-> channel
    // 16 j + 8 (i >> 2) + 4 h + (i & 3)
    for (int l = 0; l <= a.nmid; ++l) {
        const float* w = l == 0 ? a.w1 : a.wmid + (size_t)(l - 1) * a.wmid_stride;  // [k][co], k = input channel
        for (int u = tid; u < KB * 2 * CH; u += 256) {
            const int co = u % CH, h = (u / CH) & 1, j = u / (2 * CH);
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = w[(size_t)(16 * j + 8 * (i >> 2) + 4 * h + (i & 3)) * CH + co];
            const Planes pl = split8(v);
#pragma unroll
            for (int p = 0; p < 3; ++p) sA[l * LAYER + p * (KB * 2 * CH) + u] = __builtin_bit_cast(uint4, pl.p[p]);
        }
    }
    for (int l = 0; l < a.nmid; ++l)
        for (int i = tid; i < CH; i += 256) sBm[l * CH + i] = a.bmid[l * a.bmid_stride + i];
    for (int i = tid; i < CH; i += 256) sW3[i] = a.w3[i];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const float b3 = a.b3[0];
    const size_t P = (size_t)a.B * a.HW;
    auto afrag = [&](int layer, int j, int ob, bf16x8s (&wa)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            wa[p] = __builtin_bit_cast(bf16x8s, sA[layer * LAYER + p * (KB * 2 * CH) + (j * 2 + half) * CH + ob * 32 + l31]);
    };

    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        // tiles never straddle images (z differs per image): image b, 32 pixels from p0; the last tile of an image whose H * W is
        // not a multiple of 32 is ragged - its dead lanes recompute the image's last pixel and store nothing
        const int b = tile / a.tpi;
        const int pl = (tile - b * a.tpi) * 32 + l31;
        const bool live = pl < a.HW;
        const size_t pix = (size_t)b * a.HW + (live ? pl : a.HW - 1);

        // ---- feature term, once per pixel: T1[co][pix] = sum_k W1[k][co] * F[pix][k]
        f32x16 t1[NBLK];
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[cb][r] = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            Planes xb;
            if constexpr (F16 == 2) {  // fp16 storage: 11 significand bits = exactly two bf16 planes; the generic split does it
                typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
                const _Float16* f16 = reinterpret_cast<const _Float16*>(a.feat) + pix * CH + 16 * j + 4 * half;
                const f16x4s g0 = *reinterpret_cast<const f16x4s*>(f16), g1 = *reinterpret_cast<const f16x4s*>(f16 + 8);
                const float fv[8] = {(float)g0[0], (float)g0[1], (float)g0[2], (float)g0[3], (float)g1[0], (float)g1[1], (float)g1[2], (float)g1[3]};
                xb = split8(fv);
            } else if constexpr (F16 == 1) {  // already bf16: plane 0 is the value, the other planes are zero
                typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
                const __bf16* f16 = reinterpret_cast<const __bf16*>(a.feat) + pix * CH + 16 * j + 4 * half;
                const bf16x4s g0 = *reinterpret_cast<const bf16x4s*>(f16), g1 = *reinterpret_cast<const bf16x4s*>(f16 + 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) { xb.p[0][i] = g0[i]; xb.p[0][4 + i] = g1[i]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) { xb.p[1][i] = (__bf16)0.f; xb.p[2][i] = (__bf16)0.f; }
            } else {
                const float4 f0 = *reinterpret_cast<const float4*>(a.feat + pix * CH + 16 * j + 4 * half);
                const float4 f1 = *reinterpret_cast<const float4*>(a.feat + pix * CH + 16 * j + 8 + 4 * half);
                const float fv[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                xb = split8(fv);
            }
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb) {
                bf16x8s wa[3];
                afrag(0, j, cb, wa);
                t1[cb] = mfma6(wa, xb, t1[cb]);
            }
        }

        float ysum = 0.f, cnt = 0.f;
        for (int n = 0; n < a.N; ++n) {
            const float* zbp = a.zb + ((size_t)n * a.B + b) * CH;
            f32x16 ha[NBLK], hb[NBLK];
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) ha[cb][r] = fmaxf(t1[cb][r] + zbp[cb * 32 + mfma32_row(r, half)], 0.f);

            for (int l = 0; l < a.nmid; ++l) {
                const float* bl = sBm + l * CH;
#pragma unroll
                for (int ob = 0; ob < NBLK; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hb[ob][r] = 0.f;
#pragma unroll
                for (int cbk = 0; cbk < NBLK; ++cbk)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = ha[cbk][8 * jj + i];
                        const Planes xb = split8(v);
#pragma unroll
                        for (int ob = 0; ob < NBLK; ++ob) {
                            bf16x8s wa[3];
                            afrag(1 + l, 2 * cbk + jj, ob, wa);
                            hb[ob] = mfma6(wa, xb, hb[ob]);
                        }
                    }
#pragma unroll
                for (int ob = 0; ob < NBLK; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ha[ob][r] = fmaxf(hb[ob][r] + bl[ob * 32 + mfma32_row(r, half)], 0.f);
            }

            float part = 0.f;
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(ha[cb][r], sW3[cb * 32 + mfma32_row(r, half)], part);
            const float logit = part + __shfl_xor(part, 32) + b3;
            if (a.logits && half == 0 && live) a.logits[(size_t)n * P + pix] = logit;
            const float s = 1.f / (1.f + expf(-logit));
            ysum += s;
            cnt += (s >= a.upper || s <= a.lower) ? 1.f : 0.f;
        }
        if (half == 0 && live) {
            if (a.y) a.y[pix] = ysum / (float)a.N;
            if (a.z) {
                const float c = cnt / (float)a.N;
                a.z[pix] = a.masking ? (c == 1.f ? 1.f : 0.f) : c;
            }
        }
    }
}

}  // namespace

static int fcomb_sample_launch(bool x6, int f16, const float* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                               const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z, float* logits,
                               int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking, void* stream) {
    PUNET_CHECK_ARG(feat && w1 && zb && w3 && b3, "punet_fcomb_sample: null pointer");
    PUNET_CHECK_ARG(CH == 32 || CH == 64, "punet_fcomb_sample: num_filters[0] must be 32 or 64 (got %d)", CH);
    PUNET_CHECK_ARG(nmid >= 0 && nmid <= 4 && (nmid == 0 || (wmid && bmid)), "punet_fcomb_sample: bad nmid %d", nmid);
    PUNET_CHECK_ARG(N > 0 && B > 0 && HW > 0, "punet_fcomb_sample: bad shape (N %d, B %d, H*W %d)", N, B, HW);
    SamplerArgs a{};
    a.feat = feat; a.w1 = w1; a.zb = zb; a.wmid = wmid; a.bmid = bmid; a.wmid_stride = wmid_stride; a.bmid_stride = bmid_stride; a.w3 = w3; a.b3 = b3;
    a.y = y; a.z = z; a.logits = logits;
    a.N = N; a.B = B; a.HW = HW; a.nmid = nmid; a.masking = masking;
    a.tpi = (HW + 31) / 32;
    a.ntiles = B * a.tpi;
    a.upper = upper; a.lower = lower;
    // fp32 kernel: fp32 weights in LDS; x6 kernel: three bf16 planes per weight (6 B instead of 4 B per element)
    const size_t shmem = x6 ? ((size_t)CH * CH * (1 + nmid) * 6 + ((size_t)nmid * CH + CH) * sizeof(float))
                            : ((size_t)CH * CH * (1 + nmid) + (size_t)nmid * CH + CH) * sizeof(float);
    int grid = (a.ntiles + 3) / 4;
    if (grid > 2 * punet_num_cus()) grid = 2 * punet_num_cus();
    const void* fn = x6 ? (CH == 64 ? (const void*)fcomb_sampler_x6_kernel<2> : (const void*)fcomb_sampler_x6_kernel<1>)
                        : (CH == 64 ? (const void*)fcomb_sampler_kernel<2> : (const void*)fcomb_sampler_kernel<1>);
    if (shmem > 48 * 1024) {  // raise the dynamic-LDS cap (160 KiB per CU on gfx950)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
            punet_set_error("punet_fcomb_sample: cannot reserve %zu bytes of LDS", shmem);
            return PUNET_ERR_HIP;
        }
    }
    if (x6 && f16) {
        const void* fn16 = f16 == 2 ? (CH == 64 ? (const void*)fcomb_sampler_x6_kernel<2, 2> : (const void*)fcomb_sampler_x6_kernel<1, 2>)
                                    : (CH == 64 ? (const void*)fcomb_sampler_x6_kernel<2, 1> : (const void*)fcomb_sampler_x6_kernel<1, 1>);
        if (shmem > 48 * 1024 && hipFuncSetAttribute(fn16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
            punet_set_error("punet_fcomb_sample: cannot reserve %zu bytes of LDS", shmem);
            return PUNET_ERR_HIP;
        }
        if (f16 == 2) {
            if (CH == 64) fcomb_sampler_x6_kernel<2, 2><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
            else fcomb_sampler_x6_kernel<1, 2><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
        } else {
            if (CH == 64) fcomb_sampler_x6_kernel<2, 1><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
            else fcomb_sampler_x6_kernel<1, 1><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
        }
    } else if (x6) {
        if (CH == 64) fcomb_sampler_x6_kernel<2><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
        else fcomb_sampler_x6_kernel<1><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
    } else {
        if (CH == 64) fcomb_sampler_kernel<2><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
        else fcomb_sampler_kernel<1><<<grid, 256, shmem, (hipStream_t)stream>>>(a);
    }
    PUNET_LAUNCH_CHECK("fcomb_sampler");
    return PUNET_OK;
}

extern "C" int punet_fcomb_sample(const float* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                                  const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z, float* logits, int N, int B, int HW,
                                  int CH, int nmid, float upper, float lower, int masking, void* stream) {
    return fcomb_sample_launch(false, 0, feat, w1, zb, wmid, wmid_stride, bmid, bmid_stride, w3, b3, y, z, logits, N, B, HW, CH, nmid,
                               upper, lower, masking, stream);
}

extern "C" int punet_fcomb_sample_x6(const float* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                                     const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z, float* logits, int N, int B,
                                     int HW, int CH, int nmid, float upper, float lower, int masking, void* stream) {
    return fcomb_sample_launch(true, 0, feat, w1, zb, wmid, wmid_stride, bmid, bmid_stride, w3, b3, y, z, logits, N, B, HW, CH, nmid,
                               upper, lower, masking, stream);
}

// the same with the feature map stored as bf16 (the "bf16s" fast mode; opt-in, reduced precision)
extern "C" int punet_fcomb_sample_x6_bf16s(const void* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                                           const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z,
                                           float* logits, int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking,
                                           void* stream) {
    return fcomb_sample_launch(true, 1, (const float*)feat, w1, zb, wmid, wmid_stride, bmid, bmid_stride, w3, b3, y, z, logits, N, B, HW,
                               CH, nmid, upper, lower, masking, stream);
}

// ... stored as fp16 (the "f16s" fast mode)
extern "C" int punet_fcomb_sample_x6_f16s(const void* feat, const float* w1, const float* zb, const float* wmid, long wmid_stride,
                                          const float* bmid, long bmid_stride, const float* w3, const float* b3, float* y, float* z,
                                          float* logits, int N, int B, int HW, int CH, int nmid, float upper, float lower, int masking,
                                          void* stream) {
    return fcomb_sample_launch(true, 2, (const float*)feat, w1, zb, wmid, wmid_stride, bmid, bmid_stride, w3, b3, y, z, logits, N, B, HW,
                               CH, nmid, upper, lower, masking, stream);
}
