// K7t: the single-sample Fcomb of the TRAINING path - forward, BCE criterion and the whole backward (input gradient and
// every weight / bias gradient) - in ONE persistent kernel.
//
// Replaces, per student pass, what round 1 ran as ~14 launches (probabilistic_unet.py:200-214 + :360-369 and their autograd):
//   conv1x1 (x6) + ReLU  [feat -> h0, bias = folded z term]      conv1x1 input gradient x (1 + nmid)
//   conv1x1 (x6) + ReLU  [h0 -> h1] per middle layer              1x1 weight gradient (fp32 MFMA, split-K) x (1 + nmid)
//   dot + bias           [h_last -> logit]                        dot backward, per-image column sums, BCE forward/backward
// with h0, h1, dZ1, dZ0 (256 B per pixel each) written to and re-read from HBM.  Here a pixel's activations and gradients
// never leave the wave: HBM traffic = feat in (256 B/px), dfeat out (256 B/px), logits out, target / mask in.
//
// Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) everywhere, as in the sampler K7.
// Layout, as in K7: every product is computed "channels x pixels" (A = weights, B = activations), one wave = 32 pixels; the
// 32x32 accumulator of one layer (lane = pixel, 16 channels per register block) IS the B operand of the next MFMA, forward
// and backward.  The weight gradients contract over PIXELS, i.e. need both operands transposed ("32 channels of one pixel"
// per half-wave): the wave writes the two operand tiles as [pixel][channel] rows into its private LDS region (one float4 per
// accumulator quad) and reads them back row-wise; for the first layer the feature rows come straight from global memory
// (they are L1/L2-resident: the same wave has just loaded them).
// Determinism: no atomics.  Per-workgroup partial slabs (waves summed in order) + per-tile column sums, reduced by the usual
// fixed-order second pass on the host side of this file.
#include <type_traits>
#include "common.h"
#include "punet_hip.h"

#include <math.h>

// timing-only knob (tools/ab_variants.sh): bit 0 skips the two weight-gradient MFMA loops, bit 1 the dW3 reductions, bit 2 the
// input-gradient MFMAs, bit 3 the forward middle layer - WRONG results, used to see where a tile's time goes
#ifndef PUNET_FT_SKIP
#define PUNET_FT_SKIP 0
#endif

namespace {

#ifndef PUNET_FT_PITCH  // A/B: 0 = the tile pitch before round 5 (CH + 32 / CH)
#define PUNET_FT_PITCH 1
#endif
__host__ __device__ constexpr int ft_tile_pitch(int ch) { return PUNET_FT_PITCH ? ch + 4 : (ch % 64 == 0 ? ch + 32 : ch); }

struct TrainArgs {
    const float* feat;    // [P][CH]
    const float* w1;      // [CH + L][CH], rows 0..CH-1 used
    const float* zb;      // [B][CH]   b1 + W1[CH:]^T z
    const float* wmid;    // [CH][CH]  (NMID == 1)
    const float* bmid;    // [CH]
    const float* w3;      // [CH]
    const float* b3;      // [1]
    const float* target;  // [P]
    const float* consm;   // [P] or null
    float* logits;        // [P]
    float* dfeat;         // [P][CH]
    float* slabs;         // [grid][slab_stride]: dW1f [CH][CH] | dWm [CH][CH] | dbm [CH] | dW3 [CH] | db3 [1]
    float* stile;         // [ntiles][CH]: per-tile column sums of dZ0 (-> per-image sums -> z gradient, b1 gradient)
    float* loss_part;     // [grid] (BCE) / [3][grid] partial Dice sums (CRIT 1)
    const float* dice_sums;  // CRIT 2: {sum p t, sum p^2, sum t^2} over all pixels (device)
    float gscale;
    int B, HW, ntiles, slab_stride;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// sum over the 32 lanes of a half-wave: four DPP steps inside each 16-lane row (quad swaps, half-row mirror, row mirror) and one
// cross-row exchange (a butterfly of five ds_bpermutes per value had cost more than the tile's MFMAs)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half32_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v = dpp_add<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v + __shfl_xor(v, 16);
}

template <bool FIRST, typename A, typename B>
__device__ __forceinline__ auto& pick(A& a, B& b) {
    if constexpr (FIRST) return a; else return b;
}

// F16: a.feat holds 1 = bf16 (the "bf16s" fast mode), 2 = fp16 ("f16s")
// CRIT: the criterion - 0 = BCE (probabilistic_unet.py:348); the Dice criterion (:347, [recalled torch_em definition, parity
// unpinned]: 1 - 2 sum(p t) / max(sum p^2 + sum t^2, eps) over batch and pixels) is a global ratio, so it takes two launches:
// 1 = forward only (logits + the three partial sums per workgroup), 2 = forward + backward given the reduced sums.
template <int NBLK, int NMID, int F16 = 0, int CRIT = 0>
__global__ __launch_bounds__(256, 1) void fcomb_train_kernel(TrainArgs a) {
    typedef typename std::conditional<F16 == 2, _Float16, __bf16>::type ft_t16;
    // four consecutive channels of one pixel / one channel of one pixel, from either storage
    auto ldf4 = [&](size_t e) __attribute__((always_inline)) -> float4 {
        if constexpr (F16 != 0) {
            typedef ft_t16 ft_bf16x4 __attribute__((ext_vector_type(4)));
            const ft_bf16x4 h = *reinterpret_cast<const ft_bf16x4*>(reinterpret_cast<const ft_t16*>(a.feat) + e);
            return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        } else {
            return *reinterpret_cast<const float4*>(a.feat + e);
        }
    };
    auto ldf1 = [&](size_t e) __attribute__((always_inline)) -> float {
        if constexpr (F16 != 0) return (float)reinterpret_cast<const ft_t16*>(a.feat)[e];
        else return a.feat[e];
    };
    constexpr int CH = 32 * NBLK;
    constexpr int WP = CH + 1;                       // weight pitch: row-wise AND column-wise b32 reads are conflict-free
    // tile pitch CH + 4: the tiles are WRITTEN as one float4 per lane with a lane stride of one row, and every ds_write maps
    // addresses onto 32 banks (cdna_hip_programming.md, LDS): with the former pitches (96 / 32 dwords = 0 mod 32) the 16 lanes of a
    // write group fell onto the same four banks - SQ_LDS_BANK_CONFLICT was 60 % of this kernel's LDS cycles
    // (profiles/r05/pmc_lds_conflicts_adamatch_704x704x3.json).  17 (9) slots of 16 bytes per row: consecutive lanes take
    // consecutive slots mod 8.  The row-wise b32 reads (32 consecutive dwords per half-wave) are conflict-free at any pitch.
    constexpr int TP = ft_tile_pitch(CH);
    constexpr int TILE = 32 * TP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;                               // [CH][WP]
    float* sWm = sW1 + CH * WP;                      // [NMID][CH][WP]
    float* sBm = sWm + NMID * CH * WP;               // [NMID][CH]
    float* sW3 = sBm + NMID * CH;                    // [CH]
    // 4 waves x 2 tiles x [32][TP]; reused for the final reduction.  16-byte aligned by OFFSET arithmetic (smem is aligned): an
    // integer round trip of the pointer loses its LDS address space, and every tile access became a flat_load / flat_store
    // (160 + 167 of them) that counts on vmcnt AND lgkmcnt.
    constexpr int T_OFF = ((CH * WP + NMID * CH * WP + NMID * CH + CH) + 3) / 4 * 4;
    float* sT = smem + T_OFF;

    const int tid = threadIdx.x;
    for (int i = tid; i < CH * CH; i += 256) sW1[(i / CH) * WP + i % CH] = a.w1[i];
    if constexpr (NMID > 0) {
        for (int i = tid; i < CH * CH; i += 256) sWm[(i / CH) * WP + i % CH] = a.wmid[i];
        for (int i = tid; i < CH; i += 256) sBm[i] = a.bmid[i];
    }
    for (int i = tid; i < CH; i += 256) sW3[i] = a.w3[i];
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    float* TA = sT + wave * 2 * TILE;
    float* TB = TA + TILE;
    const float b3 = a.b3[0];

    // accumulators that live across tiles
    f32x16 accW1[NBLK][NBLK];                     // dW1f[k][co]: reg r <-> k = 32 kb + row(r, half), lane <-> co = 32 cb + l31
    f32x16 accWm[NMID ? NBLK : 1][NMID ? NBLK : 1];
    // channel sums kept ONE register per 32-channel block: lane (l31, half) owns channel 32 cb + l31; the two halves hold
    // partial sums over the pixels of their parity and are combined at the end
    float accBm[NBLK], accW3[NBLK];
    float accB3 = 0.f, accLoss = 0.f, accDice = 0.f;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        accBm[i] = accW3[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW1[i][j][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < (NMID ? NBLK : 1); ++i)
#pragma unroll
        for (int j = 0; j < (NMID ? NBLK : 1); ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accWm[i][j][r] = 0.f;

    // write an accumulator-layout tile (lane = pixel, reg quad q <-> channels 32 cb + 8 q + 4 half .. + 3) as [pixel][channel]
    auto put_tile = [&](float* T, const f32x16 (&v)[NBLK]) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(T + l31 * TP + cb * 32 + 8 * q + 4 * half) =
                    make_float4(v[cb][4 * q + 0], v[cb][4 * q + 1], v[cb][4 * q + 2], v[cb][4 * q + 3]);
    };

    // waves are independent inside this loop (their LDS tiles are private, no workgroup barrier): each strides over its tiles
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        const size_t pix = (size_t)tile * 32 + l31;
        const int b = (int)(((size_t)tile * 32) / a.HW);
        // everything the tile reads from global memory is issued here, before the first MFMA (one wave per SIMD: a load issued
        // where its value is needed is a fully exposed round trip)
        float4 zq[NBLK][4];
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                zq[cb][q] = *reinterpret_cast<const float4*>(a.zb + (size_t)b * CH + cb * 32 + 8 * q + 4 * half);
        const float cm = a.consm ? a.consm[pix] : 1.f;
        const float tgt = a.target[pix];

        // ---------------- forward: h0 = relu(W1f^T F + zb)
        float4 fr[CH / 8];
#pragma unroll
        for (int g = 0; g < CH / 8; ++g) fr[g] = ldf4(pix * CH + 8 * g + 4 * half);
        f32x16 h0[NBLK];
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) h0[cb][r] = 0.f;
#pragma unroll
        for (int g = 0; g < CH / 8; ++g) {
            const float fv[4] = {fr[g].x, fr[g].y, fr[g].z, fr[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * g + e + 4 * half;
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb)
                    h0[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[k * WP + cb * 32 + l31], fv[e], h0[cb], 0, 0, 0);
            }
            if (g & 1) __builtin_amdgcn_sched_barrier(0);  // bound the look-ahead of the weight reads (registers)
        }
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 zv = zq[cb][r >> 2];  // channel 32 cb + 8 (r >> 2) + 4 half + (r & 3) = row(r, half)
                const float zb_ = (r & 3) == 0 ? zv.x : (r & 3) == 1 ? zv.y : (r & 3) == 2 ? zv.z : zv.w;
                h0[cb][r] = fmaxf(h0[cb][r] + zb_, 0.f);
            }

        __builtin_amdgcn_sched_barrier(0);  // phase boundary: LDS reads of the next phase must not be hoisted (registers)
        // ---------------- middle layer: h1 = relu(Wm^T h0 + bm)
        f32x16 h1[NMID ? NBLK : 1];
        if constexpr (NMID > 0) {
#pragma unroll
            for (int ob = 0; ob < NBLK; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[ob][r] = 0.f;
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (!(PUNET_FT_SKIP & 8)) h1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(sWm[(cb * 32 + mfma32_row(r, half)) * WP + ob * 32 + l31],
                                                                      h0[cb][r], h1[ob], 0, 0, 0);
                        if (r == 7 || r == 15) __builtin_amdgcn_sched_barrier(0);
                    }
            }
#pragma unroll
            for (int ob = 0; ob < NBLK; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[ob][r] = fmaxf(h1[ob][r] + sBm[ob * 32 + mfma32_row(r, half)], 0.f);
        }
        auto& hl = pick<(NMID > 0)>(h1, h0);  // input of the last (C -> 1) layer

        __builtin_amdgcn_sched_barrier(0);  // phase boundary: LDS reads of the next phase must not be hoisted (registers)
        // ---------------- logit, BCE with the consensus mask multiplied into logits AND targets (probabilistic_unet.py:363-364)
        float part = 0.f;
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(hl[cb][r], sW3[cb * 32 + mfma32_row(r, half)], part);
        const float logit = part + __shfl_xor(part, 32) + b3;
        const float x = logit * cm, t = tgt * cm;
        float dlogit;
        if constexpr (CRIT == 0) {
            dlogit = a.gscale * (sigm(x) - t) * cm;  // both halves hold the pixel's value
            if (half == 0) {
                a.logits[pix] = logit;
                accLoss += (1.f - t) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
                accB3 += dlogit;
            }
        } else if constexpr (CRIT == 1) {  // Dice, first launch: logits and the partial sums; no backward
            const float pr = sigm(x);
            if (half == 0) {
                a.logits[pix] = logit;
                accLoss += pr * t;   // sum p t
                accB3 += pr * pr;    // sum p^2
                accDice += t * t;    // sum t^2
            }
            continue;
        } else {  // Dice, second launch: d loss / d logit from the global sums (dice_bwd_kernel's formula)
            const float sa = a.dice_sums[0], den = a.dice_sums[1] + a.dice_sums[2];
            const bool clamped = den < 1e-7f;
            const float D = clamped ? 1e-7f : den;
            const float pr = sigm(x);
            float dp = -2.f * t / D;
            if (!clamped) dp += 2.f * sa * 2.f * pr / (D * D);
            dlogit = a.gscale * dp * pr * (1.f - pr) * cm;
            if (half == 0) {
                a.logits[pix] = logit;
                accB3 += dlogit;
            }
        }
        const float dl = dlogit;

        __builtin_amdgcn_sched_barrier(0);  // phase boundary: LDS reads of the next phase must not be hoisted (registers)
        // ---------------- last layer backward: dW3 += dlogit * h_last, dZ_last = dlogit * w3 gated by (h_last > 0)
        // dW3[c] += sum_pix dlogit[pix] * h_last[c][pix]: the sum runs over the LANES of the accumulator layout; instead of 32
        // cross-lane reductions per tile the h_last tile goes through LDS ([pixel][channel] rows, as for the weight gradients)
        // and lane (c, half) adds up its 16 pixels
        f32x16 dz[NBLK];
        if (!(PUNET_FT_SKIP & 2)) {
            put_tile(TB, hl);
            if (half == 0) TA[l31] = dl;  // TA is free until the middle layer's backward
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                const float dlv = TA[2 * sidx + half];
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb) accW3[cb] = fmaf(dlv, TB[(2 * sidx + half) * TP + cb * 32 + l31], accW3[cb]);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[cb][r] = hl[cb][r] > 0.f ? dl * sW3[cb * 32 + mfma32_row(r, half)] : 0.f;

        __builtin_amdgcn_sched_barrier(0);  // phase boundary: LDS reads of the next phase must not be hoisted (registers)
        // ---------------- middle layer backward: dWm += h0 (x) dZ1 over the pixels, dbm += dZ1, dZ0 = (Wm dZ1) gated by (h0 > 0)
        if constexpr (NMID > 0) {
            // (no barrier: TA / TB are private to this wave, whose LDS operations execute in program order)
            put_tile(TA, h0);
            put_tile(TB, dz);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float av[NBLK], bv[NBLK];
#pragma unroll
                for (int i = 0; i < NBLK; ++i) {
                    av[i] = TA[(2 * s + half) * TP + i * 32 + l31];
                    bv[i] = TB[(2 * s + half) * TP + i * 32 + l31];
                    accBm[i] += bv[i];  // dbm: column sums of dZ1 (this half's pixel parity)
                }
#pragma unroll
                for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
                    for (int cb = 0; cb < NBLK; ++cb)
                        if (!(PUNET_FT_SKIP & 1)) accWm[kb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb], bv[cb], accWm[kb][cb], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 d0[NBLK];
#pragma unroll
            for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d0[kb][r] = 0.f;
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (!(PUNET_FT_SKIP & 4)) d0[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sWm[(kb * 32 + l31) * WP + cb * 32 + mfma32_row(r, half)], dz[cb][r],
                                                                      d0[kb], 0, 0, 0);
                        if (r == 7 || r == 15) __builtin_amdgcn_sched_barrier(0);
                    }
            }
#pragma unroll
            for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[kb][r] = h0[kb][r] > 0.f ? d0[kb][r] : 0.f;
        }

        __builtin_amdgcn_sched_barrier(0);  // phase boundary: LDS reads of the next phase must not be hoisted (registers)
        // ---------------- first layer backward: dW1f += F (x) dZ0 over the pixels (feature rows straight from memory), per-tile
        // column sums of dZ0 (-> z gradient and b1 gradient), dF = (W1f dZ0) gated by (F > 0)
        put_tile(TB, dz);
        float ssum[NBLK];
#pragma unroll
        for (int i = 0; i < NBLK; ++i) ssum[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float av[NBLK], bv[NBLK];
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                av[i] = ldf1(((size_t)tile * 32 + 2 * s + half) * CH + i * 32 + l31);
                bv[i] = TB[(2 * s + half) * TP + i * 32 + l31];
                ssum[i] += bv[i];
            }
#pragma unroll
            for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb)
                    if (!(PUNET_FT_SKIP & 1)) accW1[kb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb], bv[cb], accW1[kb][cb], 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            const float s2 = ssum[i] + __shfl_xor(ssum[i], 32);  // even + odd pixels
            if (half == 0) a.stile[(size_t)tile * CH + i * 32 + l31] = s2;
        }
#pragma unroll
        for (int kb = 0; kb < NBLK; ++kb) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 df;
#pragma unroll
            for (int r = 0; r < 16; ++r) df[r] = 0.f;
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!(PUNET_FT_SKIP & 4)) df = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[(kb * 32 + l31) * WP + cb * 32 + mfma32_row(r, half)], dz[cb][r], df, 0, 0, 0);
                    if (r == 7 || r == 15) __builtin_amdgcn_sched_barrier(0);
                }
            {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // channels 32 kb + 8 q + 4 half .. + 3 of this pixel, re-read (L1/L2 hit) rather than kept in 32 registers
                    const float4 f = ldf4(pix * CH + kb * 32 + 8 * q + 4 * half);
                    float4 o;
                    o.x = f.x > 0.f ? df[4 * q + 0] : 0.f; o.y = f.y > 0.f ? df[4 * q + 1] : 0.f;
                    o.z = f.z > 0.f ? df[4 * q + 2] : 0.f; o.w = f.w > 0.f ? df[4 * q + 3] : 0.f;
                    *reinterpret_cast<float4*>(a.dfeat + pix * CH + kb * 32 + 8 * q + 4 * half) = o;
                }
            }
        }
    }

    // ---------------- one slab per workgroup: waves summed in order (deterministic)
    __syncthreads();
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    float* red = sT;  // [4 waves][CH * CH] fits: 4 x 2 x 32 x TP >= 4 x CH x CH for CH in {32, 64}
    if constexpr (CRIT == 1) {  // Dice, first launch: the three partial sums of this workgroup, nothing else
        float u = accLoss, v = accB3, w = accDice;
        for (int off = 32; off > 0; off >>= 1) { u += __shfl_xor(u, off); v += __shfl_xor(v, off); w += __shfl_xor(w, off); }
        if (lane == 0) { red[wave] = u; red[4 + wave] = v; red[8 + wave] = w; }
        __syncthreads();
        if (tid < 3)
            a.loss_part[(size_t)tid * gridDim.x + blockIdx.x] = ((red[4 * tid] + red[4 * tid + 1]) + red[4 * tid + 2]) + red[4 * tid + 3];
        return;
    }
    auto reduce_matrix = [&](const f32x16 (&acc)[NBLK][NBLK], float* out) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[wave * CH * CH + (kb * 32 + mfma32_row(r, half)) * CH + cb * 32 + l31] = acc[kb][cb][r];
        __syncthreads();
        for (int i = tid; i < CH * CH; i += 256) out[i] = ((red[i] + red[CH * CH + i]) + red[2 * CH * CH + i]) + red[3 * CH * CH + i];
        __syncthreads();
    };
    // channel sums: lane (l31, half) holds its half's part of channel 32 cb + l31; halves added, then the four waves in order
    auto reduce_vector = [&](const float (&acc)[NBLK], float* out) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb) {
            const float v = acc[cb] + __shfl_xor(acc[cb], 32);
            if (half == 0) red[wave * CH + cb * 32 + l31] = v;
        }
        __syncthreads();
        for (int i = tid; i < CH; i += 256) out[i] = ((red[i] + red[CH + i]) + red[2 * CH + i]) + red[3 * CH + i];
        __syncthreads();
    };
    reduce_matrix(accW1, slab);
    float* p = slab + CH * CH;
    if constexpr (NMID > 0) {
        reduce_matrix(accWm, p);
        p += CH * CH;
        reduce_vector(accBm, p);
        p += CH;
    }
    reduce_vector(accW3, p);
    p += CH;
    {
        float v = accB3, w = accLoss;
        for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off); w += __shfl_xor(w, off); }
        if (lane == 0) { red[wave] = v; red[4 + wave] = w; }
        __syncthreads();
        if (tid == 0) {
            p[0] = ((red[0] + red[1]) + red[2]) + red[3];
            a.loss_part[blockIdx.x] = ((red[4] + red[5]) + red[6]) + red[7];
        }
    }
}

template <int NBLK, int NMID>
constexpr size_t train_smem_bytes() {
    constexpr int CH = 32 * NBLK, WP = CH + 1, TP = ft_tile_pitch(CH);
    return ((size_t)CH * WP * (1 + NMID) + (size_t)NMID * CH + CH + 4 + (size_t)4 * 2 * 32 * TP) * sizeof(float);
}

__global__ void sum_loss_parts_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

int train_grid(int ntiles) {
    int g = (ntiles + 3) / 4;
    const int cap = punet_num_cus();  // one workgroup (4 waves, one per SIMD, up to 512 VGPRs each) per CU
    return g < cap ? g : cap;
}

size_t slab_floats(int CH, int nmid) { return ((size_t)CH * CH * (1 + nmid) + (size_t)nmid * CH + CH + 1 + 3) / 4 * 4; }

}  // namespace

extern "C" int punet_fcomb_train_supported(int CH, int nmid, int HW) {
    return (CH == 32 || CH == 64) && (nmid == 0 || nmid == 1) && HW > 0 && HW % 32 == 0;
}

extern "C" size_t punet_fcomb_train_workspace_floats(int B, int HW, int CH, int nmid) {
    const int ntiles = (int)(((size_t)B * HW) / 32);
    const int grid = train_grid(ntiles);
    return (size_t)grid * slab_floats(CH, nmid) + (size_t)ntiles * CH + (size_t)3 * grid + 8;  // (+ 3 x grid: the Dice partial sums)
}

static int fcomb_train_impl(int f16, const float* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                            const float* w3, const float* b3, const float* target, const float* consm, float gscale, float* logits,
                            float* dfeat, float* loss_sum, float* workspace, int B, int HW, int CH, int nmid, void* stream_, int dice = 0);

extern "C" int punet_fcomb_train_fwd_bwd(const float* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                                         const float* w3, const float* b3, const float* target, const float* consm, float gscale,
                                         float* logits, float* dfeat, float* loss_sum, float* workspace, int B, int HW, int CH,
                                         int nmid, void* stream_) {
    return fcomb_train_impl(0, feat, w1, zb, wmid, bmid, w3, b3, target, consm, gscale, logits, dfeat, loss_sum, workspace, B, HW,
                            CH, nmid, stream_);
}

extern "C" int punet_fcomb_train_fwd_bwd_bf16s(const void* feat, const float* w1, const float* zb, const float* wmid,
                                               const float* bmid, const float* w3, const float* b3, const float* target,
                                               const float* consm, float gscale, float* logits, float* dfeat, float* loss_sum,
                                               float* workspace, int B, int HW, int CH, int nmid, void* stream_) {
    return fcomb_train_impl(1, (const float*)feat, w1, zb, wmid, bmid, w3, b3, target, consm, gscale, logits, dfeat, loss_sum,
                            workspace, B, HW, CH, nmid, stream_);
}

extern "C" int punet_fcomb_train_fwd_bwd_f16s(const void* feat, const float* w1, const float* zb, const float* wmid,
                                              const float* bmid, const float* w3, const float* b3, const float* target,
                                              const float* consm, float gscale, float* logits, float* dfeat, float* loss_sum,
                                              float* workspace, int B, int HW, int CH, int nmid, void* stream_) {
    return fcomb_train_impl(2, (const float*)feat, w1, zb, wmid, bmid, w3, b3, target, consm, gscale, logits, dfeat, loss_sum,
                            workspace, B, HW, CH, nmid, stream_);
}

// sums[3] <- the [3][n] partial Dice sums (double accumulation, deterministic); loss[0] = 1 - 2 a / max(b + c, eps)
__global__ void dice_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ sums, float* __restrict__ loss) {
    __shared__ double sh[256];
    for (int k = 0; k < 3; ++k) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[(size_t)k * n + i];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[k] = (float)sh[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float den = sums[1] + sums[2];
        loss[0] = 1.f - 2.f * sums[0] / (den < 1e-7f ? 1e-7f : den);
    }
}

static int fcomb_train_impl(int f16, const float* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                            const float* w3, const float* b3, const float* target, const float* consm, float gscale, float* logits,
                            float* dfeat, float* loss_sum, float* workspace, int B, int HW, int CH, int nmid, void* stream_, int dice) {
    hipStream_t stream = (hipStream_t)stream_;
    PUNET_CHECK_ARG(feat && w1 && zb && w3 && b3 && target && logits && dfeat && loss_sum && workspace,
                    "punet_fcomb_train_fwd_bwd: null pointer");
    PUNET_CHECK_ARG(punet_fcomb_train_supported(CH, nmid, HW) && B > 0 && (nmid == 0 || (wmid && bmid)),
                    "punet_fcomb_train_fwd_bwd: needs num_filters[0] in {32, 64}, at most one middle layer and H*W %% 32 == 0 "
                    "(CH=%d nmid=%d HW=%d)", CH, nmid, HW);
    TrainArgs a{};
    a.feat = feat; a.w1 = w1; a.zb = zb; a.wmid = wmid; a.bmid = bmid; a.w3 = w3; a.b3 = b3; a.target = target; a.consm = consm;
    a.logits = logits; a.dfeat = dfeat; a.gscale = gscale;
    a.B = B; a.HW = HW; a.ntiles = (int)(((size_t)B * HW) / 32);
    const int grid = train_grid(a.ntiles);
    a.slab_stride = (int)slab_floats(CH, nmid);
    a.slabs = workspace;
    a.stile = workspace + (size_t)grid * a.slab_stride;
    a.loss_part = a.stile + (size_t)a.ntiles * CH;
#define PUNET_TRAIN_LAUNCH1(NB, NM, F, CR)                                                                                   \
    do {                                                                                                                     \
        const size_t sh = train_smem_bytes<NB, NM>();                                                                        \
        if (hipFuncSetAttribute((const void*)fcomb_train_kernel<NB, NM, F, CR>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)sh) != hipSuccess) {                                                                    \
            punet_set_error("punet_fcomb_train_fwd_bwd: cannot reserve %zu bytes of LDS", sh);                               \
            return PUNET_ERR_HIP;                                                                                            \
        }                                                                                                                    \
        fcomb_train_kernel<NB, NM, F, CR><<<grid, 256, sh, stream>>>(a);                                                     \
    } while (0)
#define PUNET_TRAIN_LAUNCH(NB, NM, CR)                    \
    do {                                                  \
        if (f16 == 2) PUNET_TRAIN_LAUNCH1(NB, NM, 2, CR); \
        else if (f16 == 1) PUNET_TRAIN_LAUNCH1(NB, NM, 1, CR); \
        else PUNET_TRAIN_LAUNCH1(NB, NM, 0, CR);          \
    } while (0)
#define PUNET_TRAIN_DISPATCH(L, CR)                     \
    do {                                                \
        if (CH == 64 && nmid == 1) L(2, 1, CR);         \
        else if (CH == 64) L(2, 0, CR);                 \
        else if (nmid == 1) L(1, 1, CR);                \
        else L(1, 0, CR);                               \
    } while (0)
#define PUNET_TRAIN_LAUNCH_F32(NB, NM, CR) PUNET_TRAIN_LAUNCH1(NB, NM, 0, CR)  // (the Dice launches: fp32 feature storage only)
    if (!dice) {
        PUNET_TRAIN_DISPATCH(PUNET_TRAIN_LAUNCH, 0);
        PUNET_LAUNCH_CHECK("fcomb_train");
        sum_loss_parts_kernel<<<1, 256, 0, stream>>>(a.loss_part, grid, loss_sum);
        PUNET_LAUNCH_CHECK("sum_loss_parts");
        return PUNET_OK;
    }
    // Dice: a global ratio - forward launch (logits + partial sums), the reduced sums and the loss, then forward + backward
    float* sums = a.loss_part + (size_t)3 * grid;
    PUNET_CHECK_ARG(f16 == 0, "punet_fcomb_train_fwd_bwd_dice: fp32 feature storage only");
    PUNET_TRAIN_DISPATCH(PUNET_TRAIN_LAUNCH_F32, 1);
    PUNET_LAUNCH_CHECK("fcomb_train (Dice sums)");
    dice_finish_kernel<<<1, 256, 0, stream>>>(a.loss_part, grid, sums, loss_sum);
    PUNET_LAUNCH_CHECK("dice_finish");
    a.dice_sums = sums;
    PUNET_TRAIN_DISPATCH(PUNET_TRAIN_LAUNCH_F32, 2);
    PUNET_LAUNCH_CHECK("fcomb_train (Dice)");
#undef PUNET_TRAIN_LAUNCH_F32
#undef PUNET_TRAIN_DISPATCH
#undef PUNET_TRAIN_LAUNCH
#undef PUNET_TRAIN_LAUNCH1
    return PUNET_OK;
}

// The same with the Dice criterion of rl_swap=True (probabilistic_unet.py:347; every LIVECell / MitoEM script) under the recalled
// torch_em definition [parity unpinned]: loss_out[0] = 1 - 2 sum(p t) / max(sum p^2 + sum t^2, 1e-7) with p = sigmoid(logit * consm),
// t = target * consm over batch and pixels.  Two launches of the fused kernel (the ratio is global) - fp32 feature storage.
extern "C" int punet_fcomb_train_fwd_bwd_dice(const float* feat, const float* w1, const float* zb, const float* wmid, const float* bmid,
                                              const float* w3, const float* b3, const float* target, const float* consm, float gscale,
                                              float* logits, float* dfeat, float* loss_out, float* workspace, int B, int HW, int CH,
                                              int nmid, void* stream_) {
    return fcomb_train_impl(0, feat, w1, zb, wmid, bmid, w3, b3, target, consm, gscale, logits, dfeat, loss_out, workspace, B, HW, CH,
                            nmid, stream_, 1);
}

// second pass: the partial slabs of punet_fcomb_train_fwd_bwd (same workspace, same B / HW / CH / nmid) into the gradient tensors;
// S[B][CH] = per-image column sums of dZ0 (what punet_fcomb_z_bwd takes); db1 = sum_b S[b]
extern "C" int punet_fcomb_train_reduce(const float* workspace, float* dw1, int w1_rows, float* db1, float* dwmid, float* dbmid,
                                        float* dw3, float* db3, float* S, float* chan_workspace, int B, int HW, int CH, int nmid,
                                        int accumulate, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PUNET_CHECK_ARG(workspace && dw1 && db1 && dw3 && db3 && S && chan_workspace && punet_fcomb_train_supported(CH, nmid, HW) &&
                        w1_rows >= CH && (nmid == 0 || (dwmid && dbmid)), "punet_fcomb_train_reduce: bad args");
    const int ntiles = (int)(((size_t)B * HW) / 32);
    const int grid = train_grid(ntiles);
    const size_t st = slab_floats(CH, nmid);
    const float* p = workspace;
    int rc = punet_launch_reduce_slabs(p, grid, st, CH, CH, CH, w1_rows, 0, dw1, accumulate, 1.f, stream);
    if (rc != PUNET_OK) return rc;
    p += (size_t)CH * CH;
    if (nmid) {
        rc = punet_launch_reduce_slabs_pair(p, st, CH, CH, CH, CH, 0, dwmid, p + (size_t)CH * CH, st, dbmid, grid, accumulate, 1.f, stream);
        if (rc != PUNET_OK) return rc;
        p += (size_t)CH * CH + CH;
    }
    rc = punet_launch_reduce_slabs(p, grid, st, 1, CH, 1, 1, 0, dw3, accumulate, 1.f, stream);
    if (rc != PUNET_OK) return rc;
    rc = punet_launch_reduce_slabs(p + CH, grid, st, 1, 1, 1, 1, 0, db3, accumulate, 1.f, stream);
    if (rc != PUNET_OK) return rc;
    // per-image column sums of dZ0 from the per-tile sums, then the bias gradient of the first layer
    const float* stile = workspace + (size_t)grid * st;
    rc = punet_chan_sum(stile, S, chan_workspace, B, HW / 32, CH, 1.f, stream_);
    if (rc != PUNET_OK) return rc;
    return punet_launch_reduce_slabs(S, B, (size_t)CH, 1, CH, 1, 1, 0, db1, accumulate, 1.f, stream);
}
