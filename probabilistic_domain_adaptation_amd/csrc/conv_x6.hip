// K1x6 / K2x6: the 3x3 NHWC direct convolution (and its input gradient) at fp32 accuracy on the BF16 matrix cores.
//
// Same contraction, tiling and epilogue semantics as conv_mfma.hip, but every fp32 operand is split into three bf16 planes
// x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 3 x 8 significand bits = fp32's 24) and each product
// is evaluated as  hh + hm + mh + hl + lh + mm  with v_mfma_f32_32x32x16_bf16 accumulating in fp32.  The three dropped
// terms (ml, lm, ll) are <= 2^-26 relative, below the rounding of an fp32 fmaf chain, so this is a PARITY mode (same
// tolerance class as the fp32-MFMA kernel), not a reduced-precision mode.  Six bf16 MFMAs do the work of sixteen fp32
// MFMAs' worth of matrix-pipe time (bf16 issues 16x the MACs per cycle), i.e. 2.67x the fp32-MFMA peak.
//
// K blocking: one bf16 MFMA contracts K = 16.  A K block here is {two taps} x {8 channels}: lanes 0-31 carry tap 2j,
// lanes 32-63 tap 2j+1 of the same 8 channels; taps 0-7 of a chunk of 8 input channels are 4 K-steps, and tap 8 is
// contracted for an even/odd PAIR of chunks in one K-step (the activation image is double-buffered by chunk parity, the
// even chunk's tap-8 weight slab parked in a tenth slot), so 16 channels cost 9 K-steps.  (The first version paired tap 8
// with a zero slab - 10 K-steps, 10 % wasted MFMAs on a kernel that is matrix-pipe-power-bound.)  That keeps the LDS
// images small - [pixel][8 ch] = one 16-byte ds_read_b128 per lane per plane, consecutive pixels / output channels in
// consecutive 16-byte slots = conflict-free with no padding - 62 KB per workgroup, 2 workgroups per CU (<= 256 VGPRs).
// Measured and removed in round 1 (same-process A/B): staging 16 channels at once (88 KB, one workgroup per CU): 5-35 %
// slower - with one wave per SIMD nothing hides the staging phases; an 8-wave workgroup with one pixel row per wave
// (128 VGPRs, 4 waves per SIMD): 3-8 % slower - consistent with tools/mfma_peak.hip, where 4 waves per SIMD sustain
// less MFMA throughput than 1 or 2; a ping-pong MFMA/staging split of an 8-wave workgroup: 3-8 % slower.
// Weights arrive pre-split (punet_split_weights_x6, once per optimizer step) as planes[3][tap][K/8][N][8] bf16, so their
// staging is a pure copy; activations are split while being staged (v_cvt_pk_bf16_f32, ~6 VALU ops per element,
// amortised over 9 taps x 64 output channels).
// Epilogue: the MFMAs run with the operands swapped (weights as A, pixels as B), so a lane ends up with one pixel and,
// per accumulator register quad, four consecutive channels: 16-byte stores straight from the accumulators, no LDS
// staging, no barrier (history and measurements at the epilogue lambda).  ReLU gates travel as bits (punet_gate_dwords).
#include <type_traits>
#include "common.h"
#include "punet_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32, TN = 64, CK = 8;
constexpr int NUM_CU = 256;

struct ConvX6Args {
    const float* x0;
    const float* x1;
    int c0, c1;
    const __bf16* wp;   // planes[3][9][K/8][n_row][8]
    long plane_stride;  // elements between planes
    int n_row, n_off;
    const float* bias;
    int bias_img_stride;
    float* y;
    int n_total;
    const float* mask;
    const unsigned* gate_bits;  // ReLU gate of the destination as bits (instead of `mask`), layout: punet_gate_dwords
    unsigned* gate_out;         // forward with ReLU: (y > 0) bits of the output
    float* pool_out;            // forward: AvgPool2d(2, 2) of the output as well (H, W even), nullptr = off
    int accumulate, relu;
    int B, H, W, ntiles;
    int xcd_remap;
};

struct TileInfo {
    int b, ty0, tx0, n0;
};

union U4 {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// NP = number of bf16 planes per operand: 3 -> six products (fp32-accurate, the parity arithmetic); 2 -> hh + hm + mh
// (16 significand bits, opt-in fast mode); 1 -> plain bf16 (opt-in).
// DGRAD selects the epilogue: false = forward (bias, ReLU, optional gate bits of the output); true = input gradient
// (ReLU gate of the destination as fp32 mask or bits, optional accumulate).  Two instantiations instead of run-time
// flags: the kernel sits at the 256-VGPR limit and every live flag / pointer of the other mode costs registers.
// KS = 3 or 1 (the Fcomb layers): with one tap the "last tap, two chunks per K-step" pairing is the whole K loop.
template <int NP, bool DGRAD, int KS = 3>
__global__ __launch_bounds__(256, 2) void conv3x3_x6_kernel(ConvX6Args a) {
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int HTH = TH + KS - 1, HTW = TW + KS - 1;
    constexpr int NPIX = HTH * HTW;
    constexpr int XUNITS = NPIX * (CK / 4);
    constexpr int XITER = (XUNITS + 255) / 256;
    constexpr int WUNITS = NP * KK * TN;  // uint4 (8 bf16) units per chunk
    constexpr int WITER = (WUNITS + 255) / 256;
    constexpr int WSLAB = (KK + 1) * TN;  // per plane: slots 0-7 = taps 0-7 of the current chunk, slot 8 = tap 8 of the
                                          // current ODD chunk, slot 9 = tap 8 of the preceding EVEN chunk
    constexpr int XBUF = NP * NPIX;       // one activation image (NP planes); two of them, indexed by chunk parity
    constexpr int OPERANDS = 2 * XBUF + NP * WSLAB;  // in uint4 units

    // one LDS array: [2][NP][NPIX] activation planes | [NP][WSLAB] weight planes (63,360 B at NP = 3, 2 workgroups per CU)
    __shared__ uint4 smem[OPERANDS];
    uint4* sX = smem;
    uint4* sW = smem + 2 * XBUF;
    static_assert(OPERANDS * 16 <= 65536, "static LDS limit");

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
    uint4 rw[WITER];

    // Staging addresses.  Everything that depends only on the thread and on the TILE is computed once per tile
    // (`retarget`): the clamped pixel index / in-image flag of each activation unit and the element offset of each weight
    // unit; per chunk a load then costs one multiply-add (activations) or one add (weights).  The staging phase is a few
    // hundred dependent instructions of ONE wave per SIMD and makes little progress while the other resident wave issues
    // MFMAs (measured: 1.6k cycles alone, 4.1k next to a wave in its K-steps), so every instruction removed from it counts.
    // All offsets are 32-bit (uniform 64-bit base + per-lane element offset; sizes are checked on the host).
    // BRANCH-FREE on purpose: out-of-range units load from a clamped (valid) address and are zeroed / deduplicated by
    // selects.  With the loads inside divergent branches the compiler's waitcnt insertion loses track of the vmcnt order
    // across the joins and falls back to s_waitcnt vmcnt(0) - i.e. it waited for the loads it had just issued for the
    // NEXT chunk, three exposed memory latencies per commit.
    int xpix[XITER];       // (b * H + clamped y) * W + clamped x of the unit's pixel
    bool xin[XITER];       // pixel inside the image
    unsigned wbase[WITER];  // plane, tap and output-channel part of the weight offset (chunk part added per chunk)
    auto retarget = [&](const TileInfo& ti) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int u = min(tid + it * 256, XUNITS - 1);  // a clamped duplicate re-stages the last unit (same data, same slot)
            const int pix = u / (CK / 4);
            const int py = pix / HTW, px = pix % HTW;
            const int gy = ti.ty0 + py - PAD, gx = ti.tx0 + px - PAD;
            xin[it] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            xpix[it] = (ti.b * a.H + min(max(gy, 0), a.H - 1)) * a.W + min(max(gx, 0), a.W - 1);
        }
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int u = min(tid + it * 256, WUNITS - 1);
            const int n = u % TN, t = (u / TN) % KK, p = u / (TN * KK);
            const int nc = min(ti.n0 + n, N - 1);  // columns beyond N copy column N - 1: their outputs are never stored
            wbase[it] = (unsigned)(p * (int)a.plane_stride + (t * nchunks * a.n_row + a.n_off + nc) * 8);
        }
    };
    auto load_x = [&](int ch, int it) __attribute__((always_inline)) -> float4 {
        const int kbase = ch * CK;
        const float* src = kbase < a.c0 ? a.x0 : a.x1;
        const int cs = kbase < a.c0 ? a.c0 : a.c1, coff = kbase < a.c0 ? kbase : kbase - a.c0;
        const int q = min(tid + it * 256, XUNITS - 1) % (CK / 4);  // the same clamped unit index as in retarget / commit
        // raw value: the out-of-image zeroing happens where the register is consumed (split_unit) - a select right here
        // would make the wave wait for the load it has just issued
        return *reinterpret_cast<const float4*>(src + (unsigned)(xpix[it] * cs + coff + q * 4));
    };
    auto load_w = [&](int ch, int it) __attribute__((always_inline)) -> uint4 {
        return *reinterpret_cast<const uint4*>(a.wp + (wbase[it] + (unsigned)(ch * a.n_row * 8)));
    };
    auto prefetch = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XITER; ++it) rx[it] = load_x(ch, it);
#pragma unroll
        for (int it = 0; it < WITER; ++it) rw[it] = load_w(ch, it);
    };

    // Registers -> LDS images of the current chunk, and - interleaved, each staging register is re-loaded as soon as it has
    // been consumed - the global loads of the NEXT chunk (a per-phase timeline of one wave showed 3.9k cycles in a commit
    // that waits for loads issued only ~4k cycles earlier, then 1.6k cycles in a separate load-issue phase, against 4.0k
    // cycles of K-steps: issuing the loads here, under the split arithmetic, buys them ~2k cycles of extra flight time and
    // removes the separate phase)
    // fp32 -> bf16 planes of one staged activation unit, kept PACKED in registers until the commit writes them to LDS.
    // It is called from inside the K-step sequence (same instruction stream as the MFMAs): a wave's own VALU work issues in
    // the shadow of its own MFMAs, whereas VALU work of the OTHER resident wave is held up behind them.
    bf16x4 px[XITER][NP];
    auto split_unit = [&](int it) __attribute__((always_inline)) {
        const float4 r = xin[it] ? rx[it] : make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding outside the image
        const float xv[4] = {r.x, r.y, r.z, r.w};
        bf16x4 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h_, m_, l_;
            split3(xv[e], h_, m_, l_);
            ph[e] = h_; pm[e] = m_; pl[e] = l_;
        }
        px[it][0] = ph;
        if (NP > 1) px[it][NP > 1 ? 1 : 0] = pm;
        if (NP > 2) px[it][NP > 2 ? 2 : 0] = pl;
    };
    auto commit = [&](int par, int nch) __attribute__((always_inline)) {
        uint4* sXc = sX + par * XBUF;
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int u = min(tid + it * 256, XUNITS - 1);
            const int q = u % (CK / 4), pix = u / (CK / 4);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(&sXc[p * NPIX + pix]) + q * 8) = px[it][p];
            rx[it] = load_x(nch, it);  // reload the staging register for the next chunk (split during this phase's K-steps)
        }
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int u = min(tid + it * 256, WUNITS - 1);
            {
                const int n = u % TN, t = (u / TN) % KK, p = u / (TN * KK);
                const int slot = t < KK - 1 ? t : KK - par;  // tap 8: slot 8 (odd chunk) or 9 (even chunk)
                sW[p * WSLAB + slot * TN + n] = rw[it];
            }
            rw[it] = load_w(nch, it);
        }
    };

    // Epilogue straight from the accumulators.  The MFMAs are issued with the operands swapped (weights as A, pixels as
    // B), i.e. they compute the TRANSPOSED tile: lane l then holds pixel l % 32 of the row and, per accumulator register
    // quad rg, the four CONSECUTIVE channels 8 rg + 4 (l / 32) .. + 3 - a ready-made 16-byte store per quad, with no
    // transposition through LDS, no barrier in front of the epilogue and no staging region.  (The first version computed
    // pixels x channels tiles and transposed them through LDS for 256-byte-contiguous stores: measured by skipping it,
    // that epilogue was 10 % of the forward and 13 % of the input-gradient time on the 64-channel layers.)
    // ReLU gates as bits (layout: punet_gate_dwords): per {row, 32-pixel tile, 64-channel block} one dword per lane of
    // THIS mapping; bit 16 nb + 4 rg + e <-> channel 32 nb + 8 rg + 4 (l / 32) + e of pixel l % 32.  The forward ORs its
    // comparisons into one register per row; the input gradient loads its dword once per row.
    auto epilogue = [&](const TileInfo& ti) __attribute__((always_inline)) {
        const int gx = ti.tx0 + l31;
        // all eight bias quads first (one latency; per-quad loads inside the store loop had been a chain of 16 dependent L2
        // round trips per tile - 16k cycles, 17 % of a 64-channel layer's tile)
        float4 bq4[2][4];
        if (!DGRAD) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c = ti.n0 + nb * 32 + rg * 8 + 4 * half;
                    bq4[nb][rg] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.bias && c < N) bq4[nb][rg] = *reinterpret_cast<const float4*>(a.bias + (size_t)ti.b * a.bias_img_stride + c);
                }
        }
        if constexpr (!DGRAD) {
            // forward: both rows of a channel quad together - the wave's two rows are a vertical pooling pair (the tile origin
            // is even) and the horizontal partner of pixel x is lane x ^ 1, so AvgPool2d(2, 2) of the output costs one DPP
            // add per row and value.  Summation order of pool2x2_fwd_kernel: ((y00 + y01) + y10) + y11, then * 0.25.
            const int gy0 = ti.ty0 + 2 * wave;
            const bool in0 = gy0 < a.H && gx < a.W, in1 = gy0 + 1 < a.H && gx < a.W;
            const unsigned gd0 = (unsigned)(((((ti.b * a.H + gy0) * tiles_x + (ti.tx0 / TW)) * ntn + (ti.n0 / TN)) << 6) + lane);
            const unsigned gd1 = gd0 + (unsigned)((tiles_x * ntn) << 6);
            const unsigned pbase0 = (unsigned)(((ti.b * a.H + gy0) * a.W + gx) * N + ti.n0 + 4 * half);
            const unsigned pbase1 = pbase0 + (unsigned)(a.W * N);
            const unsigned qbase = (unsigned)(((ti.b * (a.H >> 1) + (gy0 >> 1)) * (a.W >> 1) + (gx >> 1)) * N + ti.n0 + 4 * half);
            unsigned gout0 = 0u, gout1 = 0u;
            auto partner = [](float v) __attribute__((always_inline)) {  // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
            };
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ch = nb * 32 + rg * 8;  // + 4 * half + e
                    const bool okc = ti.n0 + ch + 4 * half < N;  // N is a multiple of 4
                    float4 v0 = make_float4(acc[0][nb][4 * rg + 0], acc[0][nb][4 * rg + 1], acc[0][nb][4 * rg + 2], acc[0][nb][4 * rg + 3]);
                    float4 v1 = make_float4(acc[1][nb][4 * rg + 0], acc[1][nb][4 * rg + 1], acc[1][nb][4 * rg + 2], acc[1][nb][4 * rg + 3]);
                    const float4 bq = bq4[nb][rg];
                    v0.x += bq.x; v0.y += bq.y; v0.z += bq.z; v0.w += bq.w;
                    v1.x += bq.x; v1.y += bq.y; v1.z += bq.z; v1.w += bq.w;
                    if (a.relu) {
                        v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                        v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                    }
                    if (in0 && okc) {
                        *reinterpret_cast<float4*>(a.y + pbase0 + ch) = v0;
                        gout0 |= ((v0.x > 0.f ? 1u : 0u) | (v0.y > 0.f ? 2u : 0u) | (v0.z > 0.f ? 4u : 0u) | (v0.w > 0.f ? 8u : 0u))
                                 << (16 * nb + 4 * rg);
                    }
                    if (in1 && okc) {
                        *reinterpret_cast<float4*>(a.y + pbase1 + ch) = v1;
                        gout1 |= ((v1.x > 0.f ? 1u : 0u) | (v1.y > 0.f ? 2u : 0u) | (v1.z > 0.f ? 4u : 0u) | (v1.w > 0.f ? 8u : 0u))
                                 << (16 * nb + 4 * rg);
                    }
                    if (a.pool_out) {  // uniform
                        float4 q;
                        q.x = (((v0.x + partner(v0.x)) + v1.x) + partner(v1.x)) * 0.25f;
                        q.y = (((v0.y + partner(v0.y)) + v1.y) + partner(v1.y)) * 0.25f;
                        q.z = (((v0.z + partner(v0.z)) + v1.z) + partner(v1.z)) * 0.25f;
                        q.w = (((v0.w + partner(v0.w)) + v1.w) + partner(v1.w)) * 0.25f;
                        if (!(lane & 1) && in0 && okc) *reinterpret_cast<float4*>(a.pool_out + qbase + ch) = q;
                    }
                }
            if (a.gate_out) {  // forward with ReLU: (y > 0), for the consumers' backward
                if (in0) a.gate_out[gd0] = gout0;
                if (in1) a.gate_out[gd1] = gout1;
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int gy = ti.ty0 + 2 * wave + m;
            const bool pix_in = gy < a.H && gx < a.W;
            const unsigned gd = (unsigned)(((((ti.b * a.H + gy) * tiles_x + (ti.tx0 / TW)) * ntn + (ti.n0 / TN)) << 6) + lane);
            const unsigned pbase = (unsigned)(((ti.b * a.H + gy) * a.W + gx) * N + ti.n0 + 4 * half);
            unsigned gin = 0u, gout = 0u;
            if (DGRAD && a.gate_bits && pix_in) gin = a.gate_bits[gd];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ch = nb * 32 + rg * 8;  // + 4 * half + e
                    const bool ok = pix_in && ti.n0 + ch + 4 * half < N;  // N is a multiple of 4
                    float4 v = make_float4(acc[m][nb][4 * rg + 0], acc[m][nb][4 * rg + 1], acc[m][nb][4 * rg + 2], acc[m][nb][4 * rg + 3]);
                    if (!DGRAD) {
                        v.x += bq4[nb][rg].x; v.y += bq4[nb][rg].y; v.z += bq4[nb][rg].z; v.w += bq4[nb][rg].w;
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    } else if (a.gate_bits) {
                        const unsigned g4 = gin >> (16 * nb + 4 * rg);
                        v.x = g4 & 1u ? v.x : 0.f; v.y = g4 & 2u ? v.y : 0.f; v.z = g4 & 4u ? v.z : 0.f; v.w = g4 & 8u ? v.w : 0.f;
                    } else if (a.mask) {
                        if (ok) {
                            const float4 g = *reinterpret_cast<const float4*>(a.mask + pbase + ch);
                            v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f;
                            v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
                        }
                    }
                    if (ok) {
                        if (DGRAD && a.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(a.y + pbase + ch);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(a.y + pbase + ch) = v;
                        if (!DGRAD)
                            gout |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                                    << (16 * nb + 4 * rg);
                    }
                }
            if (!DGRAD && a.gate_out && pix_in) a.gate_out[gd] = gout;  // forward with ReLU: (y > 0), for the consumers' backward
        }
    };

    // XCD-aware start tile: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), while consecutive tile
    // indices share their input pixels (they differ in the output-channel block first, then in the neighbouring pixel
    // block).  Give each XCD a CONTIGUOUS range of tiles so that those re-reads hit its own L2 instead of every XCD
    // fetching the same activations from the fabric.
    int tile = blockIdx.x;
    if (a.xcd_remap) tile = (tile & 7) * ((int)gridDim.x >> 3) + (tile >> 3);
    if (tile >= a.ntiles) return;
    int ch = 0;
    TileInfo ti = decode(tile);
    retarget(ti);
    prefetch(0);
#pragma unroll
    for (int it = 0; it < XITER; ++it) split_unit(it);
    // one K-step: 24 MFMAs on the fragments at sX[offA + ...] (offset incl. the image base) and weight slot slotB; both
    // arguments are per-half-wave values (lanes 0-31 and 32-63 contract different {tap, chunk} blocks of 8 channels)
    auto kstep = [&](int offA, int slotB, bool zero_a) __attribute__((always_inline)) {
        constexpr int P1 = NP > 1 ? 1 : 0, P2 = NP > 2 ? 2 : 0;  // plane indices that exist for this NP
        U4 av[NP][2], bv[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                av[p][m].u = sX[offA + p * NPIX + (2 * wave + m) * HTW + l31];
                if (zero_a) av[p][m].u = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) bv[p][nb].u = sW[p * WSLAB + slotB * TN + nb * 32 + l31];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x16 c = acc[m][nb];
                if (NP > 2) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[P1][nb].v, av[P1][m].v, c, 0, 0, 0);  // mm
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[0][nb].v, av[P2][m].v, c, 0, 0, 0);   // lh
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[P2][nb].v, av[0][m].v, c, 0, 0, 0);   // hl
                }
                if (NP > 1) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[0][nb].v, av[P1][m].v, c, 0, 0, 0);  // mh
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[P1][nb].v, av[0][m].v, c, 0, 0, 0);  // hm
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[0][nb].v, av[0][m].v, c, 0, 0, 0);  // hh
                acc[m][nb] = c;
            }
    };
    constexpr int OFF8 = ((KK - 1) / KS) * HTW + (KK - 1) % KS;  // halo offset of tap 8

    // one chunk of 8 input channels; the parity of the chunk index is a compile-time constant so that every LDS offset
    // below is an instruction immediate.  Returns true when this workgroup has no further work.
    auto phase = [&](auto par_c) __attribute__((always_inline)) -> bool {
        constexpr int par = decltype(par_c)::value;
        int nch = ch + 1, ntile = tile;
        TileInfo nti = ti;
        if (nch == nchunks) {
            nch = 0;
            ntile = tile + (int)gridDim.x;
            if (ntile < a.ntiles) nti = decode(ntile);
        }
        const bool has_next = ntile < a.ntiles;
        if (!has_next) nch = ch;  // nothing follows: the (unconditional, branch-free) reloads re-fetch this chunk and are dropped
        if (has_next && nch == 0) retarget(nti);  // the reloads of this phase already belong to the next tile
        __syncthreads();
        // the staging phase is a few hundred dependent VALU / LDS / VMEM instructions of ONE wave per SIMD; the other
        // resident wave (the other workgroup) is usually inside its K-steps and only needs one issue slot per 32 cycles:
        // give this wave priority so that it is not queued behind MFMAs it does not depend on
        __builtin_amdgcn_s_setprio(3);
        commit(par, nch);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();

        // taps 0-7 of this chunk, two per K-step: lanes 0-31 tap 2j, lanes 32-63 tap 2j+1
#pragma unroll
        for (int j = 0; j < (KK - 1) / 2; ++j) {
            const int t0 = 2 * j, t1 = 2 * j + 1;
            const int offA0 = (t0 / KS) * HTW + t0 % KS, offA1 = (t1 / KS) * HTW + t1 % KS;
            kstep(par * XBUF + (half ? offA1 : offA0), half ? t1 : t0, false);
            // the next chunk's activations (loaded during this phase's commit) are split in the shadow of the MFMAs
            if (KS == 3 && j >= 1 && j - 1 < XITER) split_unit(j - 1);
        }
        if (KS != 3) {
#pragma unroll
            for (int it = 0; it < XITER; ++it) split_unit(it);
        }
        // tap 8 is contracted for two chunks at once (the even chunk's image is still resident in the other buffer), so 16
        // channels cost 9 K-steps, not 10; a trailing unpaired chunk (odd chunk count) pairs its tap 8 with a zero A
        if (par == 1)
            kstep((half ? XBUF : 0) + OFF8, half ? KK - 1 : KK, false);
        else if (ch == nchunks - 1)
            kstep(OFF8, KK, half != 0);
        if (ch == nchunks - 1) {  // no barrier: the epilogue touches no LDS, waves drain into it as they finish
            epilogue(ti);
            zero_acc();
        }
        tile = ntile;
        ch = nch;
        ti = nti;
        return !has_next;
    };
    while (true) {
        if (phase(std::integral_constant<int, 0>{})) break;
        if (ch == 0) continue;  // odd chunk count: the tile ended on an even chunk, the next tile starts even again
        if (phase(std::integral_constant<int, 1>{})) break;
    }
}

// planes[p][t'][k/8][n][8] <- split(w) ; mode 0 (forward): k = cin, n = cout, t' = t ;
// mode 1 (input gradient): k = cout, n = cin, t' = kk-1-t (flipped taps)
__global__ __launch_bounds__(256) void split_weights_x6_kernel(const float* __restrict__ w, __bf16* __restrict__ planes,
                                                               long plane_stride, int cin, int cout, int kk, int mode) {
    const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
    const int KC = K / 8;
    const size_t total = (size_t)kk * KC * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const int kc = (int)((i / N) % KC);
        const int tp = (int)(i / ((size_t)N * KC));
        const int t = mode == 0 ? tp : kk - 1 - tp;
        bf16x8 vh, vm, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kc * 8 + e;
            const float v = mode == 0 ? w[((size_t)t * cin + k) * cout + n] : w[((size_t)t * cin + n) * cout + k];
            __bf16 h_, m_, l_;
            split3(v, h_, m_, l_);
            vh[e] = h_; vm[e] = m_; vl[e] = l_;
        }
        *reinterpret_cast<bf16x8*>(planes + i * 8) = vh;
        *reinterpret_cast<bf16x8*>(planes + plane_stride + i * 8) = vm;
        *reinterpret_cast<bf16x8*>(planes + 2 * plane_stride + i * 8) = vl;
    }
}

// Every 3x3 slab of a flat parameter buffer in one launch.  table[j] = {w_off, cin, cout, first_unit}; a unit is one
// 16-byte (8 x bf16) group of one plane; table[nconv][3] = total units.  The planes of slab j start at planes + 3*w_off
// with plane stride 9*cin*cout (same placement the per-slab entry point is called with by the engine).
__global__ __launch_bounds__(256) void split_weights_x6_batch_kernel(const float* __restrict__ flat, __bf16* __restrict__ planes,
                                                                     const long* __restrict__ table, int nconv, int mode) {
    const long total = table[(size_t)nconv * 4 + 3];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nconv - 1;  // last slab whose first_unit <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[(size_t)mid * 4 + 3] <= i) lo = mid; else hi = mid - 1;
        }
        const long w_off = table[(size_t)lo * 4 + 0];
        const int cin = (int)table[(size_t)lo * 4 + 1], cout = (int)table[(size_t)lo * 4 + 2];
        const long u = i - table[(size_t)lo * 4 + 3];
        const float* w = flat + w_off;
        __bf16* pl = planes + 3 * w_off;
        const long plane_stride = 9l * cin * cout;
        const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
        const int KC = K / 8;
        const int n = (int)(u % N);
        const int kc = (int)((u / N) % KC);
        const int tp = (int)(u / ((long)N * KC));
        const int t = mode == 0 ? tp : 8 - tp;
        bf16x8 vh, vm, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kc * 8 + e;
            const float v = mode == 0 ? w[((size_t)t * cin + k) * cout + n] : w[((size_t)t * cin + n) * cout + k];
            __bf16 h_, m_, l_;
            split3(v, h_, m_, l_);
            vh[e] = h_; vm[e] = m_; vl[e] = l_;
        }
        *reinterpret_cast<bf16x8*>(pl + u * 8) = vh;
        *reinterpret_cast<bf16x8*>(pl + plane_stride + u * 8) = vm;
        *reinterpret_cast<bf16x8*>(pl + 2 * plane_stride + u * 8) = vl;
    }
}

int g_x6_wg_per_cu = 2;
}  // namespace

// shared with wgrad_x6.hip: bf16 planes per operand (3 = six products = the fp32-accurate parity arithmetic)
int g_punet_x6_planes = 3;

extern "C" int punet_set_x6_products(int n) {
    PUNET_CHECK_ARG(n == 6 || n == 3 || n == 1, "punet_set_x6_products: 6 (fp32-accurate), 3 (16-bit significand) or 1 (bf16)");
    g_punet_x6_planes = n == 6 ? 3 : (n == 3 ? 2 : 1);
    return PUNET_OK;
}

extern "C" int punet_split_weights_x6(const float* w, void* planes, long plane_stride, int cin, int cout, int ksize, int mode,
                                      void* stream) {
    PUNET_CHECK_ARG(w && planes && cin > 0 && cout > 0 && (ksize == 3 || ksize == 1) && (mode == 0 || mode == 1),
                    "punet_split_weights_x6: bad args");
    PUNET_CHECK_ARG((mode == 0 ? cin : cout) % 8 == 0, "punet_split_weights_x6: the contracted channel count must be a multiple of 8");
    const int kk = ksize * ksize;
    PUNET_CHECK_ARG(plane_stride >= (long)kk * cin * cout, "punet_split_weights_x6: plane_stride too small");
    const size_t total = (size_t)kk * cin * cout / 8;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    split_weights_x6_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w, (__bf16*)planes, plane_stride, cin, cout, kk, mode);
    PUNET_LAUNCH_CHECK("split_weights_x6");
    return PUNET_OK;
}

extern "C" int punet_split_weights_x6_batch(const float* flat, void* planes, const long* table, int nconv, long total_units,
                                            int mode, void* stream) {
    PUNET_CHECK_ARG(flat && planes && table && nconv > 0 && total_units > 0 && (mode == 0 || mode == 1),
                    "punet_split_weights_x6_batch: bad args");
    long grid = (total_units + 255) / 256;
    if (grid > 8192) grid = 8192;
    split_weights_x6_batch_kernel<<<(int)grid, 256, 0, (hipStream_t)stream>>>(flat, (__bf16*)planes, table, nconv, mode);
    PUNET_LAUNCH_CHECK("split_weights_x6_batch");
    return PUNET_OK;
}

extern "C" size_t punet_gate_dwords(int B, int H, int W, int C) {
    return (size_t)B * H * ((W + TW - 1) / TW) * ((C + TN - 1) / TN) * 64;
}

extern "C" int punet_conv3x3_x6(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                                int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out,
                                const float* mask, int accumulate, int relu, int B, int H, int W, void* stream) {
    return punet_conv3x3_x6_gated(x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask,
                                  nullptr, nullptr, accumulate, relu, B, H, W, stream);
}

static int conv_x6_launch(int ksize, const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                          int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                          const void* gate_bits, void* gate_out, int accumulate, int relu, int B, int H, int W, void* stream,
                          float* pool_out = nullptr);

extern "C" int punet_conv3x3_x6_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes,
                                      long plane_stride, int n_row, int n_off, const float* bias, int bias_img_stride,
                                      float* y, int n_out, const float* mask, const void* gate_bits, void* gate_out,
                                      int accumulate, int relu, int B, int H, int W, void* stream) {
    return conv_x6_launch(3, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                          gate_out, accumulate, relu, B, H, W, stream);
}

extern "C" int punet_conv3x3_x6_pooled(const float* x0, int c0, const float* x1, int c1, const void* wplanes,
                                       long plane_stride, int n_row, int n_off, const float* bias, int bias_img_stride,
                                       float* y, int n_out, void* gate_out, float* pool_out, int relu, int B, int H, int W,
                                       void* stream) {
    PUNET_CHECK_ARG(pool_out, "punet_conv3x3_x6_pooled: null pool_out");
    return conv_x6_launch(3, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, nullptr, nullptr,
                          gate_out, 0, relu, B, H, W, stream, pool_out);
}

extern "C" int punet_conv1x1_x6_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes,
                                      long plane_stride, int n_row, int n_off, const float* bias, int bias_img_stride,
                                      float* y, int n_out, const float* mask, const void* gate_bits, void* gate_out,
                                      int accumulate, int relu, int B, int H, int W, void* stream) {
    return conv_x6_launch(1, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                          gate_out, accumulate, relu, B, H, W, stream);
}

static int conv_x6_launch(int ksize, const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                          int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                          const void* gate_bits, void* gate_out, int accumulate, int relu, int B, int H, int W, void* stream,
                          float* pool_out) {
    PUNET_CHECK_ARG(x0 && wplanes && y, "punet_conv3x3_x6: null pointer");
    PUNET_CHECK_ARG(!pool_out || (H % 2 == 0 && W % 2 == 0 && !mask && !gate_bits && !accumulate),
                    "punet_conv3x3_x6_pooled: the fused 2x2 average needs even H and W (%dx%d) and the forward epilogue", H, W);
    PUNET_CHECK_ARG(c0 > 0 && c0 % CK == 0 && c1 >= 0 && c1 % CK == 0 && (c1 == 0 || x1),
                    "punet_conv3x3_x6: source channels must be multiples of %d (c0=%d c1=%d)", CK, c0, c1);
    PUNET_CHECK_ARG(n_out > 0 && n_out % 4 == 0 && n_off >= 0 && n_off + n_out <= n_row,
                    "punet_conv3x3_x6: bad output-channel window (n_out must be a multiple of 4)");
    PUNET_CHECK_ARG(B > 0 && H > 0 && W > 0, "punet_conv3x3_x6: bad shape");
    {   // the kernel does its index arithmetic in 32 bits
        const unsigned long long px = (unsigned long long)B * H * W;
        const int cmax = c0 > c1 ? (c0 > n_out ? c0 : n_out) : (c1 > n_out ? c1 : n_out);
        PUNET_CHECK_ARG(px * cmax < (1ull << 31) && 3ull * (unsigned long long)plane_stride < (1ull << 31),
                        "punet_conv3x3_x6: tensor too large for 32-bit indexing (%llu pixels x %d channels)", px, cmax);
    }
    ConvX6Args a{};
    a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1;
    a.wp = (const __bf16*)wplanes; a.plane_stride = plane_stride; a.n_row = n_row; a.n_off = n_off;
    a.bias = bias; a.bias_img_stride = bias_img_stride;
    PUNET_CHECK_ARG(!(mask && gate_bits), "punet_conv3x3_x6: give the ReLU gate either as fp32 mask or as bits, not both");
    PUNET_CHECK_ARG(!gate_out || relu, "punet_conv3x3_x6: gate_out is the ReLU gate of the output and needs relu != 0");
    a.y = y; a.n_total = n_out; a.mask = mask; a.accumulate = accumulate; a.relu = relu;
    a.gate_bits = (const unsigned*)gate_bits; a.gate_out = (unsigned*)gate_out; a.pool_out = pool_out;
    a.B = B; a.H = H; a.W = W;
    a.ntiles = B * ceil_div(H, TH) * ceil_div(W, TW) * ceil_div(n_out, TN);
    const int cap = g_x6_wg_per_cu * NUM_CU;
    const int grid = a.ntiles < cap ? a.ntiles : cap;
    // the input-gradient epilogue whenever one of its features is asked for; bias / ReLU / gate_out belong to the forward
    const bool dgrad = mask || gate_bits || accumulate;
    PUNET_CHECK_ARG(!(dgrad && (bias || relu || gate_out)),
                    "punet_conv3x3_x6: bias/relu/gate_out (forward) cannot be combined with mask/gate_bits/accumulate (input gradient)");
    hipStream_t st = (hipStream_t)stream;
    a.xcd_remap = grid % 8 == 0;  // same-process A/B: +0.5-0.8 % (the tensors of this workload fit the Infinity Cache)
    if (ksize == 1) {  // the two Fcomb layers: always the parity arithmetic (they are a rounding error of the step time)
        if (dgrad) conv3x3_x6_kernel<3, true, 1><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<3, false, 1><<<grid, 256, 0, st>>>(a);
    } else if (g_punet_x6_planes == 3) {
        if (dgrad) conv3x3_x6_kernel<3, true><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<3, false><<<grid, 256, 0, st>>>(a);
    } else if (g_punet_x6_planes == 2) {
        if (dgrad) conv3x3_x6_kernel<2, true><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<2, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (dgrad) conv3x3_x6_kernel<1, true><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<1, false><<<grid, 256, 0, st>>>(a);
    }
    PUNET_LAUNCH_CHECK("conv3x3_x6");
    return PUNET_OK;
}
