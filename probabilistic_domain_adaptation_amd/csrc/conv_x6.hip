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
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "punet_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32, TN = 64, CK = 8;
constexpr int WG_PER_CU = 2;  // persistent grid = one residency round (two workgroups per CU fit LDS and registers)

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
    // fp16-operand instantiations (F16; common.h): amax slots of the two sources and of the weight slab (device floats; nullptr =
    // scale 1) and the slot the output's amax is published to (nullable)
    const float* amax_x0;
    const float* amax_x1;
    const float* amax_w;
    float* amax_out;
    unsigned m_ntn, m_tx, m_ty;  // floor(2^32 / d) + 1 for d = channel blocks, tiles per row, tile rows: n / d = mulhi(n, m)
                                 // exactly while n * d < 2^32 (host check) - the tile decode without a division
    // Stream-K tail (conv3x3_x6b_kernel only; sk_plan below).  Work ITEMS [0, sk_whole) are whole tiles, dealt round-robin.  The
    // remaining ntiles - sk_whole tiles (fewer than one round) are a contiguous range of sk_Q = (ntiles - sk_whole) * K/32 QUADS
    // (4 chunks = 32 contracted channels, one cycle of the kernel's four phases) dealt in runs of sk_D quads to sk_P WORKERS
    // (workgroup v < sk_P takes quads [v * sk_D, (v + 1) * sk_D) after its whole tiles): a worker's run covers parts of at most two
    // tiles = tail items sk_whole + 2 v and sk_whole + 2 v + 1.  A partial item's accumulators go to slot (item - sk_whole) of
    // sk_ws (64 KB each); the workgroup whose arrival at sk_cnt[tile - sk_whole] completes the tile sums its slots in worker order
    // and runs the tile's epilogue (and leaves the counter at zero again).  Off: sk_P = 0, sk_whole = ntiles.
    float* sk_ws;
    unsigned* sk_cnt;
    int sk_whole, sk_P, sk_D, sk_Q;
    unsigned m_D, m_nq;
    // FOLD instantiations of conv3x3_x6b_kernel (round 5): tiles walk the VIRTUAL image row of B * W pixels (vtiles_x = B * W / 32
    // tiles, m_ty = its reciprocal, m_W = the reciprocal of W for "which image")
    int vtiles_x;
    unsigned m_W;
};

static inline unsigned x6_magic(int d) { return (unsigned)((1ull << 32) / (unsigned)d + 1ull); }

struct TileInfo {
    int b, ty0, tx0, n0;
    int bpos;  // FOLD: first tile column that belongs to image b + 1 (1..31), 32 = the tile lies inside image b
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
// IO16 (one-plane forward only; the opt-in "bf16s" fast mode): the activation tensors x0, x1, y and pool_out hold bf16 instead
// of fp32 - the staging loads are half the bytes and need no conversion, the epilogue stores 8 instead of 16 bytes per quad.
// F16: fp16 instead of bf16 planes (NP <= 2), operands scaled by exact powers of two from their amax slots (common.h), the
// epilogue rescales and publishes the output's amax.  IO16 && F16: the activation tensors hold fp16 (UNscaled, as under the
// reference's fp16 autocast; outputs are clamped to fp16's range).
template <int NP, bool DGRAD, int KS = 3, bool IO16 = false, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_x6_kernel(ConvX6Args a) {
    static_assert(!IO16 || (NP == 1 && !DGRAD), "16-bit activation storage exists for the one-plane forward kernel only");
    static_assert(!F16 || NP <= 2, "fp16 operands: one or two planes");
    typedef typename std::conditional<F16, _Float16, __bf16>::type S16;  // element type of 16-bit activation storage
    typedef S16 s16x4 __attribute__((ext_vector_type(4)));
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntn = (a.n_total + TN - 1) / TN;
    const int N = a.n_total;
    const int K = a.c0 + a.c1;
    const int nchunks = K / CK;
    float sx = 1.f, inv = 1.f, amax_run = 0.f;  // F16: operand scale, its inverse incl. the weights', running |output| maximum
    if constexpr (F16) {
        if (a.amax_x0 || a.amax_x1) sx = punet_f16_scale_of(fmaxf(a.amax_x0 ? *a.amax_x0 : 0.f, a.amax_x1 ? *a.amax_x1 : 0.f));
        inv = 1.f / (sx * punet_f16_scale_from(a.amax_w));
    }

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
    uint2 rx16[XITER];  // IO16: four bf16 of one unit, already in LDS format
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
    int xpix[XITER];       // clamped y * W + clamped x of the unit's pixel inside its image
    int xb = 0;            // image of the tile the staging registers are being loaded for (block-uniform)
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
            xpix[it] = min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1);  // inside image xb (32-bit); the image
        }                                                                          // base is a uniform 64-bit term
        xb = ti.b;
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
        const int cs = kbase < a.c0 ? a.c0 : a.c1, coff = kbase < a.c0 ? kbase : kbase - a.c0;
        // uniform 64-bit image base + 32-bit offset inside the image: B * H * W * C may exceed 2^31 elements, H * W * C not
        const float* src = (kbase < a.c0 ? a.x0 : a.x1) + (size_t)xb * ((size_t)(a.H * a.W) * cs);
        const int q = min(tid + it * 256, XUNITS - 1) % (CK / 4);  // the same clamped unit index as in retarget / commit
        // raw value: the out-of-image zeroing happens where the register is consumed (split_unit) - a select right here
        // would make the wave wait for the load it has just issued
        if constexpr (IO16) {
            const S16* s16 = reinterpret_cast<const S16*>(kbase < a.c0 ? a.x0 : a.x1) + (size_t)xb * ((size_t)(a.H * a.W) * cs);
            rx16[it] = *reinterpret_cast<const uint2*>(s16 + (unsigned)(xpix[it] * cs + coff + q * 4));
            return make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            return *reinterpret_cast<const float4*>(src + (unsigned)(xpix[it] * cs + coff + q * 4));
        }
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
        if constexpr (IO16) {  // already bf16: only the zero padding remains
            const uint2 z = xin[it] ? rx16[it] : make_uint2(0u, 0u);
            px[it][0] = __builtin_bit_cast(bf16x4, z);
            return;
        }
        const float4 r = xin[it] ? rx[it] : make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding outside the image
        uint2 ph, pm, pl = make_uint2(0u, 0u);
        if constexpr (F16) splitf16_quad<NP>(r, sx, ph, pm); else split3_quad<NP>(r, ph, pm, pl);
        px[it][0] = __builtin_bit_cast(bf16x4, ph);
        if (NP > 1) px[it][NP > 1 ? 1 : 0] = __builtin_bit_cast(bf16x4, pm);
        if (NP > 2) px[it][NP > 2 ? 2 : 0] = __builtin_bit_cast(bf16x4, pl);
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
        // uniform 64-bit base of image ti.b in the output-shaped tensors; offsets inside the image stay 32-bit
        const size_t img = (size_t)ti.b * ((size_t)(a.H * a.W) * N);
        // (IO16: y / pool_out hold bf16 - the element offsets below are halved in bytes by addressing through __bf16*)
        float* const yb = IO16 ? reinterpret_cast<float*>(reinterpret_cast<S16*>(a.y) + img) : a.y + img;
        const float* const mb = a.mask ? a.mask + img : nullptr;
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
            const unsigned pbase0 = (unsigned)((gy0 * a.W + gx) * N + ti.n0 + 4 * half);
            const unsigned pbase1 = pbase0 + (unsigned)(a.W * N);
            const unsigned qbase = (unsigned)(((gy0 >> 1) * (a.W >> 1) + (gx >> 1)) * N + ti.n0 + 4 * half);
            float* const qb = !a.pool_out ? nullptr  // the pooled image has a quarter of the pixels
                              : IO16 ? reinterpret_cast<float*>(reinterpret_cast<S16*>(a.pool_out) + (img >> 2)) : a.pool_out + (img >> 2);
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
                    if constexpr (F16) {  // exact: inv is a power of two
                        v0.x *= inv; v0.y *= inv; v0.z *= inv; v0.w *= inv;
                        v1.x *= inv; v1.y *= inv; v1.z *= inv; v1.w *= inv;
                    }
                    v0.x += bq.x; v0.y += bq.y; v0.z += bq.z; v0.w += bq.w;
                    v1.x += bq.x; v1.y += bq.y; v1.z += bq.z; v1.w += bq.w;
                    if (a.relu) {
                        v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                        v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                    }
                    auto store4 = [](float* base, unsigned off, const float4& v) __attribute__((always_inline)) {
                        if constexpr (IO16) {
                            s16x4 o;
                            if constexpr (F16) {  // fp16 storage: saturate instead of producing inf (65504 = fp16's largest finite value)
                                o[0] = (S16)__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f); o[1] = (S16)__builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f);
                                o[2] = (S16)__builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f); o[3] = (S16)__builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f);
                            } else {
                                o[0] = (S16)v.x; o[1] = (S16)v.y; o[2] = (S16)v.z; o[3] = (S16)v.w;
                            }
                            *reinterpret_cast<s16x4*>(reinterpret_cast<S16*>(base) + off) = o;
                        } else {
                            *reinterpret_cast<float4*>(base + off) = v;
                        }
                    };
                    if constexpr (F16) {
                        if (in0 && okc) amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(v0.x), fabsf(v0.y))), fmaxf(fabsf(v0.z), fabsf(v0.w)));
                        if (in1 && okc) amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(v1.x), fabsf(v1.y))), fmaxf(fabsf(v1.z), fabsf(v1.w)));
                    }
                    if (in0 && okc) {
                        store4(yb, pbase0 + ch, v0);
                        gout0 |= ((v0.x > 0.f ? 1u : 0u) | (v0.y > 0.f ? 2u : 0u) | (v0.z > 0.f ? 4u : 0u) | (v0.w > 0.f ? 8u : 0u))
                                 << (16 * nb + 4 * rg);
                    }
                    if (in1 && okc) {
                        store4(yb, pbase1 + ch, v1);
                        gout1 |= ((v1.x > 0.f ? 1u : 0u) | (v1.y > 0.f ? 2u : 0u) | (v1.z > 0.f ? 4u : 0u) | (v1.w > 0.f ? 8u : 0u))
                                 << (16 * nb + 4 * rg);
                    }
                    if (a.pool_out) {  // uniform
                        float4 q;
                        q.x = (((v0.x + partner(v0.x)) + v1.x) + partner(v1.x)) * 0.25f;
                        q.y = (((v0.y + partner(v0.y)) + v1.y) + partner(v1.y)) * 0.25f;
                        q.z = (((v0.z + partner(v0.z)) + v1.z) + partner(v1.z)) * 0.25f;
                        q.w = (((v0.w + partner(v0.w)) + v1.w) + partner(v1.w)) * 0.25f;
                        if (!(lane & 1) && in0 && okc) store4(qb, qbase + ch, q);
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
            const unsigned pbase = (unsigned)((gy * a.W + gx) * N + ti.n0 + 4 * half);
            unsigned gin = 0u, gout = 0u;
            if (DGRAD && a.gate_bits && pix_in) gin = a.gate_bits[gd];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ch = nb * 32 + rg * 8;  // + 4 * half + e
                    const bool ok = pix_in && ti.n0 + ch + 4 * half < N;  // N is a multiple of 4
                    float4 v = make_float4(acc[m][nb][4 * rg + 0], acc[m][nb][4 * rg + 1], acc[m][nb][4 * rg + 2], acc[m][nb][4 * rg + 3]);
                    if constexpr (F16) { v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv; }
                    if (!DGRAD) {
                        v.x += bq4[nb][rg].x; v.y += bq4[nb][rg].y; v.z += bq4[nb][rg].z; v.w += bq4[nb][rg].w;
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    } else if (a.gate_bits) {
                        const unsigned g4 = gin >> (16 * nb + 4 * rg);
                        v.x = g4 & 1u ? v.x : 0.f; v.y = g4 & 2u ? v.y : 0.f; v.z = g4 & 4u ? v.z : 0.f; v.w = g4 & 8u ? v.w : 0.f;
                    } else if (a.mask) {
                        if (ok) {
                            const float4 g = *reinterpret_cast<const float4*>(mb + pbase + ch);
                            v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f;
                            v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
                        }
                    }
                    if (ok) {
                        if (DGRAD && a.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(yb + pbase + ch);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(yb + pbase + ch) = v;
                        if constexpr (F16) amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
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
                    c = punet_mfma32<F16>(bv[P1][nb].u, av[P1][m].u, c);  // mm
                    c = punet_mfma32<F16>(bv[0][nb].u, av[P2][m].u, c);   // lh
                    c = punet_mfma32<F16>(bv[P2][nb].u, av[0][m].u, c);   // hl
                }
                if (NP > 1) {
                    c = punet_mfma32<F16>(bv[0][nb].u, av[P1][m].u, c);  // mh
                    c = punet_mfma32<F16>(bv[P1][nb].u, av[0][m].u, c);  // hm
                }
                c = punet_mfma32<F16>(bv[0][nb].u, av[0][m].u, c);  // hh
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
    if constexpr (F16) {
        if (a.amax_out) {  // block-uniform: the workgroup leaves the tile loop together
            __syncthreads();  // the slower waves' last fragment reads are done: the operand images are free
            punet_amax_publish_block256(a.amax_out, amax_run, reinterpret_cast<float*>(smem));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same convolution through v_mfma_f32_16x16x32_bf16 (input channels a multiple of 32).  The 16 x 16 x 32 shape moves
// half the accumulator data per MAC of 32 x 32 x 16 and the matrix cores are power-limited: the same
// six-product chain sustains 2.07 PF against 1.80 PF in a register-only loop, +10 % in this kernel with unchanged reads.
// One MFMA contracts FOUR {tap, 8 channels} blocks (lane group g = lane / 16 takes block g).  A chunk of 8 channels has
// 9 blocks, so 4 chunks = 36 blocks = 9 K-steps; chunks are still staged one at a time, and the blocks a chunk leaves
// over are contracted together with the next chunk's (the previous chunk's image is the other image buffer anyway):
//   phase 0: {t0 .. t7}                          leaves t8
//   phase 1: {t8', t0 .. t6}                     leaves t7 t8         (' = of the previous chunk)
//   phase 2: {t7' t8', t0 .. t5}                 leaves t6 t7 t8
//   phase 3: {t6' t7' t8', t0 .. t8}
// Left-over weight rows stay where they are: the slab has 12 slots and a chunk's rows go to the slots its predecessor's
// left-overs do not occupy (x6b_slot).  A wave's 64 pixel x 64 channel tile is 4 x 4 sub-tiles; to stay inside the register
// budget the weight planes are staged by LDS-DMA (no staging registers); then the weight fragments of all four channel
// blocks stay resident during a K-step and the pixel fragments are streamed with one block of lookahead.
// Output mapping: lane (l16 = lane % 16, g) holds pixel l16 of a 16-pixel block and channels 16 cb + 4 g .. + 3 - again a
// 16-byte store per sub-tile straight from the accumulators.  Gate bits keep the punet_gate_dwords layout: a dword's 32
// bits are held half by lane (l16, g) and half by lane (l16, g ^ 2) = lane ^ 32 (v_permlane32_swap joins them).
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int x6b_slot(int ph, int t) {
    return ph == 0 ? t : ph == 1 ? (t < 8 ? t : 9) : ph == 2 ? (t < 7 ? t : t == 7 ? 8 : 10) : (t < 6 ? t : 2 * t - 5);
}

#ifndef X6B_TIMING  // timing-only builds (WRONG results; tools/ab_variants.sh + tools/ab_layers_detail.sh): 1 = no output
#define X6B_TIMING 0  // stores, 2 = no activation loads (synthetic values), 4 = no epilogue, 8 = loads from a cache-resident window
#endif
#ifndef X6B_GENERAL_EPILOGUE  // 1 = only the general epilogue (every flag tested at run time, per store), for A/B runs
#define X6B_GENERAL_EPILOGUE 0
#endif
#ifdef X6B_TRACE  // s_memtime stamps of one workgroup's chunks 8..11 (tools/trace_conv.sh): per wave, per chunk:
__device__ unsigned long long punet_x6b_trace[4 * 16];  // top barrier passed, operand barrier passed, K-steps issued
#define X6BT(id)                                                                                             \
    do {                                                                                                     \
        if (blockIdx.x == 9 && (threadIdx.x & 63) == 0 && trace_chunk >= 8 && trace_chunk < 12)              \
            punet_x6b_trace[(threadIdx.x >> 6) * 16 + (trace_chunk - 8) * 3 + (id)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define X6BT(id)
#endif
#ifndef X6B_PIN_SPLIT
#define X6B_PIN_SPLIT 1
#endif
#ifndef X6B_PLANE_IN
#define X6B_PLANE_IN 0
#endif
#ifndef X6B_VECTOR_WAVE  // 1 = the wave index as a VGPR value (the form before round 2's scalar wave index), for A/B runs
#define X6B_VECTOR_WAVE 0
#endif
// F16: fp16 planes (NP = 2: h + l, products hh + hl + lh; NP = 1: h) with power-of-two scaling from the operands' amax slots
// (common.h) instead of bf16 planes; the epilogue rescales by the exact inverse and publishes the output's amax.
// FOLD (round 5; tile WIDTH quantization): a 32-pixel tile column does not fit image widths such as 88 or 176 (2.75 and 5.5 tiles:
// 8.3 % of those levels' MFMAs multiplied padding).  The FOLD instantiation lays the tiles over the VIRTUAL row of all B images side by
// side (B * W pixels: 352 = 11 tiles, 704 = 22), so a tile may hold the last columns of image b and the first ones of image b + 1.
// The LDS halo image of such a tile gets TWO extra columns where the images meet - the right halo of b and the left halo of b + 1,
// which the staging fills with zeros by its ordinary out-of-image test - so every tap still sits at a fixed offset from its pixel;
// lanes right of the seam read 2 columns further (one per-lane add per K-step and 16-pixel half), the epilogue maps a lane's
// tile column back to (image, x).  Per pixel the contraction runs in the same order as in the unfolded kernel: results are
// bit-identical (tests/test_gpu_fold.py).  Chosen by the host when W % 32 != 0, W >= 32 and the virtual row needs fewer tiles than
// the B images do (the row may end inside its last tile: what lies beyond is zero on the way in and not stored on the way out).
template <int NP, bool DGRAD, bool F16 = false, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void conv3x3_x6b_kernel(ConvX6Args a) {
    static_assert(!F16 || NP <= 2, "fp16 operands: one or two planes");
    constexpr int KS = 3, KK = 9, PAD = 1;
    constexpr int HTH = TH + KS - 1, HTW = TW + KS - 1 + (FOLD ? 2 : 0);
    constexpr int NPIX = HTH * HTW;
    // LDS image: rows 48 pixels apart.  ds_read_b128 is served in groups that mix lanes of lane groups (0, 1) and (2, 3);
    // two lane groups of one instruction are conflict-free iff their pixel offsets agree mod 16 (256 bytes = all 64 banks).
    // With a pitch of 48 the taps of one column kx are 0 mod 16 apart, and so are the two image buffers.
    constexpr int HTWP = 48;
    constexpr int NPIXA = (HTH - 1) * HTWP + HTW;  // 466 units per plane
#if X6B_PLANE_IN  // TIMING-ONLY experiment (wrong results): activations arrive as NP stored 16-bit planes (plane-major, NHWC, x0 / x1
                  // point at plane 0, plane p is B*H*W*c elements further): a staging unit = one {pixel, plane} = 16 bytes, no split
    constexpr int XUNITS = NPIX * NP;
#else
    constexpr int XUNITS = NPIX * (CK / 4);
#endif
    constexpr int XITER = (XUNITS + 255) / 256;
    constexpr int WSLAB = 12 * TN;   // per plane: 12 slots of one {tap, 8 channels} row (64 output channels x 16 bytes)
    constexpr int XBUF = (NP * NPIXA + 15) / 16 * 16;  // one activation image (NP planes); two of them, by chunk parity
    constexpr int OPERANDS = 2 * XBUF + NP * WSLAB;

    __shared__ uint4 smem[OPERANDS];  // 81,920 B at NP = 3: exactly two workgroups per CU
    uint4* sX = smem;
    uint4* sW = smem + 2 * XBUF;

    const int tid = threadIdx.x;
    // the wave index as a SCALAR: everything the weight DMA derives from it (plane, tap, slot, row address) then lives in
    // SGPRs and SALU instructions instead of ~30 loop-invariant VGPRs and 64-bit VALU multiplies per row
#if X6B_VECTOR_WAVE
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    __builtin_assume(wave >= 0 && wave < 4);
    const int lane = tid & 63, l16 = lane & 15, g = lane >> 4;

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntn = (a.n_total + TN - 1) / TN;
    const int N = a.n_total;
    const int K = a.c0 + a.c1;
    const int nchunks = K / CK;  // a multiple of 4 (host check)
    // F16: operand scales (exact powers of two; block-uniform scalars) and the running |output| maximum of this thread
    float sx = 1.f, inv = 1.f, amax_run = 0.f;
    if constexpr (F16) {
        if (a.amax_x0 || a.amax_x1) sx = punet_f16_scale_of(fmaxf(a.amax_x0 ? *a.amax_x0 : 0.f, a.amax_x1 ? *a.amax_x1 : 0.f));
        inv = 1.f / (sx * punet_f16_scale_from(a.amax_w));
    }

    // block-uniform and scalar: tile -> (channel block, tile column, tile row, image) by multiplications with the host's
    // reciprocals (s_mul_hi_u32) - integer division has no scalar instruction and compiles to ~30 vector instructions each
    auto decode = [&](int tile) {
        TileInfo t;
#if X6B_VECTOR_WAVE
        t.n0 = (tile % ntn) * TN;
        int sid = tile / ntn;
        t.tx0 = (sid % tiles_x) * TW;
        sid /= tiles_x;
        t.ty0 = (sid % tiles_y) * TH;
        t.b = sid / tiles_y;
#else
        const unsigned ut = (unsigned)__builtin_amdgcn_readfirstlane(tile);
        const unsigned s1 = ntn == 1 ? ut : __umulhi(ut, a.m_ntn);  // (the reciprocal of 1 does not fit 32 bits)
        t.n0 = (int)(ut - s1 * (unsigned)ntn) * TN;
        if constexpr (FOLD) {
            const unsigned s2 = a.vtiles_x == 1 ? s1 : __umulhi(s1, a.m_tx);  // (m_tx = the reciprocal of vtiles_x here)
            const unsigned vx0 = (s1 - s2 * (unsigned)a.vtiles_x) * TW;        // virtual column of the tile's first pixel
            t.ty0 = (int)s2 * TH;
            t.b = (int)__umulhi(vx0, a.m_W);                                    // vx0 / W  (W >= 32 > 1)
            t.tx0 = (int)(vx0 - (unsigned)t.b * (unsigned)a.W);
            t.bpos = min(a.W - t.tx0, TW);
        } else {
            const unsigned s2 = tiles_x == 1 ? s1 : __umulhi(s1, a.m_tx);
            const unsigned s3 = tiles_y == 1 ? s2 : __umulhi(s2, a.m_ty);
            t.tx0 = (int)(s1 - s2 * (unsigned)tiles_x) * TW;
            t.ty0 = (int)(s2 - s3 * (unsigned)tiles_y) * TH;
            t.b = (int)s3;
            t.bpos = TW;
        }
#endif
        return t;
    };

    f32x4 acc[4][4];  // [pixel block pb = 2 * row + half][channel block cb]
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // staging: identical to conv3x3_x6_kernel (see the comments there)
    // Activations go through registers (they are split on the way); the weight planes are plain copies and go global -> LDS
    // by LDS-DMA (global_load_lds_dwordx4: one wave-instruction = one {plane, tap} row of 64 output channels x 16 bytes),
    // issued at the head of the commit of the chunk that needs them - no staging registers (the register-staged version
    // kept 7 x 16 bytes + 7 offsets per thread across the K-steps, which is what the fragments of this kernel need).
    float4 rx[XITER];
    int xpix[XITER];
    const float* xbase0 = a.x0;  // image of the tile the staging registers are being loaded for, per source (block-uniform)
    const float* xbase1 = a.x1;
    bool xin[XITER];
    unsigned wlane = 0;  // this lane's output-channel offset (elements) inside a weight row
    auto retarget = [&](const TileInfo& ti) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int u = min(tid + it * 256, XUNITS - 1);
#if X6B_PLANE_IN
            const int pix = u % NPIX;
#else
            const int pix = u / (CK / 4);
#endif
            const int py = pix / HTW, px = pix % HTW;
            const int gy = ti.ty0 + py - PAD;
            int gx = ti.tx0 + px - PAD, ioff = 0;
            if constexpr (FOLD) {
                // halo columns 0 .. bpos + 1 belong to image b (column bpos + 1 = its right halo: x = W, zero), bpos + 2 .. to image
                // b + 1 (column bpos + 2 = its left halo: x = -1, zero); without a seam (bpos = 32) columns 34, 35 are never read
                const bool second = px > ti.bpos + 1;
                gx = second ? px - 3 - ti.bpos : gx;
                ioff = second && ti.b + 1 < a.B ? a.H * a.W : 0;  // (never beyond the last image: the virtual row may end inside a tile)
                gx = second && ti.b + 1 >= a.B ? -1 : gx;         // ... and what lies beyond it is zero
            }
            xin[it] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            xpix[it] = min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1) + ioff;  // inside image xb (32-bit); the image
        }                                                                                 // base is a uniform 64-bit term
#if X6B_PLANE_IN == 2
        xbase0 = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.x0) + (size_t)ti.b * ((size_t)(a.H * a.W) * 8));
        xbase1 = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.x1) + (size_t)ti.b * ((size_t)(a.H * a.W) * 8));
#elif X6B_PLANE_IN
        xbase0 = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.x0) + (size_t)ti.b * ((size_t)(a.H * a.W) * a.c0));
        xbase1 = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.x1) + (size_t)ti.b * ((size_t)(a.H * a.W) * a.c1));
#else
        xbase0 = a.x0 + (size_t)ti.b * ((size_t)(a.H * a.W) * a.c0);
        xbase1 = a.x1 + (size_t)ti.b * ((size_t)(a.H * a.W) * a.c1);
#endif
    };
    // NOT part of retarget (round 4 bug fix): the weight rows of a tile's LAST chunk are DMA'd in the commit that FOLLOWS the
    // re-targeting of the staging registers at the next tile, and that tile has another output-channel block whenever the grid is not
    // a multiple of the channel-block count (cout = 192: three blocks, 512 workgroups) - the last chunk then contracted the next
    // tile's weight columns (tests/test_gpu_x6.py, the 4 x 128 x 128, 64 -> 192 case).  wlane follows the TILE, at the end of a phase.
    auto wlane_of = [&](const TileInfo& ti) { return (unsigned)((a.n_off + min(ti.n0 + lane, N - 1)) * 8); };  // columns beyond N copy column N - 1 (never stored)
    auto load_x = [&](int ch, int it) __attribute__((always_inline)) -> float4 {
        const int kbase = ch * CK;
        const int cs = kbase < a.c0 ? a.c0 : a.c1, coff = kbase < a.c0 ? kbase : kbase - a.c0;
        const float* src = kbase < a.c0 ? xbase0 : xbase1;  // uniform image base (SGPR pair) + 32-bit lane offset
#if X6B_PLANE_IN
        {
            const int pl = min(tid + it * 256, XUNITS - 1) / NPIX;
            const __bf16* sp = reinterpret_cast<const __bf16*>(src) + (size_t)pl * ((size_t)a.B * a.H * a.W * cs);
#if X6B_PLANE_IN == 2  // chunk-major planes [plane][c / 8][B][H][W][8]: a tile row of one chunk and plane is contiguous
            return *reinterpret_cast<const float4*>(sp + ((size_t)(coff >> 3) * ((size_t)a.B * a.H * a.W) + (unsigned)xpix[it]) * 8);
#else
            return *reinterpret_cast<const float4*>(sp + (unsigned)(xpix[it] * cs + coff));  // NHWC planes: 8 channels of one plane
#endif
        }
#endif
        const int q = min(tid + it * 256, XUNITS - 1) % (CK / 4);
#if X6B_TIMING & 2
        return make_float4((float)(xpix[it] & 7), 1.f, (float)q, (float)(coff & 3));
#endif
#if X6B_TIMING & 8
        return *reinterpret_cast<const float4*>(src + (unsigned)((xpix[it] & 1023) * cs + coff + q * 4));
#endif
        return *reinterpret_cast<const float4*>(src + (unsigned)(xpix[it] * cs + coff + q * 4));
    };
    auto prefetch = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XITER; ++it) rx[it] = load_x(ch, it);
    };
    bf16x4 px[XITER][NP];
    auto split_unit = [&](int it) __attribute__((always_inline)) {
#if X6B_PLANE_IN
        return;
#endif
        uint2 ph_, pm, pl = make_uint2(0u, 0u);
        if constexpr (F16) {
            // outside the image the SCALE is zero (one select instead of four on the values: the clamped address holds a finite
            // activation, its planes come out as signed zeros)
            splitf16_quad<NP>(rx[it], xin[it] ? sx : 0.f, ph_, pm);
        } else {
            const float4 r = xin[it] ? rx[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            split3_quad<NP>(r, ph_, pm, pl);
        }
#if X6B_PIN_SPLIT
        // the planes are only needed by the NEXT commit: without a use here the optimiser sinks the whole split behind the
        // chunk's last MFMA (a ~75-instruction vector blob in front of the barrier) instead of leaving it in their shadow
        asm volatile("" : "+v"(ph_.x), "+v"(ph_.y), "+v"(pm.x), "+v"(pm.y), "+v"(pl.x), "+v"(pl.y));
#endif
        px[it][0] = __builtin_bit_cast(bf16x4, ph_);
        if (NP > 1) px[it][NP > 1 ? 1 : 0] = __builtin_bit_cast(bf16x4, pm);
        if (NP > 2) px[it][NP > 2 ? 2 : 0] = __builtin_bit_cast(bf16x4, pl);
    };
    // chunk `ch`: weight rows -> slots x6b_slot(ph, tap) by LDS-DMA (first, so that a counted vmcnt can leave the younger
    // activation loads in flight), packed activation planes -> LDS image (ph & 1), staging registers re-loaded for chunk nch
    auto commit = [&](auto ph_c, int cch, int nch) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_c)::value;
        constexpr int WROWS = NP * KK;
        // 1. packed activation planes -> LDS image (ph & 1): before any DMA is pending (the compiler orders a ds_write
        //    behind a pending LDS-DMA with vmcnt(0))
        uint4* sXc = sX + (ph & 1) * XBUF;
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int u = min(tid + it * 256, XUNITS - 1);
#if X6B_PLANE_IN
            {
                const int pl = u / NPIX, pix = u % NPIX;
                const int lpix = (pix / HTW) * HTWP + pix % HTW;
                const float4 r = xin[it] ? rx[it] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (tid + it * 256 < XUNITS) *reinterpret_cast<float4*>(&sXc[pl * NPIXA + lpix]) = r;
                continue;
            }
#endif
            const int q = u % (CK / 4), pix = u / (CK / 4);
            const int lpix = (pix / HTW) * HTWP + pix % HTW;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(&sXc[p * NPIXA + lpix]) + q * 8) = px[it][p];
        }
        // 2. this chunk's weight rows -> slots x6b_slot(ph, tap) by LDS-DMA
#pragma unroll
        for (int i = 0; i < (WROWS + 3) / 4; ++i) {
            const int r = wave + 4 * i;  // wave-uniform
            if (r < WROWS) {
                const int p = r / KK, t = r % KK;
                const int slot = ph == 0 ? t : ph == 1 ? (t < 8 ? t : 9) : ph == 2 ? (t < 7 ? t : (t == 7 ? 8 : 10)) : (t < 6 ? t : 2 * t - 5);
                const __bf16* src = a.wp + ((size_t)p * a.plane_stride + (size_t)(t * nchunks + cch) * a.n_row * 8) + wlane;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(sW + p * WSLAB + slot * TN), 16, 0, 0);
            }
        }
        // 3. the staging registers re-loaded for chunk nch: the youngest vector-memory operations, a counted vmcnt leaves
        //    them in flight
#pragma unroll
        for (int it = 0; it < XITER; ++it) rx[it] = load_x(nch, it);
    };

    // One K-step: lane group g contracts block g = {image buffer, tap, weight slot} (compile-time per group).  The nine
    // K-steps of a 4-chunk cycle have fixed blocks, so the per-lane LDS offsets of all of them are computed ONCE (18 VGPRs):
    // selected per K-step they were two 4-way selects on the lane group = ~50 instructions of exec-masked control flow
    // between every operand barrier and the first fragment read behind it.
    // block code = buffer * 4096 + tap * 16 + slot;  O(ph, t) = this chunk's tap t, P(ph, t) = the previous chunk's tap t
#define X6B_O(ph, t) (((ph) & 1) * 4096 + (t) * 16 + x6b_slot(ph, t))
#define X6B_P(ph, t) ((((ph) - 1) & 1) * 4096 + (t) * 16 + x6b_slot((ph) - 1, t))
    // lane-group pairs (0, 1) and (2, 3) take taps of one column (t, t + 3, t + 6 are 48 pixels apart) wherever the
    // 9-tap arithmetic allows: 15 of the 18 pairs per 4 chunks; the other 3 cost their reads a second LDS cycle
    constexpr int KCODE[9][4] = {
        {X6B_O(0, 0), X6B_O(0, 3), X6B_O(0, 1), X6B_O(0, 4)}, {X6B_O(0, 2), X6B_O(0, 5), X6B_O(0, 6), X6B_O(0, 7)},
        {X6B_P(1, 8), X6B_O(1, 2), X6B_O(1, 0), X6B_O(1, 3)}, {X6B_O(1, 1), X6B_O(1, 4), X6B_O(1, 5), X6B_O(1, 6)},
        {X6B_P(2, 7), X6B_O(2, 1), X6B_P(2, 8), X6B_O(2, 2)}, {X6B_O(2, 0), X6B_O(2, 3), X6B_O(2, 4), X6B_O(2, 5)},
        {X6B_P(3, 6), X6B_O(3, 0), X6B_P(3, 7), X6B_O(3, 1)}, {X6B_P(3, 8), X6B_O(3, 2), X6B_O(3, 3), X6B_O(3, 6)},
        {X6B_O(3, 4), X6B_O(3, 7), X6B_O(3, 5), X6B_O(3, 8)}};
#undef X6B_O
#undef X6B_P
    int fold_adj[2] = {0, 0};
    auto set_fold = [&](const TileInfo& t) __attribute__((always_inline)) {
        if constexpr (FOLD) {
            fold_adj[0] = l16 >= t.bpos ? 2 : 0;
            fold_adj[1] = 16 + l16 >= t.bpos ? 2 : 0;
        }
    };
    int koffA[9], koffB[9];
#pragma unroll
    for (int ki = 0; ki < 9; ++ki) {
        auto offA_of = [](int c) { return (c >> 12) * XBUF + (((c >> 4) & 15) / KS) * HTWP + ((c >> 4) & 15) % KS; };
        auto offB_of = [](int c) { return (c & 15) * TN; };
        const int c0 = KCODE[ki][0], c1 = KCODE[ki][1], c2 = KCODE[ki][2], c3 = KCODE[ki][3];
        koffA[ki] = (g == 0 ? offA_of(c0) : g == 1 ? offA_of(c1) : g == 2 ? offA_of(c2) : offA_of(c3)) + l16;
        koffB[ki] = (g == 0 ? offB_of(c0) : g == 1 ? offB_of(c1) : g == 2 ? offB_of(c2) : offB_of(c3)) + l16;
        asm volatile("" : "+v"(koffA[ki]), "+v"(koffB[ki]));  // keep them resident: do not re-derive them inside the loop
    }
    auto kstep = [&](auto ki_c, auto&& mid) __attribute__((always_inline)) {
        constexpr int P1 = NP > 1 ? 1 : 0, P2 = NP > 2 ? 2 : 0;
        const int offA = koffA[decltype(ki_c)::value], offB = koffB[decltype(ki_c)::value];
        // FOLD: lanes whose pixel lies right of the seam read two LDS columns further (fold_adj: per lane and 16-pixel half, set
        // at the tile switch)
        const int offAh[2] = {offA + (FOLD ? fold_adj[0] : 0), offA + (FOLD ? fold_adj[1] : 0)};
        // all four channel blocks' weight fragments resident (48 registers), pixel fragments streamed with one block of
        // lookahead (12 + 12): every fragment is read once per K-step (24 reads per 96 MFMAs, as in the 32 x 32 x 16 kernel)
        U4 bw[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bw[p][cb].u = sW[p * WSLAB + offB + 16 * cb];
        U4 ax[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) ax[0][p].u = sX[offAh[0] + p * NPIXA + (2 * wave) * HTWP];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            if (pb < 3) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    ax[(pb + 1) & 1][p].u = sX[offAh[(pb + 1) & 1] + p * NPIXA + (2 * wave + ((pb + 1) >> 1)) * HTWP + 16 * ((pb + 1) & 1)];
            }
            U4(&A)[NP] = ax[pb & 1];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4 c = acc[pb][cb];
                if (NP > 2) {
                    c = punet_mfma16<F16>(bw[P1][cb].u, A[P1].u, c);  // mm
                    c = punet_mfma16<F16>(bw[0][cb].u, A[P2].u, c);   // lh
                    c = punet_mfma16<F16>(bw[P2][cb].u, A[0].u, c);   // hl
                }
                if (NP > 1) {
                    c = punet_mfma16<F16>(bw[0][cb].u, A[P1].u, c);  // mh
                    c = punet_mfma16<F16>(bw[P1][cb].u, A[0].u, c);  // hm
                }
                c = punet_mfma16<F16>(bw[0][cb].u, A[0].u, c);  // hh
                acc[pb][cb] = c;
            }
            if (pb == 1) mid();  // staging work of the next chunk, in the shadow of this K-step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto nibble = [](const f32x4& v) -> unsigned {
        return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
    };
    // What the epilogue needs from memory is fetched at the head of the tile's LAST chunk (behind its operand barrier), a
    // whole chunk of MFMAs ahead of its use: the gate dwords of the input gradient (4 per lane) / the bias quads of the
    // forward (4 x 16 bytes).  Loaded inside the epilogue each of them was a memory round trip in front of the first store.
    unsigned gpre[4] = {0u, 0u, 0u, 0u};
    f32x4 bpre[4];
    auto epilogue_prefetch = [&](const TileInfo& ti) __attribute__((always_inline)) {
        if (DGRAD) {
            if (a.gate_bits) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        int gx = ti.tx0 + 16 * h + l16, bb = ti.b;
                        const int gy = ti.ty0 + 2 * wave + m;
                        if constexpr (FOLD) {
                            const bool second = 16 * h + l16 >= ti.bpos;
                            gx = second ? 16 * h + l16 - ti.bpos : gx;
                            bb += second ? 1 : 0;
                        }
                        // dword of (pixel, channel) in the punet_gate_dwords layout: block {row, x / 32, channel block}, slot x % 32
                        const unsigned gd = (unsigned)(((((bb * a.H + gy) * tiles_x + (gx >> 5)) * ntn + (ti.n0 / TN)) << 6) +
                                                       (gx & 31) + 32 * (g & 1));
                        gpre[2 * h + m] = (gy < a.H && gx < a.W && bb < a.B) ? a.gate_bits[gd] : 0u;
                    }
            }
        } else {
            const int cq = ti.n0 + 4 * g;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                bpre[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.bias && cq + 16 * cb < N)
                    bpre[cb] = *reinterpret_cast<const f32x4*>(a.bias + (size_t)ti.b * a.bias_img_stride + cq + 16 * cb);
            }
        }
    };
    // EC = compile-time knowledge about this tile's epilogue (block-uniform, decided once per tile in `epilogue`):
    //   bit 0  F: the tile lies fully inside the image and the channel block inside N, and the layer has the usual form
    //             (forward: bias + ReLU; input gradient: gated by bits) - no per-store predicates, no per-store flag tests
    //   bit 1  G: forward writes the gate bits            bit 2  P: forward writes the pooled output / input gradient accumulates
    // EC = 0 is the general path with every flag tested at run time.  The general path alone cost the 8-chunk tiles of the
    // 64-channel layers 12 % of their time: ~1100 instructions, three branches per store.
    auto epilogue_ec = [&](const TileInfo& ti, auto ec_c) __attribute__((always_inline)) {
        constexpr int EC = decltype(ec_c)::value;
        constexpr bool F = EC & 1, G = EC & 2, P = EC & 4;
        const bool f_relu = F ? true : a.relu != 0;
        const bool f_gate_out = F ? G : a.gate_out != nullptr, f_pool = F ? P : a.pool_out != nullptr;
        const bool f_gate_bits = F ? true : a.gate_bits != nullptr, f_acc = F ? P : a.accumulate != 0;
        const int cq = ti.n0 + 4 * g;  // + 16 cb: first of this lane's four consecutive channels
#if X6B_TIMING & 4
        if (acc[0][0][0] != 1.2345e-30f) return;
#endif
        // uniform 64-bit base of image ti.b in the output-shaped tensors; offsets inside the image stay 32-bit (FOLD: a lane right
        // of the seam adds one image - the host checks that two images stay below 2^31 elements)
        const size_t img = (size_t)ti.b * ((size_t)(a.H * a.W) * N);
        float* const yb = a.y + img;
        const float* const mb = a.mask ? a.mask + img : nullptr;
        const unsigned img1 = (unsigned)(a.H * a.W) * (unsigned)N;  // elements of one image
        f32x4 bq[4];
        if (!DGRAD) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bq[cb] = bpre[cb];  // epilogue_prefetch
        }
        auto partner = [](float v) __attribute__((always_inline)) {  // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // 16-pixel half of the row; rows m = 0, 1 are pixel blocks h and 2 + h
            int gx = ti.tx0 + 16 * h + l16, bb = ti.b;
            unsigned ioff = 0u;  // FOLD: this lane's pixel belongs to image b + 1
            if constexpr (FOLD) {
                const bool second = 16 * h + l16 >= ti.bpos;
                gx = second ? 16 * h + l16 - ti.bpos : gx;
                bb += second ? 1 : 0;
                ioff = second ? img1 : 0u;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int gy = ti.ty0 + 2 * wave + m;
                const bool pix_in = F || (gy < a.H && gx < a.W && bb < a.B);
                // dword of (pixel, channel) in the punet_gate_dwords layout: block {row, x / 32, channel block}, slot
                // (x % 32) + 32 * ((c % 8) / 4) (= 16 h + l16 + 32 (g & 1) without FOLD: tiles start at multiples of 32)
                const unsigned gd = (unsigned)(((((bb * a.H + gy) * tiles_x + (gx >> 5)) * ntn + (ti.n0 / TN)) << 6) +
                                               (gx & 31) + 32 * (g & 1));
                const unsigned pbase = (unsigned)((gy * a.W + gx) * N + cq) + ioff;
                unsigned gin = 0u, gout = 0u;
                if (DGRAD && f_gate_bits) gin = gpre[2 * h + m];  // epilogue_prefetch (0 outside the image)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const bool ok = F || (pix_in && cq + 16 * cb < N);
                    const int sh = 16 * (cb >> 1) + 8 * (cb & 1) + 4 * (g >> 1);  // this quad's nibble in its dword
                    f32x4 v = acc[2 * m + h][cb];
                    if constexpr (F16) v *= inv;  // exact: inv is a power of two
                    if (!DGRAD) {
                        v += bq[cb];
                        if (f_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    } else if (f_gate_bits) {
                        const unsigned g4 = gin >> sh;
                        v[0] = g4 & 1u ? v[0] : 0.f; v[1] = g4 & 2u ? v[1] : 0.f; v[2] = g4 & 4u ? v[2] : 0.f; v[3] = g4 & 8u ? v[3] : 0.f;
                    } else if (!F && a.mask) {
                        if (ok) {
                            const f32x4 gm = *reinterpret_cast<const f32x4*>(mb + pbase + 16 * cb);
                            v[0] = gm[0] > 0.f ? v[0] : 0.f; v[1] = gm[1] > 0.f ? v[1] : 0.f;
                            v[2] = gm[2] > 0.f ? v[2] : 0.f; v[3] = gm[3] > 0.f ? v[3] : 0.f;
                        }
                    }
                    if (ok) {
                        if (DGRAD && f_acc) v += *reinterpret_cast<const f32x4*>(yb + pbase + 16 * cb);
#if X6B_TIMING & 1
                        if (v[0] == 1.2345e-30f)
#endif
                        *reinterpret_cast<f32x4*>(yb + pbase + 16 * cb) = v;
                        if constexpr (F16) amax_run = fmaxf(fmaxf(amax_run, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        if (!DGRAD) gout |= nibble(v) << sh;
                    }
                }
                if (!DGRAD && f_gate_out) {  // uniform; the other half of the dword's bits sits in lane ^ 32
                    const auto sw = __builtin_amdgcn_permlane32_swap(gout, gout, false, false);
                    const unsigned full = sw[0] | sw[1];  // {own, partner} in either order
                    if (g < 2 && pix_in) a.gate_out[gd] = full;
                }
            }
            if (!DGRAD && f_pool) {  // uniform: AvgPool2d(2, 2) of the output, order ((y00 + y01) + y10) + y11
                const int gy0 = ti.ty0 + 2 * wave;
                const bool ok0 = F || (gy0 < a.H && gx < a.W && bb < a.B);
                const unsigned qbase = (unsigned)(((gy0 >> 1) * (a.W >> 1) + (gx >> 1)) * N + cq) + (ioff >> 2);
                float* const qb = a.pool_out + (img >> 2);  // the pooled image has a quarter of the pixels
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    f32x4 v0 = acc[h][cb], v1 = acc[2 + h][cb], q;  // recomputed: no live copies of the rows
                    if constexpr (F16) { v0 *= inv; v1 *= inv; }
                    v0 += bq[cb]; v1 += bq[cb];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (f_relu) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                        q[e] = (((v0[e] + partner(v0[e])) + v1[e]) + partner(v1[e])) * 0.25f;
                    }
                    if (!(lane & 1) && ok0 && (F || cq + 16 * cb < N)) *reinterpret_cast<f32x4*>(qb + qbase + 16 * cb) = q;
                }
            }
        }
    };
#define X6B_EC(v) std::integral_constant<int, v>{}
    auto epilogue = [&](const TileInfo& ti) __attribute__((always_inline)) {
#if X6B_GENERAL_EPILOGUE
        epilogue_ec(ti, X6B_EC(0));
#else
        // block-uniform (FOLD: every column of a tile is a pixel of image b or b + 1, except behind the end of the virtual row)
        const bool full = ti.ty0 + TH <= a.H && (FOLD ? (ti.bpos == TW || ti.b + 1 < a.B) : ti.tx0 + TW <= a.W) && ti.n0 + TN <= N;
        if (!DGRAD) {
            if (full && a.relu && a.bias && !a.mask) {
                if (a.gate_out) {
                    if (a.pool_out) epilogue_ec(ti, X6B_EC(7)); else epilogue_ec(ti, X6B_EC(3));
                } else {
                    if (a.pool_out) epilogue_ec(ti, X6B_EC(5)); else epilogue_ec(ti, X6B_EC(1));
                }
            } else {
                epilogue_ec(ti, X6B_EC(0));
            }
        } else {
            if (full && a.gate_bits) {
                if (a.accumulate) epilogue_ec(ti, X6B_EC(5)); else epilogue_ec(ti, X6B_EC(1));
            } else {
                epilogue_ec(ti, X6B_EC(0));
            }
        }
#endif
    };
#undef X6B_EC

#ifdef X6B_TRACE
    int trace_chunk = 0;
#endif
    // work items (block-uniform scalars): whole tiles first, then this workgroup's run of the stream-K tail (ConvX6Args)
    const int nq = nchunks >> 2;
    bool more = false;  // the current tail item is followed by the second one of the same worker
    auto item_decode = [&](int it, int& tl, int& cb, int& ce, int& pt, bool& mo) __attribute__((always_inline)) {
        if (it < a.sk_whole) {
            tl = it; cb = 0; ce = nchunks; pt = -1; mo = false;
        } else {
            const unsigned j = (unsigned)__builtin_amdgcn_readfirstlane(it - a.sk_whole);
            const unsigned v = j >> 1;
            const unsigned q0 = v * (unsigned)a.sk_D, q1 = min(q0 + (unsigned)a.sk_D, (unsigned)a.sk_Q);
            const unsigned t0 = __umulhi(q0, a.m_nq);      // q0 / nq (nq >= 2)
            const unsigned b1 = (t0 + 1u) * (unsigned)nq;  // first quad of the next tile
            unsigned tt, qb, qe;
            if (!(j & 1u)) {
                tt = t0; qb = q0 - t0 * (unsigned)nq; qe = min(q1, b1) - t0 * (unsigned)nq; mo = q1 > b1;
            } else {
                tt = t0 + 1u; qb = 0u; qe = q1 - b1; mo = false;
            }
            tl = a.sk_whole + (int)tt;
            cb = (int)(4u * qb);
            ce = (int)(4u * qe);
            pt = qe - qb == (unsigned)nq ? -1 : (int)j;  // a run that covers a whole tile needs no hand-off
        }
    };
    int wg = blockIdx.x;
    if (a.xcd_remap) wg = (wg & 7) * ((int)gridDim.x >> 3) + (wg >> 3);
    int item = wg;
    if (item >= a.sk_whole) {  // fewer whole tiles than workgroups: straight to the tail
        if (wg >= a.sk_P) return;
        item = a.sk_whole + 2 * wg;
    }
    int tile, ch, ch_end, part;
    item_decode(item, tile, ch, ch_end, part, more);
    TileInfo ti = decode(tile);
    retarget(ti);
    set_fold(ti);
    wlane = wlane_of(ti);
    prefetch(ch);
#pragma unroll
    for (int it = 0; it < XITER; ++it) split_unit(it);

    // A part of a split tile is done: publish its accumulators, count the arrival; the LAST arriver (whichever workgroup that is)
    // sums all parts in part order - bit-identical whoever it is - and returns true: it then runs the tile's epilogue.
    // Inter-workgroup hand-off as MI355X_MICROARCH.md prescribes (per-XCD L2s are not coherent): every storing wave drains its
    // stores, workgroup barrier, ONE lane: agent-scope release (L2 write-back) -> wait -> agent-scope atomic add; the last
    // arriver: agent-scope acquire -> wait -> workgroup barrier -> plain loads.  No workgroup ever waits for another one.
    auto sk_finish = [&]() __attribute__((always_inline)) -> bool {
        const unsigned tt = (unsigned)(tile - a.sk_whole);
        f32x4* const slot = reinterpret_cast<f32x4*>(a.sk_ws) + ((size_t)part * 16) * 256 + tid;
#pragma unroll
        for (int i = 0; i < 16; ++i) slot[i * 256] = acc[i >> 2][i & 3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave has stored and drained; every wave has finished its fragment reads: the LDS is free
        // the workers whose runs meet in this tile: those of its first and of its last quad, and everyone between them
        auto worker_of = [&](unsigned q) { return a.sk_D == 1 ? q : __umulhi(q, a.m_D); };
        const unsigned v_first = worker_of(tt * (unsigned)nq), v_last = worker_of((tt + 1u) * (unsigned)nq - 1u);
        volatile unsigned* const flag = reinterpret_cast<volatile unsigned*>(smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop its own wait behind the write-back)
            const unsigned old = __hip_atomic_fetch_add(a.sk_cnt + tt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = old == v_last - v_first ? 1u : 0u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.sk_cnt + tt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            }
            flag[0] = last;
        }
        __syncthreads();
        const bool last = flag[0] != 0u;  // block-uniform; the next phase's first barrier orders this read before any LDS write
        if (!last) return false;
        for (unsigned v = v_first; v <= v_last; ++v) {  // worker order, whoever arrived last: bit-reproducible
            // worker v holds this tile in its first item iff its run STARTS in this tile
            const unsigned sl = 2u * v + (__umulhi(v * (unsigned)a.sk_D, a.m_nq) == tt ? 0u : 1u);
            const f32x4* src = reinterpret_cast<const f32x4*>(a.sk_ws) + ((size_t)sl * 16) * 256 + tid;
            if (v == v_first) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = src[i * 256];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] += src[i * 256];
            }
        }
        return true;
    };

    auto phase = [&](auto ph_c) __attribute__((always_inline)) -> bool {
        constexpr int ph = decltype(ph_c)::value;
        int nch = ch + 1, nitem = item, ntile = tile, nce = ch_end, npart = part;
        bool nmore = more;
        TileInfo nti = ti;
        bool has_next = true;
        if (ph == 3 && nch == ch_end) {
            if (item < a.sk_whole) {  // after the last whole tile: this workgroup's run of the tail, if it has one
                nitem = item + (int)gridDim.x;
                if (nitem >= a.sk_whole) {
                    nitem = a.sk_whole + 2 * wg;
                    has_next = wg < a.sk_P;
                }
            } else {
                nitem = item + 1;
                has_next = more;
            }
            if (has_next) {
                item_decode(nitem, ntile, nch, nce, npart, nmore);
                nti = decode(ntile);
            }
        }
        if (!has_next) nch = ch;
        if (ph == 3 && has_next && nitem != item) retarget(nti);
        __syncthreads();
        X6BT(0);
        __builtin_amdgcn_s_setprio(3);
        commit(ph_c, ch, nch);
        __builtin_amdgcn_s_setprio(0);
        // the weight DMAs (older) must have landed, the XITER activation loads issued after them may stay in flight;
        // a __syncthreads() here would drain them too (its fence waits vmcnt(0) while an LDS-DMA is pending)
        __builtin_amdgcn_s_waitcnt(XITER | (7 << 4) | (0 << 8));  // vmcnt(XITER) expcnt(7) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        X6BT(1);
        if (ph == 3 && ch == ch_end - 1 && part < 0) epilogue_prefetch(ti);  // block-uniform
        auto s0 = [&]() __attribute__((always_inline)) { split_unit(0); };
        auto s1 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int it = 1; it < XITER; ++it) split_unit(it);
        };
        auto s_ = []() {};
#define X6B_K(i) std::integral_constant<int, i>{}
        if (ph == 0) {
            kstep(X6B_K(0), s0);
            kstep(X6B_K(1), s1);
        } else if (ph == 1) {
            kstep(X6B_K(2), s0);
            kstep(X6B_K(3), s1);
        } else if (ph == 2) {
            kstep(X6B_K(4), s0);
            kstep(X6B_K(5), s1);
        } else {
            kstep(X6B_K(6), s0);
            kstep(X6B_K(7), s1);
            kstep(X6B_K(8), s_);
        }
#undef X6B_K
        X6BT(2);
#ifdef X6B_TRACE
        ++trace_chunk;
#endif
        if (ph == 3 && ch == ch_end - 1) {  // no barrier: the epilogue touches no LDS, waves drain into it as they finish
            bool run = true;
            if (part >= 0) {  // block-uniform: a part of a split tile (always among the LAST items of a workgroup)
                run = sk_finish();
                if (run) epilogue_prefetch(ti);
            }
            if (run) epilogue(ti);
            zero_acc();
        }
        if (ph == 3 && ntile != tile) wlane = wlane_of(nti);  // the next tile's weight columns, from its first commit on
        item = nitem;
        tile = ntile;
        ch = nch;
        ch_end = nce;
        part = npart;
        more = nmore;
        if (FOLD && ph == 3 && ntile != tile) set_fold(nti);  // (after this tile's last fragment reads and its epilogue)
        ti = nti;
        return !has_next;
    };
    while (true) {
        if (phase(std::integral_constant<int, 0>{})) break;
        if (phase(std::integral_constant<int, 1>{})) break;
        if (phase(std::integral_constant<int, 2>{})) break;
        if (phase(std::integral_constant<int, 3>{})) break;
    }
    if constexpr (F16) {
        if (a.amax_out) {  // block-uniform: the workgroup leaves the tile loop together
            __syncthreads();  // the slower waves' last fragment reads are done: the operand images are free
            punet_amax_publish_block256(a.amax_out, amax_run, reinterpret_cast<float*>(smem));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// 1x1 convolution / its input gradient in the six-product arithmetic for WIDE contractions (one source, input channels a multiple
// of 32): the GEMM of K5x (upconv.hip: [B h w] x cup -> 9 cout and back).  conv3x3_x6_kernel<3, ., 1> was built for the 64-channel
// Fcomb layers: it stages 8 channels between two barriers and then issues 12 MFMAs per wave - 104-120 TFLOP/s on K = 512 ... 2304.
// Here a phase stages 32 channels (four [plane][pixel][8 ch] sub-chunk images + their weight rows, 61,440 B: two workgroups per
// CU) and issues two K-steps = 48 MFMAs per wave between the same two barriers; the fp32 -> 3 x bf16 split of the next phase's
// activations runs in their shadow as in the 3x3 kernels, the weight rows go global -> LDS by LDS-DMA (no staging registers).
// What bounds it (timing-only builds, -DX6W_TIMING=..., tools/ab_variants.sh; batch 4, K5x's GEMMs): with NO matrix instructions the
// kernel still takes 192 / 235 / 306 us, with matrix instructions but without loads, splits and stores 275 / 247 / 247 us - against a
// matrix-pipe floor of ~93 us.  Per workgroup and phase 49 KB of staging writes + 12 KB of weight rows + 98 KB of fragment reads
// cross the LDS for 1,536 cycles of MFMAs per SIMD: 81 % of the LDS's 128 bytes per clock with two workgroups per CU, where the
// 3x3 kernels (whose staging is amortised over nine taps) need 58 %.  A 64 x 64 wave tile in three planes cannot go faster as a
// GEMM; the remedy is a larger wave tile (fewer fragment reads per MFMA) with pre-split operands by LDS-DMA (no staging registers
// to pay for the accumulators) - DESIGN.md section 8.  Rejected on the way (all +-0): staging two phases ahead, sub-chunk images
// 128 bytes apart mod 256, weight units through registers, an XCD-aware tile order.
// Round 4, later (128-channel tiles, tools/trace_x6w.py + -DX6W_TIMING on them): a phase takes ~5,000 cycles for 1,536 of MFMAs per
// wave; the split moved between the MFMA groups (+1-2 %); a double-buffered form with the whole staging inside the K-step and one
// barrier per phase measured +-0 (251 / 255 / 339 vs 249 / 259 / 342 us) and was removed again.  Without its epilogue the kernel
// takes 207 / 187 / 207 us: writing z (151 / 302 / 604 MB: 9 cout floats per low-resolution pixel) costs 12 / 25 / 39 % - the
// 512 x 512 block's GEMM is bound by that write, not by anything inside the K loop.  The stores were 32-byte pieces of 32 pixels;
// as a timing-only permutation into whole 128-byte lines they measured 239 -> 234 / 255 -> 235 / 337 -> 304 us (round 5: built, X6W_LINES).
// The transposed tile (pixels as A: a lane holds one channel of 16 pixels, whole lines by dword stores) was built, passed its
// tests and measured 254 / 266 / 348 us on a box where the input gradients ran 2 % slower - four times the store instructions
// queue in front of the next tile's loads; the float4 form needs an LDS transposition for ~0.1 ms of the step: not built
// (profiles/r04/conv1x1_x6w_store_pattern_variants.txt).
// A thread stages the SAME pixel column and float4 of the 32 channels in every phase (unit = 8 rows x that column): lanes 0-7
// read the 128 contiguous bytes of one pixel.  Tile, accumulator mapping, epilogue semantics and gate-bit layout are
// conv3x3_x6_kernel's (weights as A, pixels as B: a lane holds one pixel and four consecutive channels per register quad).
#ifndef X6W_TIMING  // timing-only builds of conv1x1_x6w_kernel (WRONG results): 1 = no epilogue, 2 = no split arithmetic, 4 = no activation
#define X6W_TIMING 0  // loads, 8 = no MFMAs (tools/ab_variants.sh conv_x6.hip ...)
#endif
// Round 5: the epilogue's stores as WHOLE 128-byte lines.  In the accumulator layout a lane holds one pixel and four consecutive
// channels per register quad: a store instruction writes 32-byte pieces of 32 pixels.  Each 32-channel block of a wave's 32 pixels
// now goes through a wave-private LDS buffer ([pixel][32 channels]) and is stored as 8 pixels x 128 contiguous bytes per
// instruction (8 lanes per pixel); an accumulating input gradient reads the old values in the same pattern.  Same values, same
// operations per element: bit-identical (X6W_LINES=0: the direct stores, A/B).
#ifndef X6W_LINES
#define X6W_LINES 1
#endif
// NB = 32-channel output blocks per wave: 2 = a 64-channel tile and 32-channel phases (above); 4 = a 128-channel tile and 16-channel
// phases - the same 48 MFMAs per wave and phase, but every staged activation value and every pixel fragment serves twice the output
// channels: the LDS traffic per MFMA cycle drops from 81 % to 55 % of the LDS's rate (the accumulators take 128 registers, which
// is why the phases stage 16 channels instead of 32).  Used when the layer has at least 128 output channels.
#ifdef X6W_TRACE  // s_memtime stamps of workgroup 9's phases 8..11 (tools/trace_x6w.py): per wave and phase: top, first barrier passed,
                  // commit issued, operand barrier passed, first K-step issued, splits done.  The buffer travels in ConvX6Args::pool_out
                  // (unused by this kernel), handed in through $PUNET_X6W_TRACE_PTR by the launcher of a trace build.
#define X6WT(id)                                                                                                        \
    do {                                                                                                                \
        if (a.pool_out && blockIdx.x == 9 && (threadIdx.x & 63) == 0 && trace_phase >= 8 && trace_phase < 12)           \
            reinterpret_cast<unsigned long long*>(a.pool_out)[((threadIdx.x >> 6) * 4 + (trace_phase - 8)) * 8 + (id)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define X6WT(id)
#endif
template <bool DGRAD, int NB>
__global__ __launch_bounds__(256, 2) void conv1x1_x6w_kernel(ConvX6Args a) {
    constexpr int NP = 3, NSUB = NB == 2 ? 4 : 2, CKP = 8 * NSUB, NPIX = TH * TW, TNW = 32 * NB;
    constexpr int QN = CKP / 4;                   // float4 units per pixel and phase
    constexpr int RSTEP = 256 / (QN * TW);        // tile rows the 256 threads cover at once (1 or 2)
    constexpr int XITER = TH / RSTEP;             // units per thread: pixel (row rsub + it * RSTEP, column col), float4 q of the CKP channels
    constexpr int SUBX = NP * NPIX, SUBW = NP * TNW;  // uint4 units of one sub-chunk's activation image / weight block
    __shared__ uint4 smem[NSUB * (SUBX + SUBW)];      // 61,440 B (NB = 2) / 36,864 B (NB = 4)
    uint4* sX = smem;
    uint4* sW = smem + NSUB * SUBX;
#if X6W_LINES
    // the epilogue's transposition buffer, PRIVATE to a wave (no barrier): 32 pixels x 32 channels as rows of 9 float4 (a pitch of
    // 144 bytes: the 16 lanes of a ds_write_b128 pass hit 16 different bank quads)
    constexpr int EP = 9;
    __shared__ float4 sE[4][32 * EP];
#endif

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int q = tid % QN, col = (tid / QN) % TW, rsub = tid / (QN * TW);  // this thread's float4 of the phase's channels, its pixel column and first row

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntn = (a.n_total + TNW - 1) / TNW, ntn64 = (a.n_total + TN - 1) / TN;  // tiles / gate-dword blocks along the channels
    const int N = a.n_total, K = a.c0;
    const int nph = K / CKP;

    auto decode = [&](int tile) {
        TileInfo t;
        t.n0 = (tile % ntn) * TNW;
        int sid = tile / ntn;
        t.tx0 = (sid % tiles_x) * TW;
        sid /= tiles_x;
        t.ty0 = (sid % tiles_y) * TH;
        t.b = sid / tiles_y;
        return t;
    };

    f32x16 acc[2][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;
    };
    zero_acc();

    float4 rx[XITER];
    bf16x4 px[XITER][NP];
    const float* xbase = a.x0;  // image of the tile the staging registers are loaded for (block-uniform)
    unsigned xoff0 = 0u, xrow = 0u;  // element offset of (row 0, this column) inside the image; of one image row (0 beyond the last row)
    unsigned inmask = 0u;        // bit it: pixel (row it, this column) lies inside the image
    unsigned wlane = 0u;         // this lane's output-channel offset (elements) inside a weight row
    auto retarget = [&](const TileInfo& ti) __attribute__((always_inline)) {
        const int gx = ti.tx0 + col;
        inmask = 0u;
#pragma unroll
        for (int it = 0; it < XITER; ++it) inmask |= (ti.ty0 + rsub + it * RSTEP < a.H && gx < a.W) ? (1u << it) : 0u;
        xoff0 = (unsigned)((min(ti.ty0, a.H - 1) * a.W + min(gx, a.W - 1)) * K + q * 4);
        xrow = (unsigned)(a.W * K);
        xbase = a.x0 + (size_t)ti.b * ((size_t)(a.H * a.W) * K);
    };
    // (NOT part of retarget: the weight DMA of a tile's LAST phase is issued after the staging registers have been re-targeted at
    // the next tile, and that tile has another channel block whenever the grid is not a multiple of the channel-block count)
    // (NB = 4: a wave stages the 64-column half `wave & 1` of sub-chunk `wave >> 1`)
    auto wlane_of = [&](const TileInfo& ti) {  // columns beyond N copy column N - 1
        return (unsigned)((a.n_off + min(ti.n0 + (NB == 4 ? 64 * (wave & 1) : 0) + lane, N - 1)) * 8);
    };
    int ty_cur = 0;  // ty0 of the tile the staging registers belong to (rows beyond the image re-read the last row: never used)
    auto load_x = [&](int ph, int it) __attribute__((always_inline)) -> float4 {
#if X6W_TIMING & 4
        return make_float4((float)(ph + it), 1.f, (float)q, 0.5f);
#endif
        const int r = min(rsub + it * RSTEP, a.H - 1 - min(ty_cur, a.H - 1));  // clamped row
        return *reinterpret_cast<const float4*>(xbase + (xoff0 + (unsigned)r * xrow + (unsigned)(ph * CKP)));
    };
    auto split_unit = [&](int it) __attribute__((always_inline)) {
        const float4 r = (inmask >> it) & 1u ? rx[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 ph_, pm, pl;
#if X6W_TIMING & 2
        ph_ = make_uint2(__builtin_bit_cast(unsigned, r.x), __builtin_bit_cast(unsigned, r.y));
        pm = make_uint2(__builtin_bit_cast(unsigned, r.z), __builtin_bit_cast(unsigned, r.w));
        pl = ph_;
#else
        split3_quad<NP>(r, ph_, pm, pl);
#endif
        asm volatile("" : "+v"(ph_.x), "+v"(ph_.y), "+v"(pm.x), "+v"(pm.y), "+v"(pl.x), "+v"(pl.y));  // keep the split where it is written
        px[it][0] = __builtin_bit_cast(bf16x4, ph_);
        px[it][1] = __builtin_bit_cast(bf16x4, pm);
        px[it][2] = __builtin_bit_cast(bf16x4, pl);
    };
    // phase `ph`: packed planes -> LDS images, this phase's 12 weight rows by LDS-DMA, staging registers re-loaded for phase nph_
    auto commit = [&](int ph, int nph_) __attribute__((always_inline)) {
        char* const dst = reinterpret_cast<char*>(sX + (q >> 1) * SUBX) + (q & 1) * 8 + (rsub * TW + col) * 16;
#pragma unroll
        for (int it = 0; it < XITER; ++it)
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(dst + (p * NPIX + it * RSTEP * TW) * 16) = px[it][p];
        // one wave-instruction = one {sub-chunk, plane} row of 64 output channels; NB = 2: wave = sub-chunk, NB = 4: wave = {sub-chunk,
        // 64-column half}
        const int wsub = NB == 2 ? wave : wave >> 1, wcol = NB == 2 ? 0 : 64 * (wave & 1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const __bf16* src = a.wp + ((size_t)p * a.plane_stride + (size_t)(ph * NSUB + wsub) * a.n_row * 8) + wlane;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(sW + wsub * SUBW + p * TNW + wcol), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < XITER; ++it) rx[it] = load_x(nph_, it);
    };
    // one K-step = 16 channels: lanes 0-31 contract sub-chunk 2 ks, lanes 32-63 sub-chunk 2 ks + 1.  `mid(g)` is called after MFMA
    // group g (12 MFMAs each; NB groups per K-step): the next phase's split goes there, i.e. BETWEEN the MFMAs of this wave - a wave
    // issues in order, so vector work placed behind the K-step only started once all 48 MFMAs had been issued and ran with the matrix
    // pipe idle (s_memtime timeline, tools/trace_x6w.py: 1,900 of a phase's 5,600 cycles)
    auto kstep = [&](int ks, auto&& mid) __attribute__((always_inline)) {
        const int sub = 2 * ks + half;
        U4 av[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int m = 0; m < 2; ++m) av[p][m].u = sX[sub * SUBX + p * NPIX + (2 * wave + m) * TW + l31];
#pragma unroll
        for (int nbp = 0; nbp < NB / 2; ++nbp) {  // the weight fragments of two channel blocks at a time (NB = 4: 24 registers, not 48)
            U4 bv[NP][2];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) bv[p][nb].u = sW[sub * SUBW + p * TNW + (2 * nbp + nb) * 32 + l31];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x16 c = acc[m][2 * nbp + nb];
                    c = punet_mfma32<false>(bv[1][nb].u, av[1][m].u, c);  // mm
                    c = punet_mfma32<false>(bv[0][nb].u, av[2][m].u, c);  // lh
                    c = punet_mfma32<false>(bv[2][nb].u, av[0][m].u, c);  // hl
                    c = punet_mfma32<false>(bv[0][nb].u, av[1][m].u, c);  // mh
                    c = punet_mfma32<false>(bv[1][nb].u, av[0][m].u, c);  // hm
                    c = punet_mfma32<false>(bv[0][nb].u, av[0][m].u, c);  // hh
                    acc[m][2 * nbp + nb] = c;
                }
                mid(2 * nbp + m);
                __builtin_amdgcn_sched_barrier(0);  // keep the groups (and the split between them) where they are written
            }
        }
    };
    // epilogue straight from the accumulators (mapping and gate-bit layout: conv3x3_x6_kernel)
    auto epilogue = [&](const TileInfo& ti) __attribute__((always_inline)) {
#if X6W_TIMING & 1
        if (acc[0][0][0] != 1.2345e-30f) return;
#endif
        const int gx = ti.tx0 + l31;
        const size_t img = (size_t)ti.b * ((size_t)(a.H * a.W) * N);
        float* const yb = a.y + img;
        const float* const mb = a.mask ? a.mask + img : nullptr;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int gy = ti.ty0 + 2 * wave + m;
            const bool pix_in = gy < a.H && gx < a.W;
            // gate dwords are per 64-channel block: block ti.n0 / 64 + nb / 2, bits 16 (nb & 1) + 4 rg + e
            const unsigned gd = (unsigned)(((((ti.b * a.H + gy) * tiles_x + (ti.tx0 / TW)) * ntn64 + (ti.n0 / TN)) << 6) + lane);
            const unsigned pbase = (unsigned)((gy * a.W + gx) * N + ti.n0 + 4 * half);
            unsigned gin2[NB / 2], gout2[NB / 2];
#pragma unroll
            for (int j = 0; j < NB / 2; ++j) {
                gout2[j] = 0u;
                gin2[j] = (DGRAD && a.gate_bits && pix_in && ti.n0 + 64 * j < N) ? a.gate_bits[gd + 64 * j] : 0u;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const unsigned gin = gin2[nb >> 1];
                    unsigned& gout = gout2[nb >> 1];
                    const int ch = nb * 32 + rg * 8;  // + 4 * half + e
                    const bool ok = pix_in && ti.n0 + ch + 4 * half < N;  // N is a multiple of 4
                    float4 v = make_float4(acc[m][nb][4 * rg + 0], acc[m][nb][4 * rg + 1], acc[m][nb][4 * rg + 2], acc[m][nb][4 * rg + 3]);
                    if (!DGRAD) {
                        if (a.bias && ok) {
                            const float4 bq = *reinterpret_cast<const float4*>(a.bias + (size_t)ti.b * a.bias_img_stride + ti.n0 + ch + 4 * half);
                            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                        }
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    } else if (a.gate_bits) {
                        const unsigned g4 = gin >> (16 * (nb & 1) + 4 * rg);
                        v.x = g4 & 1u ? v.x : 0.f; v.y = g4 & 2u ? v.y : 0.f; v.z = g4 & 4u ? v.z : 0.f; v.w = g4 & 8u ? v.w : 0.f;
                    } else if (a.mask) {
                        if (ok) {
                            const float4 g = *reinterpret_cast<const float4*>(mb + pbase + ch);
                            v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f;
                            v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
                        }
                    }
#if X6W_LINES
                    sE[wave][l31 * EP + 2 * rg + half] = v;  // (values of lanes / channels outside the tensor are never stored)
                    if (!DGRAD && ok)
                        gout |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                                << (16 * (nb & 1) + 4 * rg);
                    if (rg == 3) {  // the block's 32 x 32 values are in the buffer: 8 pixels x 128 bytes per instruction
                        const int cq = ti.n0 + nb * 32 + (lane & 7) * 4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int px_ = 8 * i + (lane >> 3);
                            float4 t = sE[wave][px_ * EP + (lane & 7)];
                            if (gy < a.H && ti.tx0 + px_ < a.W && cq < N) {
                                float* const dst = yb + (unsigned)((gy * a.W + ti.tx0 + px_) * N + cq);
                                if (DGRAD && a.accumulate) {
                                    const float4 o = *reinterpret_cast<const float4*>(dst);
                                    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
                                }
                                *reinterpret_cast<float4*>(dst) = t;
                            }
                        }
                    }
#else
                    if (ok) {
                        if (DGRAD && a.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(yb + pbase + ch);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(yb + pbase + ch) = v;
                        if (!DGRAD)
                            gout |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                                    << (16 * (nb & 1) + 4 * rg);
                    }
#endif
                }
            if (!DGRAD && a.gate_out && pix_in) {
#pragma unroll
                for (int j = 0; j < NB / 2; ++j)
                    if (ti.n0 + 64 * j < N) a.gate_out[gd + 64 * j] = gout2[j];
            }
        }
    };

    // (no XCD-aware tile order here: measured +-: 277 -> 301 / 306 -> 307 / 381 -> 373 us on K5x's three forward GEMMs - the kernel is
    // bound by its LDS traffic, not by what it fetches; see the header comment)
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    int ph = 0;
#ifdef X6W_TRACE
    int trace_phase = 0;
#endif
    TileInfo ti = decode(tile);
    retarget(ti);
    wlane = wlane_of(ti);
    ty_cur = ti.ty0;
#pragma unroll
    for (int it = 0; it < XITER; ++it) rx[it] = load_x(0, it);
#pragma unroll
    for (int it = 0; it < XITER; ++it) split_unit(it);
    while (true) {
        int nph_ = ph + 1, ntile = tile;
        TileInfo nti = ti;
        if (nph_ == nph) {
            nph_ = 0;
            ntile = tile + (int)gridDim.x;
            if (ntile < a.ntiles) nti = decode(ntile);
        }
        const bool has_next = ntile < a.ntiles;
        if (!has_next) nph_ = ph;  // nothing follows: the (branch-free) reloads re-fetch this phase and are dropped
        if (has_next && nph_ == 0) {
            retarget(nti);
            ty_cur = nti.ty0;
        }
        X6WT(0);
        __syncthreads();  // every wave has read the previous phase's fragments
        X6WT(1);
        __builtin_amdgcn_s_setprio(3);
        commit(ph, nph_);
        __builtin_amdgcn_s_setprio(0);
        X6WT(2);
        // the weight DMAs (older) must have landed, the XITER activation loads issued after them may stay in flight
        __builtin_amdgcn_s_waitcnt(XITER | (7 << 4) | (0 << 8));  // vmcnt(XITER) expcnt(7) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        X6WT(3);
        // the next phase's split between the MFMA groups: NB groups per K-step, NSUB / 2 K-steps, XITER units
        constexpr int GROUPS = NB * (NSUB / 2), UPG = XITER / GROUPS;  // 4 groups: 2 units each (NB = 2) or 1 (NB = 4)
        static_assert(UPG * GROUPS == XITER, "the staged units divide evenly over the MFMA groups");
#if !(X6W_TIMING & 8)
        kstep(0, [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < UPG; ++u) split_unit(g * UPG + u);
        });
        X6WT(4);
        if constexpr (NSUB == 4)
            kstep(1, [&](int g) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < UPG; ++u) split_unit((NB + g) * UPG + u);
            });
#else
#pragma unroll
        for (int it = 0; it < XITER; ++it) split_unit(it);
#endif
        X6WT(5);
#ifdef X6W_TRACE
        ++trace_phase;
#endif
        if (ph == nph - 1) {  // no barrier: the epilogue touches no LDS
            epilogue(ti);
            zero_acc();
        }
        if (ntile != tile) wlane = wlane_of(nti);
        tile = ntile;
        ph = nph_;
        ti = nti;
        if (!has_next) break;
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

// ---- fp16 planes of the weight slabs ("f16x3" / "f16"): per-slab amax first (slot j of wamax), then h (+ l) of w * 2^k with
// the slab's power-of-two scale (common.h); same [plane][t'][k/8][n][8] layout and placement as the bf16 planes (16-bit
// elements), plane 2 unused.  A unit of the amax pass is 8 CONSECUTIVE floats of the slab (any partition does for a maximum).
__global__ __launch_bounds__(256) void slab_amax_batch_kernel(const float* __restrict__ flat, const long* __restrict__ table, int nconv,
                                                              float* __restrict__ wamax, long per) {
    // every block owns a CONTIGUOUS range of units (so that it touches one slab, rarely two or three): per-thread running
    // maxima, published when the slab changes and at the end - a few atomics per block instead of one per wave and round
    // (42 hot addresses: contended atomics serialise at the memory side)
    // `per` is ODD (host): with 2,048 units = 64 KB per block all blocks read the same offset of their ranges at the same time,
    // i.e. addresses 64 KB apart - a handful of the memory channels - and the pass ran at 1.4 TB/s
    const long total = table[(size_t)nconv * 4 + 3];
    const long u0 = (long)blockIdx.x * per, u1 = u0 + per < total ? u0 + per : total;
    if (u0 >= u1) return;  // (block-uniform)
    int lo = 0, hi = nconv - 1;  // slab of the block's first unit
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 4 + 3] <= u0) lo = mid; else hi = mid - 1;
    }
    long ub = u0;
    while (ub < u1) {  // block-uniform loop over the slabs the range touches
        const long slab_end = table[(size_t)(lo + 1) * 4 + 3];  // first unit of the next slab (row nconv holds the total)
        const long ue = slab_end < u1 ? slab_end : u1;
        const float4* w = reinterpret_cast<const float4*>(flat + table[(size_t)lo * 4 + 0] + 8 * (ub - table[(size_t)lo * 4 + 3]));
        float m = 0.f;
        for (long i = threadIdx.x; i < 2 * (ue - ub); i += 256) {  // 2 float4 per unit, consecutive in memory
            const float4 a = w[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
        }
        punet_amax_publish(wamax + lo, m);
        ub = ue;
        ++lo;
    }
}

template <int NPL>
__device__ __forceinline__ void splitf16_store(const float (&v)[8], float sc, __bf16* dst, long plane_stride) {
    uint4 h, l;
    unsigned* hp = reinterpret_cast<unsigned*>(&h);
    unsigned* lp = reinterpret_cast<unsigned*>(&l);
#pragma unroll
    for (int e = 0; e < 4; ++e) splitf16_pair<NPL>(v[2 * e], v[2 * e + 1], sc, hp[e], lp[e]);
    *reinterpret_cast<uint4*>(dst) = h;
    if (NPL > 1) *reinterpret_cast<uint4*>(dst + plane_stride) = l;
}

__global__ __launch_bounds__(256) void split_weights_f16_batch_kernel(const float* __restrict__ flat, __bf16* __restrict__ planes,
                                                                      const long* __restrict__ table, int nconv, int mode,
                                                                      const float* __restrict__ wamax) {
    const long total = table[(size_t)nconv * 4 + 3];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nconv - 1;
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
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kc * 8 + e;
            v[e] = mode == 0 ? w[((size_t)t * cin + k) * cout + n] : w[((size_t)t * cin + n) * cout + k];
        }
        splitf16_store<2>(v, punet_f16_scale_of(wamax[lo]), pl + u * 8, plane_stride);
    }
}

// one slab (unit tests / callers with their own layout): amax into slot[0], then the planes
__global__ __launch_bounds__(256) void slab_amax_kernel(const float* __restrict__ w, size_t n, float* __restrict__ slot) {
    float m = 0.f;
    const size_t rounds = (n + (size_t)gridDim.x * blockDim.x - 1) / ((size_t)gridDim.x * blockDim.x);
    for (size_t r = 0; r < rounds; ++r) {
        const size_t i = r * (size_t)gridDim.x * blockDim.x + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n) m = fmaxf(m, fabsf(w[i]));
    }
    punet_amax_publish(slot, m);
}

__global__ __launch_bounds__(256) void split_weights_f16_kernel(const float* __restrict__ w, __bf16* __restrict__ planes,
                                                                long plane_stride, int cin, int cout, int kk, int mode,
                                                                const float* __restrict__ slot) {
    const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
    const int KC = K / 8;
    const size_t total = (size_t)kk * KC * N;
    const float sc = punet_f16_scale_of(slot[0]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const int kc = (int)((i / N) % KC);
        const int tp = (int)(i / ((size_t)N * KC));
        const int t = mode == 0 ? tp : kk - 1 - tp;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kc * 8 + e;
            v[e] = mode == 0 ? w[((size_t)t * cin + k) * cout + n] : w[((size_t)t * cin + n) * cout + k];
        }
        splitf16_store<2>(v, sc, planes + i * 8, plane_stride);
    }
}

}  // namespace

extern "C" int punet_split_weights_f16(const float* w, void* planes, long plane_stride, int cin, int cout, int mode, float* amax_w,
                                       void* stream) {
    PUNET_CHECK_ARG(w && planes && amax_w && cin > 0 && cout > 0 && (mode == 0 || mode == 1), "punet_split_weights_f16: bad args");
    PUNET_CHECK_ARG((mode == 0 ? cin : cout) % 8 == 0, "punet_split_weights_f16: the contracted channel count must be a multiple of 8");
    PUNET_CHECK_ARG(plane_stride >= 9l * cin * cout, "punet_split_weights_f16: plane_stride too small");
    hipStream_t st = (hipStream_t)stream;
    if (punet_zero_fill(amax_w, sizeof(float), st) != hipSuccess) {
        punet_set_error("punet_split_weights_f16: memset failed");
        return PUNET_ERR_HIP;
    }
    const size_t n = (size_t)9 * cin * cout;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    slab_amax_kernel<<<grid, 256, 0, st>>>(w, n, amax_w);
    PUNET_LAUNCH_CHECK("slab_amax");
    grid = (int)((n / 8 + 255) / 256);
    if (grid > 4096) grid = 4096;
    split_weights_f16_kernel<<<grid, 256, 0, st>>>(w, (__bf16*)planes, plane_stride, cin, cout, 9, mode, amax_w);
    PUNET_LAUNCH_CHECK("split_weights_f16");
    return PUNET_OK;
}

extern "C" int punet_split_weights_f16_batch(const float* flat, void* planes, const long* table, int nconv, long total_units,
                                             int mode, float* wamax, int compute_amax, void* stream) {
    PUNET_CHECK_ARG(flat && planes && table && wamax && nconv > 0 && total_units > 0 && (mode == 0 || mode == 1),
                    "punet_split_weights_f16_batch: bad args");
    hipStream_t st = (hipStream_t)stream;
    long grid = (total_units + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (compute_amax) {  // (the forward and the input-gradient orientation of one weight state share the slots: compute once)
        if (punet_zero_fill(wamax, sizeof(float) * (size_t)nconv, st) != hipSuccess) {
            punet_set_error("punet_split_weights_f16_batch: memset failed");
            return PUNET_ERR_HIP;
        }
        const int ablocks = total_units / 2048 < 2048 ? (int)(total_units / 2048 + 1) : 2048;
        const long per = ((total_units + ablocks - 1) / ablocks) | 1;
        slab_amax_batch_kernel<<<ablocks, 256, 0, st>>>(flat, table, nconv, wamax, per);
        PUNET_LAUNCH_CHECK("slab_amax_batch");
    }
    split_weights_f16_batch_kernel<<<(int)grid, 256, 0, st>>>(flat, (__bf16*)planes, table, nconv, mode, wamax);
    PUNET_LAUNCH_CHECK("split_weights_f16_batch");
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
                                const float* mask, int accumulate, int relu, int B, int H, int W, int opts, void* stream) {
    return punet_conv3x3_x6_gated(x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask,
                                  nullptr, nullptr, accumulate, relu, B, H, W, opts, stream);
}

// $PUNET_CONV1X1_WIDE=0: 1x1 convolutions with wide contractions through the 8-channel-chunk kernel as before (same-box A/B);
// read once per process (a build-time choice in effect, not per-call state: both give the same products, summed in another order)
static int punet_conv1x1_wide() {  // 0: off; 1: 64-channel tiles only; 2 (default): 128-channel tiles where the layer has them
    static const int on = [] {
        const char* e = getenv("PUNET_CONV1X1_WIDE");
        return e && e[0] == '0' ? 0 : (e && e[0] == '1' ? 1 : 2);
    }();
    return on;
}

struct F16Scales {  // fp16-operand arithmetic (planes = 2 or 1; 0 = the bf16 split-plane arithmetic): amax slots, see ConvX6Args
    int planes = 0;
    const float* amax_x0 = nullptr;
    const float* amax_x1 = nullptr;
    const float* amax_w = nullptr;
    float* amax_out = nullptr;
};

static int conv_x6_launch(int ksize, const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                          int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                          const void* gate_bits, void* gate_out, int accumulate, int relu, int B, int H, int W, int opts,
                          void* stream, float* pool_out = nullptr, const F16Scales& f16 = F16Scales(), void* scratch = nullptr,
                          size_t scratch_bytes = 0);

// ---- stream-K tail of the persistent 3x3 kernels (conv3x3_x6b_kernel): what to do with the tiles of the last, partial round.
// A launch deals ntiles equal tiles to `cap` resident workgroups; when ntiles is not a multiple of cap the last round keeps a
// fraction of the chip busy for a whole tile time (704 x 704 patches: 1,056 tiles = 2.06 rounds at the 88 x 88 level, 3,872 =
// 7.56 at the 352 x 352 level).  The T tiles of that round are therefore dealt as ONE range of T * K/32 quads (a quad = 4 chunks
// = one cycle of the kernel's phases) in equal runs of D quads to P = ceil(T * nq / D) <= cap workgroups; a run covers parts of
// at most two tiles, the parts' accumulators meet in the caller's scratch (2 slots per worker) and the workgroup that completes a
// tile runs its epilogue (kernel comment).  Costs in units of one 8-channel chunk of one tile (~3-4 us): a partial item pays ~3
// for publishing 64 KB, the combining workgroup ~1.5 + 0.35 per part; an unsplit tail runs ~0.8 of a tile time when its
// workgroups have their CUs to themselves.
constexpr size_t SK_COUNTER_BYTES = 4096, SK_SLOT_BYTES = 16 * 256 * 16;
struct SkPlan {
    int whole, P, D, Q;
};
static int punet_splitk_mode() {  // $PUNET_SPLITK: 0 = never split, n >= 1 = runs of n quads wherever a launch has a partial round
    static const int m = [] {    // (A/B); unset = by the cost model; read once per process (an A/B switch like PUNET_CONV1X1_WIDE)
        const char* e = getenv("PUNET_SPLITK");
        return e ? atoi(e) : -1;
    }();
    return m;
}
static SkPlan sk_plan(int ntiles, int cap, int nchunks, size_t scratch_bytes, int force) {
    const SkPlan whole{ntiles, 0, 1, 0};
    const int nq = nchunks / 4;
    // PUNET_X6_SPLITK bits of opts: 0 = the library decides, 1 = never, n >= 2 = runs of n - 1 quads
    const int mode = force == 1 ? 0 : (force >= 2 ? force - 1 : punet_splitk_mode());
    const int full = ntiles / cap, T = ntiles - full * cap;
    if (mode == 0 || T == 0 || nq < 2 || scratch_bytes <= SK_COUNTER_BYTES || T * sizeof(unsigned) > SK_COUNTER_BYTES) return whole;
    const size_t max_slots = (scratch_bytes - SK_COUNTER_BYTES) / SK_SLOT_BYTES;
    const int Q = T * nq;
    int D = 0;
    if (mode > 0) {
        D = mode > nq ? nq : mode;
    } else {
        // Measured (profiles/r05/streamk_tail_probe.txt; unit = one chunk of one tile with the CU shared by two workgroups, ~4.3 us):
        // a workgroup alone on its CU runs its chunks in ~0.45 of that (whole tail of T <= cap / 2 tiles) resp. ~0.6 (a run of the
        // stream-K tail); a partial item costs ~3 for publishing its 64 KB + fence + ticket, the combine ~0.8 per part read.
        const double t_whole = nchunks * (2 * T <= cap ? 0.45 : 0.9);
        double best = 0.95 * t_whole;
        const int cand[2] = {(Q + cap - 1) / cap, (2 * Q + cap - 1) / cap};  // every resident workgroup / one workgroup per CU
        for (int i = 0; i < 2; ++i) {
            const int d = cand[i];
            if (d < 1 || d >= nq) continue;  // (a run of a whole tile or more: nothing to gain)
            const int p = (Q + d - 1) / d;
            const double t = 4.0 * d * (2 * p <= cap ? 0.6 : 1.0) + 3.0 + 0.8 * ((double)nq / d + 1.0);
            if (t < best) { best = t; D = d; }
        }
        if (D == 0) return whole;
    }
    const int P = (Q + D - 1) / D;
    if (P > cap || (size_t)2 * P > max_slots) return whole;
    return SkPlan{full * cap, P, D, Q};
}

extern "C" size_t punet_conv_splitk_scratch_bytes(void) {  // enough for every plan sk_plan can make on this device
    return SK_COUNTER_BYTES + (size_t)2 * WG_PER_CU * punet_num_cus() * SK_SLOT_BYTES;
}

extern "C" int punet_conv3x3_splitk(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                                    int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                                    const void* gate_bits, void* gate_out, float* pool_out, int accumulate, int relu, int B, int H, int W,
                                    int opts, int f16_planes, const float* amax_x0, const float* amax_x1, const float* amax_w,
                                    float* amax_y, void* scratch, size_t scratch_bytes, void* stream) {
    PUNET_CHECK_ARG(f16_planes >= 0 && f16_planes <= 2, "punet_conv3x3_splitk: f16_planes must be 0 (bf16 planes by `opts`), 1 or 2 (got %d)", f16_planes);
    PUNET_CHECK_ARG(!scratch || ((size_t)scratch % 16 == 0 && scratch_bytes >= SK_COUNTER_BYTES),
                    "punet_conv3x3_splitk: scratch must be 16-byte aligned and hold at least its %zu counter bytes", SK_COUNTER_BYTES);
    F16Scales f;
    f.planes = f16_planes; f.amax_x0 = amax_x0; f.amax_x1 = amax_x1; f.amax_w = amax_w; f.amax_out = amax_y;
    return conv_x6_launch(3, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                          gate_out, accumulate, relu, B, H, W, f16_planes ? (opts & 0xff0000) : opts, stream, pool_out, f, scratch, scratch_bytes);
}

extern "C" int punet_conv3x3_x6_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes,
                                      long plane_stride, int n_row, int n_off, const float* bias, int bias_img_stride,
                                      float* y, int n_out, const float* mask, const void* gate_bits, void* gate_out,
                                      int accumulate, int relu, int B, int H, int W, int opts, void* stream) {
    return conv_x6_launch(3, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                          gate_out, accumulate, relu, B, H, W, opts, stream);
}

extern "C" int punet_conv3x3_x6_pooled(const float* x0, int c0, const float* x1, int c1, const void* wplanes,
                                       long plane_stride, int n_row, int n_off, const float* bias, int bias_img_stride,
                                       float* y, int n_out, void* gate_out, float* pool_out, int relu, int B, int H, int W,
                                       int opts, void* stream) {
    PUNET_CHECK_ARG(pool_out, "punet_conv3x3_x6_pooled: null pool_out");
    return conv_x6_launch(3, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, nullptr, nullptr,
                          gate_out, 0, relu, B, H, W, opts, stream, pool_out);
}

extern "C" int punet_conv1x1_x6_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes,
                                      long plane_stride, int n_row, int n_off, const float* bias, int bias_img_stride,
                                      float* y, int n_out, const float* mask, const void* gate_bits, void* gate_out,
                                      int accumulate, int relu, int B, int H, int W, void* stream) {
    return conv_x6_launch(1, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                          gate_out, accumulate, relu, B, H, W, 0, stream);
}

// fp16-operand convolution / input gradient ("f16x3": planes = 2, "f16": planes = 1); everything else as punet_conv3x3_x6_gated /
// _pooled (pool_out nullable)
extern "C" int punet_conv3x3_f16(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                                 int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                                 const void* gate_bits, void* gate_out, float* pool_out, int accumulate, int relu, int B, int H, int W,
                                 int planes, const float* amax_x0, const float* amax_x1, const float* amax_w, float* amax_y,
                                 void* stream) {
    PUNET_CHECK_ARG(planes == 1 || planes == 2, "punet_conv3x3_f16: planes must be 2 or 1 (got %d)", planes);
    F16Scales f;
    f.planes = planes; f.amax_x0 = amax_x0; f.amax_x1 = amax_x1; f.amax_w = amax_w; f.amax_out = amax_y;
    return conv_x6_launch(3, x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                          gate_out, accumulate, relu, B, H, W, 0, stream, pool_out, f);
}

// One-plane forward convolution on bf16 activation tensors (the "bf16s" fast mode: opt-in, reduced precision): x0 / x1 / y /
// pool_out are [B,H,W,C] bf16; weights = plane 0 of punet_split_weights_x6; bias fp32; gate_out (nullable) as punet_conv3x3_x6_gated.
extern "C" int punet_conv3x3_bf16s(const void* x0, int c0, const void* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                                   int n_off, const float* bias, void* y, int n_out, void* gate_out, void* pool_out, int relu, int B,
                                   int H, int W, void* stream) {
    PUNET_CHECK_ARG(!gate_out || relu, "punet_conv3x3_bf16s: gate_out is the ReLU gate of the output and needs relu != 0");
    PUNET_CHECK_ARG(x0 && wplanes && y, "punet_conv3x3_bf16s: null pointer");
    PUNET_CHECK_ARG(c0 > 0 && c0 % CK == 0 && c1 >= 0 && c1 % CK == 0 && (c1 == 0 || x1),
                    "punet_conv3x3_bf16s: source channels must be multiples of %d (c0=%d c1=%d)", CK, c0, c1);
    PUNET_CHECK_ARG(n_out > 0 && n_out % 4 == 0 && n_off >= 0 && n_off + n_out <= n_row, "punet_conv3x3_bf16s: bad output-channel window");
    PUNET_CHECK_ARG(B > 0 && H > 0 && W > 0 && (!pool_out || (H % 2 == 0 && W % 2 == 0)), "punet_conv3x3_bf16s: bad shape %dx%d", H, W);
    {
        const unsigned long long px = (unsigned long long)H * W;
        const int cmax = c0 > c1 ? (c0 > n_out ? c0 : n_out) : (c1 > n_out ? c1 : n_out);
        PUNET_CHECK_ARG(px * cmax < (1ull << 31) && (unsigned long long)B * px < (1ull << 31), "punet_conv3x3_bf16s: image too large");
    }
    ConvX6Args a{};
    a.x0 = (const float*)x0; a.x1 = (const float*)x1; a.c0 = c0; a.c1 = c1;
    a.wp = (const __bf16*)wplanes; a.plane_stride = plane_stride; a.n_row = n_row; a.n_off = n_off;
    a.bias = bias; a.bias_img_stride = 0;
    a.y = (float*)y; a.n_total = n_out; a.relu = relu; a.pool_out = (float*)pool_out; a.gate_out = (unsigned*)gate_out;
    a.B = B; a.H = H; a.W = W;
    a.ntiles = B * ceil_div(H, TH) * ceil_div(W, TW) * ceil_div(n_out, TN);
    a.m_ntn = x6_magic(ceil_div(n_out, TN)); a.m_tx = x6_magic(ceil_div(W, TW)); a.m_ty = x6_magic(ceil_div(H, TH));
    const int cap = WG_PER_CU * punet_num_cus();
    const int grid = a.ntiles < cap ? a.ntiles : cap;
    a.xcd_remap = grid % 8 == 0;
    conv3x3_x6_kernel<1, false, 3, true><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    PUNET_LAUNCH_CHECK("conv3x3_bf16s");
    return PUNET_OK;
}

// One-plane forward convolution on fp16 activation tensors (the "f16s" fast mode): x0 / x1 / y / pool_out are [B,H,W,C] fp16,
// UNscaled (as under the reference's fp16 autocast; outputs saturate at fp16's largest finite value); weights = plane 0 of
// punet_split_weights_f16 with their slab's amax slot amax_w.
extern "C" int punet_conv3x3_f16s(const void* x0, int c0, const void* x1, int c1, const void* wplanes, long plane_stride, int n_row,
                                  int n_off, const float* bias, void* y, int n_out, void* gate_out, void* pool_out, int relu, int B,
                                  int H, int W, const float* amax_w, void* stream) {
    PUNET_CHECK_ARG(!gate_out || relu, "punet_conv3x3_f16s: gate_out is the ReLU gate of the output and needs relu != 0");
    PUNET_CHECK_ARG(x0 && wplanes && y && amax_w, "punet_conv3x3_f16s: null pointer");
    PUNET_CHECK_ARG(c0 > 0 && c0 % CK == 0 && c1 >= 0 && c1 % CK == 0 && (c1 == 0 || x1),
                    "punet_conv3x3_f16s: source channels must be multiples of %d (c0=%d c1=%d)", CK, c0, c1);
    PUNET_CHECK_ARG(n_out > 0 && n_out % 4 == 0 && n_off >= 0 && n_off + n_out <= n_row, "punet_conv3x3_f16s: bad output-channel window");
    PUNET_CHECK_ARG(B > 0 && H > 0 && W > 0 && (!pool_out || (H % 2 == 0 && W % 2 == 0)), "punet_conv3x3_f16s: bad shape %dx%d", H, W);
    {
        const unsigned long long px = (unsigned long long)H * W;
        const int cmax = c0 > c1 ? (c0 > n_out ? c0 : n_out) : (c1 > n_out ? c1 : n_out);
        PUNET_CHECK_ARG(px * cmax < (1ull << 31) && (unsigned long long)B * px < (1ull << 31), "punet_conv3x3_f16s: image too large");
    }
    ConvX6Args a{};
    a.x0 = (const float*)x0; a.x1 = (const float*)x1; a.c0 = c0; a.c1 = c1;
    a.wp = (const __bf16*)wplanes; a.plane_stride = plane_stride; a.n_row = n_row; a.n_off = n_off;
    a.bias = bias; a.bias_img_stride = 0;
    a.y = (float*)y; a.n_total = n_out; a.relu = relu; a.pool_out = (float*)pool_out; a.gate_out = (unsigned*)gate_out;
    a.B = B; a.H = H; a.W = W;
    a.ntiles = B * ceil_div(H, TH) * ceil_div(W, TW) * ceil_div(n_out, TN);
    a.m_ntn = x6_magic(ceil_div(n_out, TN)); a.m_tx = x6_magic(ceil_div(W, TW)); a.m_ty = x6_magic(ceil_div(H, TH));
    a.amax_w = amax_w;
    const int cap = WG_PER_CU * punet_num_cus();
    const int grid = a.ntiles < cap ? a.ntiles : cap;
    a.xcd_remap = grid % 8 == 0;
    conv3x3_x6_kernel<1, false, 3, true, true><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    PUNET_LAUNCH_CHECK("conv3x3_f16s");
    return PUNET_OK;
}

static int conv_x6_launch(int ksize, const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride,
                          int n_row, int n_off, const float* bias, int bias_img_stride, float* y, int n_out, const float* mask,
                          const void* gate_bits, void* gate_out, int accumulate, int relu, int B, int H, int W, int opts,
                          void* stream, float* pool_out, const F16Scales& f16, void* scratch, size_t scratch_bytes) {
    PUNET_CHECK_ARG(x0 && wplanes && y, "punet_conv3x3_x6: null pointer");
    // A 1x1 convolution does not see the image geometry: where nothing else refers to it (no gate bits in or out, no fp32 mask, no
    // per-image bias) the launch walks the flat list of B * H * W pixels as ONE image of 32-pixel rows - no partial tile columns
    // (K5x's GEMMs at 704 x 704 run on 44 / 88 / 176-pixel-wide tensors: 1.4 / 2.75 / 5.5 tiles; round 5).  Same products in the
    // same order per pixel: bit-identical.
    if (ksize == 1 && !mask && !gate_bits && !gate_out && !pool_out && bias_img_stride == 0 && B > 0 && H > 0 && W > 0 && W % TW != 0 &&
        ((long)B * H * W) % TW == 0 && (long)B * H * W / TW < (1l << 24) &&
        (unsigned long long)B * H * W * (c0 > n_out ? c0 : n_out) < (1ull << 31)) {
        H = (int)((long)B * H * W / TW);
        W = TW;
        B = 1;
    }
    PUNET_CHECK_ARG(f16.planes == 0 || ((f16.planes == 1 || f16.planes == 2) && ksize == 3),
                    "punet_conv3x3_f16: planes must be 2 (hh + hl + lh) or 1, 3x3 only");
    const int planes = punet_x6_planes_of(opts), shape = punet_x6_shape_of(opts);
    PUNET_CHECK_ARG(planes > 0 && shape > 0, "punet_conv3x3_x6: bad options word 0x%x (PUNET_X6_OPTS(products 6|3|1, shape 16|32))", opts);
    PUNET_CHECK_ARG(!pool_out || (H % 2 == 0 && W % 2 == 0 && !mask && !gate_bits && !accumulate),
                    "punet_conv3x3_x6_pooled: the fused 2x2 average needs even H and W (%dx%d) and the forward epilogue", H, W);
    PUNET_CHECK_ARG(c0 > 0 && c0 % CK == 0 && c1 >= 0 && c1 % CK == 0 && (c1 == 0 || x1),
                    "punet_conv3x3_x6: source channels must be multiples of %d (c0=%d c1=%d)", CK, c0, c1);
    PUNET_CHECK_ARG(n_out > 0 && n_out % 4 == 0 && n_off >= 0 && n_off + n_out <= n_row,
                    "punet_conv3x3_x6: bad output-channel window (n_out must be a multiple of 4)");
    PUNET_CHECK_ARG(B > 0 && H > 0 && W > 0, "punet_conv3x3_x6: bad shape");
    {   // 64-bit image base + 32-bit offsets inside an image: ONE image (not the batch) must stay below 2^31 elements
        const unsigned long long px = (unsigned long long)H * W;
        const int cmax = c0 > c1 ? (c0 > n_out ? c0 : n_out) : (c1 > n_out ? c1 : n_out);
        PUNET_CHECK_ARG(px * cmax < (1ull << 31) && 3ull * (unsigned long long)plane_stride < (1ull << 31),
                        "punet_conv3x3_x6: one image is too large for 32-bit offsets (%llu pixels x %d channels)", px, cmax);
        PUNET_CHECK_ARG((unsigned long long)B * H * ceil_div(W, TW) * ceil_div(n_out, TN) * 64ull < (1ull << 32) &&
                        (unsigned long long)B * px < (1ull << 31), "punet_conv3x3_x6: too many pixels for the tile / gate indices");
        // the tile decode divides by multiplication with 32-bit reciprocals: exact while tile index x divisor < 2^32
        const unsigned long long nt = (unsigned long long)B * ceil_div(H, TH) * ceil_div(W, TW) * ceil_div(n_out, TN);
        const int dmax = ceil_div(H, TH) > ceil_div(W, TW) ? ceil_div(H, TH) : ceil_div(W, TW);
        PUNET_CHECK_ARG(nt * (unsigned long long)(dmax > ceil_div(n_out, TN) ? dmax : ceil_div(n_out, TN)) < (1ull << 32),
                        "punet_conv3x3_x6: too many tiles for the reciprocal tile decode");
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
    a.m_ntn = x6_magic(ceil_div(n_out, TN)); a.m_tx = x6_magic(ceil_div(W, TW)); a.m_ty = x6_magic(ceil_div(H, TH));
    const int cap = WG_PER_CU * punet_num_cus();
    const int grid = a.ntiles < cap ? a.ntiles : cap;
    // the input-gradient epilogue whenever one of its features is asked for; bias / ReLU / gate_out belong to the forward
    const bool dgrad = mask || gate_bits || accumulate;
    PUNET_CHECK_ARG(!(dgrad && (bias || relu || gate_out)),
                    "punet_conv3x3_x6: bias/relu/gate_out (forward) cannot be combined with mask/gate_bits/accumulate (input gradient)");
    hipStream_t st = (hipStream_t)stream;
    a.xcd_remap = grid % 8 == 0;  // same-process A/B: +0.5-0.8 % (the tensors of this workload fit the Infinity Cache)
    a.amax_x0 = f16.amax_x0; a.amax_x1 = f16.amax_x1; a.amax_w = f16.amax_w; a.amax_out = f16.amax_out;
    // conv3x3_x6b_kernel deals work items: whole tiles, then runs of the last partial round's quads (sk_plan)
    a.sk_whole = a.ntiles; a.sk_P = 0; a.sk_D = 1; a.sk_Q = 0; a.m_D = 0u; a.m_nq = 0u;
    int gridb = grid;
    if (ksize == 3 && scratch && (c0 + c1) % 32 == 0) {
        const SkPlan pl = sk_plan(a.ntiles, cap, (c0 + c1) / CK, scratch_bytes, (opts >> 16) & 0xff);
        if (pl.P > 0) {
            a.sk_whole = pl.whole; a.sk_P = pl.P; a.sk_D = pl.D; a.sk_Q = pl.Q;
            a.m_D = pl.D > 1 ? x6_magic(pl.D) : 0u; a.m_nq = x6_magic((c0 + c1) / 32);
            a.sk_cnt = (unsigned*)scratch;
            a.sk_ws = (float*)((char*)scratch + SK_COUNTER_BYTES);
            gridb = pl.whole >= cap ? cap : (pl.whole > pl.P ? pl.whole : pl.P);
        }
    }
    // FOLD: tiles over the virtual row of all B images when the width is not whole 32-pixel tiles but B * W is (kernel comment)
    bool fold = false;
    if (ksize == 3 && shape == 16 && (c0 + c1) % 32 == 0 && (f16.planes == 0 ? planes == 3 : f16.planes == 2) && punet_fold_mode() != 0 && !((opts >> 24) & 1) &&
        W % TW != 0 && W >= TW && ceil_div(B * W, TW) < B * ceil_div(W, TW) && (!pool_out || W % 2 == 0) &&
        2ull * (unsigned long long)H * W * (c0 > c1 ? (c0 > n_out ? c0 : n_out) : (c1 > n_out ? c1 : n_out)) < (1ull << 31)) {
        fold = true;
        a.vtiles_x = ceil_div(B * W, TW);
        a.m_W = x6_magic(W);
        a.m_tx = x6_magic(a.vtiles_x);
        a.ntiles = ceil_div(H, TH) * a.vtiles_x * ceil_div(n_out, TN);
    }
    if (fold) {  // (the tile count changed: plan the stream-K tail again)
        a.sk_whole = a.ntiles; a.sk_P = 0; a.sk_D = 1; a.sk_Q = 0; a.m_D = 0u; a.m_nq = 0u;
        gridb = a.ntiles < cap ? a.ntiles : cap;
        if (scratch) {
            const SkPlan pl = sk_plan(a.ntiles, cap, (c0 + c1) / CK, scratch_bytes, (opts >> 16) & 0xff);
            if (pl.P > 0) {
                a.sk_whole = pl.whole; a.sk_P = pl.P; a.sk_D = pl.D; a.sk_Q = pl.Q;
                a.m_D = pl.D > 1 ? x6_magic(pl.D) : 0u; a.m_nq = x6_magic((c0 + c1) / 32);
                a.sk_cnt = (unsigned*)scratch;
                a.sk_ws = (float*)((char*)scratch + SK_COUNTER_BYTES);
                gridb = pl.whole >= cap ? cap : (pl.whole > pl.P ? pl.whole : pl.P);
            }
        }
    }
    const int xrb = gridb % 8 == 0;
    if (f16.planes == 2) {  // "f16x3": the 16 x 16 x 32 kernel where the channel count allows, as for the bf16 planes
        if (shape == 16 && (c0 + c1) % 32 == 0) {
            a.xcd_remap = xrb;
            if (fold) {
                if (dgrad) conv3x3_x6b_kernel<2, true, true, true><<<gridb, 256, 0, st>>>(a); else conv3x3_x6b_kernel<2, false, true, true><<<gridb, 256, 0, st>>>(a);
            } else {
                if (dgrad) conv3x3_x6b_kernel<2, true, true><<<gridb, 256, 0, st>>>(a); else conv3x3_x6b_kernel<2, false, true><<<gridb, 256, 0, st>>>(a);
            }
        } else {
            if (dgrad) conv3x3_x6_kernel<2, true, 3, false, true><<<grid, 256, 0, st>>>(a);
            else conv3x3_x6_kernel<2, false, 3, false, true><<<grid, 256, 0, st>>>(a);
        }
    } else if (f16.planes == 1) {  // "f16": one fp16 product
        if (dgrad) conv3x3_x6_kernel<1, true, 3, false, true><<<grid, 256, 0, st>>>(a);
        else conv3x3_x6_kernel<1, false, 3, false, true><<<grid, 256, 0, st>>>(a);
    } else if (ksize == 1 && c1 == 0 && c0 % 32 == 0 && punet_conv1x1_wide()) {  // K5x's GEMMs AND the Fcomb layers whose contraction is a
        // multiple of 32 (K = 32 / 64, windows n_off > 0, per-image bias: tests/test_gpu_x6.py::test_conv1x1_x6_*)
#ifdef X6W_TRACE
        if (const char* tp = getenv("PUNET_X6W_TRACE_PTR")) a.pool_out = reinterpret_cast<float*>(strtoull(tp, nullptr, 0));
#endif
        if (n_out >= 128 && punet_conv1x1_wide() >= 2) {  // 128-channel tiles, 16-channel phases (fewer LDS bytes per MFMA)
            a.ntiles = B * ceil_div(H, TH) * ceil_div(W, TW) * ceil_div(n_out, 128);
            const int grid2 = a.ntiles < cap ? a.ntiles : cap;
            if (dgrad) conv1x1_x6w_kernel<true, 4><<<grid2, 256, 0, st>>>(a); else conv1x1_x6w_kernel<false, 4><<<grid2, 256, 0, st>>>(a);
        } else {
            if (dgrad) conv1x1_x6w_kernel<true, 2><<<grid, 256, 0, st>>>(a); else conv1x1_x6w_kernel<false, 2><<<grid, 256, 0, st>>>(a);
        }
    } else if (ksize == 1) {  // the Fcomb layers: always the parity arithmetic (they are a rounding error of the step time)
        if (dgrad) conv3x3_x6_kernel<3, true, 1><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<3, false, 1><<<grid, 256, 0, st>>>(a);
    } else if (shape == 16 && (c0 + c1) % 32 == 0 && planes >= 2) {
        // 16 x 16 x 32 MFMAs: four {tap, 8 channels} blocks per instruction.  (Not for the one-plane fast mode: that one is
        // bound by its staging, not by the matrix cores, and measured 15 % slower through this kernel.)
        a.xcd_remap = xrb;
        if (planes == 3) {
            if (fold) {
                if (dgrad) conv3x3_x6b_kernel<3, true, false, true><<<gridb, 256, 0, st>>>(a); else conv3x3_x6b_kernel<3, false, false, true><<<gridb, 256, 0, st>>>(a);
            } else {
                if (dgrad) conv3x3_x6b_kernel<3, true><<<gridb, 256, 0, st>>>(a); else conv3x3_x6b_kernel<3, false><<<gridb, 256, 0, st>>>(a);
            }
        } else {
            if (dgrad) conv3x3_x6b_kernel<2, true><<<gridb, 256, 0, st>>>(a); else conv3x3_x6b_kernel<2, false><<<gridb, 256, 0, st>>>(a);
        }
    } else if (planes == 3) {
        if (dgrad) conv3x3_x6_kernel<3, true><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<3, false><<<grid, 256, 0, st>>>(a);
    } else if (planes == 2) {
        if (dgrad) conv3x3_x6_kernel<2, true><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<2, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (dgrad) conv3x3_x6_kernel<1, true><<<grid, 256, 0, st>>>(a); else conv3x3_x6_kernel<1, false><<<grid, 256, 0, st>>>(a);
    }
    PUNET_LAUNCH_CHECK("conv3x3_x6");
    return PUNET_OK;
}

#ifdef X6B_TRACE_MAIN  // standalone trace driver: 256 -> 256 channels at 128 x 128, batch 4 (tools/trace_conv.sh)
#include <cstdio>
#include <vector>
int main() {
    const int B = 4, H = 128, W = 128, C = 256;
    const size_t n = (size_t)B * H * W * C;
    std::vector<float> h(n);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - (s & 1 ? 0.5f : 1.5f); }
    float *x, *y, *w, *b;
    __bf16* planes;
    const size_t nw = (size_t)9 * C * C;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, nw * 4); hipMalloc(&b, C * 4); hipMalloc(&planes, 3 * nw * 2);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, C * 4);
    punet_split_weights_x6(w, planes, (long)nw, C, C, 3, 0, nullptr);
    for (int r = 0; r < 3; ++r) {
        const int rc = punet_conv3x3_x6(x, C, nullptr, 0, planes, (long)nw, C, 0, b, 0, y, C, nullptr, 0, 1, B, H, W, 0, nullptr);
        if (rc) { printf("error %d: %s\n", rc, punet_last_error()); return 1; }
    }
    hipDeviceSynchronize();
    unsigned long long t[4 * 16];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(punet_x6b_trace), sizeof(t));
    printf("chunks 8..11 of workgroup 9: top barrier passed / operand barrier passed / K-steps issued, shader cycles rel. to wave 0\n");
    for (int wv = 0; wv < 4; ++wv) {
        printf("wave %d:", wv);
        for (int i = 0; i < 12; ++i) printf(" %6lld%s", (long long)(t[wv * 16 + i] - t[0]), i % 3 == 2 ? " |" : "");
        printf("\n");
    }
    return 0;
}
#endif
