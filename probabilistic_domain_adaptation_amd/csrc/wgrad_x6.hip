// K3x6: weight (+ bias) gradient of the 3x3 NHWC convolutions at fp32 accuracy on the BF16 matrix cores.
//
// Same split-K structure, slab layout and deterministic second pass as wgrad_mfma.hip; the arithmetic is the
// three-plane bf16 split of conv_x6.hip (x = h + m + l, products hh + hm + mh + hl + lh + mm, fp32 accumulate).
//
//   dW[tap][ci][co] = sum_pixels X[pixel + tap][ci] * dZ[pixel][co]
//
// Per tap a GEMM with M = ci, N = co and the PIXELS as the contracted index: the MFMA operands are "8 consecutive pixels
// of one channel", i.e. the transpose of the NHWC images.  The LDS images stay [pixel][32 channels] (64-byte rows, one
// image per 32-channel block and per bf16 plane) and the operands are fetched with ds_read_b64_tr_b16, the gfx950
// transposing LDS read: a 16-lane group reads a 4-pixel x 16-channel block and each lane receives its channel's four
// pixels.  Tap shifts are whole-row offsets, so every read stays aligned; 4 rows x 2 channel groups of a half-wave
// cover the 64 banks exactly once (conflict-free).
// Workgroup = 8 waves = 4 (ci block, co block) pairs x 2 tap groups (taps 0-4 | 5-8), as in the fp32 kernel.
// The bias gradient (exact fp32 column sums of dZ) is accumulated while dZ is being staged.
#include "common.h"
#include "punet_hip.h"

// -DPUNET_WGRAD_TRACE: s_memtime stamps of one workgroup's tile (tools/trace_wgrad.sh); compiled out otherwise.  Only at
// the region boundaries (loop top 0, barrier reached 10 / passed 11): a stamp inside the MFMA stream pins the schedule and
// no longer measures the product kernel (the compiler interleaves the staging with the MFMAs of a pixel group).
#ifdef PUNET_WGRAD_TRACE
__device__ unsigned long long punet_wgrad_trace[8 * 32];
#define WGT(id)                                                                                          \
    do {                                                                                                 \
        if (blockIdx.x == 7 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && trace_iter == 5)            \
            punet_wgrad_trace[(threadIdx.x >> 6) * 32 + (id)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
#else
#define WGT(id)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int WTH = 2, WTW = 32, WC = 64;
// experiment knobs (tools/ab_variants.sh builds one library per setting): taps owned by waves 0-3 (the rest go to waves
// 4-7) and a static s_setprio for one half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4)
#ifndef PUNET_WGRAD_T0
#define PUNET_WGRAD_T0 5
#endif
#ifndef PUNET_WGRAD_PRIO
#define PUNET_WGRAD_PRIO 1  // 0: none, 1: waves 4-7 at priority 1 (-0.1 ms per step in the rolling kernel), 2: waves 0-3 at 1, 3: waves 4-7 at 3
#endif
#ifndef PUNET_WGRAD_TIMING
#define PUNET_WGRAD_TIMING 0
#endif
#ifndef PUNET_WGRAD_PREFETCH  // n >= 1: fragment reads n steps ahead of the MFMAs (hand-pipelined); 0: left to the scheduler
#define PUNET_WGRAD_PREFETCH 1
#endif
#ifndef PUNET_WGRAD_SB  // scheduling regions of the rolling kernel's tile: 1 = one per pixel group, 2 = per two, 0 = whole tile
#define PUNET_WGRAD_SB 0  // (one region per pixel group was needed at 255 VGPRs; at 180 the free schedule is 0.5-1 % faster)
#endif
constexpr int GT0 = PUNET_WGRAD_T0, GT1 = 9 - GT0, GTMAX = GT0 > GT1 ? GT0 : GT1;
constexpr int HTH = WTH + 2, HTW = WTW + 2, HTW_BASE = HTW;
constexpr int NPX = HTH * HTW;   // 136 halo pixels
constexpr int NPD = WTH * WTW;   // 64 pixels
constexpr int ROWB = 64;         // bytes per [pixel][32 ch] bf16 row
constexpr int XIMG = NPX * ROWB;  // one (plane, ci block) image
constexpr int DIMG = NPD * ROWB;  // one (plane, co block) image

struct WgradX6Args {
    const float* x;
    int cs;
    const float* dz;
    int cout;
    float* partial;
    float* bias_partial;
    int B, H, W, nslices;
    const float* amax_x;   // fp16-operand instantiations (F16; common.h): amax slots of the two operands, nullptr = scale 1
    const float* amax_dz;
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

union Frag {
    bf16x8 v;
    s16x4 half4[2];
    uint4 u;
};

__device__ __forceinline__ s16x4 tr_read(const char* lds_base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_base + byte_off));
}

// taps [T0, T0 + NT) of one staged tile.  aoff/boff: this lane's byte offset inside an image (see kernel body).
// `between(s)` is called after the MFMAs of pixel group s have been issued: the kernel stages the NEXT tile there (split +
// LDS writes into the other image set), i.e. in the same instruction stream as the MFMAs - a wave's own VALU / LDS work
// issues in the shadow of its own MFMAs, whereas a separate staging phase leaves the matrix pipe idle.
template <int NP, int T0, int NT, typename F>
__device__ __forceinline__ void tile_mfma(f32x16 (&acc)[GTMAX], const char* sX, const char* sD, int cib, int cob, int aoff, int boff,
                                          F&& between) {
    constexpr int P1 = NP > 1 ? 1 : 0, P2 = NP > 2 ? 2 : 0;  // plane indices that exist for this NP
#pragma unroll
    for (int s = 0; s < NPD / 16; ++s) {
        const int row = (16 * s) / WTW, col0 = (16 * s) % WTW;
        Frag b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                b[p].half4[r] = tr_read(sD, (p * 2 + cob) * DIMG + (16 * s + 4 * r) * ROWB + boff);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = T0 + tt, ky = t / 3, kx = t % 3;
            Frag a[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    a[p].half4[r] = tr_read(sX, (p * 2 + cib) * XIMG + ((row + ky) * HTW + col0 + kx + 4 * r) * ROWB + aoff);
            f32x16 c = acc[tt];
            if (NP > 2) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P1].v, b[P1].v, c, 0, 0, 0);  // mm
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P2].v, b[0].v, c, 0, 0, 0);   // lh
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[P2].v, c, 0, 0, 0);   // hl
            }
            if (NP > 1) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P1].v, b[0].v, c, 0, 0, 0);  // mh
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[P1].v, c, 0, 0, 0);  // hm
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, c, 0, 0, 0);  // hh
            acc[tt] = c;
        }
        between(s);
        __builtin_amdgcn_sched_barrier(0);  // keep the four unrolled pixel groups apart (hoisted fragment reads cost registers)
    }
}

template <int T0, int NT>
__device__ __forceinline__ void store_partial(const f32x16 (&acc)[GTMAX], const WgradX6Args& a, int slice, int ci_base, int co, int half,
                                              float inv = 1.f) {  // inv: exact power of two (fp16 operands), 1 otherwise
    if (co >= a.cout) return;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci_base + mfma32_row(r, half);
            if (ci < a.cs) a.partial[(((size_t)slice * 9 + T0 + tt) * a.cs + ci) * a.cout + co] = acc[tt][r] * inv;
        }
}

// NP = bf16 planes per operand (3: six products, fp32-accurate; 2: hh + hm + mh; 1: plain bf16), see conv_x6.hip.
// ONE workgroup per CU (2 waves per SIMD, 200 VGPRs), two operand image sets in LDS (2 x 76,800 B at NP = 3): the next
// tile's global loads are issued into registers BEFORE the MFMA phase of the current tile and split / written into the
// other image set right after it, so a tile costs one barrier and no exposed HBM latency.  (The first version was
// single-buffered with two workgroups per CU covering each other's staging; same-process A/B: this one is 2.4 % faster
// and, at half the split-K slices, halves the slab traffic of the reduction pass.)
template <int NP>
__global__ __launch_bounds__(512, 2) void wgrad3x3_x6_kernel(WgradX6Args a) {
    constexpr int XUNITS = NPX * (WC / 4), DUNITS = NPD * (WC / 4);
    constexpr int XIT = (XUNITS + 511) / 512, DIT = (DUNITS + 511) / 512;
    constexpr int BUF = NP * 2 * (XIMG + DIMG);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sBias = reinterpret_cast<float*>(smem);  // reused after the last tile

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int pair = wave & 3, group = wave >> 2;
    const int cib = pair >> 1, cob = pair & 1;

    const int co_tiles = (a.cout + WC - 1) / WC;
    const int ci_tile = blockIdx.y / co_tiles, co_tile = blockIdx.y % co_tiles;
    const int ci0 = ci_tile * WC, co0 = co_tile * WC;
    const int slice = blockIdx.x;
    const int tiles_x = (a.W + WTW - 1) / WTW, tiles_y = (a.H + WTH - 1) / WTH;
    const int ntiles = a.B * tiles_x * tiles_y;

    f32x16 acc[GTMAX];
#pragma unroll
    for (int t = 0; t < GTMAX; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    const int l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, cb = (lane >> 4) & 1;
    const int lane_off = (8 * half + q) * ROWB + (16 * cb + 4 * p4) * 2;

    float4 rx[XIT], rd[DIT];
    unsigned inmask = 0u;  // bit per staging register: its unit lies inside the image / channel range
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        int sid = tile;
        const int txi = sid % tiles_x;
        sid /= tiles_x;
        const int tyi = sid % tiles_y;
        const int b = sid / tiles_y;
        const int ty0 = tyi * WTH, tx0 = txi * WTW;
        // branch-free (clamped address + select) with 32-bit offsets: one basic block, exact vmcnt waits.  The image base is
        // a uniform 64-bit term, so only ONE image (not the batch) has to stay below 2^31 elements.
        const float* const xb = a.x + (size_t)b * ((size_t)(a.H * a.W) * a.cs);
        const float* const db = a.dz + (size_t)b * ((size_t)(a.H * a.W) * a.cout);
        inmask = 0u;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int u = min(tid + it * 512, XUNITS - 1);
            const int c4 = u % (WC / 4), pix = u / (WC / 4);
            const int py = pix / HTW, px = pix % HTW;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            const int c = ci0 + c4 * 4;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.cs;
            const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1), cc = min(c, a.cs - 4);
            rx[it] = *reinterpret_cast<const float4*>(xb + (unsigned)((cy * a.W + cx) * a.cs + cc));
            inmask = in ? inmask | (1u << it) : inmask;  // zeroing happens at the consumer: no wait on the load just issued
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int u = tid + it * 512;
            const int c4 = u % (WC / 4), pix = u / (WC / 4);
            const int py = pix / WTW, px = pix % WTW;
            const int gy = ty0 + py, gx = tx0 + px;
            const int c = co0 + c4 * 4;
            const bool in = gy < a.H && gx < a.W && c < a.cout;
            const int cy = min(gy, a.H - 1), cx = min(gx, a.W - 1), cc = min(c, a.cout - 4);
            rd[it] = *reinterpret_cast<const float4*>(db + (unsigned)((cy * a.W + cx) * a.cout + cc));
            inmask = in ? inmask | (1u << (XIT + it)) : inmask;
        }
    };
    static_assert(DUNITS % 512 == 0, "every dZ unit is owned by exactly one thread (the bias sums rely on it)");
    auto commit_x = [&](int buf, int it) __attribute__((always_inline)) {
        char* sX = smem + buf * BUF;
        const int u = min(tid + it * 512, XUNITS - 1);  // a clamped duplicate re-writes the last unit with the same data
        const int c4 = u % (WC / 4), pix = u / (WC / 4);
        const float4 r = (inmask >> it) & 1u ? rx[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float xv[4] = {r.x, r.y, r.z, r.w};
        bf16x4 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h_, m_, l_;
            split3(xv[e], h_, m_, l_);
            ph[e] = h_; pm[e] = m_; pl[e] = l_;
        }
        const int off = (c4 >> 3) * XIMG + pix * ROWB + (c4 & 7) * 8;
        *reinterpret_cast<bf16x4*>(sX + 0 * 2 * XIMG + off) = ph;
        if (NP > 1) *reinterpret_cast<bf16x4*>(sX + (NP > 1) * 2 * XIMG + off) = pm;
        if (NP > 2) *reinterpret_cast<bf16x4*>(sX + (NP > 2) * 4 * XIMG + off) = pl;
    };
    auto commit_d = [&](int buf, int it) __attribute__((always_inline)) {
        char* sD = smem + buf * BUF + NP * 2 * XIMG;
        const int u = tid + it * 512;
        const int c4 = u % (WC / 4), pix = u / (WC / 4);
        const float4 v = (inmask >> (XIT + it)) & 1u ? rd[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        const float xv[4] = {v.x, v.y, v.z, v.w};
        bf16x4 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h_, m_, l_;
            split3(xv[e], h_, m_, l_);
            ph[e] = h_; pm[e] = m_; pl[e] = l_;
        }
        const int off = (c4 >> 3) * DIMG + pix * ROWB + (c4 & 7) * 8;
        *reinterpret_cast<bf16x4*>(sD + 0 * 2 * DIMG + off) = ph;
        if (NP > 1) *reinterpret_cast<bf16x4*>(sD + (NP > 1) * 2 * DIMG + off) = pm;
        if (NP > 2) *reinterpret_cast<bf16x4*>(sD + (NP > 2) * 4 * DIMG + off) = pl;
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) commit_x(buf, it);
#pragma unroll
        for (int it = 0; it < DIT; ++it) commit_d(buf, it);
    };

    int cur = 0;
#ifdef PUNET_WGRAD_TRACE
    int trace_iter = 0;
#endif
#if PUNET_WGRAD_PRIO == 1
    if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
#elif PUNET_WGRAD_PRIO == 2
    if (__builtin_amdgcn_readfirstlane(tid) < 256) __builtin_amdgcn_s_setprio(1);
#endif
    if (slice < ntiles) {
        prefetch(slice);
        commit(0);
    }
    __syncthreads();
    for (int tile = slice; tile < ntiles; tile += a.nslices) {
        const int next = tile + a.nslices;
        const bool has_next = next < ntiles;  // block-uniform
        WGT(0);
        prefetch(has_next ? next : tile);     // unconditional (the loop body stays one basic block); a re-fetch is dropped
        const char* sX = smem + cur * BUF;
        const char* sD = sX + NP * 2 * XIMG;
        const int nbuf = cur ^ 1;  // the other image set: its last readers passed the previous barrier
        // the next tile is staged between the four pixel groups of this tile's MFMAs, late enough for its loads to land
        auto between = [&](int s) __attribute__((always_inline)) {
            if (s == 1) { commit_x(nbuf, 0); commit_x(nbuf, 1); }
            if (s == 2) { commit_x(nbuf, 2); commit_x(nbuf, 3); }
            if (s == 3) {
#pragma unroll
                for (int it = 4; it < XIT; ++it) commit_x(nbuf, it);
#pragma unroll
                for (int it = 0; it < DIT; ++it) commit_d(nbuf, it);
            }
        };
        static_assert(XIT >= 4 && NPD / 16 == 4, "staging schedule assumes four pixel groups and at least four X units");
        if (has_next) {
            if (group == 0)
                tile_mfma<NP, 0, GT0>(acc, sX, sD, cib, cob, lane_off, lane_off, between);
            else
                tile_mfma<NP, GT0, GT1>(acc, sX, sD, cib, cob, lane_off, lane_off, between);
        } else {  // last tile of this slice: nothing to stage (and the dZ column sums must not count a tile twice)
            auto nothing = [](int) {};
            if (group == 0)
                tile_mfma<NP, 0, GT0>(acc, sX, sD, cib, cob, lane_off, lane_off, nothing);
            else
                tile_mfma<NP, GT0, GT1>(acc, sX, sD, cib, cob, lane_off, lane_off, nothing);
        }
        WGT(10);
        __syncthreads();
        WGT(11);
        cur ^= 1;
#ifdef PUNET_WGRAD_TRACE
        ++trace_iter;
#endif
    }

    const int co = co0 + cob * 32 + l31;
    if (group == 0)
        store_partial<0, GT0>(acc, a, slice, ci0 + cib * 32, co, half);
    else
        store_partial<GT0, GT1>(acc, a, slice, ci0 + cib * 32, co, half);

    if (a.bias_partial && ci_tile == 0) {  // block-uniform
        __syncthreads();
        sBias[tid * 4 + 0] = bsum.x; sBias[tid * 4 + 1] = bsum.y; sBias[tid * 4 + 2] = bsum.z; sBias[tid * 4 + 3] = bsum.w;
        __syncthreads();
        if (tid < WC) {
            const int c4 = tid >> 2, e = tid & 3;
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += sBias[(k * 16 + c4) * 4 + e];
            if (co0 + tid < a.cout) a.bias_partial[(size_t)slice * a.cout + co0 + tid] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 2: the same kernel with a ROLLING ROW BUFFER.  The kernel above re-stages the whole 4 x 34-pixel halo for every
// 2 x 32-pixel tile, i.e. every input row is loaded, split into planes and written to LDS 2.1 times (once as a tile's own
// rows, once as its neighbour's halo).  Here a workgroup sweeps DOWN a 32-pixel column strip: the X rows live in a ring of
// six row slots (four in use, two being filled), the dZ rows in a ring of four, and a tile stages only its successor's two
// NEW rows of each operand - 68 + 64 instead of 136 + 64 pixels per tile (-34 % of the loads, splits and LDS writes that
// share the instruction stream with the MFMAs), 127.5 KB of LDS instead of 153.6 KB.  A split-K slice owns a CONTIGUOUS run
// of tiles in strip-major order; where a run starts (slice start, top of a strip) its first tile is staged in full by a
// prologue.  Fragment reads take the ring slot of their row from four (X) + two (dZ) per-tile base offsets.
constexpr int XSLOTS = 6, DSLOTS = 4;
// The staging writes are 8 bytes per lane and the 16 lanes of a ds_write_b64 group are one pixel's 16 channel quads: quads 0-7 go
// to the image of channel block 0, quads 8-15 to block 1's.  Every ds_write maps onto 32 banks (128 bytes): with images a multiple
// of 128 bytes apart the two halves of a group shared 16 banks (2 LDS cycles per group instead of 1; SQ_LDS_BANK_CONFLICT was 13 %
// of this kernel's LDS cycles, profiles/r05/pmc_lds_conflicts_adamatch_704x704x3.json).  64 bytes of padding per image put block
// 1's quads on the other 16 banks.  (The fragment reads of a wave stay inside one image: unaffected.)
#ifndef PUNET_WGRAD_IMGPAD
#define PUNET_WGRAD_IMGPAD 64
#endif
constexpr int XIMGR = XSLOTS * HTW * ROWB + PUNET_WGRAD_IMGPAD;  // one (plane, ci block) ring image: 13,056 B + padding
constexpr int DIMGR = DSLOTS * WTW * ROWB + PUNET_WGRAD_IMGPAD;  // one (plane, co block) ring image: 8,192 B + padding

// xa / db: per-lane byte offsets of the tile's four X rows / two dZ rows, INCLUDING the wave's channel-block image (the
// remaining compile-time part - plane, column, tap - then fits the 16-bit offset field of the LDS instruction)
// XR = bytes of one (plane, ci block) ring image (XIMGR; larger in the FOLD instantiation, whose rows have two more pixels);
// xs (FOLD only): per lane, the byte shift of its pixels where they lie right of the image seam, by 16-pixel column half
template <int NP, int T0, int NT, bool F16, int XR, bool FOLD, typename F>
__device__ __forceinline__ void tile_mfma_r(f32x16 (&acc)[GTMAX], const char* sX, const char* sD, const int (&xa)[4],
                                            const int (&db)[2], const int (&xs)[2], F&& between) {
    constexpr int P1 = NP > 1 ? 1 : 0, P2 = NP > 2 ? 2 : 0;
    constexpr int NS = NPD / 16;
    constexpr int XIMGR = XR;  // (shadows the namespace constant inside this function)
#if PUNET_WGRAD_PREFETCH
    // Software-pipelined by hand: the fragments of step i + 1 (one tap of one 16-pixel group; the dZ fragments too where a
    // new pixel group begins) are read BEFORE the six MFMAs of step i are issued, into the other half of a double buffer.
    // Left to the scheduler the reads of a tap followed the previous tap's MFMAs and only their last ~170 cycles of
    // execution hid the LDS latency: R x 6, wait, M x 6, R x 6, wait, ... - the two waves of a SIMD kept its matrix pipe
    // 67 % busy (s_memtime, tools/trace_wgrad.sh).  Needs 24 more registers than the kernel had at 255 VGPRs.
    constexpr int D = PUNET_WGRAD_PREFETCH;  // steps of lookahead
    Frag a[D + 1][NP], b[2][NP];
    auto load_b = [&](int s, Frag(&bb)[NP]) __attribute__((always_inline)) {
        const int row = (16 * s) / WTW, col0 = (16 * s) % WTW;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 2; ++r) bb[p].half4[r] = tr_read(sD, p * 2 * DIMGR + (col0 + 4 * r) * ROWB + db[row]);
    };
    auto load_a = [&](int s, int tt, Frag(&aa)[NP]) __attribute__((always_inline)) {
        const int row = (16 * s) / WTW, col0 = (16 * s) % WTW;
        const int t = T0 + tt, ky = t / 3, kx = t % 3;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                aa[p].half4[r] = tr_read(sX, p * 2 * XIMGR + (col0 + kx + 4 * r) * ROWB + xa[row + ky] + (FOLD ? xs[col0 / 16] : 0));
    };
    load_b(0, b[0]);
#pragma unroll
    for (int j = 0; j < D; ++j) load_a(j / NT, j % NT, a[j % (D + 1)]);
#pragma unroll
    for (int i = 0; i < NS * NT; ++i) {
        const int s = i / NT, tt = i % NT;
        if (i + D < NS * NT) {
            const int s1 = (i + D) / NT, tt1 = (i + D) % NT;
            if (tt1 == 0) load_b(s1, b[s1 & 1]);
            load_a(s1, tt1, a[(i + D) % (D + 1)]);
        }
        __builtin_amdgcn_sched_barrier(0);  // the prefetch stays in front of this step's MFMAs
        const Frag(&aa)[NP] = a[i % (D + 1)];
        const Frag(&bb)[NP] = b[s & 1];
        f32x16 c = acc[tt];
        if (NP > 2) {
            c = punet_mfma32<F16>(aa[P1].u, bb[P1].u, c);  // mm
            c = punet_mfma32<F16>(aa[P2].u, bb[0].u, c);   // lh
            c = punet_mfma32<F16>(aa[0].u, bb[P2].u, c);   // hl
        }
        if (NP > 1) {
            c = punet_mfma32<F16>(aa[P1].u, bb[0].u, c);  // mh
            c = punet_mfma32<F16>(aa[0].u, bb[P1].u, c);  // hm
        }
        c = punet_mfma32<F16>(aa[0].u, bb[0].u, c);  // hh
        acc[tt] = c;
        if (tt == NT - 1) between(s);
    }
#else
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int row = (16 * s) / WTW, col0 = (16 * s) % WTW;
        Frag b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 2; ++r) b[p].half4[r] = tr_read(sD, p * 2 * DIMGR + (col0 + 4 * r) * ROWB + db[row]);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = T0 + tt, ky = t / 3, kx = t % 3;
            Frag a[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    a[p].half4[r] = tr_read(sX, p * 2 * XIMGR + (col0 + kx + 4 * r) * ROWB + xa[row + ky] + (FOLD ? xs[col0 / 16] : 0));
            f32x16 c = acc[tt];
            if (NP > 2) {
                c = punet_mfma32<F16>(a[P1].u, b[P1].u, c);  // mm
                c = punet_mfma32<F16>(a[P2].u, b[0].u, c);   // lh
                c = punet_mfma32<F16>(a[0].u, b[P2].u, c);   // hl
            }
            if (NP > 1) {
                c = punet_mfma32<F16>(a[P1].u, b[0].u, c);  // mh
                c = punet_mfma32<F16>(a[0].u, b[P1].u, c);  // hm
            }
            c = punet_mfma32<F16>(a[0].u, b[0].u, c);  // hh
            acc[tt] = c;
        }
        between(s);
#if PUNET_WGRAD_SB == 1
        __builtin_amdgcn_sched_barrier(0);
#elif PUNET_WGRAD_SB == 2
        if (s & 1) __builtin_amdgcn_sched_barrier(0);
#endif
    }
#endif
}

// X16 (one plane only; the "bf16s" / "f16s" fast modes): the activation operand x holds bf16 / fp16 - its staging is a plain
// 8-byte copy.  F16: fp16 planes (NP <= 2) scaled by exact powers of two from the operands' amax slots (common.h; an operand
// STORED as fp16 is unscaled), the slabs are written rescaled.
// FOLD (round 5; see conv3x3_x6b_kernel): the 32-pixel column strips run over the VIRTUAL row of all B images side by side where
// the width is not whole strips (88 = 2.75, 176 = 5.5); a strip that holds the end of image b and the start of image b + 1 gets two
// zero pixels per X row at the seam (the right halo of b, the left halo of b + 1: filled by the ordinary out-of-image test), and
// the fragment reads of pixels right of the seam are shifted by those two pixels.  Widths that are multiples of 8 only (a
// fragment's 4-pixel sub-blocks then never straddle a seam).  Sums run over the same pixels; only their ORDER differs from the
// per-image launch (strips are cut elsewhere), i.e. results agree to rounding, not bit for bit.
template <int NP, bool X16 = false, bool F16 = false, bool FOLD = false>
__global__ __launch_bounds__(512, 2) void wgrad3x3_x6r_kernel(WgradX6Args a) {
    static_assert(!X16 || NP == 1, "16-bit activation storage exists for the one-plane arithmetic only");
    static_assert(!F16 || NP <= 2, "fp16 operands: one or two planes");
    constexpr int HTW = ::HTW_BASE + (FOLD ? 2 : 0);            // (shadow the namespace constants: rows of the X ring are 36 pixels)
    constexpr int XIMGR = XSLOTS * HTW * ROWB + PUNET_WGRAD_IMGPAD;
    float sxs = 1.f, sds = 1.f;  // F16: operand scales (block-uniform)
    if constexpr (F16) {
        if (!X16) sxs = punet_f16_scale_from(a.amax_x);
        sds = punet_f16_scale_from(a.amax_dz);
    }
    constexpr int C4 = WC / 4;                                   // float4 units per pixel
    constexpr int XNEW = 2 * HTW * C4, DNEW = 2 * WTW * C4;      // units of a tile's two new rows: 1,088 and 1,024
    constexpr int XIT = (XNEW + 511) / 512, DIT = DNEW / 512;    // 3 and 2 staging registers per thread
    static_assert(DNEW % 512 == 0, "every dZ unit is owned by exactly one thread (the bias sums rely on it)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                       // [NP][2 ci blocks] ring images
    char* sD = smem + NP * 2 * XIMGR;      // [NP][2 co blocks] ring images
    float* sBias = reinterpret_cast<float*>(smem);  // reused after the last tile

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int pair = wave & 3, group = wave >> 2;
#if PUNET_WGRAD_PRIO == 1  // waves 4-7 lose the issue arbitration against 0-3 and reach the tile's barrier ~1.5k cycles later
    if (group == 1) __builtin_amdgcn_s_setprio(1);
#elif PUNET_WGRAD_PRIO == 2
    if (group == 0) __builtin_amdgcn_s_setprio(1);
#elif PUNET_WGRAD_PRIO == 3
    if (group == 1) __builtin_amdgcn_s_setprio(3);
#endif
    const int cib = pair >> 1, cob = pair & 1;
    const int co_tiles = (a.cout + WC - 1) / WC;
    const int ci_tile = blockIdx.y / co_tiles, co_tile = blockIdx.y % co_tiles;
    const int ci0 = ci_tile * WC, co0 = co_tile * WC;
    const int slice = blockIdx.x;
    const int tiles_x = FOLD ? (a.B * a.W + WTW - 1) / WTW : (a.W + WTW - 1) / WTW, tiles_y = (a.H + WTH - 1) / WTH;
    const int ntiles = (FOLD ? 1 : a.B) * tiles_x * tiles_y;  // FOLD: tiles_x strips over the virtual row of all images
    const int t_begin = (int)((long)slice * ntiles / a.nslices), t_end = (int)((long)(slice + 1) * ntiles / a.nslices);

    f32x16 acc[GTMAX];
#pragma unroll
    for (int t = 0; t < GTMAX; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    const int l16 = lane & 15, q = l16 >> 2, p4 = l16 & 3, cb = (lane >> 4) & 1;
    const int lane_off = (8 * half + q) * ROWB + (16 * cb + 4 * p4) * 2;

    // fp32 -> planes of one float4 unit, written at byte offset `off` inside the (plane, block 0) image of `img` bytes each
    auto put_planes = [&](char* base, int img, int off, const float4& r, float sc) __attribute__((always_inline)) {
        if constexpr (F16) {
            uint2 h, l;
            splitf16_quad<NP>(r, sc, h, l);
            *reinterpret_cast<uint2*>(base + off) = h;
            if (NP > 1) *reinterpret_cast<uint2*>(base + (NP > 1) * 2 * img + off) = l;
            return;
        }
        const float xv[4] = {r.x, r.y, r.z, r.w};
        bf16x4 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h_, m_, l_;
            split3(xv[e], h_, m_, l_);
            ph[e] = h_; pm[e] = m_; pl[e] = l_;
        }
        *reinterpret_cast<bf16x4*>(base + 0 * 2 * img + off) = ph;
        if (NP > 1) *reinterpret_cast<bf16x4*>(base + (NP > 1) * 2 * img + off) = pm;
        if (NP > 2) *reinterpret_cast<bf16x4*>(base + (NP > 2) * 4 * img + off) = pl;
    };
    auto load_x = [&](const float* xb, int gy, int px, int c4) __attribute__((always_inline)) -> float4 {
        const int gx_ = px;  // already offset by the strip
        const int c = ci0 + c4 * 4;
        const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx_, 0), a.W - 1), cc = min(c, a.cs - 4);
        if constexpr (X16) {  // four bf16 travel in the first two dwords of the staging register (bit pattern, not values)
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(xb) + (unsigned)((cy * a.W + cx) * a.cs + cc));
            return make_float4(__builtin_bit_cast(float, h.x), __builtin_bit_cast(float, h.y), 0.f, 0.f);
        } else {
#if PUNET_WGRAD_TIMING & 8  // timing-only (WRONG results): operands from a cache-resident window, same instructions
            return *reinterpret_cast<const float4*>(xb + (unsigned)((((cy * a.W + cx) & 2047)) * a.cs + cc));
#endif
            return *reinterpret_cast<const float4*>(xb + (unsigned)((cy * a.W + cx) * a.cs + cc));
        }
    };
    // the X operand's unit -> LDS: split (fp32 storage) or copied as it is (bf16 storage)
    auto put_x = [&](int off, const float4& r, bool in) __attribute__((always_inline)) {
        if constexpr (X16) {
            const uint2 h = in ? make_uint2(__builtin_bit_cast(unsigned, r.x), __builtin_bit_cast(unsigned, r.y)) : make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(sX + off) = h;
        } else {
            if constexpr (F16) put_planes(sX, XIMGR, off, r, in ? sxs : 0.f);  // zero SCALE outside the image: one select, not four
            else put_planes(sX, XIMGR, off, in ? r : make_float4(0.f, 0.f, 0.f, 0.f), sxs);
        }
    };
    auto load_d = [&](const float* dbp, int gy, int gx_, int c4) __attribute__((always_inline)) -> float4 {
        const int c = co0 + c4 * 4;
        const int cy = min(gy, a.H - 1), cx = min(gx_, a.W - 1), cc = min(c, a.cout - 4);
#if PUNET_WGRAD_TIMING & 8
        return *reinterpret_cast<const float4*>(dbp + (unsigned)((((cy * a.W + cx) & 2047)) * a.cout + cc));
#endif
        return *reinterpret_cast<const float4*>(dbp + (unsigned)((cy * a.W + cx) * a.cout + cc));
    };

    // The two tap groups run the WHOLE sweep in their own instantiation (one uniform branch per launch): with the branch
    // around each tile's MFMA block the two paths ended in different accumulator registers and every wave paid ~40
    // v_mov_b64 of accumulator copies per tile at the join.
    auto sweep = [&](auto ta_c, auto tb_c) __attribute__((always_inline)) {
        constexpr int TA = decltype(ta_c)::value, TB = decltype(tb_c)::value;
        for (int t = t_begin; t < t_end;) {
            const int sid = t / tiles_y, ty_s = t - sid * tiles_y;
            const int run = min(t_end - t, tiles_y - ty_s);  // tiles of this run: down the strip until its end or the slice's
            // FOLD: the strip starts at pixel tx0 of image b and holds `seam` pixels of it (>= 32: no seam); the rest is image b + 1
            const int b = FOLD ? (sid * WTW) / a.W : sid / tiles_x;
            const int tx0 = FOLD ? sid * WTW - b * a.W : (sid % tiles_x) * WTW;
            const int seam = a.W - tx0;
            const bool have2 = b + 1 < a.B;
            auto x_image = [&](int bb) __attribute__((always_inline)) {
                return X16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(a.x) + (size_t)bb * ((size_t)(a.H * a.W) * a.cs))
                           : a.x + (size_t)bb * ((size_t)(a.H * a.W) * a.cs);
            };
            const float* const xb = x_image(b);
            const float* const dbp = a.dz + (size_t)b * ((size_t)(a.H * a.W) * a.cout);
            const float* const xb2 = FOLD ? x_image(have2 ? b + 1 : b) : xb;
            const float* const dbp2 = FOLD ? a.dz + (size_t)(have2 ? b + 1 : b) * ((size_t)(a.H * a.W) * a.cout) : dbp;
            // position p of an X ring row (0 .. HTW - 1) / pixel j of a dZ row -> x inside its image; returns "of the second image"
            // (the positions seam + 1 and seam + 2 come out as x = W and x = -1: the two halo zeros at the seam)
            auto xpos = [&](int p, int& gx_) __attribute__((always_inline)) -> bool {
                const bool second = FOLD && p >= seam + 2;
                gx_ = second ? p - seam - 3 : tx0 + p - 1;
                return second;
            };
            auto dpos = [&](int j, int& gx_) __attribute__((always_inline)) -> bool {
                const bool second = FOLD && j >= seam;
                gx_ = second ? j - seam : tx0 + j;
                return second;
            };
            int xs[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) xs[c] = FOLD && 16 * c + 8 * half >= seam ? 2 * ROWB : 0;

            // ---- prologue of the run: the first tile's four X rows and two dZ rows in full (slots 0..3 / 0..1)
            __syncthreads();  // the previous run's last readers are done
            for (int u = tid; u < 4 * HTW * C4; u += 512) {
                const int c4 = u % C4, pix = u / C4, hr = pix / HTW, px = pix % HTW;
                int gx_;
                const bool second = xpos(px, gx_);
                const int gy = 2 * ty_s - 1 + hr;
                const bool in = gy >= 0 && gy < a.H && gx_ >= 0 && gx_ < a.W && ci0 + c4 * 4 < a.cs && (!second || have2);
                const float4 v = load_x(second ? xb2 : xb, gy, gx_, c4);
                put_x((c4 >> 3) * XIMGR + (hr * HTW + px) * ROWB + (c4 & 7) * 8, v, in);
            }
            for (int u = tid; u < 2 * WTW * C4; u += 512) {
                const int c4 = u % C4, pix = u / C4, row = pix / WTW, px = pix % WTW;
                int gx_;
                const bool second = dpos(px, gx_);
                const int gy = 2 * ty_s + row;
                const bool in = gy < a.H && gx_ < a.W && co0 + c4 * 4 < a.cout && (!second || have2);
                float4 v = load_d(second ? dbp2 : dbp, gy, gx_, c4);
                if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
                bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
                put_planes(sD, DIMGR, (c4 >> 3) * DIMGR + (row * WTW + px) * ROWB + (c4 & 7) * 8, v, sds);
            }
            __syncthreads();

            for (int k = 0; k < run; ++k) {
#ifdef PUNET_WGRAD_TRACE
                const int trace_iter = k;  // tiles 5, 6, 7 of the first run: ids 0 / 10 / 11 (+ 3 per tile)
#define WGTR(id)                                                                                              \
    do {                                                                                                      \
        if (blockIdx.x == 7 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && t == t_begin && trace_iter >= 5 && trace_iter < 8) \
            punet_wgrad_trace[(threadIdx.x >> 6) * 32 + (trace_iter - 5) * 3 + (id)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define WGTR(id)
#endif
                WGTR(0);
                const int ty = ty_s + k;
                const bool has_next = k + 1 < run;  // block-uniform
                // ring slots of this tile's rows and of the successor's two new rows
                const int sx = (2 * k) % XSLOTS, sd = (2 * k) % DSLOTS;
                int xa[4], db_[2];
    #pragma unroll
                for (int hr = 0; hr < 4; ++hr) xa[hr] = ((sx + hr) % XSLOTS) * (HTW * ROWB) + lane_off + cib * XIMGR;
    #pragma unroll
                for (int r = 0; r < 2; ++r) db_[r] = ((sd + r) % DSLOTS) * (WTW * ROWB) + lane_off + cob * DIMGR;
                const int nx0 = ((sx + 4) % XSLOTS) * (HTW * ROWB), nx1 = ((sx + 5) % XSLOTS) * (HTW * ROWB);
                const int nd0 = ((sd + 2) % DSLOTS) * (WTW * ROWB), nd1 = ((sd + 3) % DSLOTS) * (WTW * ROWB);

                // the successor's new rows: X rows 2 ty + 3, 2 ty + 4 (its halo rows 2, 3), dZ rows 2 ty + 2, 2 ty + 3.
                // Unconditional and branch-free (clamped addresses; a run's last tile re-fetches valid rows and drops them).
                float4 rx[XIT], rd[DIT];
                unsigned inmask = 0u;
    #pragma unroll
                for (int it = 0; it < XIT; ++it) {
                    const int u = min(tid + it * 512, XNEW - 1);
                    const int c4 = u % C4, pix = u / C4, nr = pix / HTW, px = pix % HTW;
                    int gx_;
                    const bool second = xpos(px, gx_);
                    const int gy = 2 * ty + 3 + nr;
                    const bool in = gy < a.H && gx_ >= 0 && gx_ < a.W && ci0 + c4 * 4 < a.cs && (!second || have2);
                    rx[it] = load_x(second ? xb2 : xb, gy, gx_, c4);
                    inmask = in ? inmask | (1u << it) : inmask;
                }
    #pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const int u = tid + it * 512;
                    const int c4 = u % C4, pix = u / C4, nr = pix / WTW, px = pix % WTW;
                    int gx_;
                    const bool second = dpos(px, gx_);
                    const int gy = 2 * ty + 2 + nr;
                    // (the run's last tile stages too - no branch in the MFMA stream, the rows land in ring slots nobody reads
                    // - but its dZ rows count as outside: they must not enter the column sums a second time)
                    const bool in = has_next && gy < a.H && gx_ < a.W && co0 + c4 * 4 < a.cout && (!second || have2);
                    rd[it] = load_d(second ? dbp2 : dbp, gy, gx_, c4);
                    inmask = in ? inmask | (1u << (XIT + it)) : inmask;
                }
                auto commit_x = [&](int it) __attribute__((always_inline)) {
                    const int u = min(tid + it * 512, XNEW - 1);  // a clamped duplicate re-writes the last unit with the same data
                    const int c4 = u % C4, pix = u / C4, nr = pix / HTW, px = pix % HTW;
                    put_x((c4 >> 3) * XIMGR + (nr ? nx1 : nx0) + px * ROWB + (c4 & 7) * 8, rx[it], (inmask >> it) & 1u);
                };
                auto commit_d = [&](int it) __attribute__((always_inline)) {
                    const int u = tid + it * 512;
                    const int c4 = u % C4, pix = u / C4, nr = pix / WTW, px = pix % WTW;
                    const float4 v = (inmask >> (XIT + it)) & 1u ? rd[it] : make_float4(0.f, 0.f, 0.f, 0.f);
                    bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
                    put_planes(sD, DIMGR, (c4 >> 3) * DIMGR + (nr ? nd1 : nd0) + px * ROWB + (c4 & 7) * 8, v, sds);
                };
                auto between = [&](int s) __attribute__((always_inline)) {
#ifdef PUNET_WGRAD_TRACE_FINE
                    if (blockIdx.x == 7 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && t == t_begin && trace_iter == 6)
                        punet_wgrad_trace[(threadIdx.x >> 6) * 32 + 9 + s] = __builtin_amdgcn_s_memtime();
#endif
                    if (s == 1) commit_x(0);
                    if (s == 2) {
    #pragma unroll
                        for (int it = 1; it < XIT; ++it) commit_x(it);
                    }
                    if (s == 3) {
    #pragma unroll
                        for (int it = 0; it < DIT; ++it) commit_d(it);
                    }
                };
                tile_mfma_r<NP, TA, TB, F16, XIMGR, FOLD>(acc, sX, sD, xa, db_, xs, between);
                WGTR(1);
                __syncthreads();
                WGTR(2);
            }
            t += run;
        }

        const int co = co0 + cob * 32 + l31;
        store_partial<TA, TB>(acc, a, slice, ci0 + cib * 32, co, half, F16 ? 1.f / (sxs * sds) : 1.f);
    };
    if (group == 0)
        sweep(std::integral_constant<int, 0>{}, std::integral_constant<int, GT0>{});
    else
        sweep(std::integral_constant<int, GT0>{}, std::integral_constant<int, GT1>{});

    if (a.bias_partial && ci_tile == 0) {  // block-uniform
        __syncthreads();
        sBias[tid * 4 + 0] = bsum.x; sBias[tid * 4 + 1] = bsum.y; sBias[tid * 4 + 2] = bsum.z; sBias[tid * 4 + 3] = bsum.w;
        __syncthreads();
        if (tid < WC) {
            // thread t owns channel t: every thread's units have c4 = tid % 16 (512 and the unit counts are multiples of 16)
            const int c4 = tid >> 2, e = tid & 3;
            float sacc = 0.f;
            for (int kk = 0; kk < 32; ++kk) sacc += sBias[(kk * 16 + c4) * 4 + e];
            if (co0 + tid < a.cout) a.bias_partial[(size_t)slice * a.cout + co0 + tid] = sacc;
        }
    }
}

constexpr int smem_bytes_r(int np, bool fold = false) {
    const int operands = np * 2 * (XIMGR + (fold ? XSLOTS * 2 * ROWB : 0) + DIMGR);  // 127,488 B at NP = 3 (FOLD: 132,096 B)
    return operands > 512 * 16 ? operands : 512 * 16;
}

constexpr int smem_bytes(int np) {
    const int operands = np * 2 * (XIMG + DIMG);  // 76,800 B at NP = 3
    return operands > 512 * 16 ? operands : 512 * 16;  // the bias staging (512 threads x float4) re-uses the region
}

#ifndef PUNET_WGRAD_ROLLING
#define PUNET_WGRAD_ROLLING 1  // 1: the rolling-row-buffer kernel (round 2), 0: the whole-halo-per-tile kernel of round 1
#endif

int launch_wgrad_x16(const WgradX6Args& a, dim3 grid, hipStream_t stream) {
    if (hipFuncSetAttribute((const void*)wgrad3x3_x6r_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes_r(1)) !=
        hipSuccess) {
        punet_set_error("punet_conv_wgrad_bf16s: cannot reserve %d bytes of LDS", smem_bytes_r(1));
        return PUNET_ERR_HIP;
    }
    wgrad3x3_x6r_kernel<1, true><<<grid, 512, smem_bytes_r(1), stream>>>(a);
    return PUNET_OK;
}

template <int NP, bool X16>
int launch_wgrad_f16(const WgradX6Args& a, dim3 grid, hipStream_t stream) {
    if (hipFuncSetAttribute((const void*)wgrad3x3_x6r_kernel<NP, X16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes_r(NP)) !=
        hipSuccess) {
        punet_set_error("punet_conv_wgrad_f16: cannot reserve %d bytes of LDS", smem_bytes_r(NP));
        return PUNET_ERR_HIP;
    }
    wgrad3x3_x6r_kernel<NP, X16, true><<<grid, 512, smem_bytes_r(NP), stream>>>(a);
    return PUNET_OK;
}

// the FOLD instantiations: the three-plane bf16 and the two-plane fp16 arithmetic (fp32 storage)
template <int NP, bool F16>
int launch_wgrad_fold(const WgradX6Args& a, dim3 grid, hipStream_t stream) {
    if (hipFuncSetAttribute((const void*)wgrad3x3_x6r_kernel<NP, false, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem_bytes_r(NP, true)) != hipSuccess) {
        punet_set_error("punet_conv_wgrad_x6: cannot reserve %d bytes of LDS", smem_bytes_r(NP, true));
        return PUNET_ERR_HIP;
    }
    wgrad3x3_x6r_kernel<NP, false, F16, true><<<grid, 512, smem_bytes_r(NP, true), stream>>>(a);
    return PUNET_OK;
}

template <int NP>
int launch_wgrad(const WgradX6Args& a, dim3 grid, hipStream_t stream) {
#if PUNET_WGRAD_ROLLING
    if (hipFuncSetAttribute((const void*)wgrad3x3_x6r_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes_r(NP)) !=
        hipSuccess) {
        punet_set_error("punet_conv_wgrad_x6: cannot reserve %d bytes of LDS", smem_bytes_r(NP));
        return PUNET_ERR_HIP;
    }
    wgrad3x3_x6r_kernel<NP><<<grid, 512, smem_bytes_r(NP), stream>>>(a);
#else
    if (hipFuncSetAttribute((const void*)wgrad3x3_x6_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * smem_bytes(NP)) !=
        hipSuccess) {
        punet_set_error("punet_conv_wgrad_x6: cannot reserve %d bytes of LDS", 2 * smem_bytes(NP));
        return PUNET_ERR_HIP;
    }
    wgrad3x3_x6_kernel<NP><<<grid, 512, 2 * smem_bytes(NP), stream>>>(a);
#endif
    return PUNET_OK;
}

}  // namespace

extern "C" int punet_wgrad_x6_slices(int cs, int cout, int B, int H, int W) {
    const int ntiles = B * ceil_div(H, WTH) * ceil_div(W, WTW);
    const int pairs = ceil_div(cs, WC) * ceil_div(cout, WC);
    if (pairs < 1 || ntiles < 1) return 1;  // (a shape the launch would refuse)
    int ns = punet_num_cus() / pairs;  // one resident workgroup per CU: one full round of the CUs
    if (ns > ntiles) ns = ntiles;
    return ns < 1 ? 1 : ns;
}

static int wgrad_x6_impl(bool x16, const float* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                         float* dbias, float* workspace, int nslices, int B, int H, int W, int accumulate, float scale, int opts,
                         void* stream_, int f16planes = 0, const float* amax_x = nullptr, const float* amax_dz = nullptr);

extern "C" int punet_conv_wgrad_x6(const float* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                                   float* dbias, float* workspace, int nslices, int B, int H, int W, int accumulate,
                                   float scale, int opts, void* stream_) {
    return wgrad_x6_impl(false, x, cs, dz, cout, dw, cin_total, ci_off, dbias, workspace, nslices, B, H, W, accumulate, scale, opts,
                         stream_);
}

// the same in the one-plane arithmetic with the activation operand x stored as bf16 (the "bf16s" fast mode); dz stays fp32
extern "C" int punet_conv_wgrad_bf16s(const void* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                                      float* dbias, float* workspace, int nslices, int B, int H, int W, int accumulate, float scale,
                                      void* stream_) {
    return wgrad_x6_impl(true, (const float*)x, cs, dz, cout, dw, cin_total, ci_off, dbias, workspace, nslices, B, H, W, accumulate,
                         scale, PUNET_X6_OPTS(1, 16), stream_);
}

// fp16-operand weight gradient ("f16x3": planes = 2, "f16": planes = 1; x16 != 0: x is STORED as fp16, planes = 1 only);
// amax_x / amax_dz: the operands' amax slots (common.h; nullptr = scale 1)
extern "C" int punet_conv_wgrad_f16(const void* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                                    float* dbias, float* workspace, int nslices, int B, int H, int W, int accumulate, float scale,
                                    int planes, int x16, const float* amax_x, const float* amax_dz, void* stream_) {
    PUNET_CHECK_ARG((planes == 1 || planes == 2) && !(x16 && planes != 1), "punet_conv_wgrad_f16: planes must be 2 or 1 (fp16 storage: 1)");
    return wgrad_x6_impl(x16 != 0, (const float*)x, cs, dz, cout, dw, cin_total, ci_off, dbias, workspace, nslices, B, H, W, accumulate,
                         scale, PUNET_X6_OPTS(planes == 2 ? 3 : 1, 16), stream_, planes, amax_x, amax_dz);
}

static int wgrad_x6_impl(bool x16, const float* x, int cs, const float* dz, int cout, float* dw, int cin_total, int ci_off,
                         float* dbias, float* workspace, int nslices, int B, int H, int W, int accumulate, float scale, int opts,
                         void* stream_, int f16planes, const float* amax_x, const float* amax_dz) {
    hipStream_t stream = (hipStream_t)stream_;
    const int planes = punet_x6_planes_of(opts);
    PUNET_CHECK_ARG(planes > 0, "punet_conv_wgrad_x6: bad options word 0x%x", opts);
    PUNET_CHECK_ARG(x && dz && dw && workspace, "punet_conv_wgrad_x6: null pointer");
    PUNET_CHECK_ARG(cs > 0 && cs % 4 == 0 && cout > 0 && cout % 4 == 0,
                    "punet_conv_wgrad_x6: channels must be multiples of 4 (cs=%d cout=%d)", cs, cout);
    PUNET_CHECK_ARG(ci_off >= 0 && ci_off + cs <= cin_total, "punet_conv_wgrad_x6: bad channel window");
    PUNET_CHECK_ARG(nslices >= 1, "punet_conv_wgrad_x6: nslices must be >= 1");
    PUNET_CHECK_ARG((unsigned long long)H * W * (cs > cout ? cs : cout) < (1ull << 31) && (unsigned long long)B * H * W < (1ull << 31),
                    "punet_conv_wgrad_x6: one image is too large for 32-bit offsets");
    WgradX6Args a{};
    a.x = x; a.cs = cs; a.dz = dz; a.cout = cout;
    a.partial = workspace;
    a.bias_partial = dbias ? workspace + (size_t)nslices * 9 * cs * cout : nullptr;
    a.B = B; a.H = H; a.W = W; a.nslices = nslices;
    a.amax_x = amax_x; a.amax_dz = amax_dz;
    dim3 grid(nslices, ceil_div(cs, WC) * ceil_div(cout, WC));
    // FOLD (see the kernel): strips over the virtual row of all images where that saves strips; x6 and f16x3 only
    const bool fold = punet_fold_mode() == 1 && !(opts & PUNET_X6_NOFOLD) && !x16 && (f16planes == 2 || (f16planes == 0 && planes == 3)) &&
                      W >= WTW && W % 8 == 0 && ceil_div(B * W, WTW) < B * ceil_div(W, WTW) && PUNET_WGRAD_ROLLING;
    const int rc0 = fold ? (f16planes == 2 ? launch_wgrad_fold<2, true>(a, grid, stream) : launch_wgrad_fold<3, false>(a, grid, stream))
                    : f16planes == 2 ? launch_wgrad_f16<2, false>(a, grid, stream)
                    : f16planes == 1 ? (x16 ? launch_wgrad_f16<1, true>(a, grid, stream) : launch_wgrad_f16<1, false>(a, grid, stream))
                    : x16 ? launch_wgrad_x16(a, grid, stream)
                    : planes == 3 ? launch_wgrad<3>(a, grid, stream)
                    : planes == 2 ? launch_wgrad<2>(a, grid, stream) : launch_wgrad<1>(a, grid, stream);
    if (rc0 != PUNET_OK) return rc0;
    PUNET_LAUNCH_CHECK("wgrad3x3_x6");
    const size_t n = (size_t)9 * cs * cout;
    if (dbias)
        return punet_launch_reduce_slabs_pair(a.partial, n, 9 * cs, cout, cs, cin_total, ci_off, dw, a.bias_partial, (size_t)cout,
                                              dbias, nslices, accumulate, scale, stream);
    const int rc = punet_launch_reduce_slabs(a.partial, nslices, n, 9 * cs, cout, cs, cin_total, ci_off, dw, accumulate, scale, stream);
    return rc;
}

#ifdef PUNET_WGRAD_TRACE_MAIN  // standalone trace driver: 64 -> 64 channels at 512 x 512, batch 4 (tools/trace_wgrad.sh)
#include <cstdio>
#include <vector>
int main() {
    const int B = 4, H = 512, W = 512, C = 64;
    const size_t n = (size_t)B * H * W * C;
    std::vector<float> h(n);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - (s & 1 ? 0.5f : 1.5f); }
    float *x, *dz, *dw, *db, *ws;
    const int ns = punet_wgrad_x6_slices(C, C, B, H, W);
    hipMalloc(&x, n * 4); hipMalloc(&dz, n * 4); hipMalloc(&dw, 9 * C * C * 4); hipMalloc(&db, C * 4);
    hipMalloc(&ws, ((size_t)ns * 9 * C * C + (size_t)ns * C) * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dz, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) {
        const int rc = punet_conv_wgrad_x6(x, C, dz, C, dw, C, 0, db, ws, ns, B, H, W, 0, 1.f, 0, nullptr);
        if (rc) { printf("error %d\n", rc); return 1; }
    }
    hipDeviceSynchronize();
    unsigned long long t[8 * 32];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(punet_wgrad_trace), sizeof(t));
    printf("slices %d; stamps relative to wave 0's loop top (id 0), shader cycles (s_memtime)\n", ns);
    for (int w = 0; w < 8; ++w) {  // rolling kernel: tiles 5, 6, 7 of the workgroup's first run: top / barrier reached / passed
        printf("wave %d:", w);
        for (int i = 0; i < 9; ++i) printf(" %6lld%s", (long long)(t[w * 32 + i] - t[0]), i % 3 == 2 ? " |" : "");
#ifdef PUNET_WGRAD_TRACE_FINE
        printf("  tile 6, pixel groups issued:");
        for (int i = 9; i < 13; ++i) printf(" %6lld", (long long)(t[w * 32 + i] - t[w * 32 + 3]));
#endif
        printf("\n");
    }
    return 0;
}
#endif
