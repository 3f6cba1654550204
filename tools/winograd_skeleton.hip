// Timing-only skeleton of a Winograd F(2x2, 3x3) form of the split-plane 3x3 convolution: NOT a convolution -
// the addresses, the data volumes, the instruction mix and the accumulator footprint are those of the real thing, the
// arithmetic is not checked.  Purpose: a go / no-go number against the product kernel (tools/bench_x6_layers.py) before any
// correctness work - go needs >= 1.25x on 256->256 @128x128 and 64->64 @512x512 (batch 4).
//
// F(2x2, 3x3): a 4 x 4 input patch d and the 3 x 3 filter g give a 2 x 2 output tile as A^T [(G g G^T) o (B^T d B)] A - 16
// multiplies instead of 36, i.e. 16 independent "position" GEMMs  M_p[tile][cout] += V_p[tile][cin] * U_p[cin][cout]  and
// 16 accumulators per 2 x 2 outputs (4x the output's footprint).  With the six-product bf16 arithmetic every fp32 value of
// V = B^T d B and U = G g G^T is three bf16 planes.
//
// Structure measured here (what the 512-register budget of one wave per SIMD allows):
//   * one 256-thread workgroup per CU; a wave owns 16 tiles (= 2 x 32 output pixels) x 64 output channels x 16 positions
//     = 256 accumulator registers; the workgroup's four waves share the weight stream (8 x 32 pixels x 64 channels);
//   * U_p (pre-transformed, pre-split: [chunk of 32 cin][position][3 planes][4 k-groups][64 cout][8 cin] bf16, 12 KB per
//     position and chunk) streams global -> LDS by LDS-DMA through a ring of `STAGES` stages of `PPS` positions, one barrier
//     per stage; weight fragments come from LDS (12 ds_read_b128 per position for 24 MFMAs);
//   * V_p is produced in REGISTERS by the wave that consumes it: lane (tile, k-group) loads its 4 x 4 x 8-channel patch of the
//     fp32 image (32 float4 loads per chunk), transforms it (32 adds per channel) and splits the 16 x 8 results into planes
//     (-DWINO_TRANSFORM=1; 0 = fragments from constants: the upper bound with free V);
//   * epilogue: A^T M A (24 adds per output quad), ReLU, 16-byte stores of the 2 x 2 pixels.
//   hipcc --offload-arch=gfx950 -O3 tools/winograd_skeleton.hip -o /tmp/wino && /tmp/wino 256 256 128 && /tmp/wino 64 64 512
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef WINO_TRANSFORM
#define WINO_TRANSFORM 1
#endif
#ifndef PPS
#define PPS 2  // positions per ring stage
#endif
#ifndef STAGES
#define STAGES 4
#endif

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int SLAB = 3 * 4 * 64;          // uint4 units of one position's weights for one 32-channel chunk (12 KB)
constexpr int STAGE_UNITS = PPS * SLAB;
constexpr int LDS_UNITS = STAGES * STAGE_UNITS;

union U4 {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// fp32 pair -> three bf16 planes (the product kernels' packed split: 9 vector instructions per pair)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 x = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
    const f32x2 r1 = x - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
    const f32x2 r2 = r1 - mf;
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// x: [H][W][C] fp32 (the batch stacked along y); up: transformed weight planes, layout above; y: [H][W][N] fp32
__global__ __launch_bounds__(256, 1) void wino_skeleton(const float* __restrict__ x, const __bf16* __restrict__ up,
                                                        float* __restrict__ y, int H, int W, int C, int N, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int nchunks = C / 32;
    const int tiles_x = W / 32, tiles_y = H / 8, ntn = N / 64;
    const int nstage_per_chunk = 16 / PPS;

    // weight stream: stage s of the whole K loop = (chunk, position group); one stage = PPS slabs = PPS * 12 wave-instructions
    auto issue_stage = [&](int nb, int s, int slot) __attribute__((always_inline)) {
        const int ch = s / nstage_per_chunk, pg = s % nstage_per_chunk;
        const __bf16* src = up + ((((size_t)nb * nchunks + ch) * 16 + pg * PPS) * SLAB) * 8;
#pragma unroll
        for (int r = 0; r < PPS * 3; ++r) {  // 12 * PPS rows of 1 KB over 4 waves
            const int row = wave + 4 * r;
            glds16(src + ((size_t)row * 64 + lane) * 8, smem + slot * STAGE_UNITS + row * 64);
        }
    };

    f32x4 acc[16][4];
    int tile = blockIdx.x;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int nb = tile % ntn;
        int sid = tile / ntn;
        const int tx0 = (sid % tiles_x) * 32, ty0 = ((sid / tiles_x) % tiles_y) * 8;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[p][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nst = nchunks * nstage_per_chunk;
        // prologue: STAGES - 1 stages in flight
#pragma unroll
        for (int s0 = 0; s0 < STAGES - 1; ++s0)
            if (s0 < nst) issue_stage(nb, s0, s0);
        // this lane's tile: wave w owns output rows 2w, 2w+1 of the 8-row tile, tile column l16 (2 pixels each)
        const int py0 = ty0 + 2 * wave - 1, px0 = tx0 + 2 * l16 - 1;  // top-left of the 4 x 4 input patch
        int s = 0;
        for (int ch = 0; ch < nchunks; ++ch) {
            // ---- this lane's 4 x 4 x 8-channel patch, row-transformed in place: t = B^T d (128 registers for the chunk); the
            //      column step and the plane split of a position happen right in front of its MFMAs (V never exists as a whole:
            //      16 x 3 x 4 fragment registers next to 256 accumulators would not fit)
#if WINO_TRANSFORM
            float t[4][4][8];
            {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        int gy = py0 + r, gx = px0 + c;
                        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
                        gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
                        const float4* p = reinterpret_cast<const float4*>(x + ((size_t)gy * W + gx) * C + ch * 32 + g * 8);
                        const float4 a = p[0], b = p[1];
                        t[r][c][0] = a.x; t[r][c][1] = a.y; t[r][c][2] = a.z; t[r][c][3] = a.w;
                        t[r][c][4] = b.x; t[r][c][5] = b.y; t[r][c][6] = b.z; t[r][c][7] = b.w;
                    }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d0 = t[0][c][e], d1 = t[1][c][e], d2 = t[2][c][e], d3 = t[3][c][e];
                        t[0][c][e] = d0 - d2; t[1][c][e] = d1 + d2; t[2][c][e] = d2 - d1; t[3][c][e] = d1 - d3;
                    }
            }
#endif
            // ---- the 16 position GEMMs of this chunk, PPS per ring stage (fully unrolled: positions are compile-time)
            static_assert((16 / PPS) % STAGES == 0, "the ring slot of a stage must be a compile-time constant");
#pragma unroll
            for (int pg = 0; pg < 16 / PPS; ++pg, ++s) {
                // the oldest outstanding stage (this one) must have landed: at most STAGES - 2 younger stages stay in flight
                constexpr int NV = (STAGES - 2) * PPS * 3;
                if (s + STAGES - 1 <= nst) __builtin_amdgcn_s_waitcnt(0x0f70 | (NV & 15) | ((NV >> 4) << 14));  // vmcnt(NV)
                else __builtin_amdgcn_s_waitcnt(0x0f70);                                                        // tail: vmcnt(0)
                __builtin_amdgcn_s_barrier();  // everybody's rows of stage s are in LDS; everybody is done with stage s - 1
                if (s + STAGES - 1 < nst) issue_stage(nb, s + STAGES - 1, (pg + STAGES - 1) % STAGES);
                const uint4* sW = smem + (pg % STAGES) * STAGE_UNITS;
#pragma unroll
                for (int pp = 0; pp < PPS; ++pp) {
                    const int p = pg * PPS + pp, pr = p >> 2, pc = p & 3;  // compile-time after unrolling
                    U4 b[3][4];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) b[pl][cb].u = sW[pp * SLAB + (pl * 4 + g) * 64 + 16 * cb + l16];
                    U4 vf[3];
#if WINO_TRANSFORM
                    {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e)  // column step of B^T d B for column pc
                            v[e] = pc == 0 ? t[pr][0][e] - t[pr][2][e] : pc == 1 ? t[pr][1][e] + t[pr][2][e]
                                 : pc == 2 ? t[pr][2][e] - t[pr][1][e] : t[pr][1][e] - t[pr][3][e];
                        unsigned h[4], m[4], l[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) split3_pair(v[2 * q], v[2 * q + 1], h[q], m[q], l[q]);
                        vf[0].u = make_uint4(h[0], h[1], h[2], h[3]);
                        vf[1].u = make_uint4(m[0], m[1], m[2], m[3]);
                        vf[2].u = make_uint4(l[0], l[1], l[2], l[3]);
                    }
#else
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned v = 0x3f803f80u ^ (unsigned)((lane * 131 + p * 17 + pl * 5 + ch) & 0x7f7f);
                        vf[pl].u = make_uint4(v, v ^ 0x11u, v ^ 0x1100u, v ^ 0x0101u);
                    }
#endif
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        f32x4 c = acc[p][cb];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][cb].v, vf[1].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][cb].v, vf[2].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2][cb].v, vf[0].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][cb].v, vf[1].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][cb].v, vf[0].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][cb].v, vf[0].v, c, 0, 0, 0);
                        acc[p][cb] = c;
                    }
                }
            }
        }
        // ---- epilogue: Y = A^T M A per (tile, channel quad), ReLU, stores (lane (l16, g): tile l16, channels 16 cb + 4 g ..)
        const int oy = ty0 + 2 * wave, ox = tx0 + 2 * l16;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 r0[4], r1[4];  // A^T M: rows
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r0[c] = acc[0 + c][cb] + acc[4 + c][cb] + acc[8 + c][cb];
                r1[c] = acc[4 + c][cb] - acc[8 + c][cb] - acc[12 + c][cb];
            }
            f32x4 y00 = r0[0] + r0[1] + r0[2], y01 = r0[1] - r0[2] - r0[3];
            f32x4 y10 = r1[0] + r1[1] + r1[2], y11 = r1[1] - r1[2] - r1[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y00[e] = fmaxf(y00[e], 0.f); y01[e] = fmaxf(y01[e], 0.f); y10[e] = fmaxf(y10[e], 0.f); y11[e] = fmaxf(y11[e], 0.f);
            }
            float* o = y + ((size_t)oy * W + ox) * N + nb * 64 + 16 * cb + 4 * g;
            *reinterpret_cast<f32x4*>(o) = y00;
            *reinterpret_cast<f32x4*>(o + N) = y01;
            *reinterpret_cast<f32x4*>(o + (size_t)W * N) = y10;
            *reinterpret_cast<f32x4*>(o + (size_t)W * N + N) = y11;
        }
        __builtin_amdgcn_s_barrier();  // the ring is reused by the next tile's prologue
        tile += (int)gridDim.x;
    }
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 256, HW = argc > 3 ? atoi(argv[3]) : 128;
    const int B = 4;
    const int H = B * HW, W = HW;  // the batch stacked along y (a skeleton: no image borders inside)
    const int ntiles = (H / 8) * (W / 32) * (N / 64);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount < ntiles ? prop.multiProcessorCount : ntiles;
    const int tiles_per_wg = ntiles / grid;  // whole rounds only
    const size_t nx = (size_t)H * W * C, nu = (size_t)(N / 64) * (C / 32) * 16 * SLAB * 8;
    std::vector<float> hx(nx);
    std::vector<unsigned short> hu(nu);
    srand(1);
    for (auto& v : hx) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += (float)rand() / RAND_MAX;
        v = s - 2.f < 0.f ? 0.f : s - 2.f;  // post-ReLU-like: half zeros
    }
    for (size_t i = 0; i < nu; ++i) {
        float v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
        const int plane = (int)((i / 8 / 64 / 4) % 3);
        v *= plane == 0 ? 1.f : (plane == 1 ? 1.f / 256 : 1.f / 65536);
        unsigned u;
        __builtin_memcpy(&u, &v, 4);
        hu[i] = (unsigned short)(u >> 16);
    }
    float *x, *y;
    __bf16* up;
    CK(hipMalloc(&x, nx * 4));
    CK(hipMalloc(&up, nu * 2));
    CK(hipMalloc(&y, (size_t)H * W * N * 4));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(up, hu.data(), nu * 2, hipMemcpyHostToDevice));
    const size_t lds = (size_t)LDS_UNITS * 16;
    CK(hipFuncSetAttribute((const void*)wino_skeleton, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k) wino_skeleton<<<grid, 256, lds>>>(x, up, y, H, W, C, N, tiles_per_wg);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        const double mf = (double)grid * tiles_per_wg * 4 * (C / 32) * 16 * 24;             // MFMAs executed
        const double alg = 2.0 * (double)grid * tiles_per_wg * 256 * 64 * C * 9;            // direct-convolution FLOPs covered
        printf("wino transform=%d PPS=%d STAGES=%d  C=%d N=%d %dx%d  grid %d x %d tiles  %.3f ms  executed %.0f TFLOP/s bf16 = %.1f "
               "TFLOP/s of x6 work (weights streamed %.1f MB per launch)\n", WINO_TRANSFORM, PPS, STAGES, C, N, HW, HW, grid,
               tiles_per_wg, ms, mf * 2 * 16 * 16 * 32 / ms * 1e-9, alg / ms * 1e-9,
               (double)grid * tiles_per_wg * (C / 32) * 16 * SLAB * 16 / 1e6);
    }
    return 0;
}
