// Throughput skeleton of the conv structure DESIGN.md §8 proposes (NOT a convolution: addresses are realistic, the
// arithmetic is not checked): one 8-wave workgroup per CU, operands as bf16 planes in global memory staged by LDS-DMA
// (global_load_lds_dwordx4, no staging registers, no splits), v_mfma_f32_16x16x32_bf16, a 16 x 32 pixel x 64 channel
// tile, chunks of 8 input channels, 9 K-steps (of four {tap, 8 channel} blocks) per 4 chunks, one barrier per chunk.
// LDS: ring of 3 activation images [3 planes][18 x 34 px][8 ch] + 2 weight slabs [3 planes][<= 12 rows][64 n][8 k].
// Prints the executed bf16 MFMA rate; compare with the product kernel's (tools/bench_x6_layers.py: TFLOP/s x 6).
//   hipcc --offload-arch=gfx950 -O3 tools/proto_skeleton.hip -o /tmp/proto && /tmp/proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int NPIX = 18 * 34;                    // halo image of a 16 x 32 tile
constexpr int XROWS = (3 * NPIX + 63) / 64;      // 1 KiB DMA rows per image (3 planes)
constexpr int XIMG = XROWS * 64;                 // uint4 units per image
constexpr int WROWS = 3 * 12;                    // <= 12 {tap} rows per plane and chunk
constexpr int WSLAB = WROWS * 64;                // uint4 units per weight slab
constexpr int LDS_UNITS = 3 * XIMG + 2 * WSLAB;  // 162,816 B

union U4 {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// H x W image of C channels as 3 bf16 planes [plane][pixel][C]; weights [plane][tap][C/8][N][8]
__global__ __launch_bounds__(512, 2) void skeleton(const __bf16* __restrict__ xp, const __bf16* __restrict__ wp,
                                                   float* __restrict__ y, int H, int W, int C, int N, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4* sX = smem;
    uint4* sW = smem + 3 * XIMG;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int nchunks = C / 8;
    const size_t plane = (size_t)H * W * C;
    const size_t wplane = (size_t)9 * (C / 8) * N * 8;
    const int tiles_x = W / 32, tiles_y = H / 16, ntn = N / 64;

    f32x4 acc[4][4];
    auto stage = [&](int tile, int ch, int ring, int wbuf) __attribute__((always_inline)) {
        const int nb = tile % ntn;
        int sid = tile / ntn;
        const int tx0 = (sid % tiles_x) * 32, ty0 = ((sid / tiles_x) % tiles_y) * 16;
        // activations: unit u = plane * NPIX + pixel; 16 bytes = the chunk's 8 channels of one pixel
        for (int r = wave; r < XROWS; r += 8) {
            int u = r * 64 + lane;
            u = u < 3 * NPIX ? u : 3 * NPIX - 1;
            const int p = u / NPIX, pix = u % NPIX;
            int gy = ty0 + pix / 34 - 1, gx = tx0 + pix % 34 - 1;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);  // the real kernel zero-fills out-of-image pixels (buffer OOB)
            gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            glds16(xp + p * plane + ((size_t)gy * W + gx) * C + ch * 8, sX + ring * XIMG + r * 64);
        }
        // weights: row = (plane, tap): 64 output channels x 8 input channels; 8, 8, 8, 12 rows per plane over 4 chunks
        const int rows = (ch & 3) == 3 ? 12 : 8;
        for (int r = wave; r < 3 * rows; r += 8) {
            const int p = r / rows, t = (r % rows) % 9;
            glds16(wp + p * wplane + (((size_t)t * nchunks + ch) * N + nb * 64 + lane) * 8, sW + wbuf * WSLAB + r * 64);
        }
    };

    int tile = blockIdx.x;
    int ring = 0, wbuf = 0;
    stage(tile, 0, 0, 0);
    for (int it = 0; it < tiles_per_wg; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();  // with an LDS-DMA in flight the fence in front of the barrier waits vmcnt(0): this chunk has landed
            const bool last = ch + 1 == nchunks;
            if (!(last && it + 1 == tiles_per_wg))
                stage(last ? tile + (int)gridDim.x : tile, last ? 0 : ch + 1, (ring + 1) % 3, wbuf ^ 1);
            const uint4* cX = sX + ring * XIMG;
            const uint4* pX = sX + ((ring + 2) % 3) * XIMG;  // previous chunk's image (carried blocks)
            const uint4* cW = sW + wbuf * WSLAB;
            const int ksteps = (ch & 3) == 3 ? 3 : 2;
            for (int ks = 0; ks < ksteps; ++ks) {
                // four {tap, 8 channel} blocks per K-step: lane group g takes block g; group 0 of the first K-step of
                // phases 1-3 reads the previous image (the carried blocks)
                const int tap = (4 * ks + g) % 9;
                const uint4* img = (ks == 0 && (ch & 3) != 0 && g == 0) ? pX : cX;
                const int toff = (tap / 3) * 34 + tap % 3;
                U4 a[3][4], b[3][4];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb)  // pixel blocks: rows 2 wave + (pb >> 1), columns 16 (pb & 1) ..
                        b[p][pb].u = img[p * NPIX + (2 * wave + (pb >> 1)) * 34 + 16 * (pb & 1) + l16 + toff];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)  // channel blocks
                        a[p][cb].u = cW[(p * (ksteps * 4) + 4 * ks + g) * 64 + 16 * cb + l16];
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        f32x4 c = acc[pb][cb];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][cb].v, b[1][pb].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][cb].v, b[2][pb].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][cb].v, b[0][pb].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][cb].v, b[1][pb].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][cb].v, b[0][pb].v, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][cb].v, b[0][pb].v, c, 0, 0, 0);
                        acc[pb][cb] = c;
                    }
            }
            ring = (ring + 1) % 3;
            wbuf ^= 1;
        }
        // epilogue: lane (l16, g) holds pixel 16 (pb & 1) + l16 of row 2 wave + (pb >> 1), channels 16 cb + 4 g .. + 3
        const int nb = tile % ntn;
        int sid = tile / ntn;
        const int tx0 = (sid % tiles_x) * 32, ty0 = ((sid / tiles_x) % tiles_y) * 16;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int gy = ty0 + 2 * wave + (pb >> 1), gx = tx0 + 16 * (pb & 1) + l16;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4 v = acc[pb][cb];
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                *reinterpret_cast<f32x4*>(y + ((size_t)gy * W + gx) * N + nb * 64 + 16 * cb + 4 * g) = v;
            }
        }
        tile += (int)gridDim.x;
    }
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 256, HW = argc > 3 ? atoi(argv[3]) : 128;
    const int B = 4;
    const int H = B * HW, W = HW;  // the batch stacked along y (a skeleton: no image borders inside)
    const int ntiles = (H / 16) * (W / 32) * (N / 64);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount < ntiles ? prop.multiProcessorCount : ntiles;
    const int tiles_per_wg = ntiles / grid;  // whole rounds only
    const size_t nx = (size_t)3 * H * W * C, nw = (size_t)3 * 9 * C * N;
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(1);
    auto rnd = [](float scale) {
        float v = 0.f;
        for (int k = 0; k < 12; ++k) v += (float)rand() / RAND_MAX;
        v = (v - 6.f) * scale;
        unsigned u;
        __builtin_memcpy(&u, &v, 4);
        return (unsigned short)(u >> 16);
    };
    for (size_t i = 0; i < nx; ++i) hx[i] = rnd(i < nx / 3 ? 1.f : (i < 2 * nx / 3 ? 1.f / 256 : 1.f / 65536));
    for (size_t i = 0; i < nw; ++i) hw[i] = rnd(i < nw / 3 ? 0.05f : (i < 2 * nw / 3 ? 0.05f / 256 : 0.05f / 65536));
    __bf16 *xp, *wp;
    float* y;
    CK(hipMalloc(&xp, nx * 2));
    CK(hipMalloc(&wp, nw * 2));
    CK(hipMalloc(&y, (size_t)H * W * N * 4));
    CK(hipMemcpy(xp, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wp, hw.data(), nw * 2, hipMemcpyHostToDevice));
    const size_t lds = (size_t)LDS_UNITS * 16;
    CK(hipFuncSetAttribute((const void*)skeleton, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k) skeleton<<<grid, 512, lds>>>(xp, wp, y, H, W, C, N, tiles_per_wg);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        // MFMAs actually executed: per chunk-quad 9 K-steps x 96 per wave
        const double mf = (double)grid * tiles_per_wg * 8 * (C / 32) * 9 * 96;
        const double alg = 2.0 * (double)grid * tiles_per_wg * 512 * 64 * C * 9;  // FLOPs of the convolution tiles covered
        printf("C=%d N=%d %dx%d  grid %d x %d tiles  %.3f ms  executed %.0f TFLOP/s bf16  = %.1f TFLOP/s of x6 work\n", C, N, HW, HW,
               grid, tiles_per_wg, ms, mf * 2 * 16 * 16 * 32 / ms * 1e-9, alg / ms * 1e-9);
    }
    return 0;
}
