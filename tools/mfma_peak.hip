// Practical bf16 MFMA ceiling of the device under sustained load: register-only v_mfma_f32_32x32x16_bf16 chains with
// (a) random normal operands, (b) all-zero operands (data-dependent power), at 1, 2 and 4 waves per SIMD, and a variant
// that adds the conv_x6 kernel's LDS read mix (12 ds_read_b128 per 24 MFMAs).  Standalone: hipcc --offload-arch=gfx950
// tools/mfma_peak.hip -o gpurun_out/mfma_peak && ./gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

template <int WITH_LDS, int ORDER>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ uint4 sm[WITH_LDS ? 2048 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    if (WITH_LDS) {
        for (int i = tid; i < 2048; i += 256) sm[i] = src[i];
        __syncthreads();
    }
    union U { uint4 u; bf16x8 v; };
    U a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        a[i].u = src[(lane + 64 * i) & 2047];
        b[i].u = src[(lane + 64 * (i + 6)) & 2047];
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) {
            const int base = (it * 7) & 1023;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[i].u = sm[(base + lane + 64 * i) & 2047];
                b[i].u = sm[(base + lane + 64 * (i + 6)) & 2047];
            }
        }
        // 24 MFMAs: the conv_x6 K-step (2x2 tiles x 6 plane products)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int mb = t >> 1, nb = t & 1;
            f32x16 c = acc[t];
            if (ORDER == 0) {  // the kernels' order: small terms first (mm lh hl mh hm hh)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 + mb].v, b[2 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4 + mb].v, b[0 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0 + mb].v, b[4 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 + mb].v, b[0 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0 + mb].v, b[2 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0 + mb].v, b[0 + nb].v, c, 0, 0, 0);
            } else {  // consecutive MFMAs share one operand (lh | mh mm | hm hh hl)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4 + mb].v, b[0 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 + mb].v, b[0 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 + mb].v, b[2 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0 + mb].v, b[2 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0 + mb].v, b[0 + nb].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0 + mb].v, b[4 + nb].v, c, 0, 0, 0);
            }
            acc[t] = c;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;  // keep the chains alive
}


// the same operand registers through v_mfma_f32_16x16x32_bf16 (half the accumulator traffic per MAC, twice the operand
// traffic): 48 MFMAs of 16 cycles = the 24 x 32 cycles of the K-step above
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
template <int WITH_LDS>
__global__ __launch_bounds__(256) void mfma_loop_16(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ uint4 sm[WITH_LDS ? 2048 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    if (WITH_LDS) {
        for (int i = tid; i < 2048; i += 256) sm[i] = src[i];
        __syncthreads();
    }
    union U { uint4 u; bf16x8 v; };
    U a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        a[i].u = src[(lane + 64 * i) & 2047];
        b[i].u = src[(lane + 64 * (i + 6)) & 2047];
    }
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) {  // the same 12 x 16-byte fragment reads per 768 MFMA cycles as the 32x32 variant
            const int base = (it * 7) & 1023;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[i].u = sm[(base + lane + 64 * i) & 2047];
                b[i].u = sm[(base + lane + 64 * (i + 6)) & 2047];
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // eight DIFFERENT operand combinations: tiles 4-7 swap the roles of the a and b registers
            const int mb = (t >> 1) & 1, nb = t & 1;
            const bf16x8 *A = t < 4 ? &a[0].v : &b[0].v, *B = t < 4 ? &b[0].v : &a[0].v;
            f32x4 c = acc[t];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 + mb], B[2 + nb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[4 + mb], B[0 + nb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0 + mb], B[4 + nb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2 + mb], B[0 + nb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0 + mb], B[2 + nb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0 + mb], B[0 + nb], c, 0, 0, 0);
            acc[t] = c;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}

template <int WITH_LDS>
static void run16(const char* label, const uint4* src, float* out, int wg_per_cu, int iters, int reps) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    mfma_loop_16<WITH_LDS><<<grid, 256>>>(src, out, iters);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        mfma_loop_16<WITH_LDS><<<grid, 256>>>(src, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4 * iters * 48 * 2.0 * 16 * 16 * 32;
        printf("%-34s waves/SIMD=%d rep=%d  %.3f ms  %.1f TFLOP/s bf16  (x6-equivalent %.1f)\n", label, wg_per_cu, r, ms,
               flops / ms * 1e-9, flops / ms * 1e-9 / 6);
    }
}

template <int WITH_LDS, int ORDER = 0>
static void run(const char* label, const uint4* src, float* out, int wg_per_cu, int iters, int reps) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    mfma_loop<WITH_LDS, ORDER><<<grid, 256>>>(src, out, iters);  // warm
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        mfma_loop<WITH_LDS, ORDER><<<grid, 256>>>(src, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
        printf("%-34s waves/SIMD=%d rep=%d  %.3f ms  %.1f TFLOP/s bf16  (x6-equivalent %.1f)\n", label, wg_per_cu, r, ms,
               flops / ms * 1e-9, flops / ms * 1e-9 / 6);
    }
}

int main() {
    const int n = 2048 * 8;  // bf16 elements
    std::vector<unsigned short> h(n);
    srand(1);
    auto fill = [&](int mode) {
        for (int i = 0; i < n; ++i) {
            if (mode == 0) { h[i] = 0; continue; }
            // ~N(0,1) via sum of uniforms, truncated to bf16
            float v = 0.f;
            for (int k = 0; k < 12; ++k) v += (float)rand() / RAND_MAX;
            v -= 6.f;
            if (mode == 2) v *= 1e-3f * (float)(1 + rand() % 1000);
            unsigned u;
            __builtin_memcpy(&u, &v, 4);
            h[i] = (unsigned short)(u >> 16);
        }
    };
    uint4* src;
    float* out;
    CK(hipMalloc(&src, n * 2));
    CK(hipMalloc(&out, 64));
    const int iters = 20000;  // ~50-100 ms launches: long enough for the power controller to settle
    for (int mode = 1; mode >= 0; --mode) {
        fill(mode);
        CK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
        const char* d = mode ? "normal data" : "zero data";
        char label[64];
        for (int w : {1, 2, 4}) {
            snprintf(label, sizeof label, "registers only, %s", d);
            run<0>(label, src, out, w, iters, 3);
        }
        for (int w : {1, 2}) {
            snprintf(label, sizeof label, "+12 ds_read_b128/24 mfma, %s", d);
            run<1>(label, src, out, w, iters, 3);
        }
        snprintf(label, sizeof label, "16x16x32 MFMAs, %s", d);
        run16<0>(label, src, out, 2, iters, 3);
        snprintf(label, sizeof label, "16x16x32 + 12 ds_read_b128, %s", d);
        run16<1>(label, src, out, 2, iters, 3);
        snprintf(label, sizeof label, "operand-sharing order, %s", d);
        run<0, 1>(label, src, out, 2, iters, 3);
        snprintf(label, sizeof label, "sharing order + LDS, %s", d);
        run<1, 1>(label, src, out, 2, iters, 3);
    }
    return 0;
}
