// Shared helpers for libpunet_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PUNET_OK 0
#define PUNET_ERR_ARG (-1)
#define PUNET_ERR_HIP (-2)

void punet_set_error(const char* fmt, ...);

#define PUNET_CHECK_ARG(cond, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            punet_set_error(__VA_ARGS__);   \
            return PUNET_ERR_ARG;           \
        }                                   \
    } while (0)

#define PUNET_LAUNCH_CHECK(name)                                                  \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            punet_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return PUNET_ERR_HIP;                                                 \
        }                                                                         \
    } while (0)

// out[(row / cs) * cin_total + ci_off + row % cs][col] (+)= scale * sum_s partial[s * slab_stride + row * cols + col]
// (slabs summed in slice order: deterministic).  Defined in pointwise.hip.
int punet_launch_reduce_slabs(const float* partial, int nslices, size_t slab_stride, int rows, int cols, int cs,
                              int cin_total, int ci_off, float* out, int accumulate, float scale, hipStream_t stream);

// weight slabs (layout as above) and bias slabs (1 x cols per slice) of one layer in ONE launch; bit-identical to two calls
int punet_launch_reduce_slabs_pair(const float* partial_w, size_t stride_w, int rows, int cols, int cs, int cin_total, int ci_off,
                                   float* out_w, const float* partial_b, size_t stride_b, float* out_b, int nslices,
                                   int accumulate, float scale, hipStream_t stream);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// compute units of the current device (hipDeviceProp_t::multiProcessorCount; 256 on an MI355X), queried once per device;
// 256 when no device can be queried (host-only calls such as punet_wgrad_x6_slices in a container without a GPU)
int punet_num_cus();

// decode the per-call options word of the split-plane kernels (PUNET_X6_OPTS in punet_hip.h); 0 = defaults
static inline int punet_x6_planes_of(int opts) {  // bf16 planes per operand: 3 (six products), 2 (three), 1 (one)
    const int p = opts & 0xff;
    return p == 0 || p == 6 ? 3 : (p == 3 ? 2 : (p == 1 ? 1 : -1));
}
static inline int punet_x6_shape_of(int opts) {  // 16 or 32
    const int s = (opts >> 8) & 0xff;
    return s == 0 ? 16 : (s == 16 || s == 32 ? s : -1);
}

// C/D fragment map of v_mfma_f32_32x32x2_f32 (dtype independent on gfx950):
//   col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int mfma32_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ReLU gate bits of a [B,H,W,C] activation (layout documented at punet_gate_dwords in punet_hip.h): the 4 gate bits of
// channels c .. c+3 (c a multiple of 4) of pixel (b, y, x).
__device__ __forceinline__ unsigned punet_gate_nibble(const unsigned* __restrict__ bits, size_t b, int H, int W, int C, int y,
                                                      int x, int c) {
    const int tiles = (W + 31) >> 5, cbs = (C + 63) >> 6;
    const size_t idx = (((((size_t)b * H + y) * tiles + (x >> 5)) * cbs + (c >> 6)) << 6) + (x & 31) + 32 * ((c & 7) >> 2);
    return (bits[idx] >> (16 * ((c & 63) >> 5) + 4 * ((c & 31) >> 3))) & 0xFu;
}

// Bilinear x2 interpolation with align_corners=True (unet_blocks.py:51): source index of output index o along one axis as ATen
// computes it - scale = (in - 1) / (out - 1) in fp32, src = scale * o ROUNDED (a contracted fma(scale, o, -i0) is more exact but
// moves the interpolation weight by ulp(src) ~ 2e-6, which the cancellation src - i0 exposes).  i0 = left / upper neighbour,
// ip = 1 when the second neighbour exists, l1 = its weight (the first neighbour's is 1 - l1).  Shared by K5 (pointwise.hip) and the
// low-resolution form of the up-blocks' first convolution (upconv.hip) so that both interpolate with the same weights.
__device__ __forceinline__ void punet_up_src(int o, float scale, int in_size, int& i0, int& ip, float& l1) {
#pragma clang fp contract(off)
    const float src = __fmul_rn(scale, (float)o);
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    ip = (i0 < in_size - 1) ? 1 : 0;
    l1 = src - (float)i0;
}
// weight with which output index o reads input index i along one axis
__device__ __forceinline__ float punet_up_weight(int o, int i, float scale, int in_size) {
    int i0, ip;
    float l1;
    punet_up_src(o, scale, in_size, i0, ip, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i0 + ip == i) w += l1;
    return w;
}

// fp32 -> bf16 planes h + m + l (round to nearest each, exact residuals) for a PAIR of values in packed arithmetic:
// v_cvt_pk_bf16_f32 converts both, the packed dword IS the LDS image's element pair, its halves shifted / masked back to
// fp32 feed one v_pk_add_f32 per level - 9 vector instructions per pair instead of ~12 for two scalar splits, bit-identical.
typedef float punet_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 punet_bf16x2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const punet_f32x2 x = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, punet_bf16x2));
    m = l = 0u;
    if (NP > 1) {
        const punet_f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
        const punet_f32x2 r1 = x - hf;
        m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, punet_bf16x2));
        if (NP > 2) {
            const punet_f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
            const punet_f32x2 r2 = r1 - mf;
            l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, punet_bf16x2));
        }
    }
}
// one float4 unit -> its planes as packed 8-byte units (4 x bf16 each)
template <int NP>
__device__ __forceinline__ void split3_quad(const float4& r, uint2& h, uint2& m, uint2& l) {
    split3_pair<NP>(r.x, r.y, h.x, m.x, l.x);
    split3_pair<NP>(r.z, r.w, h.y, m.y, l.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16-operand arithmetic ("f16x3" = two fp16 planes, hh + hl + lh: 22-23 significand bits; "f16" = one plane: the analogue of
// the reference scripts' fp16 autocast, LIVECell/livecell_mt.py:107).  fp16 has 5 exponent bits, so every tensor is brought into
// range by an EXACT power of two taken from its amax (max |v|, published by the producing kernel into a device slot with an
// order-independent atomic max on the float's bits): s = 2^(14 - floor(log2(amax))), i.e. the largest magnitude lands in
// [2^14, 2^15) and everything down to 2^-24 (fp16's smallest subnormal) of that keeps absolute precision 2^-25 - 39 binades
// below the tensor's maximum.  Products are rescaled by the exact inverse in the epilogue.  (The reference's GradScaler does
// the same job with one global loss scale; a per-tensor scale cannot overflow and needs no skipped steps.)
typedef _Float16 punet_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 punet_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 punet_bf16x8 __attribute__((ext_vector_type(8)));

// scale (a power of two) for a tensor whose amax slot holds `amax` (>= 0; 0 or non-finite -> 1)
__device__ __forceinline__ float punet_f16_scale_of(float amax) {
    const unsigned b = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((b >> 23) & 0xffu);            // biased exponent of amax (0: zero / subnormal, 255: inf / nan)
    if (e == 0 || e == 255) return 1.f;
    int se = 127 + 14 - (e - 127);                     // biased exponent of 2^(14 - floor(log2 amax))
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
__device__ __forceinline__ float punet_f16_scale_from(const float* slot) { return slot ? punet_f16_scale_of(*slot) : 1.f; }

// One lane's contribution to an amax slot.  Thousands of workgroups publish into ONE address and contended atomics queue at
// the memory side (~12 ns each: 8,192 blocks of the first-layer conv cost 30 us of a 78 us kernel): look first (an L2 read that
// bypasses the L1) and skip the atomic when the slot already holds at least this value - after the first few publications almost
// every one does.  A stale look only costs a redundant atomic; the maximum itself is still decided by atomicMax (order-independent).
__device__ __forceinline__ void punet_amax_atomic(float* slot, float m) {
    const unsigned mb = __builtin_bit_cast(unsigned, m);
    if (mb > __hip_atomic_load(reinterpret_cast<unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<unsigned*>(slot), mb);
}

// amax slot update: |v| as unsigned bits (monotone for non-negative floats); NaN-free inputs assumed (a NaN would publish a
// huge pattern -> scale 2^-113..: the consumer's values underflow to zero, the NaN itself still propagates through the MFMAs)
__device__ __forceinline__ void punet_amax_publish(float* slot, float wave_local_max) {
    float m = wave_local_max;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) punet_amax_atomic(slot, m);
}

// the same with ONE atomic per 256-thread workgroup: `lds` = 4 floats of LDS nobody else touches any more (the caller has passed a
// barrier behind the last use of that memory); every thread of the workgroup must call
__device__ __forceinline__ void punet_amax_publish_block256(float* slot, float thread_local_max, float* lds) {
    float m = thread_local_max;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) punet_amax_atomic(slot, fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3])));
}

// scaled fp32 -> fp16 planes h (+ l) of a PAIR of values: four mixed-precision fmas (v_fma_mixlo / mixhi_f16: fp32 fma, result
// rounded into one half of the destination; the residual reads h straight from its fp16 half) instead of v_pk_mul_f32 +
// v_cvt_pk_f16_f32 + 2 x v_cvt_f32_f16 + v_pk_fma_f32 + v_cvt_pk_f16_f32.  Same values bit for bit (s is a power of two: a * s is
// exact) except that a * s = -0 gives h = +0.
template <int NP>
__device__ __forceinline__ void splitf16_pair(float a, float b, float s, unsigned& h, unsigned& l) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PUNET_NO_FMA_MIX)
    if (NP > 1) {  // (one plane: the packed multiply + packed conversion below is as short and has no half-register dependency)
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(s), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(s), "v"(h));
        return;
    }

#endif
    {
    const punet_f32x2 x = {a * s, b * s};
    const punet_f16x2 h2 = __builtin_convertvector(x, punet_f16x2);
    h = __builtin_bit_cast(unsigned, h2);
    l = 0u;
    if (NP > 1) {
        const punet_f32x2 r = x - __builtin_convertvector(h2, punet_f32x2);
        l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, punet_f16x2));
    }
    }
}
template <int NP>
__device__ __forceinline__ void splitf16_quad(const float4& r, float s, uint2& h, uint2& l) {
    splitf16_pair<NP>(r.x, r.y, s, h.x, l.x);
    splitf16_pair<NP>(r.z, r.w, s, h.y, l.y);
}

// one MFMA on 16-byte fragments held as uint4, element type selected at compile time
template <bool F16>
__device__ __forceinline__ f32x4 punet_mfma16(const uint4& a, const uint4& b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(punet_f16x8, a), __builtin_bit_cast(punet_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(punet_bf16x8, a), __builtin_bit_cast(punet_bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16 punet_mfma32(const uint4& a, const uint4& b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(punet_f16x8, a), __builtin_bit_cast(punet_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(punet_bf16x8, a), __builtin_bit_cast(punet_bf16x8, b), c, 0, 0, 0);
}

// Zero `bytes` (a multiple of 4) of device memory with a KERNEL, not hipMemsetAsync: every pass of the step-level runtime zeroes a
// few small regions (amax slots, arrival counters) between kernels, and inside a captured hipGraph a memset node is executed by
// another path than the kernel nodes around it - round 5 saw a replayed f16x3 step read amax slots that its memset node had not
// cleared yet (tests/test_gpu_graph.py::test_graph_replay_across_conv_mode_switches_equals_eager under PUNET_CONV_MODE=f16x3:
// wrong losses, then NaN, depending on what else the process had enqueued).  As a kernel node it is ordered like everything else.
static __global__ void punet_zero_fill_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t punet_zero_fill(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (!n) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    punet_zero_fill_kernel<<<grid, 256, 0, st>>>((unsigned*)p, n);
    return hipGetLastError();
}


// $PUNET_FOLD (an A/B switch, read once per process): tiles / strips laid over the virtual row of all B images where the width is
// not whole 32-pixel tiles (round 5; conv3x3_x6b_kernel<..., FOLD> and wgrad3x3_x6r_kernel<..., FOLD>).  unset / 1 = wherever it
// saves tiles, 0 = per image as before round 5, 2 = the forward / input-gradient kernel only (whose results are bit-identical either
// way; the weight gradient's agree to rounding).
static inline int punet_fold_mode() {
    static const int m = [] {
        const char* e = getenv("PUNET_FOLD");
        return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
    }();
    return m;
}
