// K13: weak / strong view generation for the self-training steps, on the device.
//
// The reference builds its two views per patch on DataLoader workers: my_standardize_torch (prob_utils/my_utils/util.py:9-14)
// followed by randomly applied GaussianBlur / AdditiveGaussianNoise / RandomContrast (LIVECell/livecell_fm.py:43-67,
// MitoEM/common.py:50-68).  Once the training step runs ~300x faster than on the CPU those workers are the bottleneck, so
// the deterministic arithmetic lives here and only the random DECISIONS (apply or not, kernel size, sigma, noise scale,
// alpha) are drawn on the host, per item, and handed over as a small parameter table.
//
// All of it is HBM/L2-bound plane arithmetic on NCHW fp32 planes (the loaders' layout; C = 1 or 3):
//   stats   : per item mean and unbiased std over C*H*W, fp64 accumulation, fixed-order two-level reduction
//   blur    : separable Gaussian, reflect padding, taps exp(-0.5 (i/sigma)^2) / sum as torchvision builds them;
//             the horizontal pass standardises on load, the vertical pass carries the noise + contrast epilogue
// Algorithmic bytes: 4 B read + 4 B written per pixel and pass (+ 4 B noise); taps hit L1/L2.
#include "common.h"
#include "punet_hip.h"

namespace {

constexpr int MAXK = 64;      // largest supported kernel size (torch_em draws at most 2*24+1 = 49)
constexpr int STAT_SLICES = 64;

__global__ __launch_bounds__(256) void aug_stats_partial_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                                long n) {
    __shared__ double sh[2][256];
    const int s = blockIdx.x, b = blockIdx.y;
    const long per = (n + STAT_SLICES - 1) / STAT_SLICES;
    const long i0 = s * per, i1 = i0 + per < n ? i0 + per : n;
    double a = 0.0, q = 0.0;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const double v = (double)x[(size_t)b * n + i];
        a += v;
        q += v * v;
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[((size_t)b * STAT_SLICES + s) * 2 + 0] = sh[0][0];
        partial[((size_t)b * STAT_SLICES + s) * 2 + 1] = sh[1][0];
    }
}

__global__ void aug_stats_finalize_kernel(const double* __restrict__ partial, float* __restrict__ stats, long n, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a = 0.0, q = 0.0;
    for (int s = 0; s < STAT_SLICES; ++s) {
        a += partial[((size_t)b * STAT_SLICES + s) * 2 + 0];
        q += partial[((size_t)b * STAT_SLICES + s) * 2 + 1];
    }
    const double mean = a / (double)n;
    double var = n > 1 ? (q - (double)n * mean * mean) / (double)(n - 1) : 0.0;  // torch.std(): Bessel-corrected
    if (var < 0.0) var = 0.0;
    stats[b * 2 + 0] = (float)mean;
    stats[b * 2 + 1] = (float)sqrt(var);
}

__device__ __forceinline__ int reflect(int i, int n) {  // torch 'reflect' padding: -1 -> 1, n -> n-2
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// taps of this item into LDS; returns the radius
__device__ __forceinline__ int load_taps(const float* __restrict__ params, int b, float* sw, int rmax) {
    int ks = (int)params[b * 4 + 0];
    // the host validates the table; clamp anyway so that a bad entry can never index out of bounds
    ks = ks < 1 ? 1 : (ks > MAXK - 1 ? MAXK - 1 : ks) | 1;
    if ((ks - 1) / 2 > rmax) ks = 2 * rmax + 1;
    const float sigma = params[b * 4 + 1];
    if (threadIdx.x == 0) {
        const float half = (ks - 1) * 0.5f;
        float tot = 0.f;
        for (int k = 0; k < ks; ++k) {
            const float t = (-half + (float)k) / sigma;
            sw[k] = ks == 1 ? 1.f : expf(-0.5f * t * t);
            tot += sw[k];
        }
        for (int k = 0; k < ks; ++k) sw[k] /= tot;
    }
    __syncthreads();
    return (ks - 1) / 2;
}

// horizontal pass: tmp[b][c][y][x] = sum_k w[k] * standardised(x[b][c][y][reflect(x + k - r)])
__global__ __launch_bounds__(256) void aug_blur_h_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                         const float* __restrict__ params, float* __restrict__ tmp, int C,
                                                         int H, int W, float eps) {
    __shared__ float sw[MAXK];
    const int b = blockIdx.y;
    const int r = load_taps(params, b, sw, W - 1);
    const float mean = stats ? stats[b * 2 + 0] : 0.f;
    const float den = stats ? stats[b * 2 + 1] + eps : 1.f;
    const long n = (long)C * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int px = (int)(i % W);
        const float* row = x + (size_t)b * n + (i - px);
        float acc = 0.f;
        for (int k = -r; k <= r; ++k) {
            float v = row[reflect(px + k, W)];
            if (stats) v = (v - mean) / den;
            acc += sw[k + r] * v;
        }
        tmp[(size_t)b * n + i] = acc;
    }
}

// vertical pass + epilogue: out = cmean + alpha * ((blur + noise_scale * noise) - cmean)
__global__ __launch_bounds__(256) void aug_blur_v_kernel(const float* __restrict__ tmp, const float* __restrict__ noise,
                                                         const float* __restrict__ params, float* __restrict__ out, int C,
                                                         int H, int W, float cmean) {
    __shared__ float sw[MAXK];
    const int b = blockIdx.y;
    const int r = load_taps(params, b, sw, H - 1);
    const float nscale = params[b * 4 + 2], alpha = params[b * 4 + 3];
    const long n = (long)C * H * W, plane = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long pl = i / plane, rem = i % plane;
        const int py = (int)(rem / W), px = (int)(rem % W);
        const float* col = tmp + (size_t)b * n + pl * plane + px;
        float acc = 0.f;
        for (int k = -r; k <= r; ++k) acc += sw[k + r] * col[(size_t)reflect(py + k, H) * W];
        if (noise && nscale != 0.f) acc += nscale * noise[(size_t)b * n + i];
        out[(size_t)b * n + i] = cmean + alpha * (acc - cmean);
    }
}

}  // namespace

extern "C" size_t punet_standardize_workspace_doubles(int B) { return (size_t)B * STAT_SLICES * 2; }

extern "C" int punet_standardize_stats(const float* x, float* stats, double* workspace, int B, long n, void* stream) {
    PUNET_CHECK_ARG(x && stats && workspace && B > 0 && n > 0, "punet_standardize_stats: bad args");
    hipStream_t st = (hipStream_t)stream;
    aug_stats_partial_kernel<<<dim3(STAT_SLICES, B), 256, 0, st>>>(x, workspace, n);
    PUNET_LAUNCH_CHECK("aug_stats_partial");
    aug_stats_finalize_kernel<<<(B + 63) / 64, 64, 0, st>>>(workspace, stats, n, B);
    PUNET_LAUNCH_CHECK("aug_stats_finalize");
    return PUNET_OK;
}

extern "C" int punet_augment_view(const float* x, const float* stats, const float* noise, const float* params,
                                  int max_ksize, float contrast_mean, float eps, float* tmp, float* out, int B, int C, int H,
                                  int W, void* stream) {
    PUNET_CHECK_ARG(x && params && tmp && out && B > 0 && C > 0 && H > 0 && W > 0, "punet_augment_view: bad args");
    PUNET_CHECK_ARG(max_ksize >= 1 && max_ksize <= MAXK && max_ksize % 2 == 1, "punet_augment_view: kernel sizes must be odd and <= %d", MAXK);
    PUNET_CHECK_ARG(max_ksize / 2 < H && max_ksize / 2 < W, "punet_augment_view: reflect padding needs radius < image size");
    PUNET_CHECK_ARG(tmp != x && tmp != out, "punet_augment_view: tmp must not alias x or out");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)C * H * W;
    long gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    aug_blur_h_kernel<<<dim3((unsigned)gx, B), 256, 0, st>>>(x, stats, params, tmp, C, H, W, eps);
    PUNET_LAUNCH_CHECK("aug_blur_h");
    aug_blur_v_kernel<<<dim3((unsigned)gx, B), 256, 0, st>>>(tmp, noise, params, out, C, H, W, contrast_mean);
    PUNET_LAUNCH_CHECK("aug_blur_v");
    return PUNET_OK;
}
