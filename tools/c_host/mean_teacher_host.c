/* A Mean-Teacher training loop driven from plain C through the step-level C ABI (include/punet_hip.h, "Step-level runtime"):
 * no Python, no torch - the HIP runtime for device memory and libpunet_hip.so for everything else.  What a non-Python host of
 * the reference's hot path looks like (MeanTeacherTrainer._train_epoch_impl, mean_teacher_trainer.py:101-131: pseudo-labels
 * from the teacher on view 1, consensus mask, student ELBO forward / backward on view 2, regulariser, Adam, EMA).
 *
 *   mean_teacher_host <case.bin> <out.bin>
 * case.bin: int32 header {input_channels, levels, latent_dim, no_convs_fcomb, nf[levels], B, H, W, N, iterations, dice, products},
 *           float64 lr, then float32 arrays: params [P] (state-dict order, conv weights [ky][kx][cin][cout]), x1 [B][H][W][C],
 *           x2 [B][H][W][C], eps_pl [N][B][latent], eps_s [B][latent]   (the same batch and draws every iteration: a parity case)
 * out.bin:  float32 student [P], teacher [P], y [B][H][W], z [B][H][W], scalars [3] of the last iteration
 * tests/test_gpu_runtime.py::test_c_host_runs_the_training_loop compares out.bin with the Python trainers' fused step, bit for bit.
 *
 * Data parallelism without Python (SURVEY.md 8(b)'s punet_allreduce_grads sketch; built with -DPUNET_HOST_RCCL): with
 * $PUNET_C_HOST_RCCL=1 the host creates an RCCL communicator (ncclCommInitRank over $PUNET_C_HOST_WORLD ranks, default ONE: all a
 * one-GPU box can hold) and passes PunetRtHooks to the step: `after_unet` starts ncclAllReduce of the U-Net's slice
 * [0, punet_rt_unet_floats) of the flat gradient on a communication stream as soon as that slice is final (it then runs under the
 * encoders' backward), `before_update` reduces the remainder and makes the compute stream wait for both - the same split exchange
 * as my_trainer/fused.py::allreduce_grads, with 1 / world_size in the optimizer's state block.  With one rank the collectives are
 * the identity and out.bin must equal the run without them bit for bit (the same test, parametrised). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "punet_hip.h"
#ifdef PUNET_HOST_RCCL
#include <rccl/rccl.h>
typedef struct {
    ncclComm_t comm;
    hipStream_t compute, comm_stream;
    hipEvent_t ready, done;
    float* grad;
    size_t unet_floats, all_floats;
    int failed;
} DpCtx;
/* the U-Net's gradient slice is final and enqueued on the compute stream: reduce it on the communication stream from there on */
static void dp_after_unet(void* p) {
    DpCtx* c = (DpCtx*)p;
    if (hipEventRecord(c->ready, c->compute) != hipSuccess || hipStreamWaitEvent(c->comm_stream, c->ready, 0) != hipSuccess ||
        ncclAllReduce(c->grad, c->grad, c->unet_floats, ncclFloat, ncclSum, c->comm, c->comm_stream) != ncclSuccess)
        c->failed = 1;
}
/* the whole gradient is final: reduce the remainder, then the optimizer (next on the compute stream) waits for both reductions */
static void dp_before_update(void* p) {
    DpCtx* c = (DpCtx*)p;
    if (hipEventRecord(c->ready, c->compute) != hipSuccess || hipStreamWaitEvent(c->comm_stream, c->ready, 0) != hipSuccess ||
        ncclAllReduce(c->grad + c->unet_floats, c->grad + c->unet_floats, c->all_floats - c->unet_floats, ncclFloat, ncclSum, c->comm,
                      c->comm_stream) != ncclSuccess ||
        hipEventRecord(c->done, c->comm_stream) != hipSuccess || hipStreamWaitEvent(c->compute, c->done, 0) != hipSuccess)
        c->failed = 1;
}
#endif

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define PUNET(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, punet_last_error()); return 3; } } while (0)

static int read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static float* upload(FILE* f, size_t n) { /* n floats from the file to the device */
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    if (!h || read_exact(f, h, n * sizeof(float)) || hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess ||
        hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "upload of %zu floats failed\n", n);
        exit(4);
    }
    free(h);
    return d;
}

static int download(FILE* f, const float* d, size_t n) {
    float* h = (float*)malloc(n * sizeof(float));
    if (!h || hipMemcpy(h, d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess || fwrite(h, sizeof(float), n, f) != n) return -1;
    free(h);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    PunetRtConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    int head[4], dims[7];
    if (read_exact(f, head, sizeof head)) return 1;
    cfg.input_channels = head[0]; cfg.num_classes = 1; cfg.levels = head[1]; cfg.latent_dim = head[2]; cfg.no_convs_fcomb = head[3];
    if (cfg.levels < 1 || cfg.levels > 8 || read_exact(f, cfg.num_filters, cfg.levels * sizeof(int)) || read_exact(f, dims, sizeof dims)) return 1;
    const int B = dims[0], H = dims[1], W = dims[2], N = dims[3], iterations = dims[4], dice = dims[5], products = dims[6];
    double lr;
    if (read_exact(f, &lr, sizeof lr)) return 1;

    HIP(hipSetDevice(0));
    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    PunetRt* rt = NULL;
    PUNET(punet_rt_create(&cfg, &rt));
    PUNET(punet_rt_set_arithmetic(rt, products));
    PUNET(punet_rt_set_criterion(rt, dice));
    {   /* the same implementation switch the Python side reads: K5x on (default) or the up-blocks as written */
        const char* e = getenv("PUNET_UPCONV");
        PUNET(punet_rt_set_upconv(rt, !(e && e[0] == '0')));
    }
    const size_t P = (size_t)punet_rt_param_floats(rt), px = (size_t)B * H * W, L = (size_t)cfg.latent_dim;

    float *student = upload(f, P), *x1 = upload(f, px * cfg.input_channels), *x2 = upload(f, px * cfg.input_channels);
    float *eps_pl = upload(f, (size_t)N * B * L), *eps_s = upload(f, (size_t)B * L);
    fclose(f);
    float *teacher, *grad, *m, *v, *y, *z, *scalars, *hyper;
    double* hstate;
    void *tables, *arena;
    HIP(hipMalloc((void**)&teacher, P * 4)); HIP(hipMemcpy(teacher, student, P * 4, hipMemcpyDeviceToDevice)); /* deepcopy(model) */
    HIP(hipMalloc((void**)&grad, P * 4));    HIP(hipMemset(grad, 0, P * 4));
    HIP(hipMalloc((void**)&m, P * 4));       HIP(hipMemset(m, 0, P * 4));
    HIP(hipMalloc((void**)&v, P * 4));       HIP(hipMemset(v, 0, P * 4));
    HIP(hipMalloc((void**)&y, px * 4));      HIP(hipMalloc((void**)&z, px * 4));
    HIP(hipMalloc((void**)&scalars, 8 * 4)); HIP(hipMalloc((void**)&hyper, 4 * 4));
    HIP(hipMalloc((void**)&hstate, 8 * 8));
    /* optimizer state block: {adam steps, lr, beta1, beta2, EMA momentum, EMA rule (1 = constant), iteration, 1 / world_size} */
    const double h0[8] = {0.0, lr, 0.9, 0.999, 0.999, 1.0, 0.0, 1.0};
    HIP(hipMemcpy(hstate, h0, sizeof h0, hipMemcpyHostToDevice));
    HIP(hipMalloc(&tables, punet_rt_tables_bytes(rt)));
    PUNET(punet_rt_init_tables(rt, tables, stream));
    size_t ab = punet_rt_arena_bytes(rt, B, H, W, N, 1), ab0 = punet_rt_arena_bytes(rt, B, H, W, N, 0);
    if (ab0 > ab) ab = ab0;
    HIP(hipMalloc(&arena, ab));

    const PunetRtHooks* hooks = NULL;
#ifdef PUNET_HOST_RCCL
    PunetRtHooks hk;
    DpCtx dp;
    memset(&dp, 0, sizeof dp);
    const char* want = getenv("PUNET_C_HOST_RCCL");
    if (want && want[0] == '1') {
        const char* ws = getenv("PUNET_C_HOST_WORLD");
        const char* rk = getenv("PUNET_C_HOST_RANK");
        const int world = ws ? atoi(ws) : 1, rank = rk ? atoi(rk) : 0;
        ncclUniqueId id;
        if (world != 1) { fprintf(stderr, "this demonstration host exchanges the unique id in-process: one rank only\n"); return 6; }
        if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&dp.comm, world, id, rank) != ncclSuccess) {
            fprintf(stderr, "RCCL communicator could not be built\n");
            return 6;
        }
        HIP(hipStreamCreate(&dp.comm_stream));
        HIP(hipEventCreateWithFlags(&dp.ready, hipEventDisableTiming));
        HIP(hipEventCreateWithFlags(&dp.done, hipEventDisableTiming));
        dp.compute = stream; dp.grad = grad; dp.unet_floats = (size_t)punet_rt_unet_floats(rt); dp.all_floats = P;
        hk.after_unet = dp_after_unet; hk.before_update = dp_before_update; hk.ctx = &dp;
        hooks = &hk;
        const double scale = 1.0 / world;  /* gradients are summed over ranks; the optimizer's state block carries 1 / world_size */
        HIP(hipMemcpy((char*)hstate + 7 * sizeof(double), &scale, sizeof scale, hipMemcpyHostToDevice));
        printf("data parallel: RCCL communicator of %d rank(s), U-Net slice %zu of %zu gradient floats reduced under the encoders' backward\n",
               world, dp.unet_floats, P);
    }
#endif
    for (int it = 0; it < iterations; ++it)
        PUNET(punet_rt_self_training_step(rt, student, teacher, grad, m, v, hstate, hyper, /*ema=*/1, x1, x2, B, H, W, eps_pl, N, eps_s,
                                          0.9f, 0.1f, /*masking=*/1, /*use_consensus=*/1, /*beta=*/1.f, /*reg_weight=*/1e-5f, 0.9f,
                                          0.999f, 1e-8f, y, z, scalars, hooks, arena, ab, stream));
    HIP(hipStreamSynchronize(stream));
#ifdef PUNET_HOST_RCCL
    if (hooks) {
        if (dp.failed) { fprintf(stderr, "a collective could not be enqueued\n"); return 7; }
        HIP(hipStreamSynchronize(dp.comm_stream));
        ncclCommDestroy(dp.comm);
    }
#endif

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    if (download(o, student, P) || download(o, teacher, P) || download(o, y, px) || download(o, z, px) || download(o, scalars, 3)) return 5;
    fclose(o);
    float sc[3];
    HIP(hipMemcpy(sc, scalars, sizeof sc, hipMemcpyDeviceToHost));
    printf("%d iterations, %zu parameters, arena %.1f MB (used %.1f): rec %.6f kl %.6f reg %.6f\n", iterations, P, ab / 1e6,
           punet_rt_last_arena_used(rt) / 1e6, sc[0], sc[1], sc[2]);
    punet_rt_destroy(rt);
    return 0;
}
