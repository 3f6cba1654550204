// Step-level runtime of libpunet_hip.so: the graph walk of the Probabilistic U-Net - forward(patch, segm, training)
// (probabilistic_unet.py:285-293), sample / consensus (mean_teacher_trainer.py:72-88), elbo (probabilistic_unet.py:341-371) and the
// whole backward that autograd performs in the reference - as NATIVE host code behind the C ABI (punet_create,
// punet_forward_features, punet_elbo_fwd_bwd, ...).  A host in any language drives a training step with four calls and no
// knowledge of the layer graph; the Python engine (engine.py) remains for the parts of the constructor space this runtime does
// not cover and enqueues exactly the same kernels in the same order, so the two agree bit for bit (tests/test_gpu_runtime.py).
//
// Covered: num_classes = 1, num_filters[0] in {32, 64} (the fused Fcomb kernels), no_convs_fcomb in {2, 3}, every other filter
// count a multiple of 8, input_channels + 1 <= 7, the BCE and Dice criteria, the six-product parity arithmetic ("x6") and the
// fp16-operand fast modes - i.e. every configuration the reference's scripts build, and the bench workload.
//
// Memory: the runtime owns nothing on the device.  The caller hands every call an ARENA (device bytes, punet_rt_arena_bytes) that
// is carved up by a bump allocator - activations, gate bits, gradients, weight planes, split-K slabs - and a small TABLES buffer
// (punet_rt_tables_bytes, filled once by punet_rt_init_tables).  Nothing is allocated, freed or synchronised inside a call, so a
// call is hipGraph-capturable as it stands.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "punet_hip.h"

namespace {

constexpr int SMALL_CIN = 7;
constexpr int X6_OPTS = PUNET_X6_OPTS(6, 16);

struct Conv {
    std::string name;
    int cin, cout, k;
    long w_off, b_off;
    int x6_index = -1;  // row of the weight-split table (= slot of its amax in the fp16-operand arithmetics)
    long w_numel() const { return (long)k * k * cin * cout; }
};

struct Entry {
    std::string name;
    long off, numel;
    int shape[4];
    int ndim;
};

struct Act {
    float* t = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    bool relu_out = false, needs_grad = true;
    unsigned* bits = nullptr;
    float* grad = nullptr;  // gradient buffer once one exists (the "gm" map of engine.py)
    float* amax = nullptr;   // fp16-operand arithmetics: amax slot of t (common.h); gamax: of grad
    float* gamax = nullptr;
    Act* up_of = nullptr;        // this tensor is the bilinear x2 of that one (an as-written up-block: K5x's input gradient applies)
    bool up_bwd_done = false;    // ... and its consumer's backward has already produced that tensor's gradient in the low-resolution form
    size_t numel() const { return (size_t)B * H * W * C; }
};

struct Op {
    int kind;  // 0 conv, 1 pool, 2 up, 3 up-block entry conv in the low-resolution form (K5x: x0 = the LOW-resolution source, x1 = bridge)
    int conv;
    Act *x0, *x1, *y;
};

struct Arena {
    char* base;
    size_t cap, used = 0;
    void* take(size_t bytes) {  // (a real pass runs only after a dry pass has shown that everything fits: no overflow here)
        const size_t a = (used + 255) & ~(size_t)255;
        used = a + bytes;
        return base + a;
    }
    float* floats(size_t n) { return (float*)take(n * sizeof(float)); }
};

// take() aligns OFFSETS to 256 bytes and the kernels read what it carves with 16-byte vector loads (planes by LDS-DMA rows): the
// base the caller hands over has to be 256-byte aligned itself (hipMalloc and torch's allocator are; a sub-allocating C host may
// not be - ADVICE r3).  Checked by every entry point that takes an arena, before anything is launched.
inline bool arena_aligned(const void* arena) { return ((uintptr_t)arena & 255) == 0; }

}  // namespace

struct PunetRt {
    PunetRtConfig cfg;
    std::vector<Conv> convs;
    std::vector<Entry> entries;
    std::vector<std::vector<int>> unet_down, unet_up, enc_prior, enc_post;
    int head_prior = -1, head_post = -1, fc1 = -1, fcmid = -1, fc3 = -1;
    long numel = 0, unet_numel = 0;
    // host copies of the device tables; device pointers once punet_rt_init_tables has run
    std::vector<long> x6_table;  // [nconv + 1][4]
    int x6_nconv = 0;
    long x6_units = 0;
    std::vector<long> l2_table;  // [nchunks][3]
    int l2_nchunks = 0, l2_ntensors = 0;
    long* d_x6 = nullptr;
    long* d_l2 = nullptr;
    int f16_planes = 0;  // 0: six-product bf16 parity arithmetic; 2: "f16x3"; 1: "f16" (punet_rt_set_arithmetic)
    int dice = 0;        // criterion of punet_rt_elbo_fwd_bwd: 0 BCE, 1 Dice (punet_rt_set_criterion)
    int upconv = 1;      // K5x: up-block entry convs with the `up` source contracted at low resolution (punet_rt_set_upconv; x6 only)
    // per-call state
    size_t last_arena_used = 0;
    std::vector<Act*> acts;
    Act* new_act(float* t, int B, int H, int W, int C, bool relu, bool needs_grad = true) {
        Act* a = new Act();
        a->t = t; a->B = B; a->H = H; a->W = W; a->C = C; a->relu_out = relu; a->needs_grad = needs_grad;
        acts.push_back(a);
        return a;
    }
    void clear_acts() {
        for (Act* a : acts) delete a;
        acts.clear();
    }
};

namespace {

int add_conv(PunetRt* rt, const std::string& name, int cin, int cout, int k) {
    Conv c;
    c.name = name; c.cin = cin; c.cout = cout; c.k = k; c.w_off = c.b_off = -1;
    rt->convs.push_back(c);
    return (int)rt->convs.size() - 1;
}

// the reference's nn.Sequential indices of the Encoder's convs (probabilistic_unet.py:44-63): a pool in front of blocks 1.., a
// ReLU behind every conv
std::vector<std::vector<int>> encoder_indices(int n_blocks) {
    std::vector<std::vector<int>> out;
    int idx = 0;
    for (int i = 0; i < n_blocks; ++i) {
        if (i) ++idx;
        std::vector<int> blk;
        for (int k = 0; k < 3; ++k) { blk.push_back(idx); idx += 2; }
        out.push_back(blk);
    }
    return out;
}

#define RT_CALL(expr)                  \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_ != PUNET_OK) return rc_; \
    } while (0)
// a kernel launch through the per-op ABI: skipped in a DRY pass (which only walks the graph and sizes the arena)
#define RT_LAUNCH(expr)                \
    do {                               \
        if (!dry) RT_CALL(expr);       \
    } while (0)

struct Pass {  // one forward / backward walk
    PunetRt* rt;
    const float* flat;
    float* gflat;
    Arena* ar;
    void* stream;
    const __bf16* planes0 = nullptr;  // forward-orientation weight planes of `flat`
    const __bf16* planes1 = nullptr;  // input-gradient orientation
    float* ws_wgrad = nullptr;
    float* ws_chan = nullptr;
    bool fuse_pool = true;
    bool dry = false;
    int f16 = 0;              // fp16 planes per operand (0: bf16 six-product arithmetic)
    float* wamax = nullptr;   // [nconv] amax slots of the weight slabs
    float* slots = nullptr;   // zeroed pool of activation / gradient amax slots
    int nslots = 0, next_slot = 0;

    // K5x (csrc/upconv.hip; engine.py upconv_fwd / _upconv_backward): per up-block the re-laid-out entry slab and its planes
    struct UpW {
        float *w9 = nullptr, *wb = nullptr;
        __bf16 *p9 = nullptr, *pb = nullptr, *p9d = nullptr;
    };
    std::vector<UpW> upw;
    const int UPCONV_MIN_CUP_GRAD = punet_upconv_min_cup_grad();
    bool up_eligible(const Conv& c, int cup) const {  // the library's rule, as engine.py's _upconv_ok: the two walkers agree bit for bit
        return rt->upconv && !f16 && punet_upconv_eligible(cup, c.cin - cup, c.cout, c.k);
    }
    int up_index(int ci) const {  // up-block of conv ci (its entry conv), -1 otherwise
        for (size_t j = 0; j < rt->unet_up.size(); ++j)
            if (rt->unet_up[j][0] == ci) return (int)j;
        return -1;
    }
    int split_up(int mode) {
        const int n = rt->cfg.levels;
        upw.resize(rt->unet_up.size());
        for (size_t j = 0; j < rt->unet_up.size(); ++j) {
            const Conv& c = rt->convs[rt->unet_up[j][0]];
            const int cb = rt->cfg.num_filters[n - 2 - (int)j], cup = c.cin - cb;
            if (!up_eligible(c, cup)) continue;
            UpW& u = upw[j];
            const size_t n9 = (size_t)cup * 9 * c.cout, nb = (size_t)9 * cb * c.cout;
            if (!u.w9) {
                u.w9 = ar->floats(n9);
                u.wb = ar->floats(nb);
            }
            RT_LAUNCH(punet_upconv_relayout(w(c), cup, cb, c.cout, u.w9, u.wb, stream));
            if (mode == 0) {
                u.p9 = (__bf16*)ar->take(3 * n9 * 2);
                u.pb = (__bf16*)ar->take(3 * nb * 2);
                RT_LAUNCH(punet_split_weights_x6(u.w9, u.p9, (long)n9, cup, 9 * c.cout, 1, 0, stream));
                RT_LAUNCH(punet_split_weights_x6(u.wb, u.pb, (long)nb, cb, c.cout, 3, 0, stream));
            } else {
                u.p9d = (__bf16*)ar->take(3 * n9 * 2);
                RT_LAUNCH(punet_split_weights_x6(u.w9, u.p9d, (long)n9, cup, 9 * c.cout, 1, 1, stream));
            }
        }
        return PUNET_OK;
    }
    // ReLU(conv3x3(cat[up(x), bridge]) + b): 1x1 conv of x with w9 at the low resolution, 3x3 conv of the bridge, gather
    int upconv_fwd(int j, Act* x, Act* bridge, std::vector<Op>* tape, Act** out) {
        const int ci = rt->unet_up[j][0];
        const Conv& c = rt->convs[ci];
        const UpW& u = upw[j];
        const int B = x->B, h = x->H, wd = x->W, cup = x->C, cb = bridge->C;
        float* z = ar->floats((size_t)B * h * wd * 9 * c.cout);
        RT_LAUNCH(punet_conv1x1_x6_gated(x->t, cup, nullptr, 0, u.p9, (long)cup * 9 * c.cout, 9 * c.cout, 0, nullptr, 0, z, 9 * c.cout, nullptr,
                                         nullptr, nullptr, 0, 0, B, h, wd, stream));
        Act* y = rt->new_act(ar->floats((size_t)B * 4 * h * wd * c.cout), B, 2 * h, 2 * wd, c.cout, true);
        RT_LAUNCH(c3_gated(bridge->t, cb, nullptr, 0, u.pb, (long)9 * cb * c.cout, c.cout, 0, b(c), 0, y->t, c.cout, nullptr,
                                         nullptr, nullptr, 0, 0, B, 2 * h, 2 * wd, X6_OPTS, stream));
        if (tape) y->bits = (unsigned*)ar->take(punet_gate_dwords(B, 2 * h, 2 * wd, c.cout) * 4);
        RT_LAUNCH(punet_upconv_gather_fwd(z, y->t, y->bits, 1, B, h, wd, c.cout, stream));
        if (tape) tape->push_back(Op{3, ci, x, bridge, y});
        *out = y;
        return PUNET_OK;
    }
    // everything the `up` source receives from dz in the low-resolution form: the nine adjoint-interpolated maps of dz ONCE, the weight
    // gradient of the `up` rows (x^T . maps), d loss / d x (one 1x1 input gradient with w9, x's ReLU gate) - engine.py _upconv_backward
    int upconv_backward(int j, Act* x, const float* dz, int accumulate) {
        const Conv& c = rt->convs[rt->unet_up[j][0]];
        const UpW& u = upw[j];
        const int B = x->B, h = x->H, wd = x->W, cup = x->C;
        float* dzt = ar->floats((size_t)B * h * wd * 9 * c.cout);
        RT_LAUNCH(punet_upconv_gather_bwd(dz, dzt, B, h, wd, c.cout, stream));
        const int ns = punet_upconv_wgrad_slices(cup, c.cout, B, h, wd);
        RT_LAUNCH(punet_upconv_wgrad_x6(x->t, cup, dzt, c.cout, gflat + c.w_off, c.cin, ws_wgrad, ns, B, h, wd, accumulate, 1.f, stream));
        if (!x->needs_grad) return PUNET_OK;
        int acc = 0;
        float* g = grad_into(x, &acc);
        const bool use_bits = x->bits && x->relu_out;
        RT_LAUNCH(punet_conv1x1_x6_gated(dzt, 9 * c.cout, nullptr, 0, u.p9d, (long)cup * 9 * c.cout, cup, 0, nullptr, 0, g, cup,
                                         (x->relu_out && !use_bits) ? x->t : nullptr, use_bits ? x->bits : nullptr, nullptr, acc, 0, B, h, wd,
                                         stream));
        return PUNET_OK;
    }

    // Stream-K scratch of this pass's 3x3 launches (punet_conv3x3_splitk, conv_x6.hip: the tiles of a launch's last partial round
    // are dealt as K-runs over the workgroups): carved from the arena, its arrival counters zeroed once per pass (every launch
    // leaves them zero; the memset makes a pass independent of whatever an aborted one left behind).  The 3x3 entry points of the
    // per-op ABI are called through their superset with this scratch - same arguments otherwise (engine.py does the same).
    void* sk = nullptr;
    size_t sk_bytes = 0;
    int begin_splitk() {
        sk_bytes = punet_conv_splitk_scratch_bytes();
        sk = ar->take(sk_bytes);
        if (!dry && punet_zero_fill(sk, 4096, (hipStream_t)stream) != hipSuccess) {
            punet_set_error("punet_rt: memset of the stream-K counters failed");
            return PUNET_ERR_HIP;
        }
        return PUNET_OK;
    }
    int c3_gated(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row, int n_off,
                 const float* bias, int bias_img_stride, float* y, int n_out, const float* mask, const void* gate_bits, void* gate_out,
                 int accumulate, int relu, int B, int H, int W, int opts, void* st) {
        return punet_conv3x3_splitk(x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                                    gate_out, nullptr, accumulate, relu, B, H, W, opts, 0, nullptr, nullptr, nullptr, nullptr, sk, sk_bytes, st);
    }
    int c3_pooled(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row, int n_off,
                  const float* bias, int bias_img_stride, float* y, int n_out, void* gate_out, float* pool_out, int relu, int B, int H,
                  int W, int opts, void* st) {
        return punet_conv3x3_splitk(x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, nullptr, nullptr,
                                    gate_out, pool_out, 0, relu, B, H, W, opts, 0, nullptr, nullptr, nullptr, nullptr, sk, sk_bytes, st);
    }
    int c3_f16(const float* x0, int c0, const float* x1, int c1, const void* wplanes, long plane_stride, int n_row, int n_off,
               const float* bias, int bias_img_stride, float* y, int n_out, const float* mask, const void* gate_bits, void* gate_out,
               float* pool_out, int accumulate, int relu, int B, int H, int W, int planes, const float* amax_x0, const float* amax_x1,
               const float* amax_w, float* amax_y, void* st) {
        return punet_conv3x3_splitk(x0, c0, x1, c1, wplanes, plane_stride, n_row, n_off, bias, bias_img_stride, y, n_out, mask, gate_bits,
                                    gate_out, pool_out, accumulate, relu, B, H, W, 0, planes, amax_x0, amax_x1, amax_w, amax_y, sk, sk_bytes, st);
    }

    int begin_slots(int n) {  // (fp16-operand arithmetics) a zeroed pool for this pass
        if (!f16) return PUNET_OK;
        wamax = ar->floats(rt->x6_nconv);
        slots = ar->floats(n);
        nslots = n;
        next_slot = 0;
        if (!dry && punet_zero_fill(slots, (size_t)n * 4, (hipStream_t)stream) != hipSuccess) {
            punet_set_error("punet_rt: memset of the amax slots failed");
            return PUNET_ERR_HIP;
        }
        return PUNET_OK;
    }
    bool slots_exhausted = false;
    float* slot() {
        if (!f16) return nullptr;
        if (next_slot >= nslots) {  // (never with <= 8 levels; checked at the end of the walk: a missing slot would silently mean scale 1)
            slots_exhausted = true;
            return nullptr;
        }
        return slots + next_slot++;
    }
    int check_slots() const {
        if (!slots_exhausted) return PUNET_OK;
        punet_set_error("punet_rt: the amax slot pool (%d) is too small for this graph", nslots);
        return PUNET_ERR_ARG;
    }
    const float* wslot(const Conv& c) const { return wamax + c.x6_index; }

    const float* w(const Conv& c) const { return flat + c.w_off; }
    const float* b(const Conv& c) const { return flat + c.b_off; }

    int split(int mode, const __bf16** out) {
        __bf16* p = (__bf16*)ar->take((size_t)3 * rt->numel * 2);
        if (f16)
            RT_LAUNCH(punet_split_weights_f16_batch(flat, p, rt->d_x6, rt->x6_nconv, rt->x6_units, mode, wamax, mode == 0 ? 1 : 0, stream));  // (mode 1: same weights as this pass's forward planes, their amax stands)
        else
            RT_LAUNCH(punet_split_weights_x6_batch(flat, p, rt->d_x6, rt->x6_nconv, rt->x6_units, mode, stream));
        *out = p;
        return PUNET_OK;
    }

    // Conv2d + ReLU; tape != nullptr records the op and publishes the gate bits; pooled != nullptr asks for the fused 2x2 average
    int conv(int ci, Act* x0, Act* x1, bool relu, std::vector<Op>* tape, Act** pooled, Act** out) {
        const Conv& c = rt->convs[ci];
        const int B = x0->B, H = x0->H, W = x0->W;
        const int cin = x0->C + (x1 ? x1->C : 0);
        if (cin != c.cin) {
            punet_set_error("punet_rt: %s expects %d input channels, got %d", c.name.c_str(), c.cin, cin);
            return PUNET_ERR_ARG;
        }
        Act* y = rt->new_act(ar->floats((size_t)B * H * W * c.cout), B, H, W, c.cout, relu);
        if (cin <= SMALL_CIN) {
            if (f16) {
                y->amax = slot();
                RT_LAUNCH(punet_conv3x3_small_fwd_amax(x0->t, cin, w(c), b(c), y->t, c.cout, B, H, W, relu ? 1 : 0, y->amax, stream));
            } else {
                RT_LAUNCH(punet_conv3x3_small_fwd(x0->t, cin, w(c), b(c), y->t, c.cout, B, H, W, relu ? 1 : 0, stream));
            }
        } else if (f16) {  // fp16 planes scaled from the operands' amax slots; publishes y's amax (engine.py, conv())
            if (relu && tape) y->bits = (unsigned*)ar->take(punet_gate_dwords(B, H, W, c.cout) * 4);
            Act* q = nullptr;
            if (pooled && H % 2 == 0 && W % 2 == 0) {
                q = rt->new_act(ar->floats((size_t)B * (H / 2) * (W / 2) * c.cout), B, H / 2, W / 2, c.cout, false);
                *pooled = q;
            }
            y->amax = slot();
            if (q) q->amax = y->amax;  // an average never exceeds the maximum
            RT_LAUNCH(c3_f16(x0->t, x0->C, x1 ? x1->t : nullptr, x1 ? x1->C : 0, planes0 + 3 * c.w_off, c.w_numel(), c.cout, 0,
                                        b(c), 0, y->t, c.cout, nullptr, nullptr, y->bits, q ? q->t : nullptr, 0, relu ? 1 : 0, B, H, W, f16,
                                        x0->amax, x1 ? x1->amax : nullptr, wslot(c), y->amax, stream));
        } else {
            if (relu && tape) y->bits = (unsigned*)ar->take(punet_gate_dwords(B, H, W, c.cout) * 4);
            const void* wp = planes0 + 3 * c.w_off;
            if (pooled && H % 2 == 0 && W % 2 == 0) {
                Act* q = rt->new_act(ar->floats((size_t)B * (H / 2) * (W / 2) * c.cout), B, H / 2, W / 2, c.cout, false);
                *pooled = q;
                RT_LAUNCH(c3_pooled(x0->t, x0->C, x1 ? x1->t : nullptr, x1 ? x1->C : 0, wp, c.w_numel(), c.cout, 0, b(c), 0,
                                                y->t, c.cout, y->bits, q->t, relu ? 1 : 0, B, H, W, X6_OPTS, stream));
            } else {
                RT_LAUNCH(c3_gated(x0->t, x0->C, x1 ? x1->t : nullptr, x1 ? x1->C : 0, wp, c.w_numel(), c.cout, 0, b(c), 0,
                                               y->t, c.cout, nullptr, nullptr, y->bits, 0, relu ? 1 : 0, B, H, W, X6_OPTS, stream));
            }
        }
        if (tape) tape->push_back(Op{0, ci, x0, x1, y});
        *out = y;
        return PUNET_OK;
    }

    int pool(Act* x, std::vector<Op>* tape, Act** out) {
        const int OH = (x->H + 1) / 2, OW = (x->W + 1) / 2;
        Act* y = rt->new_act(ar->floats((size_t)x->B * OH * OW * x->C), x->B, OH, OW, x->C, false);
        RT_LAUNCH(punet_pool2x2_fwd(x->t, y->t, x->B, x->H, x->W, x->C, stream));
        y->amax = x->amax;
        if (tape) tape->push_back(Op{1, -1, x, nullptr, y});
        *out = y;
        return PUNET_OK;
    }

    // the contracting stack shared by Unet (unet.py:50-55) and Encoder (probabilistic_unet.py:67-69)
    int encoder_stack(const std::vector<std::vector<int>>& blocks, Act* x, std::vector<Op>* tape, std::vector<Act*>* outs, Act** last) {
        Act* pooled = nullptr;
        for (size_t i = 0; i < blocks.size(); ++i) {
            if (i) {
                if (pooled) {
                    if (tape) tape->push_back(Op{1, -1, x, nullptr, pooled});
                    x = pooled;
                    pooled = nullptr;
                } else {
                    RT_CALL(pool(x, tape, &x));
                }
            }
            for (size_t k = 0; k < blocks[i].size(); ++k) {
                const bool want_pool = k + 1 == blocks[i].size() && i + 1 < blocks.size() && fuse_pool;
                RT_CALL(conv(blocks[i][k], x, nullptr, true, tape, want_pool ? &pooled : nullptr, &x));
            }
            if (outs) outs->push_back(x);
        }
        *last = x;
        return PUNET_OK;
    }

    int unet(Act* x, std::vector<Op>* tape, Act** feat) {
        const int div = 1 << (rt->cfg.levels - 1);
        if (x->H % div || x->W % div) {
            punet_set_error("punet_rt: H and W must be divisible by %d (got %dx%d)", div, x->H, x->W);
            return PUNET_ERR_ARG;
        }
        std::vector<Act*> outs;
        RT_CALL(encoder_stack(rt->unet_down, x, tape, &outs, &x));
        for (size_t j = 0; j < rt->unet_up.size(); ++j) {
            Act* bridge = outs[outs.size() - 2 - j];
            const bool elig = up_eligible(rt->convs[rt->unet_up[j][0]], x->C) && x->C + bridge->C == rt->convs[rt->unet_up[j][0]].cin;
            if (elig && (!tape || x->C >= UPCONV_MIN_CUP_GRAD)) {
                RT_CALL(upconv_fwd((int)j, x, bridge, tape, &x));
            } else {
                Act* up = rt->new_act(ar->floats((size_t)x->B * 4 * x->H * x->W * x->C), x->B, 2 * x->H, 2 * x->W, x->C, false);
                up->amax = x->amax;  // a bound: bilinear interpolation is a convex combination
                if (elig) up->up_of = x;
                RT_LAUNCH(punet_upsample2x_fwd(x->t, up->t, x->B, x->H, x->W, x->C, stream));
                if (tape) tape->push_back(Op{2, -1, x, nullptr, up});
                RT_CALL(conv(rt->unet_up[j][0], up, bridge, true, tape, nullptr, &x));  // concat order [up, bridge] (unet_blocks.py:56)
            }
            RT_CALL(conv(rt->unet_up[j][1], x, nullptr, true, tape, nullptr, &x));
            RT_CALL(conv(rt->unet_up[j][2], x, nullptr, true, tape, nullptr, &x));
        }
        *feat = x;
        return PUNET_OK;
    }

    // AxisAlignedConvGaussian.forward (probabilistic_unet.py:113-142): encoder, mean over H and W, 1x1 conv -> (mu, log sigma)
    int latent(const std::vector<std::vector<int>>& blocks, int head, Act* x, std::vector<Op>* tape, float** muls, float** mean, Act** enc) {
        RT_CALL(encoder_stack(blocks, x, tape, nullptr, enc));
        const Conv& c = rt->convs[head];
        const int P = (*enc)->H * (*enc)->W;
        *mean = ar->floats((size_t)(*enc)->B * (*enc)->C);
        *muls = ar->floats((size_t)(*enc)->B * c.cout);
        RT_LAUNCH(punet_latent_head_fwd((*enc)->t, w(c), b(c), *mean, *muls, ws_chan, (*enc)->B, P, (*enc)->C, c.cout, stream));
        return PUNET_OK;
    }

    float* grad_into(Act* a, int* acc) {
        if (!a->grad) {
            a->grad = ar->floats(a->numel());
            *acc = 0;
        } else {
            *acc = 1;
        }
        return a->grad;
    }

    int wgrad(const Conv& c, Act* x, int ci_off, const float* dz, const float* dz_amax, bool with_bias, int accumulate) {
        float* dw = gflat + c.w_off;
        float* db = gflat + c.b_off;
        if (dry) return PUNET_OK;
        if (c.cin <= SMALL_CIN)
            return punet_conv3x3_small_wgrad(x->t, c.cin, dz, c.cout, dw, db, ws_wgrad, x->B, x->H, x->W, accumulate, 1.f, stream);
        const int ns = punet_wgrad_x6_slices(x->C, c.cout, x->B, x->H, x->W);
        if (f16)
            return punet_conv_wgrad_f16(x->t, x->C, dz, c.cout, dw, c.cin, ci_off, with_bias ? db : nullptr, ws_wgrad, ns, x->B, x->H, x->W,
                                        accumulate, 1.f, f16, 0, x->amax, dz_amax, stream);
        return punet_conv_wgrad_x6(x->t, x->C, dz, c.cout, dw, c.cin, ci_off, with_bias ? db : nullptr, ws_wgrad, ns, x->B, x->H, x->W,
                                   accumulate, 1.f, X6_OPTS, stream);
    }

    int backward_tape(std::vector<Op>& tape, int accumulate) {
        for (size_t i = tape.size(); i-- > 0;) {
            Op& op = tape[i];
            if (op.kind == 0) {
                const Conv& c = rt->convs[op.conv];
                const float* dz = op.y->grad;
                if (!dz && !dry) {
                    punet_set_error("punet_rt: no gradient reached %s", c.name.c_str());
                    return PUNET_ERR_ARG;
                }
                const bool lowres = op.x1 && op.x0->up_of;  // x0 = up(low-res tensor): K5x's backward of the `up` rows
                if (lowres) {
                    RT_CALL(upconv_backward(up_index(op.conv), op.x0->up_of, dz, accumulate));
                    RT_CALL(wgrad(c, op.x1, op.x0->C, dz, op.y->gamax, true, accumulate));
                    op.x0->up_bwd_done = true;  // the `up` op's own backward is skipped
                } else {
                    RT_CALL(wgrad(c, op.x0, 0, dz, op.y->gamax, true, accumulate));
                    if (op.x1) RT_CALL(wgrad(c, op.x1, op.x0->C, dz, op.y->gamax, false, accumulate));
                }
                int off = 0;
                for (Act* src : {op.x0, op.x1}) {
                    if (!src) continue;
                    if (src == op.x0 && lowres) {
                        // (done above)
                    } else if (src->needs_grad) {
                        int acc = 0;
                        float* g = grad_into(src, &acc);
                        const float* mask = src->relu_out ? src->t : nullptr;
                        const bool use_bits = src->bits && mask;
                        if (f16) {
                            if (!src->gamax) src->gamax = slot();  // (accumulating: the slot keeps max(previous, new total) - a bound)
                            RT_LAUNCH(c3_f16(dz, c.cout, nullptr, 0, planes1 + 3 * c.w_off, c.w_numel(), c.cin, off, nullptr, 0, g,
                                                        src->C, use_bits ? nullptr : mask, use_bits ? src->bits : nullptr, nullptr, nullptr,
                                                        acc, 0, src->B, src->H, src->W, f16, op.y->gamax, nullptr, wslot(c), src->gamax,
                                                        stream));
                        } else {
                            RT_LAUNCH(c3_gated(dz, c.cout, nullptr, 0, planes1 + 3 * c.w_off, c.w_numel(), c.cin, off, nullptr, 0,
                                                           g, src->C, use_bits ? nullptr : mask, use_bits ? src->bits : nullptr, nullptr,
                                                           acc, 0, src->B, src->H, src->W, X6_OPTS, stream));
                        }
                    }
                    off += src->C;
                }
            } else if (op.kind == 3) {  // K5x forward: everything that concerns the `up` source in the low-resolution form (the up-sampled
                const Conv& c = rt->convs[op.conv];  // tensor never exists), the bridge's gradients by the ordinary kernels
                const float* dz = op.y->grad;
                if (!dz && !dry) {
                    punet_set_error("punet_rt: no gradient reached %s", c.name.c_str());
                    return PUNET_ERR_ARG;
                }
                Act *x = op.x0, *bridge = op.x1;
                RT_CALL(upconv_backward(up_index(op.conv), x, dz, accumulate));
                RT_CALL(wgrad(c, bridge, x->C, dz, op.y->gamax, true, accumulate));
                if (bridge->needs_grad) {
                    int acc = 0;
                    float* g = grad_into(bridge, &acc);
                    const float* mask = bridge->relu_out ? bridge->t : nullptr;
                    const bool use_bits = bridge->bits && mask;
                    RT_LAUNCH(c3_gated(dz, c.cout, nullptr, 0, planes1 + 3 * c.w_off, c.w_numel(), c.cin, x->C, nullptr, 0, g,
                                                   bridge->C, use_bits ? nullptr : mask, use_bits ? bridge->bits : nullptr, nullptr, acc, 0,
                                                   bridge->B, bridge->H, bridge->W, X6_OPTS, stream));
                }
            } else if (op.kind == 1) {
                int acc = 0;
                float* g = grad_into(op.x0, &acc);
                const void* bits = (op.x0->relu_out && op.x0->bits) ? op.x0->bits : nullptr;
                RT_LAUNCH(punet_pool2x2_bwd_gated(op.y->grad, (op.x0->relu_out && !bits) ? op.x0->t : nullptr, bits, g, op.x0->B, op.x0->H,
                                                op.x0->W, op.x0->C, acc, stream));
                if (f16) {  // |dx| <= |dy| / 4 (+ what was there): a bound from the slots
                    float* prev = acc ? op.x0->gamax : nullptr;
                    float* sl = prev ? prev : slot();
                    RT_LAUNCH(punet_amax_bound(op.y->gamax, 0.25f, prev, 1.f, sl, stream));
                    op.x0->gamax = sl;
                }
            } else {
                if (op.y->up_bwd_done) continue;  // (the consumer's backward produced op.x0's gradient in the low-resolution form)
                int acc = 0;
                float* g = grad_into(op.x0, &acc);
                const void* bits = (op.x0->relu_out && op.x0->bits) ? op.x0->bits : nullptr;
                RT_LAUNCH(punet_upsample2x_bwd_gated(op.y->grad, (op.x0->relu_out && !bits) ? op.x0->t : nullptr, bits, g, op.x0->B, op.x0->H,
                                                   op.x0->W, op.x0->C, stream));
                if (f16) {  // the transposed x2 interpolation sums weights of at most 2 per dimension
                    op.x0->gamax = slot();
                    RT_LAUNCH(punet_amax_bound(op.y->gamax, 4.f, nullptr, 0.f, op.x0->gamax, stream));
                }
            }
        }
        return PUNET_OK;
    }
};

size_t max_wgrad_ws_floats(const PunetRt* rt, int B, int H, int W) {
    // the largest split-K slab workspace any weight gradient of one pass needs: every 3x3 layer at the level it runs at, a concat
    // convolution once per source (its sources are reduced one at a time)
    size_t m = 0;
    auto one = [&](const Conv& c, int cs, int lv) {
        const int h = H >> lv, w = W >> lv;
        size_t v;
        if (c.cin <= SMALL_CIN) {
            v = punet_small_wgrad_workspace_floats(c.cin, c.cout);
        } else {
            const int ns = punet_wgrad_x6_slices(cs, c.cout, B, h, w);
            v = punet_wgrad_workspace_floats(cs, c.cout, 3, ns);
        }
        m = v > m ? v : m;
    };
    const int n = rt->cfg.levels;
    for (const auto* blocks : {&rt->unet_down, &rt->enc_prior, &rt->enc_post})
        for (int i = 0; i < n; ++i)
            for (int ci : (*blocks)[i]) one(rt->convs[ci], rt->convs[ci].cin, i);
    for (int j = 0; j < (int)rt->unet_up.size(); ++j) {
        const int lv = n - 2 - j;
        const Conv& c0 = rt->convs[rt->unet_up[j][0]];
        const int bridge = rt->cfg.num_filters[lv];
        one(c0, c0.cin - bridge, lv);  // the up-sampled source
        one(c0, bridge, lv);           // the bridge
        {   // K5x: the `up` rows' weight gradient at the low resolution (four slabs per slice)
            const int cup = c0.cin - bridge, h = H >> (lv + 1), w = W >> (lv + 1);
            const size_t v = punet_upconv_wgrad_workspace_floats(cup, c0.cout, punet_upconv_wgrad_slices(cup, c0.cout, B, h, w));
            m = v > m ? v : m;
        }
        one(rt->convs[rt->unet_up[j][1]], rt->convs[rt->unet_up[j][1]].cin, lv);
        one(rt->convs[rt->unet_up[j][2]], rt->convs[rt->unet_up[j][2]].cin, lv);
    }
    return m;
}

}  // namespace

extern "C" int punet_rt_create(const PunetRtConfig* cfg, PunetRt** out) {
    PUNET_CHECK_ARG(cfg && out, "punet_rt_create: null pointer");
    const int n = cfg->levels;
    PUNET_CHECK_ARG(n >= 2 && n <= 8, "punet_rt_create: levels must be 2..8 (got %d)", n);
    PUNET_CHECK_ARG(cfg->num_classes == 1, "punet_rt_create: the native runtime covers num_classes = 1 (got %d); use the per-op ABI", cfg->num_classes);
    PUNET_CHECK_ARG(cfg->input_channels >= 1 && cfg->input_channels + 1 <= SMALL_CIN, "punet_rt_create: input_channels + 1 must be <= %d", SMALL_CIN);
    PUNET_CHECK_ARG(cfg->num_filters[0] == 32 || cfg->num_filters[0] == 64, "punet_rt_create: num_filters[0] must be 32 or 64 (the fused Fcomb kernels)");
    PUNET_CHECK_ARG(cfg->no_convs_fcomb == 2 || cfg->no_convs_fcomb == 3, "punet_rt_create: no_convs_fcomb must be 2 or 3");
    PUNET_CHECK_ARG(cfg->latent_dim >= 1 && cfg->latent_dim <= 64, "punet_rt_create: latent_dim out of range");
    for (int i = 0; i < n; ++i) PUNET_CHECK_ARG(cfg->num_filters[i] >= 8 && cfg->num_filters[i] % 8 == 0, "punet_rt_create: num_filters must be multiples of 8");
    PunetRt* rt = new PunetRt();
    rt->cfg = *cfg;
    const int* nf = cfg->num_filters;
    char nm[128];
    for (int i = 0; i < n; ++i) {
        const int ci = i == 0 ? cfg->input_channels : nf[i - 1], off = i == 0 ? 0 : 1;
        std::vector<int> blk;
        for (int k = 0; k < 3; ++k) {
            snprintf(nm, sizeof nm, "unet.contracting_path.%d.layers.%d", i, off + 2 * k);
            blk.push_back(add_conv(rt, nm, k == 0 ? ci : nf[i], nf[i], 3));
        }
        rt->unet_down.push_back(blk);
    }
    int outc = nf[n - 1];
    for (int j = 0, i = n - 2; i >= 0; --i, ++j) {
        const int ci = outc + nf[i];
        outc = nf[i];
        std::vector<int> blk;
        for (int k = 0; k < 3; ++k) {
            snprintf(nm, sizeof nm, "unet.upsampling_path.%d.conv_block.layers.%d", j, 2 * k);
            blk.push_back(add_conv(rt, nm, k == 0 ? ci : outc, outc, 3));
        }
        rt->unet_up.push_back(blk);
    }
    const auto idx = encoder_indices(n);
    for (int which = 0; which < 2; ++which) {
        const char* net = which == 0 ? "prior" : "posterior";
        const int extra = which == 0 ? 0 : cfg->num_classes;
        std::vector<std::vector<int>> blocks;
        for (int i = 0; i < n; ++i) {
            const int ci = i == 0 ? cfg->input_channels + extra : nf[i - 1];
            std::vector<int> blk;
            for (int k = 0; k < 3; ++k) {
                snprintf(nm, sizeof nm, "%s.encoder.layers.%d", net, idx[i][k]);
                blk.push_back(add_conv(rt, nm, k == 0 ? ci : nf[i], nf[i], 3));
            }
            blocks.push_back(blk);
        }
        snprintf(nm, sizeof nm, "%s.conv_layer", net);
        const int head = add_conv(rt, nm, nf[n - 1], 2 * cfg->latent_dim, 1);
        if (which == 0) { rt->enc_prior = blocks; rt->head_prior = head; } else { rt->enc_post = blocks; rt->head_post = head; }
    }
    rt->fc1 = add_conv(rt, "fcomb.layers.0", nf[0] + cfg->latent_dim, nf[0], 1);
    if (cfg->no_convs_fcomb == 3) rt->fcmid = add_conv(rt, "fcomb.layers.2", nf[0], nf[0], 1);
    rt->fc3 = add_conv(rt, "fcomb.last_layer", nf[0], cfg->num_classes, 1);
    long off = 0;
    for (Conv& c : rt->convs) {
        c.w_off = off;
        Entry ew; ew.name = c.name + ".weight"; ew.off = off; ew.numel = c.w_numel(); ew.ndim = 4;
        ew.shape[0] = c.cout; ew.shape[1] = c.cin; ew.shape[2] = c.k; ew.shape[3] = c.k;
        rt->entries.push_back(ew);
        off += c.w_numel();
        c.b_off = off;
        Entry eb; eb.name = c.name + ".bias"; eb.off = off; eb.numel = c.cout; eb.ndim = 1;
        eb.shape[0] = c.cout; eb.shape[1] = eb.shape[2] = eb.shape[3] = 0;
        rt->entries.push_back(eb);
        off += c.cout;
        off = (off + 3) / 4 * 4;  // every tensor 16-byte aligned
    }
    rt->numel = off;
    rt->unet_numel = off;
    for (const Entry& e : rt->entries)
        if (e.name.compare(0, 5, "unet.") != 0) { rt->unet_numel = e.off; break; }
    long units = 0;
    for (Conv& c : rt->convs)
        if (c.k == 3 && c.cin > SMALL_CIN) {
            c.x6_index = rt->x6_nconv;
            rt->x6_table.insert(rt->x6_table.end(), {c.w_off, (long)c.cin, (long)c.cout, units});
            units += c.w_numel() / 8;
            ++rt->x6_nconv;
        }
    rt->x6_table.insert(rt->x6_table.end(), {0, 0, 0, units});
    rt->x6_units = units;
    int t = 0;
    for (const Entry& e : rt->entries) {  // posterior, prior, fcomb.layers (mean_teacher_trainer.py:115-116), chunks of 16384
        if (e.name.compare(0, 10, "posterior.") == 0 || e.name.compare(0, 6, "prior.") == 0 || e.name.compare(0, 13, "fcomb.layers.") == 0) {
            for (long s = 0; s < e.numel; s += 16384) {
                rt->l2_table.insert(rt->l2_table.end(), {e.off + s, e.numel - s < 16384 ? e.numel - s : 16384, (long)t});
                ++rt->l2_nchunks;
            }
            ++t;
        }
    }
    rt->l2_ntensors = t;
    *out = rt;
    return PUNET_OK;
}

// 3x3 conv arithmetic of every later call on this handle: 6 = the six-product bf16 parity arithmetic (default), 3 = "f16x3" (two
// fp16 planes, per-tensor power-of-two scaling: what the trainers' mixed_precision=True selects), 1 = "f16" (one fp16 plane)
extern "C" int punet_rt_set_arithmetic(PunetRt* rt, int products) {
    PUNET_CHECK_ARG(rt && (products == 6 || products == 3 || products == 1), "punet_rt_set_arithmetic: products must be 6, 3 (f16x3) or 1 (f16)");
    rt->f16_planes = products == 6 ? 0 : (products == 3 ? 2 : 1);
    return PUNET_OK;
}

// criterion of every later punet_rt_elbo_fwd_bwd on this handle: 0 = BCEWithLogits (rl_swap=False, probabilistic_unet.py:348),
// 1 = Dice (rl_swap=True, :347; recalled torch_em definition, parity unpinned)
extern "C" int punet_rt_set_criterion(PunetRt* rt, int dice) {
    PUNET_CHECK_ARG(rt && (dice == 0 || dice == 1), "punet_rt_set_criterion: 0 (BCE) or 1 (Dice)");
    rt->dice = dice;
    return PUNET_OK;
}

// K5x for every later call on this handle: 1 (default) = up-block entry convs with the `up` source contracted at low resolution
// (csrc/upconv.hip; the parity arithmetic only), 0 = the layer as written, on the materialised up-sampled tensor (an A/B switch)
extern "C" int punet_rt_set_upconv(PunetRt* rt, int on) {
    PUNET_CHECK_ARG(rt && (on == 0 || on == 1), "punet_rt_set_upconv: 0 or 1");
    rt->upconv = on;
    return PUNET_OK;
}

extern "C" void punet_rt_destroy(PunetRt* rt) {
    if (!rt) return;
    rt->clear_acts();
    delete rt;
}

extern "C" long punet_rt_param_floats(const PunetRt* rt) { return rt ? rt->numel : 0; }
extern "C" long punet_rt_unet_floats(const PunetRt* rt) { return rt ? rt->unet_numel : 0; }
extern "C" int punet_rt_num_entries(const PunetRt* rt) { return rt ? (int)rt->entries.size() : 0; }

extern "C" int punet_rt_param_entry(const PunetRt* rt, int i, char* name, int cap, long* off, long* numel, int* shape4) {
    PUNET_CHECK_ARG(rt && i >= 0 && i < (int)rt->entries.size() && name && cap > 0 && off && numel && shape4, "punet_rt_param_entry: bad args");
    const Entry& e = rt->entries[i];
    snprintf(name, (size_t)cap, "%s", e.name.c_str());
    *off = e.off;
    *numel = e.numel;
    for (int k = 0; k < 4; ++k) shape4[k] = e.shape[k];
    return PUNET_OK;
}

extern "C" size_t punet_rt_tables_bytes(const PunetRt* rt) {
    return rt ? (rt->x6_table.size() + rt->l2_table.size()) * sizeof(long) + 256 : 0;
}

extern "C" int punet_rt_init_tables(PunetRt* rt, void* device_tables, void* stream) {
    PUNET_CHECK_ARG(rt && device_tables, "punet_rt_init_tables: null pointer");
    long* d = (long*)device_tables;
    rt->d_x6 = d;
    rt->d_l2 = d + ((rt->x6_table.size() + 31) / 32) * 32;
    if (hipMemcpyAsync(rt->d_x6, rt->x6_table.data(), rt->x6_table.size() * sizeof(long), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipMemcpyAsync(rt->d_l2, rt->l2_table.data(), rt->l2_table.size() * sizeof(long), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        punet_set_error("punet_rt_init_tables: copy failed");
        return PUNET_ERR_HIP;
    }
    return PUNET_OK;
}

// forward(patch, None, training=False) (probabilistic_unet.py:292-293) + N x sigmoid(sample()) + thresholds + stack / sum
// (mean_teacher_trainer.py:72-88, fixmatch_trainer.py:37-54, punet_predictions.py:29-33): y = mean sigmoid, z = consensus
// (masking != 0: its == 1 indicator), logits[N][B][H][W] optional.  x: [B][H][W][C] fp32 (NHWC), eps: [N][B][latent].
static int pseudo_labels_body(PunetRt* rt, const float* flat, const float* x, int B, int H, int W, const float* eps, int N, float upper,
                              float lower, int masking, float* y, float* z, float* logits, Arena& ar, bool dry, void* stream) {
    rt->clear_acts();
    Pass p{rt, flat, nullptr, &ar, stream};
    p.dry = dry;
    p.f16 = rt->f16_planes;
    const int n = rt->cfg.levels, CH = rt->cfg.num_filters[0], L = rt->cfg.latent_dim, nmid = rt->cfg.no_convs_fcomb - 2;
    RT_CALL(p.begin_slots(256));
    RT_CALL(p.begin_splitk());
    p.ws_chan = ar.floats(punet_chan_sum_workspace_floats(B, H * W, rt->cfg.num_filters[n - 1]));
    Act* xin = rt->new_act(const_cast<float*>(x), B, H, W, rt->cfg.input_channels, false, false);
    float* zs = ar.floats((size_t)N * B * L);
    float* zb = ar.floats((size_t)N * B * CH);
    RT_CALL(p.split(0, &p.planes0));
    RT_CALL(p.split_up(0));
    float *muls = nullptr, *mean = nullptr;
    Act *enc = nullptr, *feat = nullptr;
    RT_CALL(p.latent(rt->enc_prior, rt->head_prior, xin, nullptr, &muls, &mean, &enc));
    RT_CALL(p.unet(xin, nullptr, &feat));
    RT_CALL(p.check_slots());
    if (dry) return PUNET_OK;
    const Conv &c1 = rt->convs[rt->fc1], &c3 = rt->convs[rt->fc3];
    RT_CALL(punet_reparam(muls, eps, zs, N, B, L, stream));
    RT_CALL(punet_fcomb_zbias(zs, p.w(c1), p.b(c1), zb, N * B, L, CH, stream));
    const float* wmid = nmid ? p.w(rt->convs[rt->fcmid]) : nullptr;
    const float* bmid = nmid ? p.b(rt->convs[rt->fcmid]) : nullptr;
    return punet_fcomb_sample_x6(feat->t, p.w(c1), zb, wmid, 0, bmid, 0, p.w(c3), p.b(c3), y, z, logits, N, B, H * W, CH, nmid, upper, lower,
                                 masking, stream);
}

static char* const DRY_BASE = (char*)(uintptr_t)4096;  // a dry pass never dereferences what it carves

extern "C" int punet_rt_pseudo_labels(PunetRt* rt, const float* flat, const float* x, int B, int H, int W, const float* eps, int N, float upper,
                                      float lower, int masking, float* y, float* z, float* logits, void* arena, size_t arena_bytes,
                                      void* stream) {
    PUNET_CHECK_ARG(rt && flat && x && eps && arena && (y || logits) && (!y == !z), "punet_rt_pseudo_labels: null pointer");
    PUNET_CHECK_ARG(rt->d_x6, "punet_rt_pseudo_labels: punet_rt_init_tables has not run");
    PUNET_CHECK_ARG(B > 0 && N > 0 && H > 0 && W > 0, "punet_rt_pseudo_labels: bad shape");
    PUNET_CHECK_ARG(arena_aligned(arena), "punet_rt_pseudo_labels: the arena must be 256-byte aligned (got %p)", arena);
    Arena dry{DRY_BASE, (size_t)-1};  // first walk: sizes only, nothing is launched
    RT_CALL(pseudo_labels_body(rt, flat, x, B, H, W, eps, N, upper, lower, masking, y, z, logits, dry, true, stream));
    PUNET_CHECK_ARG(dry.used <= arena_bytes, "punet_rt_pseudo_labels: the arena holds %zu bytes, this call needs %zu (punet_rt_arena_bytes)",
                    arena_bytes, dry.used);
    Arena ar{(char*)arena, arena_bytes};
    const int rc = pseudo_labels_body(rt, flat, x, B, H, W, eps, N, upper, lower, masking, y, z, logits, ar, false, stream);
    rt->last_arena_used = ar.used;
    return rc;
}

// forward(patch, segm, training=True); elbo(segm, consm) with the BCE criterion (probabilistic_unet.py:285-293, 341-371) and the
// gradient of gscale * (reconstruction_loss + beta * kl) w.r.t. every parameter into gflat (written, or accumulated when
// accumulate != 0).  target = segm (the posterior's input AND the criterion's target, as in every reference trainer), consm nullable,
// eps: [B][latent].  rec_out / kl_out: device scalars.  after_unet(ctx) (nullable) is called on the host once the U-Net's slice
// [0, punet_rt_unet_floats) of gflat is final and enqueued - a data-parallel caller starts that slice's all-reduce there.
static int elbo_body(PunetRt* rt, const float* flat, float* gflat, const float* x, const float* target, const float* consm, const float* eps,
                     int B, int H, int W, float beta, float gscale, int accumulate, float* rec_out, float* kl_out, float* logits_out,
                     void (*after_unet)(void*), void* ctx, Arena& ar, bool dry, void* stream) {
    const int n = rt->cfg.levels, CH = rt->cfg.num_filters[0], L = rt->cfg.latent_dim, nmid = rt->cfg.no_convs_fcomb - 2;
    const int C = rt->cfg.input_channels;
    rt->clear_acts();
    Pass p{rt, flat, gflat, &ar, stream};
    p.dry = dry;
    p.f16 = rt->f16_planes;
    RT_CALL(p.begin_slots(1024));
    RT_CALL(p.begin_splitk());
    p.ws_chan = ar.floats(punet_chan_sum_workspace_floats(B, H * W, rt->cfg.num_filters[n - 1] > CH ? rt->cfg.num_filters[n - 1] : CH));
    p.ws_wgrad = ar.floats(max_wgrad_ws_floats(rt, B, H, W));
    // ---- forward: posterior on [patch, segm], prior, U-Net (tapes recorded)
    RT_CALL(p.split(0, &p.planes0));
    RT_CALL(p.split_up(0));
    Act* xin = rt->new_act(const_cast<float*>(x), B, H, W, C, false, false);
    Act* xq = rt->new_act(ar.floats((size_t)B * H * W * (C + 1)), B, H, W, C + 1, false, false);
    if (!dry) RT_CALL(punet_concat_nhwc(x, C, target, 1, xq->t, (long)B * H * W, stream));
    std::vector<Op> tq, tp, tu;
    float *muls_q, *mean_q, *muls_p, *mean_p;
    Act *enc_q, *enc_p, *feat;
    RT_CALL(p.latent(rt->enc_post, rt->head_post, xq, &tq, &muls_q, &mean_q, &enc_q));
    RT_CALL(p.latent(rt->enc_prior, rt->head_prior, xin, &tp, &muls_p, &mean_p, &enc_p));
    RT_CALL(p.unet(xin, &tu, &feat));
    // ---- elbo head: z = posterior.rsample(), KL, Fcomb + BCE + Fcomb backward in one kernel
    const Conv &c1 = rt->convs[rt->fc1], &c3 = rt->convs[rt->fc3];
    const Conv* cm = nmid ? &rt->convs[rt->fcmid] : nullptr;
    float* zq = ar.floats((size_t)B * L);
    float* kl_rows = ar.floats(B);
    float* zb = ar.floats((size_t)B * CH);
    float* ws_f = ar.floats(punet_fcomb_train_workspace_floats(B, H * W, CH, nmid));
    float* logits = logits_out ? logits_out : ar.floats((size_t)B * H * W);
    float* dfeat = ar.floats((size_t)B * H * W * CH);
    float* S = ar.floats((size_t)B * CH);
    float* dz = ar.floats((size_t)B * L);
    float* dq = ar.floats((size_t)B * 2 * L);
    float* dp = ar.floats((size_t)B * 2 * L);
    float* dmean = ar.floats((size_t)B * rt->cfg.num_filters[n - 1]);
    if (!dry) {
        RT_CALL(punet_reparam(muls_q, eps, zq, 1, B, L, stream));
        RT_CALL(punet_latent_kl_rows(muls_q, muls_p, nullptr, kl_rows, kl_out, B, L, stream));
        RT_CALL(punet_fcomb_zbias(zq, p.w(c1), p.b(c1), zb, B, L, CH, stream));
        RT_CALL((rt->dice ? punet_fcomb_train_fwd_bwd_dice : punet_fcomb_train_fwd_bwd)(
            feat->t, p.w(c1), zb, cm ? p.w(*cm) : nullptr, cm ? p.b(*cm) : nullptr, p.w(c3), p.b(c3), target, consm, gscale, logits, dfeat,
            rec_out, ws_f, B, H * W, CH, nmid, stream));
    }
    // ---- backward
    RT_CALL(p.split(1, &p.planes1));
    RT_CALL(p.split_up(1));
    if (!dry) {
        RT_CALL(punet_fcomb_train_reduce(ws_f, gflat + c1.w_off, c1.cin, gflat + c1.b_off, cm ? gflat + cm->w_off : nullptr,
                                         cm ? gflat + cm->b_off : nullptr, gflat + c3.w_off, gflat + c3.b_off, S, p.ws_chan, B, H * W, CH,
                                         nmid, accumulate, stream));
        RT_CALL(punet_fcomb_z_bwd(S, zq, p.w(c1), gflat + c1.w_off, dz, B, L, CH, accumulate, stream));
    }
    feat->grad = dfeat;
    if (p.f16) {  // d loss / d features comes from the Fcomb kernel (fp32 arithmetic): it publishes no amax
        feat->gamax = p.slot();
        if (!dry) RT_CALL(punet_amax(dfeat, (long)B * H * W * CH, feat->gamax, 0, stream));
    }
    RT_CALL(p.backward_tape(tu, accumulate));
    if (after_unet && !dry) after_unet(ctx);
    if (!dry)
        RT_CALL(punet_latent_kl_bwd_ex(muls_q, muls_p, eps, dz, gscale * beta / (float)B, 0, 0, dq, dp, B, L, stream));
    for (int which = 0; which < 2; ++which) {  // posterior first, then prior (the order of engine.py)
        const Conv& ch = rt->convs[which == 0 ? rt->head_post : rt->head_prior];
        Act* enc = which == 0 ? enc_q : enc_p;
        int acc = 0;
        float* denc = p.grad_into(enc, &acc);
        if (!dry)
            RT_CALL(punet_latent_head_bwd(which == 0 ? dq : dp, which == 0 ? mean_q : mean_p, p.w(ch), enc->t, gflat + ch.w_off,
                                          gflat + ch.b_off, dmean, denc, B, enc->H * enc->W, enc->C, ch.cout, accumulate, stream));
        if (p.f16) {
            enc->gamax = p.slot();
            if (!dry) RT_CALL(punet_amax(denc, (long)enc->numel(), enc->gamax, 0, stream));
        }
        RT_CALL(p.backward_tape(which == 0 ? tq : tp, accumulate));
    }
    return p.check_slots();
}

extern "C" int punet_rt_elbo_fwd_bwd(PunetRt* rt, const float* flat, float* gflat, const float* x, const float* target, const float* consm,
                                     const float* eps, int B, int H, int W, float beta, float gscale, int accumulate, float* rec_out,
                                     float* kl_out, float* logits_out, void (*after_unet)(void*), void* ctx, void* arena, size_t arena_bytes,
                                     void* stream) {
    PUNET_CHECK_ARG(rt && flat && gflat && x && target && eps && rec_out && kl_out && arena, "punet_rt_elbo_fwd_bwd: null pointer");
    PUNET_CHECK_ARG(rt->d_x6, "punet_rt_elbo_fwd_bwd: punet_rt_init_tables has not run");
    PUNET_CHECK_ARG(arena_aligned(arena), "punet_rt_elbo_fwd_bwd: the arena must be 256-byte aligned (got %p)", arena);
    PUNET_CHECK_ARG(B > 0 && punet_fcomb_train_supported(rt->cfg.num_filters[0], rt->cfg.no_convs_fcomb - 2, H * W),
                    "punet_rt_elbo_fwd_bwd: H * W must be a multiple of 32");
    Arena dry{DRY_BASE, (size_t)-1};  // first walk: sizes only, nothing is launched, the callback is not called
    RT_CALL(elbo_body(rt, flat, gflat, x, target, consm, eps, B, H, W, beta, gscale, accumulate, rec_out, kl_out, logits_out, after_unet, ctx,
                      dry, true, stream));
    PUNET_CHECK_ARG(dry.used <= arena_bytes, "punet_rt_elbo_fwd_bwd: the arena holds %zu bytes, this call needs %zu (punet_rt_arena_bytes)",
                    arena_bytes, dry.used);
    Arena ar{(char*)arena, arena_bytes};
    const int rc = elbo_body(rt, flat, gflat, x, target, consm, eps, B, H, W, beta, gscale, accumulate, rec_out, kl_out, logits_out, after_unet,
                             ctx, ar, false, stream);
    rt->last_arena_used = ar.used;
    return rc;
}

// EXACT bytes the corresponding call carves out of its arena (a dry walk of the same graph): training != 0: punet_rt_elbo_fwd_bwd,
// else punet_rt_pseudo_labels with N samples
extern "C" size_t punet_rt_arena_bytes(PunetRt* rt, int B, int H, int W, int N, int training) {
    if (!rt || B <= 0 || H <= 0 || W <= 0) return 0;
    Arena dry{DRY_BASE, (size_t)-1};
    const float* f = (const float*)DRY_BASE;
    float* g = (float*)DRY_BASE;
    const int rc = training ? elbo_body(rt, f, g, f, f, nullptr, f, B, H, W, 1.f, 1.f, 0, g, g, nullptr, nullptr, nullptr, dry, true, nullptr)
                            : pseudo_labels_body(rt, f, f, B, H, W, f, N > 0 ? N : 1, 0.9f, 0.1f, 0, g, g, nullptr, dry, true, nullptr);
    return rc == PUNET_OK ? dry.used + 256 : 0;
}

// bytes the last punet_rt_pseudo_labels / punet_rt_elbo_fwd_bwd call carved out of its arena (diagnostic: compare with the bound)
extern "C" size_t punet_rt_last_arena_used(const PunetRt* rt) { return rt ? rt->last_arena_used : 0; }

// reg_out[0] = l2_regularisation(posterior) + (prior) + (fcomb.layers) (utils.py:32-40: un-squared 2-norms, biases included) and
// gflat += coef * d reg / d theta (mean_teacher_trainer.py:115-117)
extern "C" int punet_rt_l2_reg_grad(PunetRt* rt, const float* flat, float* gflat, float coef, float* reg_out, void* arena, size_t arena_bytes,
                                    void* stream) {
    PUNET_CHECK_ARG(rt && flat && reg_out && arena && rt->d_l2, "punet_rt_l2_reg_grad: null pointer / tables not initialised");
    PUNET_CHECK_ARG(arena_aligned(arena), "punet_rt_l2_reg_grad: the arena must be 256-byte aligned (got %p)", arena);
    Arena ar{(char*)arena, arena_bytes};
    float* partial = ar.floats(rt->l2_nchunks);
    float* norms = ar.floats(rt->l2_ntensors);
    PUNET_CHECK_ARG(ar.used <= arena_bytes, "punet_rt_l2_reg_grad: arena too small (%zu bytes needed)", ar.used);
    RT_CALL(punet_l2_reg(flat, (const int64_t*)rt->d_l2, rt->l2_nchunks, rt->l2_ntensors, partial, norms, reg_out, stream));
    if (gflat) RT_CALL(punet_l2_reg_add_grad(flat, gflat, (const int64_t*)rt->d_l2, rt->l2_nchunks, norms, coef, stream));
    return PUNET_OK;
}

// ONE call per self-training iteration (mean_teacher_trainer.py:101-131; fixmatch_trainer.py:67-95 with labeller == student and
// ema == 0): pseudo-labels from `labeller` on view x1, ELBO forward / backward of `student` on view x2 against them (consensus z as
// criterion mask when use_consensus != 0), regulariser, Adam on the flat buffers with the EMA of the teacher fused in (ema != 0:
// labeller is updated as the student's EMA).  hstate / hyper: the device state block of punet_adam_hyper (the host keeps lr, step
// count, EMA rule and the gradient scale 1 / world_size there).  scalars[3] = {reconstruction loss, kl, reg} (device);
// loss = scalars[0] + beta * scalars[1] + reg_weight * scalars[2].  hooks (nullable): after_unet = the U-Net slice of gflat is
// final and enqueued; before_update = all of gflat is - a data-parallel host enqueues its all-reduces there.
extern "C" int punet_rt_self_training_step(PunetRt* rt, float* student, float* labeller, float* gflat, float* exp_avg, float* exp_avg_sq,
                                           double* hstate, float* hyper, int ema, const float* x1, const float* x2, int B, int H, int W,
                                           const float* eps_pl, int N, const float* eps_s, float upper, float lower, int masking,
                                           int use_consensus, float beta, float reg_weight, float adam_beta1, float adam_beta2,
                                           float adam_eps, float* y, float* z, float* scalars, const PunetRtHooks* hooks, void* arena,
                                           size_t arena_bytes, void* stream) {
    PUNET_CHECK_ARG(rt && student && labeller && gflat && exp_avg && exp_avg_sq && hstate && hyper && x1 && x2 && eps_pl && eps_s && y && z &&
                        scalars && arena, "punet_rt_self_training_step: null pointer");
    RT_CALL(punet_rt_pseudo_labels(rt, labeller, x1, B, H, W, eps_pl, N, upper, lower, masking, y, z, nullptr, arena, arena_bytes, stream));
    RT_CALL(punet_rt_elbo_fwd_bwd(rt, student, gflat, x2, y, use_consensus ? z : nullptr, eps_s, B, H, W, beta, 1.f, 0, scalars, scalars + 1,
                                  nullptr, hooks ? hooks->after_unet : nullptr, hooks ? hooks->ctx : nullptr, arena, arena_bytes, stream));
    RT_CALL(punet_rt_l2_reg_grad(rt, student, gflat, reg_weight, scalars + 2, arena, arena_bytes, stream));
    if (hooks && hooks->before_update) hooks->before_update(hooks->ctx);
    RT_CALL(punet_adam_hyper(hstate, hyper, stream));
    return punet_adam_ema_dev(student, gflat, exp_avg, exp_avg_sq, (ema && labeller != student) ? labeller : nullptr, rt->numel, hyper,
                              adam_beta1, adam_beta2, adam_eps, stream);
}

// ONE call per supervised iteration (PUNetTrainer / PseudoTrainerPUNet, punet_trainer.py:24-36, punet_target_trainer.py:105-118):
// ELBO forward / backward on (x, y[, consensus weights z]), regulariser, Adam.  scalars[3] = {reconstruction loss, kl, reg}.
extern "C" int punet_rt_supervised_step(PunetRt* rt, float* model, float* gflat, float* exp_avg, float* exp_avg_sq, double* hstate,
                                        float* hyper, const float* x, const float* y, const float* z, int B, int H, int W, const float* eps,
                                        float beta, float reg_weight, float adam_beta1, float adam_beta2, float adam_eps, float* scalars,
                                        const PunetRtHooks* hooks, void* arena, size_t arena_bytes, void* stream) {
    PUNET_CHECK_ARG(rt && model && gflat && exp_avg && exp_avg_sq && hstate && hyper && x && y && eps && scalars && arena,
                    "punet_rt_supervised_step: null pointer");
    RT_CALL(punet_rt_elbo_fwd_bwd(rt, model, gflat, x, y, z, eps, B, H, W, beta, 1.f, 0, scalars, scalars + 1, nullptr,
                                  hooks ? hooks->after_unet : nullptr, hooks ? hooks->ctx : nullptr, arena, arena_bytes, stream));
    RT_CALL(punet_rt_l2_reg_grad(rt, model, gflat, reg_weight, scalars + 2, arena, arena_bytes, stream));
    if (hooks && hooks->before_update) hooks->before_update(hooks->ctx);
    RT_CALL(punet_adam_hyper(hstate, hyper, stream));
    return punet_adam_ema_dev(model, gflat, exp_avg, exp_avg_sq, nullptr, rt->numel, hyper, adam_beta1, adam_beta2, adam_eps, stream);
}

// ONE call per joint iteration (AdaMTTrainer / AdaMatchTrainer, adamt_trainer.py:89-128, adamatch_trainer.py:62-107): supervised
// ELBO on the source batch and self-training ELBO on the target batch at scale 1/2 each (two accumulating gradient passes, one
// optimizer step), pseudo-labels from `labeller` (the EMA teacher; == student and ema == 0 for AdaMatch).  The EMA rule - constant
// or AdaMT's warm-up min(1 - 1/(it + 1), m) - lives in hstate.  scalars[5] = {rec_src, kl_src, rec_tgt, kl_tgt, reg};
// loss = (scalars[0] + beta scalars[1] + scalars[2] + beta scalars[3]) / 2 + reg_weight scalars[4].
extern "C" int punet_rt_joint_step(PunetRt* rt, float* student, float* labeller, float* gflat, float* exp_avg, float* exp_avg_sq,
                                   double* hstate, float* hyper, int ema, const float* xs, const float* ys, int Bs, const float* xt1,
                                   const float* xt2, int B, int H, int W, const float* eps_sup, const float* eps_pl, int N,
                                   const float* eps_s, float upper, float lower, int masking, int use_consensus, float beta,
                                   float reg_weight, float adam_beta1, float adam_beta2, float adam_eps, float* y, float* z, float* scalars,
                                   const PunetRtHooks* hooks, void* arena, size_t arena_bytes, void* stream) {
    PUNET_CHECK_ARG(rt && student && labeller && gflat && exp_avg && exp_avg_sq && hstate && hyper && xs && ys && xt1 && xt2 && eps_sup &&
                        eps_pl && eps_s && y && z && scalars && arena, "punet_rt_joint_step: null pointer");
    RT_CALL(punet_rt_elbo_fwd_bwd(rt, student, gflat, xs, ys, nullptr, eps_sup, Bs, H, W, beta, 0.5f, 0, scalars, scalars + 1, nullptr, nullptr,
                                  nullptr, arena, arena_bytes, stream));
    RT_CALL(punet_rt_pseudo_labels(rt, labeller, xt1, B, H, W, eps_pl, N, upper, lower, masking, y, z, nullptr, arena, arena_bytes, stream));
    RT_CALL(punet_rt_elbo_fwd_bwd(rt, student, gflat, xt2, y, use_consensus ? z : nullptr, eps_s, B, H, W, beta, 0.5f, 1, scalars + 2,
                                  scalars + 3, nullptr, hooks ? hooks->after_unet : nullptr, hooks ? hooks->ctx : nullptr, arena, arena_bytes,
                                  stream));
    RT_CALL(punet_rt_l2_reg_grad(rt, student, gflat, reg_weight, scalars + 4, arena, arena_bytes, stream));  // (reg + reg) / 2
    if (hooks && hooks->before_update) hooks->before_update(hooks->ctx);
    RT_CALL(punet_adam_hyper(hstate, hyper, stream));
    return punet_adam_ema_dev(student, gflat, exp_avg, exp_avg_sq, (ema && labeller != student) ? labeller : nullptr, rt->numel, hyper,
                              adam_beta1, adam_beta2, adam_eps, stream);
}
