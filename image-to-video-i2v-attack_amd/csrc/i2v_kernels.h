// Internal interface between the planner/executor (i2v_engine.cpp) and a kernel backend.
// The product backend is i2v_kernels.hip (gfx950).  tests/hostsim/ holds a scalar host backend of
// the same interface that exists ONLY so the planner can be unit-tested without a GPU.
#pragma once
#include <initializer_list>
#include "i2v_params.h"

typedef void* i2v_stream_t;

const char* be_name();
long long be_stat(const char* name);          // launch counters ("conv_launches", "pws_launches"), -1 for an unknown name
int  be_set_device(int device);
void* be_malloc(size_t bytes);
void be_free(void* p);
int  be_h2d(void* dst, const void* src, size_t bytes);                       // synchronous upload
int  be_d2d_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
               i2v_stream_t s);                                               // async strided copy
int  be_memset0(void* p, size_t bytes, i2v_stream_t s);
const char* be_error();
// timing instrumentation (bench.py roofline): stream-ordered event pairs around a launch
void* be_event_create();
void be_event_destroy(void* ev);
int  be_event_record(void* ev, i2v_stream_t s);
int  be_event_elapsed_ms(void* start, void* stop, float* ms);           // both must have completed
int  be_stream_sync(i2v_stream_t s);
i2v_stream_t be_stream_create();                                          // a non-blocking stream of its own (side stream of run_list's launch overlap); null on failure
void be_stream_destroy(i2v_stream_t s);
int  be_stream_wait_event(i2v_stream_t s, void* ev);                       // work queued on `s` after this call starts after `ev` has completed
int  be_device_sync();                                                       // last backend error or null

int k_conv(const I2VConvParams& p, i2v_stream_t s);
// fused pair (3x3 convolution `a` -> pointwise convolution `b` over its channels, one launch, a's output never stored):
// eligibility of the parameter pair (0 no, 1 plain staging, 3 also halo staging) and the launch (both vec_epilogue, same N)
int k_conv_fusable(const I2VConvParams& a, const I2VConvParams& b);
// The STRUCTURAL half of that answer, shared by every backend (the device adds only the halo-staging bit): `a` a 3x3 / stride-1 /
// same-size tap-uniform image convolution with 64 or 128 output channels and a dense epilogue, `b` the pointwise convolution that
// reads exactly a's output, both with identity frame maps (the executor derives `temporal` from exactly this at run time).
inline bool i2v_conv_pair_fusable(const I2VConvParams& a, const I2VConvParams& b) {
    if (a.pointwise || !a.tap_uniform || a.temporal || a.quad || a.pre_scale || a.gate_scale || a.blk > 1 || a.blkt > 1) return false;
    for (const I2VConvParams* q : {&a, &b})
        if (!(q->Tg == q->Ts && q->Ts == q->To && q->st == 1 && q->ost == 1 && q->ot0 == 0 && q->blkt <= 1)) return false;
    if (a.sh != 1 || a.sw != 1 || a.Hs != a.Hg || a.Ws != a.Wg || a.Hg != a.Ho || a.Wg != a.Wo || a.osh != 1 || a.osw != 1 || a.oh0 || a.ow0) return false;
    if ((a.Cd != 64 && a.Cd != 128) || a.Kpad != a.K || a.add0_stride > 1) return false;
    if (!b.pointwise || b.temporal || b.quad || b.pre_scale || b.gate_scale || b.blk > 1 || b.blkt > 1 || b.add0_stride > 1) return false;
    if (b.K != a.Cd || b.Kpad != b.K || b.Cs != a.Cd || b.src != a.dst || b.src_nstride != a.dst_nstride) return false;
    if (b.Hs != a.Ho || b.Ws != a.Wo || b.Hg != a.Hg || b.Wg != a.Wg || b.Ho != a.Ho || b.Wo != a.Wo || b.sh != 1 || b.sw != 1 || b.osh != 1 || b.osw != 1 ||
        b.oh0 || b.ow0 || b.Cd < 64 || b.Tg != a.Tg) return false;
    return true;
}
int k_conv_fused(const I2VConvParams& a, const I2VConvParams& b, int halo, i2v_stream_t s);
int k_conv_candidates(const I2VConvParams& p, int* out);
// Fused fast-pathway block (I2VFastBlockParams; i2v_fastblock.hip): `a`, `b` and -- forward -- `c` (+ `d`, the projection shortcut, or
// null) are the PREPPED parameters of the separate launches it replaces; backward: c == d == null.  The structural rule below is shared
// by every backend (the k-tables are not inspected: that `a` has temporal / pointwise taps only and `b` is a 3 x 3 / pad-1 convolution
// is the planner's statement, made from the graph nodes).  Returns the rows per block R (> 0) or 0.
inline int i2v_fastblock_rows(const I2VConvParams& a, const I2VConvParams& b, const I2VConvParams* c, const I2VConvParams* d) {
    auto plain = [](const I2VConvParams& q) {
        return !q.pre_scale && !q.gate_scale && q.blk <= 1 && q.blkt <= 1 && q.sh == 1 && q.sw == 1 && q.osh == 1 && q.osw == 1 && q.oh0 == 0 &&
               q.ow0 == 0 && q.Hs == q.Hg && q.Hg == q.Ho && q.Ws == q.Wg && q.Wg == q.Wo && q.Tg == q.Ts && q.Ts == q.To && q.st == 1 && q.ost == 1 && q.ot0 == 0 &&
               !q.add1 && !q.mask && q.Kpad % 8 == 0 && q.Kpad <= 512 && q.gate_pix0 == 0 && q.gate_out_pix0 == 0;
    };
    if (!plain(a) || !plain(b) || a.quad || a.add0 || b.add0) return 0;
    const int CM = a.Cd;
    if ((CM != 4 && CM != 8) || b.Cd != CM || b.Cs != CM || b.src != a.dst || b.src_nstride != a.dst_nstride) return 0;
    if (b.Hg != a.Hg || b.Wg != a.Wg || b.Tg != a.Tg || b.N != a.N) return 0;
    const int H = a.Hg, W = a.Wg;
    if ((H * W) % 32 != 0 || (int64_t)a.N * a.src_nstride * 4 >= (1ll << 31)) return 0;
    if (c) {        // forward
        if (!plain(*c) || c->quad || !a.relu || !b.relu || a.gate || b.gate || c->gate || !c->relu || !a.shift || !b.shift || !c->shift || !a.gate_out || !b.gate_out || !c->gate_out) return 0;
        if (c->src != b.dst || c->src_nstride != b.dst_nstride || c->Cs != CM || c->K != CM || c->Cd != 4 * CM || c->Hg != H || c->Wg != W || c->Tg != a.Tg || c->N != a.N) return 0;
        if (!c->add0 || c->add0_stride != 1) return 0;
        if (d) {
            if (!plain(*d) || d->quad || d->relu || d->add0 || d->gate || d->gate_out || !d->shift || d->src != a.src || d->src_nstride != a.src_nstride || d->Cs != a.Cs ||
                d->K != a.Cs || d->K != CM || d->Cd != c->Cd || d->Hg != H || d->Wg != W || d->Tg != a.Tg || d->N != a.N || c->add0 != d->dst || c->add0_nstride != d->dst_nstride) return 0;
        }
    } else {        // backward: plain input gradients (their gates, no ReLU)
        if (d || a.relu || b.relu || a.gate_out || b.gate_out || a.shift || b.shift || !a.gate || !b.gate || a.K != a.Cs || a.K > 64) return 0;
    }
    int g = W, r32 = 32; while (r32) { const int t = g % r32; g = r32; r32 = t; }      // gcd(W, 32)
    const int Rq = 32 / g;
    if (H % Rq != 0) return 0;
    int R = Rq;
    while (R * W < 192 && R * 2 <= H && H % (R * 2) == 0) R *= 2;
    if ((size_t)CM * (R + 2) * (W + 2) * 4 > 60 * 1024) return 0;
    return R;
}
int k_fastblock(const I2VConvParams& a, const I2VConvParams& b, const I2VConvParams* c, const I2VConvParams* d, i2v_stream_t s);
   // tile configurations valid for p (ids 0..5, +8 = no epilogue prefetch), returns count
int k_pool_fwd(const I2VPoolParams& p, i2v_stream_t s);
int k_pool_bwd(const I2VPoolParams& p, i2v_stream_t s);
int k_pool3d_fwd(const I2VPoolParams& p, i2v_stream_t s);  // video max pooling (kt/stride_t/pad_t honoured)
int k_pool3d_bwd(const I2VPoolParams& p, i2v_stream_t s);
int k_avgpool_fwd(const I2VPoolParams& p, i2v_stream_t s);
int k_avgpool_bwd(const I2VPoolParams& p, i2v_stream_t s);
int k_addmask(const I2VAddMaskParams& p, i2v_stream_t s);
int k_cos(const I2VCosParams& p, i2v_stream_t s);
int k_std_reduce(const I2VStdParams& p, i2v_stream_t s);   // -> p.sums[0..1]
int k_std_grad(const I2VStdParams& p, i2v_stream_t s);     // p.sums, p.total_count -> std_out, grad
int k_ilaf_reduce(const I2VIlafParams& p, i2v_stream_t s); // -> p.sums[0..1]
int k_ilaf_grad(const I2VIlafParams& p, i2v_stream_t s);   // p.sums, p.init_norm -> loss_out, grad
int k_head_ce(const I2VHeadParams& p, i2v_stream_t s);
int k_clip_from_u8(const uint8_t* frames, float* video, int b, int t, int h, int w, i2v_stream_t s);
int k_clip_resize_crop(const uint8_t* frames, float* video, const int32_t* xtab, const int32_t* ytab, int b, int t, int H, int W,
                       int cy, int cx, int oh, int ow, i2v_stream_t s);
int k_frames_from_video(const float* video, float* x, float* u, int b, int f, int h, int w, i2v_stream_t s);
int k_compose(const float* u, const float* delta, float* x, int b, int f, int h, int w, float eps,
              int video_layout, i2v_stream_t s);
int k_adam(float* delta, float* m, float* v, const float* gx, const float* u, int64_t n, int hw,
           float eps, float step_size, float bc2_sqrt, float w1, float beta2, float w2, float adam_eps,
           i2v_stream_t s);   // w1 = 1-beta1, w2 = 1-beta2 (rounded from double)
int k_sign_bim(float* adv, const float* u, const float* grad, int64_t n, int64_t chan_stride,
               float step, float eps, i2v_stream_t s);
int k_sign_delta(float* delta, const float* grad, int64_t n, float step, i2v_stream_t s);
int k_sign_delta_gx(float* delta, const float* gx, const float* u, int64_t n, float eps, float step,
                    i2v_stream_t s);   // delta -= step * sign(d cost / d delta) from the gradient w.r.t. the composed frames
int k_clip_resample_crop(const uint8_t* frames, float* video, const int32_t* xb, const int32_t* xk, int kx, const int32_t* yb, const int32_t* yk,
                         int ky, int b, int t, int H, int W, int cy, int cx, int oh, int ow, i2v_stream_t s);
int k_tt_grad_mix(const float* grads, float* out, const float* kern /*host [D]*/, const int* moves /*host [D]*/, int D, int64_t NC, int T,
                  int HW, float w1, float w, i2v_stream_t s);
int k_attn_gemm(const I2VAttnGemm& p, i2v_stream_t s);          // the three product forms of the non-local block
int k_softmax_rows(const I2VSoftmaxRows& p, i2v_stream_t s);
// base_attacks.py input / gradient transforms (DI-FGSM, TI-FGSM, TI-FGSM-3D): nearest resampling through index maps and its
// transpose, depthwise 1-D convolution along one axis of a dense tensor (maps on the device, taps on the host, k <= 64)
int k_resample_nearest(const float* src, float* dst, int64_t planes, int Hs, int Ws, int Hd, int Wd, const int32_t* map_y,
                       const int32_t* map_x, i2v_stream_t s);
int k_resample_nearest_bwd(const float* g, float* gsrc, int64_t planes, int Hd, int Wd, int Hs, int Ws, const int32_t* ylo,
                           const int32_t* yhi, const int32_t* xlo, const int32_t* xhi, i2v_stream_t s);
// gradient post-processing (i2v_grad_post_f32): groups / splits of the mean-abs reduction, then the fused apply pass
int k_grad_post_groups(int b, int c, int f, int h, int w, int mode, int64_t* group_elems);     // number of groups, elements per group
int k_grad_post_splits(int64_t group_elems);
int k_grad_post(const float* g, float* momentum, float* out, int b, int c, int f, int h, int w, int frame_major, int mode, float decay,
                double* partial, i2v_stream_t s);
int k_tap_perts(const float* adv, const float* videos, float* out, int b, int c, int f, int h, int w, i2v_stream_t s);
int k_tap_sign_abs(const float* smooth, float* sign_out, float* reg, int64_t n, double* partial, i2v_stream_t s);     // partial: [1024] doubles
int k_tap_grad(const float* gx, const float* boxsign, float* out, int b, int c, int f, int h, int w, float weight, i2v_stream_t s);
int k_dwconv1d(const float* src, float* dst, int64_t outer, int len, int64_t inner, const float* taps /*host [k]*/, int k, i2v_stream_t s);
int k_aens_coeffs(const float* prev, float* coeffs, float momentum, int L, i2v_stream_t s);
int k_aens_reduce(const float* cos, const float* coeffs, int L, int frames, float* feat_sum,
                  float* weighted, i2v_stream_t s);
int cos_nblk(int64_t D);   // blocks per frame used by k_cos / k_std (scratch sizing)
