/* libi2v_hip.so -- C ABI of the MI355X (gfx950) I2V adversarial-perturbation engine.
 *
 * The reference (zhipeng-wei/Image-to-Video-I2V-attack) has NO native boundary: its hot path is
 * Python calling PyTorch/cuDNN.  This header is the boundary a binding
 * for that path attaches to; every entry point cites the reference lines it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; `i2v_last_error()` gives the text;
 *    nothing throws across the ABI;
 *  - tensors passed by the caller are caller-owned DEVICE pointers, contiguous fp32;
 *  - `stream` is a `hipStream_t` passed as `void*` (0 = default stream); all work is enqueued
 *    asynchronously on it, nothing synchronises the host;
 *  - the library owns only what hangs off the opaque handle: packed weights, the plan and the
 *    activation/gradient arena of each backbone.  One handle per (process, device).
 *  - threads: describing / planning / destroying backbones must be serialised by the caller (the Python mirror
 *    holds a lock); DIFFERENT planned backbones may be executed concurrently from different threads on different
 *    streams (a planned backbone owns its arena; the library has no other mutable state; errors are per thread).
 *    One backbone must not be executed by two threads at once.
 */
#ifndef I2V_HIP_H
#define I2V_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct i2v_ctx* i2v_handle;

/* ---- lifetime ------------------------------------------------------------------------- */
int i2v_create(int device, i2v_handle* out);
int i2v_destroy(i2v_handle h);
const char* i2v_last_error(void);
int i2v_abi_version(void);
/* "hip:gfx950" for the product library.  (The planner unit tests build a host simulation of
 * the same ABI that answers "hostsim"; the Python package refuses to load anything else than
 * the HIP build.) */
const char* i2v_backend(void);
/* Launch counters of the kernel backend since the library was loaded: "conv_launches" (every conv_igemm-family launch),
 * "pws_launches" (of those, the persistent role-split pointwise kernel conv_pw_stream), "bf3_launches" (of those, launches on the
 * split-bf16 K loop of the opt-in I2V_MATH=bf16x3 mode), anything else -1.  Diagnostics for the
 * tests (a forced configuration must really have run); results never depend on it.  No counterpart in the reference. */
long long i2v_backend_stat(const char* name);

/* ---- backbone description --------------------------------------------------------------
 * Replaces `get_model`/`get_models` + `.cuda()` (image_attacks.py:84-115) and the hook lookup
 * `_find_target_layer` (image_attacks.py:260-271, TPAMI_attack.py:176-200): the host walks its
 * graph IR and declares buffers, tensor views and nodes in execution order. */
typedef struct {
    int32_t src, dst;             /* tensor ids */
    int32_t cin, cout, kh, kw, stride, pad;
    int32_t relu;                 /* ReLU after scale/shift (+residual) */
    int32_t residual;             /* tensor id added before the ReLU, or -1 */
} i2v_conv_desc;

typedef struct {
    int32_t src, dst;
    int32_t k, stride, pad;
} i2v_pool_desc;

int i2v_net_create(i2v_handle h, int* net);
/* Frees the backbone's packed weights and arena (net ids are not reused). */
int i2v_net_destroy(i2v_handle h, int net);
int i2v_net_add_buffer(i2v_handle h, int net, int C, int H, int W, int* buf);
int i2v_net_add_tensor(i2v_handle h, int net, int buf, int c_off, int C, int post_relu, int* tensor);
int i2v_net_set_input(i2v_handle h, int net, int tensor);
/* Backward gain of the ReLU that produced `tensor` (a post_relu tensor; before i2v_net_plan): the gradient that passes back through
 * that ReLU is multiplied by `gain` -- the Skip Gradient Method's hook on the model's ReLU modules
 * (base_attacks.py:495-513: `gamma ** 0.5 * grad_in[0]` on every module named *relu* but not *0.relu*).  The forward pass is unchanged. */
int i2v_net_set_relu_gain(i2v_handle h, int net, int tensor, float gain);
/* weight: host [cout][cin][kh][kw]; scale/shift: host [cout] -- the node computes
 * relu(conv(x,W)*scale + shift + residual): eval-mode BatchNorm folded (image_attacks.py:253-256)
 * or the conv bias. */
int i2v_net_add_conv(i2v_handle h, int net, const i2v_conv_desc* d, const float* weight,
                     const float* scale, const float* shift);
int i2v_net_add_maxpool(i2v_handle h, int net, const i2v_pool_desc* d);
/* DenseNet extension (not a backbone the reference can hook; BASELINE.json configs[2] names DenseNet-121):
 * a 1x1/stride-1 convolution that reads relu(x*pre_scale + pre_shift) -- torchvision `_DenseLayer`
 * norm1->relu1->conv1 and `_Transition` norm->relu->conv (host arrays of `cin` floats) -- and average
 * pooling.  The gradient of a tensor read this way ACCUMULATES over its readers (dense connectivity). */
int i2v_net_add_conv_preact(i2v_handle h, int net, const i2v_conv_desc* d, const float* weight,
                            const float* scale, const float* shift, const float* pre_scale,
                            const float* pre_shift);
int i2v_net_add_avgpool(i2v_handle h, int net, const i2v_pool_desc* d);
/* ---- video (3-D) backbones: the white-box model of ILAF (image_attacks.py:513-519,553-611) -------------
 * A (clips, C, T, H, W) activation is held FRAME-MAJOR -- clips*T frames of (C, H, W) -- so the input is the
 * same (b*f, 3, h, w) frame tensor the image path uses (i2v_frames_from_video_f32), spatial (1 x k x k) and
 * pointwise convolutions ARE the image kernels, and a temporal tap is a whole-frame offset.  `T` is the number
 * of frames per clip at that depth.  A network mixes these calls freely with the 2-D ones (2-D == T of the
 * source, kt 1).  Frame counts passed to plan/forward are INPUT frames (a multiple of the input's T).
 *   conv3d: weight host [cout][cin][kt][kh][kw]; output frame t reads input frames t*stride_t - pad_t + q*dil_t
 *           (dil_t > 1 with stride_t == dil_t is a convolution over every dil_t-th input frame, SlowFast's
 *           `x[:, :, ::stride]` pathway inputs). */
typedef struct {
    int32_t src, dst;
    int32_t cin, cout, kt, kh, kw;
    int32_t stride_t, stride, pad_t, pad, dil_t;
    int32_t relu, residual;
} i2v_conv3d_desc;
typedef struct {
    int32_t src, dst;
    int32_t kt, k, stride_t, stride, pad_t, pad;
} i2v_pool3d_desc;
int i2v_net_add_buffer3d(i2v_handle h, int net, int C, int T, int H, int W, int* buf);
int i2v_net_add_conv3d(i2v_handle h, int net, const i2v_conv3d_desc* d, const float* weight,
                       const float* scale, const float* shift);
int i2v_net_add_maxpool3d(i2v_handle h, int net, const i2v_pool3d_desc* d);
/* Core of a non-local (self-attention) block -- gluoncv's `i3d_nl5_resnet50/101_v1`, which ARE the reference's I3D models
 * (`/root/reference/utils.py:9-10`; `image_attacks.py:513-514` hooks `res_layers[1]`, which holds two of the five blocks):
 *   dst[c][i] = sum_j g[c][j] * softmax_j(scale * sum_c' theta[c'][i] * phi[c'][j])     per clip,
 * i over the T*H*W positions of `theta` (= those of `dst`), j over the positions of `phi` and `g` (the block max-pools their input
 * 1x2x2).  theta / phi / g / dst are tensors of the graph with equal channel counts, produced / consumed by ordinary nodes: the
 * 1x1x1 embeddings, the max-pool and `W` + BatchNorm + residual are convolution / pooling nodes.  The attention matrix is kept in
 * the arena for the input-gradient pass, which produces the gradients of theta, phi and g (each must have no other consumer). */
typedef struct {
    int32_t theta, phi, g, dst;
    float scale;
} i2v_attn_desc;
int i2v_net_add_attention(i2v_handle h, int net, const i2v_attn_desc* d);
/* Frames per clip of a tensor (1 for image networks). */
int i2v_net_tensor_frames(i2v_handle h, int net, int tensor, int* T);
/* Freeze the graph: pack weights for forward and input-gradient, plan both passes for up to
 * `max_frames` frames and allocate the arena.  `hook_tensors` are the hooked layer outputs in
 * the order the reference's forward hooks fire (image_attacks.py:281-283). */
int i2v_net_plan(i2v_handle h, int net, const int* hook_tensors, int n_hooks, int max_frames);
size_t i2v_net_workspace_bytes(i2v_handle h, int net);
/* Fused pairs of a planned net (a 3x3 convolution and the pointwise convolution over its output as ONE launch, the intermediate
 * kept in LDS -- a bottleneck's conv2 -> conv3, `torchvision.models.resnet.Bottleneck.forward`, and the mirrored pair of input
 * gradients): out[0..1] = pairs that qualify in the forward / backward launch list, out[2..3] = pairs the plan-time autotuner
 * (or I2V_FORCE_FUSE) fused at the planned batch size.  Diagnostics; results never depend on it. */
int i2v_net_fusion_info(i2v_handle h, int net, int32_t out[4]);

/* ---- backbone execution ----------------------------------------------------------------
 * `_ = self.model(x)` up to the deepest hook (image_attacks.py:318,334).  x: (frames,3,H,W). */
int i2v_net_forward(i2v_handle h, int net, const float* x, int frames, void* stream);
/* Where hook `hook` lives: activation view and gradient view (frame stride in elements, D
 * contiguous elements per frame).  The gradient view is what `i2v_cossim_fwd_bwd_f32` fills. */
int i2v_net_hook_info(i2v_handle h, int net, int hook, float** act, int64_t* act_stride,
                      float** grad, int64_t* grad_stride, int64_t* D, int32_t* post_relu);
/* `cost.backward()` restricted to d(cost)/d(input) (image_attacks.py:352): consumes the hook
 * gradient views, writes (accumulate=0) or adds (accumulate=1) gx: (frames,3,H,W). */
int i2v_net_backward(i2v_handle h, int net, float* gx, int accumulate, void* stream);
/* Test/debug: copy a tensor's activation (which=0) or gradient (which=1) to a contiguous
 * caller buffer (frames,C,H,W). */
int i2v_net_read_tensor(i2v_handle h, int net, int tensor, int which, float* out, int frames,
                        void* stream);

/* ---- measurement (bench.py `roofline`) ---------------------------------------------------
 * When enabled, every backbone launch is bracketed by a HIP event pair on its own stream;
 * `i2v_timing_collect` synchronises that stream and returns, per kernel kind (0 conv_igemm forward,
 * 1 first-layer image gradient, 2 pool fwd, 3 pool bwd, 4 addmask, 5 conv_igemm input-gradient; n_kinds >= 6),
 * the summed device time,
 * the summed ALGORITHMIC flops (2*pixels*Cout*Cin*kh*kw per convolution launch) and the launch
 * count since the last collect.  `enable` = 2: one event pair per SEGMENT -- a run of consecutive launches of one kind in a launch
 * list -- instead of per launch (the same per-kind totals at a fraction of the event records; the low-intensity split of
 * `i2v_timing_collect_ex` and the per-launch dump need mode 1). */
int i2v_timing_enable(i2v_handle h, int enable);
int i2v_timing_collect(i2v_handle h, double* ms_by_kind, double* flops_by_kind,
                       int64_t* launches_by_kind, int n_kinds);
/* The same with the launches' ALGORITHMIC bytes (every operand of a convolution launch once: source view, packed weights,
 * output, epilogue addends, ReLU gate) so that bench.py can state both rooflines: out[kind*n_fields + f], n_fields >= 8,
 * f = 0 ms, 1 flops, 2 launches, 3 bytes, and over the launches whose flops/byte is below the machine balance
 * (157.3 TFLOP/s / 8 TB/s = 19.7 -- the HBM-bound ones): 4 ms, 5 bytes, 6 launches, 7 flops. */
int i2v_timing_collect_ex(i2v_handle h, double* out, int n_kinds, int n_fields);

/* ---- loop kernels ------------------------------------------------------------------------ */
/* Decoded, resized and cropped uint8 frames (b,t,h,w,3) -> normalised clip (b,3,t,h,w): the tail of
 * the reference's loader, `ClipToTensor` (/255) + `Normalize(mean,std)` (datasets.py:88-93), fused
 * with the layout change (decoding/resizing stay on the host). */
int i2v_clip_from_u8_f32(const uint8_t* frames, float* video, int b, int t, int h, int w, void* stream);
/* The WHOLE validation transform of the reference's loader (datasets.py:86-93: `Resize(short_side, 'bilinear')` ->
 * `CenterCrop` -> `ClipToTensor` -> `Normalize`) on decoded uint8 frames (b,t,H,W,3) in one pass: only the pixels that
 * survive the crop are computed.  The resize is OpenCV's 8-bit bilinear (gluoncv runs cv2.resize on decord's numpy
 * frames): xtab (rw x 3) / ytab (rh x 3) int32 DEVICE tables hold, per resized column / row, (source index, weight of
 * it, weight of the next index) in 1/2048 units, built by the host as cv::resize builds them (i2v_amd/clips.py).
 * Output (b,3,t,out_h,out_w), normalised. */
int i2v_clip_resize_crop_u8_f32(const uint8_t* frames, float* video, const int32_t* xtab, const int32_t* ytab, int b, int t,
                                int H, int W, int rh, int rw, int crop_y, int crop_x, int out_h, int out_w, void* stream);
/* The UCF-101 loader's validation transform (`dataset_ucf101.py:113-126`, `transforms_ucf101.py`): Scale(size) = PIL
 * `Image.resize(BILINEAR)` (antialiased two-pass resampler, 8-bit intermediates) -> CornerCrop(size, 'c') -> ToTensor -> Normalize,
 * on decoded uint8 frames (b,t,H,W,3) -> (b,3,t,out_h,out_w).  xbounds/ybounds: int32 (rw|rh, 2) = first source index and tap
 * count per RESIZED column / row; xcoef/ycoef: int32 (rw|rh, kx|ky) 22-bit fixed-point taps -- built on the host exactly as
 * Pillow's `precompute_coeffs` / `normalize_coeffs_8bpc` (i2v_amd/clips.py:pil_resample_table; pinned bit for bit against Pillow). */
int i2v_clip_resample_crop_u8_f32(const uint8_t* frames, float* video, const int32_t* xbounds, const int32_t* xcoef, int kx,
                                  const int32_t* ybounds, const int32_t* ycoef, int ky, int b, int t, int H, int W, int rh, int rw,
                                  int crop_y, int crop_x, int out_h, int out_w, void* stream);
/* videos (b,3,f,h,w) normalised -> frames x:(b*f,3,h,w), frame n = b_idx*f + f_idx
 * (image_attacks.py:300-301) and u = x*std + mean (`_transform_video(...,'back')`, :62,308). */
int i2v_frames_from_video_f32(const float* video, float* x, float* u, int b, int f, int h, int w,
                              void* stream);
/* x = (clamp(u + clamp(delta,-eps,eps),0,1) - mean)/std (image_attacks.py:331-332).
 * video_layout=1 writes (b,3,f,h,w) instead of (b*f,3,h,w) (final output, :360-363). */
int i2v_compose_f32(const float* u, const float* delta, float* x, int b, int f, int h, int w,
                    float eps, int video_layout, void* stream);
/* Per-frame cosine similarity of `a` against the detached clean feature `b` and its gradient
 * (image_attacks.py:341-347; F.cosine_similarity eps 1e-8):
 *   cos_out[n] = <a_n,b_n>/(max(|a_n|,1e-8) max(|b_n|,1e-8))
 *   grad_n (+)= coef * (b_n/(|a_n||b_n|) - cos_n a_n/|a_n|^2) [gated by a>0 if mask_relu]
 * coef = coef_host * (coef_dev ? coef_dev[coef_index] : 1).  `scratch` >= i2v_cossim_scratch_bytes. */
size_t i2v_cossim_scratch_bytes(int64_t D, int frames);
int i2v_cossim_fwd_bwd_f32(const float* a, int64_t a_stride, const float* b, int64_t b_stride,
                           int64_t D, int frames, const float* coef_dev, int coef_index,
                           float coef_host, int mask_relu, int accumulate, float* cos_out,
                           float* grad, int64_t grad_stride, void* scratch, void* stream);
/* Dispersion-Reduction loss `activation.std()` over the WHOLE tensor, unbiased
 * (image_attacks.py:218), and its gradient.  Same scratch size as the cosine kernel. */
int i2v_std_fwd_bwd_f32(const float* a, int64_t a_stride, int64_t D, int frames, int mask_relu,
                        int accumulate, float* std_out, float* grad, int64_t grad_stride,
                        void* scratch, void* stream);
/* The same in two halves for clip-sharded runs: `reduce` leaves (sum, sum of squares) of the LOCAL
 * frames as two doubles at the start of `scratch` (device); the host all-reduces them, and `grad`
 * forms std and gradient from the (global) sums over `total_count` elements. */
int i2v_std_reduce_f32(const float* a, int64_t a_stride, int64_t D, int frames, void* scratch, void* stream);
int i2v_std_grad_f32(const float* a, int64_t a_stride, int64_t D, int frames, int64_t total_count,
                     int mask_relu, int accumulate, float* std_out, float* grad, int64_t grad_stride,
                     void* scratch, void* stream);
/* Compose backward + `torch.optim.Adam.step` on delta (image_attacks.py:306,351-353):
 *   g = gx/std[c] where -eps<=delta<=eps and 0<=u+clamp(delta)<=1 (inclusive), else 0
 *   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; delta -= lr/(1-b1^t) * m/(sqrt(v)/sqrt(1-b2^t)+1e-8)
 * lr/betas/eps are doubles as in torch.optim: 1-b1, 1-b2 and the bias corrections are formed in
 * double on the host and only then rounded to fp32 scalars. */
int i2v_adam_step_f32(float* delta, float* m, float* v, const float* gx, const float* u,
                      int64_t frames, int hw, float eps, double lr, double beta1, double beta2,
                      double adam_eps, int step_t, void* stream);
/* BIM-style update (base_attacks.py:289-293) on a normalised clip of any rank whose channel
 * index is (i / chan_stride) % 3:  a = adv*std+mean + step*sign(g); d = clamp(a-u,+-eps);
 * adv = (clamp(u+d,0,1)-mean)/std. */
int i2v_sign_step_f32(float* adv, const float* u, const float* grad, int64_t n, int64_t chan_stride,
                      float step, float eps, void* stream);
/* ILAF update `modifier -= step*sign(grad)` (image_attacks.py:617). */
int i2v_sign_step_delta_f32(float* delta, const float* grad, int64_t n, float step, void* stream);
/* The same from the gradient w.r.t. the COMPOSED frames (what i2v_net_backward produces): the compose
 * backward only gates (inclusive clamp masks, image_attacks.py:589) and scales by 1/std > 0, so
 * delta -= step * (pass ? sign(gx) : 0). */
int i2v_sign_step_delta_gx_f32(float* delta, const float* gx, const float* u, int64_t n, float eps,
                               float step, void* stream);
/* ILAF loss over one hooked tensor and its gradient (image_attacks.py:595-611):
 *   d = a - ori, d0 = adv0 - ori (fp32), s = |d|, n0 = |d0| over the WHOLE tensor (all frames),
 *   loss = -(0.5*s/n0 + <d0/n0, d/s>),  grad (+)= d loss / d a  [gated by a>0 if mask_relu]
 * `ori`/`adv0` are dense (frames, D) copies of the clean / initial adversarial features.  `reduce` leaves
 * (sum d*d, sum d*d0) as two doubles at the start of `scratch` (so n0 = sqrt(first) when a == adv0);
 * `grad` consumes them.  Scratch size as for the cosine kernel. */
int i2v_ilaf_reduce_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                        int frames, void* scratch, void* stream);
int i2v_ilaf_grad_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                      int frames, double init_norm, int mask_relu, int accumulate, float* loss_out,
                      float* grad, int64_t grad_stride, void* scratch, void* stream);
/* Segmented form: the reference fine-tunes ONE clip per call (image_fine_tune_attack.py:73-79) and every norm of its loss runs
 * over that clip alone (image_attacks.py:563-567,595-613), so K clips batched into one launch list are K independent
 * problems.  Frames [k*frames_per_seg, (k+1)*frames_per_seg) form segment k: `reduce` leaves (sum d*d, sum d*d0) of segment k at
 * doubles [2k, 2k+1] of `scratch`; `grad` takes |d0|^2 per segment from DEVICE memory (`init_sq[k]`: what a reduce with
 * a == adv0 left), writes loss_out[k] and the gradient.  Every segment's result is bit-identical to the one-clip call.
 * Scratch: i2v_ilaf_scratch_bytes(D, frames, frames_per_seg). */
size_t i2v_ilaf_scratch_bytes(int64_t D, int frames, int frames_per_seg);
int i2v_ilaf_reduce_seg_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                            int frames, int frames_per_seg, void* scratch, void* stream);
int i2v_ilaf_grad_seg_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                          int frames, int frames_per_seg, const double* init_sq, int mask_relu, int accumulate,
                          float* loss_out, float* grad, int64_t grad_stride, void* scratch, void* stream);
/* Feature-distance term of base_attacks.TAP (`/root/reference/base_attacks.py:770-776`) over one hooked stage: per segment (clip) of
 * frames_per_seg frames  dist = || r(a) - r(clean) ||_2 with r(x) = sign(x) sqrt|x|, written to dist_out[seg], and
 * grad (+)= coef * d dist / d a  (0 where a == 0: the stage's own ReLU selects 0 there in the reference; 0 while dist == 0, as
 * torch's norm backward).  `clean`: dense [frames][D] copy of the stage's activation on the clean clip; scratch as
 * i2v_ilaf_scratch_bytes(D, frames, frames / frames_per_seg). */
int i2v_tap_distance_f32(const float* a, int64_t a_stride, const float* clean, int64_t D, int frames, int frames_per_seg,
                         double coef, int mask_relu, int accumulate, float* dist_out, float* grad, int64_t grad_stride,
                         void* scratch, void* stream);
/* Classifier head of a white-box video model and the cross-entropy gradient the BIM family starts from
 * (`attack.py:63-96` builds the classifier, `base_attacks.py:282-284`: `cost = targeted * CrossEntropyLoss()(model(adv), labels)`):
 * over the hooked LAST feature map (frame-major, clips*T frames of (C, HW)): global average pool over (T,H,W) ->
 * Linear W[K][C] + bias -> softmax cross-entropy, mean over the clips; writes the logits (clips,K), the per-clip losses,
 * and  grad (+)= scale * d(mean loss)/d(feature)  [gated by a>0 if mask_relu] into the hook's gradient view, from where
 * `i2v_net_backward` takes it to the input.  scratch >= i2v_head_scratch_bytes(C, clips); W / bias / labels on the device. */
size_t i2v_head_scratch_bytes(int C, int clips);
int i2v_head_ce_f32(const float* a, int64_t a_stride, int C, int HW, int T, int clips, const float* W, const float* bias,
                    int K, const int32_t* labels, float scale, int mask_relu, int accumulate, float* logits, float* loss_each,
                    float* grad, int64_t grad_stride, void* scratch, void* stream);
/* The same head over SEVERAL features -- SlowFast pools its slow and fast pathway separately and concatenates them in front of
 * `fc` -- in three steps sharing one scratch block (>= i2v_head_scratch_bytes(Ctot, clips)): every feature is pooled into
 * columns [c_off, c_off + C) of the Ctot-wide vector, one call forms logits / losses / d loss / d pooled (W[K][Ctot]), and every
 * feature's share is spread into its gradient view (divided by its own T*HW, the average pool's backward). */
int i2v_head_pool_f32(const float* a, int64_t a_stride, int C, int HW, int T, int clips, int Ctot, int c_off, void* scratch, void* stream);
int i2v_head_logits_ce_f32(int Ctot, int clips, const float* W, const float* bias, int K, const int32_t* labels, float scale,
                           float* logits, float* loss_each, void* scratch, void* stream);
int i2v_head_grad_f32(const float* a, int64_t a_stride, int C, int HW, int T, int clips, int Ctot, int c_off, int mask_relu,
                      int accumulate, float* grad, int64_t grad_stride, void* scratch, void* stream);
/* Temporal-translation attack (`video_attacks.py:14-229`, the white-box video attack `attack.py --attack_type video` runs): the
 * gradient augmentation `_grad_augmentation` (:160-175) in one pass.  grads: (D, NC, T, HW) device floats = input gradients of the D
 * cyclically frame-shifted copies of a clip (NC = clips * channels); kernel[D] (host: the temporal Gaussian / linear / uniform
 * weights, :49-80), moves[D] (host: `cycle_move_list`, :46-48); out (NC, T, HW) =
 * (1 - weight) * sum_d kernel[d] * grads[d]  +  weight * sum_d kernel[d] * roll(grads[d], -moves[d] along T). */
int i2v_tt_grad_mix_f32(const float* grads, float* out, const float* kernel, const int32_t* moves, int D, int64_t NC, int T, int HW,
                        float weight, void* stream);
/* DI-FGSM's input diversity (`base_attacks.py:357-376`): nearest resize to rnd x rnd, zero pad to 250 x 250 at a random offset, nearest
 * resize to 224 x 224 -- composed by the caller into ONE index map per axis (`map[d]` = source index, < 0 for padding):
 * dst[pl][y][x] = src[pl][map_y[y]][map_x[x]] or 0.  `_bwd` is its transpose (the gradient w.r.t. the un-diversified clip, what
 * autograd returns at :391-392): gsrc[pl][sy][sx] = sum of g over the output rows [ylo[sy], yhi[sy]) x columns [xlo[sx], xhi[sx])
 * that read (sy, sx) -- contiguous ranges because the maps are monotone; summed row-major in fp32.  Maps on the device. */
int i2v_resample_nearest_f32(const float* src, float* dst, int64_t planes, int Hs, int Ws, int Hd, int Wd, const int32_t* map_y,
                             const int32_t* map_x, void* stream);
int i2v_resample_nearest_bwd_f32(const float* g, float* gsrc, int64_t planes, int Hd, int Wd, int Hs, int Ws, const int32_t* ylo,
                                 const int32_t* yhi, const int32_t* xlo, const int32_t* xhi, void* stream);
/* One depthwise 1-D convolution pass (zero padding k/2, odd k <= 64, taps on the host) along the middle axis of a dense
 * [outer][len][inner] tensor, out of place: dst[o][i][j] = sum_t taps[t] * src[o][i + t - k/2][j].  TI-FGSM's 15 x 15 and
 * TI-FGSM-3D's 15 x 15 x 15 Gaussians (`base_attacks.py:412-441`, `:613-651`) are outer products of one 1-D kernel: two / three passes. */
int i2v_dwconv1d_f32(const float* src, float* dst, int64_t outer, int len, int64_t inner, const float* taps, int k, void* stream);
/* Gradient post-processing of the momentum / normalisation family between the input gradient and the sign step, fused (two launches, no
 * atomics; replaces a chain of framework elementwise / reduction kernels):
 *   utils.norm_grads (`utils.py:58-67`, MI-FGSM `base_attacks.py:326`, TI-FGSM-3D `:650`)  mode 1: g / mean|g| over (c,h,w), per (clip, frame)
 *                                                                     (frame_level=False)    mode 2: ... over (c,f,h,w), per clip
 *   TI-FGSM (`:440`, sic: mean over channels, frames and ROWS)                              mode 3: ... over (c,f,h), per (clip, column)
 *   the L1 form of DI / SI / SGM (`:394`, `:545`, `:599`: grad / torch.norm(grad, p=1))     mode 4: g / sum|g| over everything
 *   mode 0: no normalisation
 * then, when `momentum` is given (`:327-329`, `:395-397`): out = g_normalised + decay * momentum; momentum = out.
 * `g` is the gradient either in the clip layout (b,c,f,h,w) or -- `frame_major` != 0 -- as the backbone's input gradient (b*f,c,h,w);
 * `out` / `momentum` are always (b,c,f,h,w), so the layout change rides along.  True division by the fp32 mean / norm (sums in
 * double, fixed order).  `scratch`: i2v_grad_post_scratch_bytes(...) bytes. */
int64_t i2v_grad_post_scratch_bytes(int b, int c, int f, int h, int w, int mode);
int i2v_grad_post_f32(const float* g, float* momentum, float* out, int b, int c, int f, int h, int w, int frame_major, int mode, float decay,
                      void* scratch, void* stream);
/* TAP's elementwise steps around the box filter (`base_attacks.py:724-731, 777-790`), clip layout (b,c,f,h,w), c == 3:
 *   _perts:     out = (adv - videos) / std[c]                      (`_transform_perts`, sic: divided)
 *   _sign_abs:  sign_out = sign(smooth);  *reg = sum |smooth|      (the regulariser's value, double sums in a fixed order; `scratch`
 *               holds i2v_tap_scratch_bytes(n) bytes)
 *   _grad:      out = gx + weight * boxsign / std[c], gx being the backbone's FRAME-major input gradient (b*f,c,h,w) */
int64_t i2v_tap_scratch_bytes(int64_t n);
int i2v_tap_perts_f32(const float* adv, const float* videos, float* out, int b, int c, int f, int h, int w, void* stream);
int i2v_tap_sign_abs_f32(const float* smooth, float* sign_out, float* reg, int64_t n, void* scratch, void* stream);
int i2v_tap_grad_f32(const float* gx, const float* boxsign, float* out, int b, int c, int f, int h, int w, float weight, void* stream);
/* Adaptive ENS-I2V re-weighting `coeffs = softmax(softmax(prev) + momentum*coeffs)`
 * (TPAMI_attack.py:265), L <= 64, in place on device. */
int i2v_aens_coeffs_f32(const float* prev, float* coeffs, float momentum, int L, void* stream);
/* sum_l coeffs[l]*sum_n cos[l][n] bookkeeping for AENS (TPAMI_attack.py:289-297):
 * feat_sum[l] = sum_n cos[l*frames+n]; weighted[l] = coeffs[l]*feat_sum[l]. */
int i2v_aens_reduce_f32(const float* cos, const float* coeffs, int L, int frames, float* feat_sum,
                        float* weighted, void* stream);

#ifdef __cplusplus
}
#endif
#endif
