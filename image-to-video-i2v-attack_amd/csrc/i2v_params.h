// Launch-parameter structs shared by the planner (i2v_engine.cpp) and the gfx950 kernels
// (i2v_kernels.hip).  Plain data, no HIP types, so the planner can be unit-tested on the host.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define I2V_KC 16          // K-chunk of the implicit GEMM; packed weights/k-tables are padded to it

// One row of the implicit-GEMM reduction axis: which source plane and which spatial tap.
struct I2VKEntry {
    int32_t chan_off;      // element offset of the source channel plane (c * Hs * Ws)
    int32_t dh, dw;        // tap offset in source coordinates
    int32_t valid;         // bit 0: 0 for padding rows; bits 1..31: signed temporal tap offset dt (source frames,
                           // video networks only; the kernel adds dt * src_nstride to chan_off)
};

// Implicit-GEMM convolution, forward and input-gradient alike:
//   acc[cd][p] = sum_k wp[k][cd] * src[n(p)][ktab[k].chan][i(p)*sh + dh_k][j(p)*sw + dw_k]
//   v = acc + shift[cd] + add0[..] + add1[..];  relu -> max(v,0);  mask -> (mask[..] > 0 ? v : 0)
//   dst[n][cd][i*osh + oh0][j*osw + ow0] = v
// p enumerates (n, i, j) over an N x Hg x Wg grid.
//
// Video (3-D) networks keep every activation FRAME-MAJOR: a (clips, C, T, H, W) tensor is stored as
// clips*T frames of (C, H, W), so spatial and pointwise convolutions are the image kernels unchanged and a
// temporal tap is a whole-frame offset.  Grid frame n = clip*Tg + tg reads source frame clip*Ts + tg*st + dt_k
// (taps outside [0, Ts) contribute zeros) and writes destination frame clip*To + tg*ost + ot0.
// Image networks have Tg = Ts = To = st = ost = 1, ot0 = 0.
struct I2VConvParams {
    const float* src;  int64_t src_nstride;  int32_t Hs, Ws;
    int32_t Cs;             // channels of the source view
    int32_t src_span_bytes; // bytes spanned by the N frames of the source view (< 2 GiB per launch; set by the executor)
    const float* wp;   const I2VKEntry* ktab; int32_t K, Kpad, Cd, Cdpad;   // K real rows, Kpad padded
    int32_t N, Hg, Wg, sh, sw;
    float* dst;        int64_t dst_nstride;  int32_t Ho, Wo, osh, osw, oh0, ow0;
    const float* shift;
    const float* add0; int64_t add0_nstride; int32_t add0_stride, add0_H, add0_W;  // stride s>1: compact addend
    const float* add1; int64_t add1_nstride;
    const float* mask; int64_t mask_nstride;
    // 1-bit ReLU gates.  A post-ReLU tensor keeps one bit per element, (value > 0), in rows of 32-bit words per
    // CHANNEL: bit (pix0 + n*H*W + pixel) of row c, n counting the launch's frames of that tensor -- 1/32 of the
    // bytes of re-reading the fp32 activation as its own gate.  `gate_out` (forward launches with relu, dense
    // output) is written by the epilogue; `gate` (input-gradient launches) replaces `mask`: v = gate bit ? v : 0.
    // Row strides in words; pix0 is a multiple of 32 (a launch sliced over clips starts mid-row).
    uint32_t* gate_out; int32_t gate_out_stride; int32_t gate_out_pix0;
    const uint32_t* gate; int32_t gate_stride; int32_t gate_pix0;
    // DenseNet pre-activation (norm -> relu -> 1x1 conv), pointwise launches only:
    //   forward:  the B operand is relu(src * pre_scale[k] + pre_shift[k])        (arrays of Kpad floats, 0-padded)
    //   backward: the gate is (mask * gate_scale[cd] + gate_shift[cd] > 0) instead of (mask > 0) and it gates the
    //             accumulator BEFORE the addends (the gradient of a dense buffer accumulates: add1 == dst)
    const float* pre_scale; const float* pre_shift;
    const float* gate_scale; const float* gate_shift;
    int32_t relu;
    int32_t pointwise;  // 1: 1x1 / stride 1 / no padding, planes 16-byte aligned -> vector path
    int32_t tap_uniform; // 1: K rows are (tap-major, channel-minor) with channels % I2V_KC == 0, so every
                         //    K-chunk has ONE spatial tap: ktab[k0] describes the whole chunk
    // blk > 1: the Cd axis packs blk*blk output-position classes: cd = (ph*blk + pw)*(Cd/blk^2) + c is
    // stored to channel c at (i*osh + ph, j*osw + pw).  Used for the gradient w.r.t. the 3-channel image,
    // where 3 output channels alone would waste 29/32 of every MFMA.
    int32_t blk;
    int32_t blkt;           // temporal classes in front of the blk*blk spatial ones (video stem); 1 otherwise
    int32_t Tg, Ts, To, st, ost, ot0;   // frames per clip of grid / source / destination, temporal strides (see above)
    int32_t oct;            // frames between two temporal classes of a class-packed launch (1; the stride of a frame-skipping stem's paired gradient)
    int32_t temporal;       // 1: some k-table row has dt != 0 or the frame mapping is not the identity -> the kernel's video
                            //    variant runs; 0 (all image launches, spatial / pointwise launches of video networks): the
                            //    temporal fields are ignored altogether
    // 1: output is dense over the pixel grid (osh=osw=1, Hg x Wg == Ho x Wo, plane % 4 == 0, every
    // plane 16-byte aligned, plain addends): the epilogue may use 16-byte accesses along W
    int32_t vec_epilogue;
    // quad > 0: "quad rows" packing of a few-channel stem (kernel MODE 4): K rows come in groups of four that share channel,
    // frame and row tap and differ by dw = dw0 .. dw0+3 -- four ADJACENT source pixels -- so one 16-byte DMA per lane stages
    // four K rows of a pixel ([quad][pixel][4] LDS image) instead of four 4-byte pieces; quad = quads per row run
    // (ceil(kw / 4), 1 or 2), quad_kw = kw (rows with dw0 + e beyond the kernel have zero weights and are masked), the tap of
    // element e of run-quad qi is dw = quad_dw0 + 4 qi + e.  The source view needs 64 readable bytes on either side.
    int32_t quad, quad_kw, quad_dw0;
    // halo = 9 (kernel MODE 5): a tap-uniform 3x3 / stride-1 / pad-1 launch whose K rows are ordered (16-channel group, tap, channel).
    // The nine chunks of a group read the SAME 16 channel planes shifted by their tap, so the kernel may stage one halo row per
    // channel and group (tile pixels + a row and a pixel on either side) instead of nine shifted copies of the tile.  Same
    // products in the same order: results are bit-identical.  halo = 1: every tap at (0, 0) (a k x 1 x 1 kernel: conv_vfma_kernel);
    // halo = 49: the 7x7 / 2 / pad-3 stem over 3 channels in (tap, channel)
    // order (conv_stem64_halo); halo = 77: a quad-row stem with SEVEN row taps from -3 (7x7, pad 3: the window conv_stem_halo stages).
    int32_t halo;
    // ig_th > 0: a class-packed image-gradient packing (pack_img) in tap-uniform order (16-channel group, frame tap, row tap, column tap,
    // channel) with ig_tt x ig_th x ig_tw union taps, the first k-table row being the lowest tap of each axis: conv_imggrad_halo may
    // stage one 2-D halo tile per (group, frame tap) instead of one shifted copy per tap.  Same products, same order.
    int32_t ig_tt, ig_th, ig_tw;
    int32_t ig_p77;        // quad-row image gradient of a stride-2 7 x 7 kernel whose packed weights are zero exactly where that geometry says (pack_img checks): conv_igvfma_kernel skips those pairs
    const float* wpc;      // quad-row image gradient with 12 class rows: the packed weights once more with rows of 16 floats (64 bytes) -- conv_igvfma_kernel's scalar loads;
                           // at wp's 512-byte row stride its 640 rows fall into one scalar-cache set in eight (null: no such copy)
    // exact division of a pixel index (< 2^31) by Hg*Wg, Wg, Tg and Wo as multiply-high + shift: filled in by k_conv (the
    // hardware has no integer divide; the 64-bit software divisions of round 1 cost a block more VALU issue slots than a
    // K = 64 tile spends on its MFMAs)
    uint32_t dv_hw_m, dv_hw_s, dv_w_m, dv_w_s, dv_t_m, dv_t_s, dv_wo_m, dv_wo_s;
    // Split-bf16 arithmetic (round 5; opt-in, I2V_MATH=bf16x3): `wp3` = the same weights pre-split into three bf16 terms w = w1 + w2 + w3, in
    // the 32x32x16 bf16 MFMA's fragment order: [Kpad / 16][Cdpad / 32][3 terms][64 lanes][8 bf16] -- lane l of a (chunk, 32-row tile) holds
    // rows cd = 32 tile + (l & 31), K rows 16 chunk + 8 (l >> 5) .. + 7.  bf3 != 0: the launch's K loop runs on it (conv_tile, BF3).
    const void* wp3; int32_t bf3;
    int32_t cfg;            // 0: pick the tile configuration with the cost model; c+1: use configuration c (autotuned; bit 3 of c = no epilogue-operand prefetch)
};

// One convolution stage of a fused fast-pathway block (I2VFastBlockParams): its packed weights / k-table and the parts of its epilogue.
struct I2VFastStage {
    const float* wp; const I2VKEntry* ktab; int32_t Kpad, Cdpad;
    const float* shift; int32_t relu;
    const uint32_t* gate; int32_t gate_stride, gate_pix0;          // input-gradient stages: the 1-bit gates of the tensor whose gradient this is
    uint32_t* gate_out; int32_t gate_out_stride, gate_out_pix0;    // forward stages: this tensor's own gates
};
// A bottleneck of SlowFast's FAST pathway (8 mid channels: half a 16-row MFMA fragment, 17-45 us launches at 1-17 TFLOP/s when run one
// convolution at a time) as ONE launch on packed-fp32 vector FMAs, intermediates in LDS / registers (k_fastblock, i2v_fastblock.hip):
//   forward  (mode 0):  A = conv1 (k x 1 x 1 temporal or pointwise, Cs -> CM) -> B = conv2 (1 x 3 x 3, CM -> CM) ->
//                       C = conv3 (pointwise, CM -> 4 CM) + residual (a tensor, or the projection D = downsample(x) computed here) + ReLU
//   backward (mode 1):  A = conv3's input gradient (pointwise, 4 CM -> CM) -> B = conv2's input gradient (3 x 3) -> stored (CM channels)
// Every output element is the same k-ordered fmaf chain over the same fp32 values as in the separate conv_igemm launches (an fp32 MFMA
// IS such a chain; rows with zero weights or zero operands add exact zeros), followed by the same epilogue operations in the same
// order: bit-identical, which is what lets the autotuner choose.  A block takes R whole rows of one frame (R W a multiple of 32: gate
// words are never shared between blocks); stage A is computed on R + 2 rows (the 3 x 3 stage's halo).
struct I2VFastBlockParams {
    int32_t mode, CM;
    const float* src; int64_t src_nstride; int32_t Cs;      // stage A's source view (x, or the gradient of the block's output)
    int32_t N, T, H, W, R;                                   // frames, frames per clip, plane, rows per block
    I2VFastStage a, b, c, d;                                 // c / d: forward only (d.wp == nullptr: no projection)
    const float* add0; int64_t add0_nstride;                 // forward without projection: the residual tensor
    float* dst; int64_t dst_nstride;                         // forward: the block's output (4 CM channels); backward: stage B's output (CM)
    uint32_t dv_w_m, dv_w_s, dv_hw_m, dv_hw_s;               // exact division by W and by H W
    int32_t U, S, G, BU;                                     // XCD mapping: units (= clips x S strip groups of G strips), blocks per unit (T G)
    uint32_t dv_bu_m, dv_bu_s, dv_s_m, dv_s_s, dv_g_m, dv_g_s;
    int32_t delay;                                           // probe (I2V_FB_DELAY): every other block sleeps this many x 127 x 64 clocks first
};

struct I2VPoolParams {
    const float* x;    int64_t x_nstride;    int32_t C, Hs, Ws;
    float* y;          int64_t y_nstride;    int32_t Ho, Wo;       // fwd: output; bwd: upstream grad
    float* gx;         int64_t gx_nstride;                           // bwd only
    const float* yact; int64_t yact_nstride;                         // bwd, optional: the pooled ACTIVATION (forward output); with
                                                                     // mask_relu the gate x > 0 is then evaluated as y > 0 on it
    uint8_t* idx;      // [N][C][Ho][Wo] window-relative arg-max (r*k+s), written by fwd, read by bwd
    int32_t N, k, stride, pad, mask_relu;
    // video pooling (k_pool3d_*): window kt x k x k over frame-major clips; N counts OUTPUT frames (clips*To)
    int32_t kt, stride_t, pad_t, Ts, To;
};   // average pooling reuses it: fwd x -> y, bwd y (upstream gradient) -> gx

// out = (a0 + a1 + a2) gated by (mask > 0); planes of HW elements, C channels, N frames
struct I2VAddMaskParams {
    float* out;        int64_t out_nstride;
    const float* a[3]; int64_t a_nstride[3];
    const float* mask; int64_t mask_nstride;
    const uint32_t* gate; int32_t gate_stride;     // 1-bit gates instead of `mask` (rows per channel, bit n*HW + i)
    int32_t N, C, HW;
    float gain;                                    // != 0: the result times this (backward gain of a ReLU, i2v_net_set_relu_gain)
};

struct I2VCosParams {
    const float* a;    int64_t a_nstride;    // current feature (view: frame stride, D contiguous)
    const float* b;    int64_t b_nstride;    // clean feature
    int64_t D;         int32_t N;
    float* partial;    int32_t nblk;         // [N][nblk][4] partial sums (dot, aa, bb, -)
    float* cos_out;                          // [N]
    float* grad;       int64_t grad_nstride;
    const float* coef_dev; int32_t coef_index; float coef_host;
    int32_t mask_relu, accumulate;
};

// ILAF loss over a whole hooked tensor (image_attacks.py:579-611): d = a - ori, d0 = adv0 - ori,
//   loss = -(0.5 * |d| / |d0| + <d0/|d0|, d/|d|>);  sums = (sum d*d, sum d*d0) in double
struct I2VIlafParams {
    const float* a;    int64_t a_nstride;
    const float* ori;  const float* adv0;    // dense [N][D] copies of the clean / initial adversarial features
    int64_t D;         int32_t N;
    double* partial;   int32_t nblk;         // [N*nblk][2]
    double* sums;                            // [nseg][2]
    double init_norm;                        // |d0| (used when init_sq is null)
    float* loss_out;                         // [nseg]
    float* grad;       int64_t grad_nstride;
    int32_t mask_relu, accumulate;
    // Segments: the reference attacks ONE clip per ILAF call (image_fine_tune_attack.py:73-79) and its norms run over that clip
    // only, so K clips batched into one launch list are K independent problems: frames [k*fps, (k+1)*fps) form segment k with its
    // own sums, initial norm and loss.  fps = 0: one segment of all N frames.  init_sq (device, [nseg]): |d0|^2 per segment as
    // left by the reduction over the initial adversarial features (the kernel takes the square root: correctly rounded, the
    // value the host path passes in init_norm).
    int32_t fps;
    const double* init_sq;
    // mode 1: the feature-distance term of base_attacks.TAP (:770-776) instead of the ILAF loss: per segment
    //   dist = || r(a) - r(ori) ||_2,  r(x) = sign(x) sqrt|x|   (sums[2 seg] = dist^2, loss_out[seg] = dist),
    //   grad = coef * (r(a) - r(ori)) / dist * 1 / (2 sqrt|a|),  0 where a == 0 (the ReLU behind a hooked stage selects 0 there in
    //   the reference, which is what keeps its NaN from sqrt'(0) out of the gradient) and 0 while dist == 0 (torch's norm backward).
    // adv0 / init_* are unused.
    int32_t mode; double coef;
};

// Non-local (self-attention) block core of gluoncv's `i3d_nl5_*` models -- the reference's I3D configurations
// (`/root/reference/utils.py:9-10`; two of the five blocks lie inside the stage ILAF hooks, `image_attacks.py:513-514`):
//   S[i][j] = scale * sum_c theta[c][i] * phi[c][j],  P = softmax_j(S),  y[c][i] = sum_j g[c][j] * P[i][j]
// per clip; i runs over the T*H*W positions of theta, j over the Tk*Hk*Wk positions of phi / g (max-pooled 1x2x2).
// A channel-major operand is a frame-major activation view: element (clip b, channel c, position i) lives at
//   p + (b * T + i / HW) * nstride + c * HW + i % HW.
struct I2VActMat { const float* p; int64_t nstride; int32_t T, HW; };
// The three product forms the block needs, forward and backward (fp32 MFMA, every output element one k-ordered fmaf chain, no atomics):
//   form 1  D[i][j] = scale * sum_c A[c][i] * B[c][j]     (D dense [clips][M][N];  S = theta^T phi,  dP = dY^T g)
//   form 2  C[c][i] = sum_j A[c][j] * D[i][j]             (C an activation view;   y = g P^T,        dtheta = phi dS^T)
//   form 3  C[c][j] = sum_i A[c][i] * D[i][j]             (                        dg = dY P,        dphi = theta dS)
struct I2VAttnGemm {
    int32_t form, clips, Cc, M, N;        // channels, rows (i) and columns (j) of the dense matrix
    I2VActMat A, B;                        // B only in form 1
    float* Cact; int64_t C_nstride; int32_t C_T, C_HW;   // forms 2 / 3: the output activation view
    float* D; const float* Din;            // dense matrix: output (form 1) / input (forms 2, 3), [clips][M][N]
    float scale; int32_t accumulate;       // forms 2 / 3: Cact += ...
    // forms 2 / 3 with few output tiles and a long reduction: the K axis in `ksplit` segments of attn_kseg(K, ksplit) rows, one block
    // each; segment sums go to part[clips][ksplit][Cc][cols] and are added in segment order (fixed by the shape: deterministic)
    int32_t ksplit; float* part;
};
// rows of one K segment: a whole number of the kernel's 32-row chunks
#ifdef __HIPCC__
__host__ __device__
#endif
inline int attn_kseg(int K, int ksplit) { return ksplit <= 1 ? K : ((K + ksplit - 1) / ksplit + 31) / 32 * 32; }
// row-wise softmax (mode 0: P = softmax(S) in place) and its backward (mode 1: dS = P o (dP - rowsum(dP o P)), in place over dP)
struct I2VSoftmaxRows { float* X; const float* P; int64_t rows; int32_t N; int32_t mode; };

// Classifier head of a white-box video model over its last feature map (frame-major: clips*T frames of (C, HW)):
// global average pool over (T, H, W) -> Linear(C -> K) -> softmax cross-entropy against `labels`, mean over the clips
// (attack.py:63-96 builds the gluoncv classifier, base_attacks.py:282-284 takes `CrossEntropyLoss()(model(adv), labels)`),
// and d(scale * loss) / d(feature):  grad[n][c][p] = scale/clips * (W^T (softmax - onehot))[clip(n)][c] / (T*HW)
struct I2VHeadParams {
    const float* a;    int64_t a_nstride;     // feature view, frame stride; C*HW contiguous per frame
    int32_t C, HW, T, clips, K;
    const float* W;    const float* bias;     // [K][C], [K]
    const int32_t* labels;                    // [clips]
    float scale;                              // `_targeted` of base_attacks.py:229-231 (+1 / -1)
    float* pooled;                            // [clips][Ctot]   (scratch)
    float* dpooled;                           // [clips][Ctot]   (scratch): d(scale * mean loss) / d pooled
    float* logits;                            // [clips][K]      (output: the model's prediction)
    float* loss_each;                         // [clips]         (-log softmax[label]; the host averages)
    float* grad;       int64_t grad_nstride;
    int32_t mask_relu, accumulate;
    // A head may read SEVERAL features (SlowFast: slow and fast pathway, pooled separately and concatenated): this feature's
    // channels are columns [c_off, c_off + C) of the Ctot-wide pooled vector / fc weight rows.
    int32_t Ctot, c_off;
    int32_t phase;                            // bit 0: pool this feature, bit 1: logits + loss + dpooled, bit 2: this feature's gradient
};

struct I2VStdParams {                          // Dispersion-Reduction loss: unbiased std of a tensor
    const float* a;    int64_t a_nstride; int64_t D; int32_t N;
    double* partial;   int32_t nblk;         // [N*nblk][2] (sum, sumsq)
    double* sums;                            // [2] (sum, sumsq) over the LOCAL frames -- all-reduced by the host
    double total_count;                      // element count behind `sums` when the gradient is formed (global)
    float* std_out;                          // [1]
    float* grad;       int64_t grad_nstride;
    int32_t mask_relu, accumulate;
};
