// Graph planner + executor behind the C ABI of include/i2v_hip.h.
//
// The host declares a backbone as buffers, channel-slice tensor views and conv/maxpool nodes.
// `i2v_net_plan` then
//   * packs every convolution twice -- forward [K=(tap,cin)][cout] and input-gradient
//     [K=(tap,cout)][cin] per stride-parity class -- with the eval-mode BatchNorm scale folded in
//     (reference: image_attacks.py:253-256 freezes BN; autograd's BN backward is a per-channel
//     multiply by the same scale),
//   * lays out one arena (activations + gradients + temporaries) for `max_frames` frames,
//   * emits a flat launch list for the forward pass to the deepest hook and one for the
//     input-gradient pass.  Only d(cost)/d(input) is produced: the reference's weight gradients
//     (image_attacks.py:352, wasted -- weights are frozen) are never computed.
//
// Backward fusion rule: every tensor's gradient is finalised by exactly one launch -- the dgrad of
// its FIRST consumer in forward order -- whose epilogue adds the pending contributions of the
// other consumers (residual alias, downsample dgrad, hook gradient) and applies the tensor's own
// ReLU mask.  Non-final contributions are written raw into temporaries.
#include "../../include/i2v_hip.h"
#include "i2v_kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#define I2V_MAX_NETS 4096
static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define CHECK_BE(expr)                                                            \
    do {                                                                          \
        if ((expr) != 0) return fail("%s: %s", #expr, be_error() ? be_error() : "backend error"); \
    } while (0)

namespace {

struct Buffer { int C, H, W; int T = 1; size_t act_off = 0, grad_off = 0; bool is_input = false;    // T: frames per clip (video networks)
                size_t gate_off = 0; int gate_words = 0; bool gated = false; };   // 1-bit ReLU gates: C rows of gate_words 32-bit words
struct Tensor { int buf, c_off, C; bool post_relu; float bwd_gain = 1.f; };   // bwd_gain: i2v_net_set_relu_gain

struct Packed {               // one implicit-GEMM operand set
    float* wp = nullptr; I2VKEntry* ktab = nullptr;
    float* wpc = nullptr;     // compact copy of wp for conv_igvfma_kernel (I2VConvParams::wpc)
    uint16_t* wp3 = nullptr;  // split-bf16 copy of `wp` (I2VConvParams::wp3), only in the bf16x3 math mode
    int K = 0, Kpad = 0, Cd = 0, Cdpad = 0, tap_uniform = 0;
    int halo = 0;           // 9 for a 3x3 / stride-1 / pad-1 packing in (16-channel group, tap, channel) order (kernel MODE 5), else 0
    int ph = 0, pw = 0, Hg = 0, Wg = 0;
    int pt = 0, Tg = 1;       // temporal parity class / grid frames per clip (video networks)
    int has_dt = 0;           // some k-table row carries a temporal tap offset
    int quad = 0, quad_kw = 0, quad_dw0 = 0;     // "quad rows" packing (I2VConvParams::quad): quads per row run, taps per run, first tap
    int tpair = 0;            // forward packing with TWO output frames per grid frame (rows = (frame class, channel)): see pack_fwd
    int ig_tt = 0, ig_th = 0, ig_tw = 0;      // image-gradient packing in tap-uniform order: union taps per axis (I2VConvParams::ig_*)
    int ig_p77 = 0;           // quad-row image gradient whose zero weights follow the stride-2 7 x 7 pattern (I2VConvParams::ig_p77)
};

struct Node {
    int type;                 // 0 conv, 1 maxpool, 2 avgpool, 3 attention core (non-local block)
    i2v_attn_desc ad{};       // type 3
    size_t p_off = 0;         // type 3: the attention matrix P [clips][M][N] (kept for the input-gradient pass), arena offset
    int src0() const { return type == 0 ? cd.src : type == 3 ? ad.theta : pd.src; }
    int dst0() const { return type == 0 ? cd.dst : type == 3 ? ad.dst : pd.dst; }
    i2v_conv3d_desc cd; i2v_pool3d_desc pd;        // image nodes are stored as kt = 1 video nodes
    std::vector<float> w;     // [cout][cin][kt][kh][kw] with scale folded
    std::vector<float> shift;
    std::vector<float> pre_scale, pre_shift;      // pre-activation conv (DenseNet): per input channel
    float* shift_d = nullptr; float* pre_scale_d = nullptr; float* pre_shift_d = nullptr;   // *_d padded to Kpad
    bool preact() const { return !pre_scale.empty(); }
    Packed fwd; std::vector<Packed> bwd;
    // input gradient of a convolution that reads the network input (class-packed): one launch -- or one per temporal class (pack_img)
    struct ImgGrad {
        Packed P; int blk = 0, sh = 1, blkt = 1;
        int ost = 1, ot0 = 0;                                // temporal output stride / offset
        int st = 1, oct = 1; bool skips = false;             // dz frames per grid frame, frames between its temporal classes, frames left to a memset
        double flop_share = 1.0;                             // this launch's part of the node's algorithmic flops
    };
    std::vector<ImgGrad> imgs;
    size_t idx_off = 0;                           // maxpool: arg-max bytes, arena offset in floats
};

enum Kind { L_CONV, L_IMGGRAD, L_POOLF, L_POOLB, L_ADDMASK, L_CONVB_UNUSED, L_AVGF, L_AVGB, L_MEMSET, L_POOL3F, L_POOL3B, L_AGEMM, L_SOFTMAX };   // L_IMGGRAD: conv_igemm with class-packed Cd
struct Launch {
    Kind kind;
    I2VConvParams conv; I2VPoolParams pool; I2VAddMaskParams am;
    I2VAttnGemm ag; I2VSoftmaxRows sm; int sm_rows_per_clip = 0;    // L_AGEMM / L_SOFTMAX (clips are filled in at run time)
    int T = 1;                     // frames per clip of the launch's iteration space (conv launches: conv.Tg)
    bool src_is_input = false;     // conv: src pointer patched with the caller's x
    bool img_accumulate = false;   // L_IMGGRAD of a second convolution reading the input (two-pathway stems): gx += ...
    float* ms_ptr = nullptr; size_t ms_floats_per_frame = 0;   // L_MEMSET
    bool ms_gx = false;            // L_MEMSET of the caller's gradient output (skipped when accumulating)
    double alg_flops_per_frame = 0; // L_IMGGRAD: algorithmic (not class-padded) flops
    // conv launches: the autotuner's tile configuration (conv.cfg encoding) per batch bucket b = clips in (max >> (b + 1), max >> b];
    // 0: not tuned (conv.cfg as planned).  Every configuration computes the same bits, so the choice never shows in a result.
    int cfg_b[4] = {0, 0, 0, 0};
    // Fused pair (k_conv_fused): this 3x3 launch and the NEXT launch of its list, the pointwise convolution over its output, may run as
    // one kernel that never stores the intermediate.  fuse_ok: k_conv_fusable's bits, 0 when anything else reads the intermediate
    // (mark_fusable); fuse_b[bucket]: what the autotuner measured -- 0 two launches, 1 fused (plain staging), 2 fused (halo staging).
    int fuse_ok = 0;
    int fuse_b[4] = {0, 0, 0, 0};
    // Fused fast-pathway block (k_fastblock, round 6): this launch and the next fb_ok - 1 launches of its list -- forward 3: conv1, conv2,
    // conv3; 4: conv1, conv2, the projection shortcut, conv3; backward 2: the input gradients of conv3 and conv2 -- may run as ONE kernel
    // that never stores the intermediates.  0 when anything else reads an intermediate (mark_fastblocks); fb_b[bucket]: what the
    // autotuner measured (0: separate launches, 1: fused).
    int node = -1;                 // conv launches: the graph node they belong to
    int fb_ok = 0;
    int fb_b[4] = {0, 0, 0, 0};
    // Launch overlap (mark_overlap, round 6): a convolution launch that does not depend on its predecessors back to launch ov_after
    // (-1: on nothing in its list) may run on the net's SIDE stream, issued right after launch ov_after, while the main stream goes on;
    // ov_join is the first later launch that touches what it writes or reads (list size: none in this list) and waits for it.
    // ov_after == -2: runs in place.  No launch changes, so no result changes.
    int ov_after = -2, ov_join = -1;
};
static int cfg_bucket(int clips, int max_clips) {
    int b = 0;
    while (b < 3 && (max_clips >> (b + 1)) >= clips && (max_clips >> (b + 1)) >= 1) ++b;
    return b;
}

struct Addend { const float* p; int64_t nstride; int stride, H, W; };

struct Net {
    std::vector<Buffer> bufs; std::vector<Tensor> tens; std::vector<Node> nodes;
    int input = -1; std::vector<int> hooks; int maxN = 0; bool planned = false;
    float* arena = nullptr; size_t arena_floats = 0; std::vector<void*> dev_allocs;
    std::vector<Launch> fwd, bwd;
    int frames = 0;
    int Tin() const { return bufs[tens[input].buf].T; }     // frames per clip of the input
    size_t weight_bytes = 0;
    std::vector<float*> hook_tmp;  // per hook: separate gradient buffer when the hooked tensor is also consumed
    size_t in_stage_off = 0; bool stage_input = false;   // quad-row stems read up to 64 bytes around a view: the caller's frames are
                                                         // copied into the arena (slack on both sides) before the forward pass
    // launch overlap (mark_overlap / run_list): the side stream, its event pool, and per list (0 forward, 1 backward) the hoisted
    // launches to issue right after main launch p (index p + 1; index 0: at the start of the list)
    i2v_stream_t side = nullptr; std::vector<void*> ov_ev; size_t ov_used = 0;
    std::vector<std::vector<int>> ov_at[2]; int ov_max_frames = 0;
};

}  // namespace

// `chain`: the launch follows the previous timed launch back to back on the same stream (same launch list), so its
// start IS that launch's stop event -- one event record per launch instead of two (the records cost ~2 us of stream
// time each, 4 % of the headline bench when every launch carried a pair).
struct TimedLaunch { void* start; void* stop; int kind; double flops; int Cd, K, HWg, frames, pw; void* chain_from; double bytes; int count = 1; };
struct i2v_ctx {
    int device; std::vector<Net*> nets; std::mutex nets_mu;     // the table: created / destroyed under the lock, ids of destroyed nets are handed out again
    // nets may be executed from several threads on several streams (clip lanes): entries are handed out under a lock,
    // live in a deque (stable addresses) and chain to an explicit event, never to "the previous entry"
    int timing = 0;          // 0 off, 1 one event pair per launch, 2 one per SEGMENT (run of consecutive launches of one kind)
    std::deque<TimedLaunch> timed; size_t timed_used = 0; std::mutex timing_mu;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
static int upload(Net& n, const std::vector<T>& host, T** dev) {
    size_t bytes = host.size() * sizeof(T);
    void* d = be_malloc(bytes ? bytes : 16);
    if (!d) return fail("device allocation of %zu bytes failed", bytes);
    n.dev_allocs.push_back(d);
    n.weight_bytes += bytes;
    if (bytes) CHECK_BE(be_h2d(d, host.data(), bytes));
    *dev = (T*)d;
    return 0;
}

// Math mode of a plan.  Default: every convolution on fp32-input MFMAs (exact fp32: a k-ordered fmaf chain, the bit-exact rungs of
// the parity ladder).  I2V_MATH=bf16x3 (opt-in, round 5): launches the split-bf16 K loop admits (conv_bf3_ok) run on three-term bf16
// operands -- six bf16 MFMAs per 16 K rows in place of eight fp32 ones at twice the cycles, every product term down to 2^-26 of |w||x|
// kept, fp32 accumulation.  Read when a net is PLANNED (one process may hold plans of both kinds); results of a plan do not depend on
// the autotuner's tile choices in either mode.
static bool math_bf16x3() { const char* e = getenv("I2V_MATH"); return e && !strcmp(e, "bf16x3"); }

// w = w1 + w2 + w3 in bf16 (round to nearest even at every level; the residuals are exact in fp32), laid out in the 32x32x16 bf16 MFMA's
// A-fragment order: [Kpad / 16][Cdpad / 32][term][lane][8]: lane l holds row 32 tile + (l & 31), K rows 16 chunk + 8 (l >> 5) + j.
static int upload_split_bf16(Net& n, const std::vector<float>& wp, Packed& P) {
    if (!math_bf16x3() || P.quad || P.Kpad % I2V_KC || P.Cdpad % 32 || P.Kpad == 0) return 0;
    auto bf = [](float x) { uint32_t u; memcpy(&u, &x, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; float y; memcpy(&y, &u, 4); return y; };
    const int nch = P.Kpad / I2V_KC, nt = P.Cdpad / 32;
    std::vector<uint16_t> w3((size_t)nch * nt * 3 * 64 * 8);
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < nt; ++t)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    float w = wp[(size_t)(16 * c + 8 * (l >> 5) + j) * P.Cdpad + 32 * t + (l & 31)];
                    for (int term = 0; term < 3; ++term) {
                        const float b = bf(w); uint32_t u; memcpy(&u, &b, 4);
                        w3[((((size_t)c * nt + t) * 3 + term) * 64 + l) * 8 + j] = (uint16_t)(u >> 16);
                        w -= b;
                    }
                }
    return upload(n, w3, &P.wp3);
}

static Net* get_net(i2v_handle h, int id) {
    if (!h || id < 0 || id >= (int)h->nets.size() || !h->nets[id]) { fail("bad net id %d", id); return nullptr; }
    return h->nets[id];
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
static int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
static int posmod(int a, int b) { int m = a % b; return m < 0 ? m + b : m; }

static int pack_fwd(Net& n, Node& nd) {
    const i2v_conv3d_desc& c = nd.cd;
    const Buffer& sb = n.bufs[n.tens[c.src].buf];
    int K = c.kt * c.kh * c.kw * c.cin;
    Packed& P = nd.fwd;
    P.K = K; P.Kpad = (int)align_up(K, I2V_KC); P.Cd = c.cout; P.Cdpad = (int)align_up(c.cout, 128);
    P.tap_uniform = (c.cin % I2V_KC == 0) ? 1 : 0;
    if (P.tap_uniform && c.kt == 1 && c.kh == 3 && c.kw == 3 && c.stride == 1 && c.stride_t == 1 && c.pad == 1 && !nd.preact()) P.halo = 9;
    // "Quad rows" for narrow stems (few input channels AND few output channels: SlowFast's fast pathway, 3 -> 8): such a launch
    // spends its time ISSUING the 4-byte im2col DMA of the per-row path (one instruction per K row and 64 pixels; 17 TFLOP/s),
    // not in the matrix pipe.  K rows ordered (channel, frame tap, row tap, column-tap quad x 4) put four ADJACENT source pixels
    // in consecutive rows, which the kernel (MODE 4) stages with ONE 16-byte DMA per pixel: 1.6x on that launch
    // (tools/conv_microbench.cpp "fast stem").  Wide stems (64 output channels) are bound elsewhere and measured no gain.
    static const bool no_quad = [] { const char* e = getenv("I2V_QUAD"); return e && e[0] == '0'; }();
    if (!no_quad && c.cin < I2V_KC && c.cout <= 32 && c.kw >= 2 && c.kw <= 8 && !nd.preact()) {
        const int kwq = (c.kw + 3) / 4;
        // Frame PAIRS (round 3): with <= 8 output channels (SlowFast's fast stem) half of even a 16-row fragment is empty.  Two
        // consecutive output frames share most of their source frames when the kernel spans time (5 taps at stride = dilation 2:
        // six distinct source frames for the pair instead of ten), so one grid frame computes BOTH -- rows (frame class, channel), K
        // rows over the UNION of the pair's frame taps, zero weights where a class has no tap there -- with 0.6x the matrix work
        // and im2col traffic of two half-empty launches.  A zero weight adds an exact +0 to the k-ordered chain and the real taps
        // keep their order: same bits as the unpaired packing.  The epilogue is the class-packed one of the image gradient
        // (I2VConvParams::blkt = 2: row -> channel, frame 2 tau + class).
        static const bool no_tpair = [] { const char* e = getenv("I2V_TPAIR"); return e && e[0] == '0'; }();
        const Buffer& dbuf = n.bufs[n.tens[c.dst].buf];
        std::vector<int> taps;                       // frame offsets relative to the grid frame's base source frame
        const int classes = (!no_tpair && c.kt > 1 && 2 * c.cout <= 16 && dbuf.T >= 2) ? 2 : 1;
        for (int ct = 0; ct < classes; ++ct)
            for (int q = 0; q < c.kt; ++q) {
                const int u = ct * c.stride_t + q * c.dil_t - c.pad_t;
                if (std::find(taps.begin(), taps.end(), u) == taps.end()) taps.push_back(u);
            }
        std::sort(taps.begin(), taps.end());
        const int NU = (int)taps.size();
        if (classes == 2) { P.tpair = 2; P.Cd = 2 * c.cout; P.Cdpad = (int)align_up(P.Cd, 128); }
        P.K = c.cin * NU * c.kh * kwq * 4; P.Kpad = (int)align_up(P.K, I2V_KC); P.tap_uniform = 0;
        P.quad = kwq; P.quad_kw = c.kw; P.quad_dw0 = -c.pad;
        if (c.kh == 7 && c.kw == 7 && c.pad == 3) P.halo = 77;      // seven row taps from -3: what conv_stem_halo's window (37 rows from 2 y0 - 3, 56 K rows per plane) is built for
        std::vector<float> wq((size_t)P.Kpad * P.Cdpad, 0.f);
        std::vector<I2VKEntry> kq(P.Kpad, I2VKEntry{0, 0, 0, 0});
        for (int ci = 0; ci < c.cin; ++ci)
            for (int ui = 0; ui < NU; ++ui)
                for (int r = 0; r < c.kh; ++r)
                    for (int s4 = 0; s4 < kwq * 4; ++s4) {
                        const int k = (((ci * NU + ui) * c.kh + r) * kwq) * 4 + s4;
                        kq[k] = I2VKEntry{ci * sb.H * sb.W, r - c.pad, s4 - c.pad, (s4 < c.kw ? 1 : 0) + 2 * taps[ui]};
                        if (s4 >= c.kw) continue;
                        for (int ct = 0; ct < classes; ++ct) {
                            const int num = taps[ui] - ct * c.stride_t + c.pad_t;       // = q * dil_t for this class's tap q
                            if (num < 0 || num % c.dil_t || num / c.dil_t >= c.kt) continue;
                            const int q = num / c.dil_t;
                            for (int co = 0; co < c.cout; ++co)
                                wq[(size_t)k * P.Cdpad + ct * c.cout + co] = nd.w[((((size_t)co * c.cin + ci) * c.kt + q) * c.kh + r) * c.kw + s4];
                        }
                    }
        for (const I2VKEntry& e : kq) if (e.valid >> 1) P.has_dt = 1;
        if (upload(n, wq, &P.wp)) return 1;
        return upload(n, kq, &P.ktab);
    }
    // the 7x7 / stride-2 / pad-3 stem over 3 channels in (tap, channel) order: conv_stem64_halo may walk it without the k-table
    if (!P.tap_uniform && c.cin == 3 && c.kt == 1 && c.kh == 7 && c.kw == 7 && c.stride == 2 && c.pad == 3 && c.dil_t == 1 && c.pad_t == 0 && !nd.preact()) P.halo = 49;
    if (c.kh == 1 && c.kw == 1 && c.pad == 0 && c.stride == 1 && !nd.preact()) P.halo = 1;      // every tap at (0, 0): k x 1 x 1 (conv_vfma_kernel's mark)
    std::vector<float> wp((size_t)P.Kpad * P.Cdpad, 0.f);
    std::vector<I2VKEntry> kt(P.Kpad, I2VKEntry{0, 0, 0, 0});
    // K order: (16-channel chunk, tap, channel in chunk) when the channel count allows -- each 16-row chunk keeps a
    // single tap (MODE 2) and consecutive chunks re-read the same 16 channels at the next tap, a few KB apart in
    // the cache instead of a full channel sweep apart -- else (tap, channel).
    const int NT = c.kt * c.kh * c.kw;
    for (int q = 0; q < c.kt; ++q)
        for (int r = 0; r < c.kh; ++r)
            for (int s = 0; s < c.kw; ++s)
                for (int ci = 0; ci < c.cin; ++ci) {
                    const int tap = (q * c.kh + r) * c.kw + s;
                    int k = P.tap_uniform ? ((ci / I2V_KC) * NT + tap) * I2V_KC + ci % I2V_KC : tap * c.cin + ci;
                    kt[k] = I2VKEntry{ci * sb.H * sb.W, r - c.pad, s - c.pad, 1 + 2 * (q * c.dil_t - c.pad_t)};
                    for (int co = 0; co < c.cout; ++co)
                        wp[(size_t)k * P.Cdpad + co] = nd.w[((((size_t)co * c.cin + ci) * c.kt + q) * c.kh + r) * c.kw + s];
                }
    for (const I2VKEntry& e : kt) if (e.valid >> 1) P.has_dt = 1;
    if (upload(n, wp, &P.wp) || upload_split_bf16(n, wp, P)) return 1;
    return upload(n, kt, &P.ktab);
}

// Input-gradient operands, one per stride-parity class (pt, ph, pw): the source positions congruent to the
// class modulo the stride receive exactly the taps with (class + pad - tap) % stride == 0.
static int pack_bwd(Net& n, Node& nd) {
    const i2v_conv3d_desc& c = nd.cd;
    const Buffer& sb = n.bufs[n.tens[c.src].buf];
    const Buffer& db = n.bufs[n.tens[c.dst].buf];
    int st = c.stride, stt = c.stride_t;
    for (int pt = 0; pt < stt; ++pt)
    for (int ph = 0; ph < st; ++ph)
        for (int pw = 0; pw < st; ++pw) {
            Packed P;
            P.pt = pt; P.ph = ph; P.pw = pw;
            P.Tg = (sb.T - pt + stt - 1) / stt;
            P.Hg = (sb.H - ph + st - 1) / st; P.Wg = (sb.W - pw + st - 1) / st;
            std::vector<int> tq, tr, ts;
            for (int q = 0; q < c.kt; ++q) if (posmod(pt + c.pad_t - q * c.dil_t, stt) == 0) tq.push_back(q);
            for (int r = 0; r < c.kh; ++r) if (posmod(ph + c.pad - r, st) == 0) tr.push_back(r);
            for (int s = 0; s < c.kw; ++s) if (posmod(pw + c.pad - s, st) == 0) ts.push_back(s);
            int K = (int)(tq.size() * tr.size() * ts.size()) * c.cout;
            P.K = K; P.Kpad = (int)align_up(K, I2V_KC); P.Cd = c.cin; P.Cdpad = (int)align_up(c.cin, 128);
            P.tap_uniform = (c.cout % I2V_KC == 0) ? 1 : 0;
            if (P.tap_uniform && st == 1 && stt == 1 && c.kt == 1 && c.kh == 3 && c.kw == 3 && c.pad == 1 && !nd.preact()) P.halo = 9;
            if (c.kh == 1 && c.kw == 1 && c.pad == 0 && st == 1 && !nd.preact()) P.halo = 1;      // every tap at (0, 0)
            std::vector<float> wp((size_t)P.Kpad * P.Cdpad, 0.f);
            std::vector<I2VKEntry> kt(P.Kpad ? P.Kpad : 1, I2VKEntry{0, 0, 0, 0});
            int t = 0;
            const int NTc = (int)(tq.size() * tr.size() * ts.size());
            for (int q : tq)
            for (int r : tr)
                for (int s : ts) {
                    int dt = floordiv(pt + c.pad_t - q * c.dil_t, stt);
                    int dh = floordiv(ph + c.pad - r, st), dw = floordiv(pw + c.pad - s, st);
                    for (int co = 0; co < c.cout; ++co) {
                        int k = P.tap_uniform ? ((co / I2V_KC) * NTc + t) * I2V_KC + co % I2V_KC : t * c.cout + co;
                        kt[k] = I2VKEntry{co * db.H * db.W, dh, dw, 1 + 2 * dt};
                        for (int ci = 0; ci < c.cin; ++ci)
                            wp[(size_t)k * P.Cdpad + ci] =
                                nd.w[((((size_t)co * c.cin + ci) * c.kt + q) * c.kh + r) * c.kw + s] * (nd.preact() ? nd.pre_scale[ci] : 1.f);
                    }
                    ++t;
                }
            for (const I2VKEntry& e : kt) if (e.valid >> 1) P.has_dt = 1;
            if (upload(n, wp, &P.wp) || upload_split_bf16(n, wp, P)) return 1;
            if (upload(n, kt, &P.ktab)) return 1;
            nd.bwd.push_back(P);
        }
    return 0;
}

// Gradient of the FIRST convolution w.r.t. the image.  GEMM-N would be Cin = 3; instead the output is
// cut into B x B position blocks (B = stride, or 2 for stride 1) and the B*B*Cin (class, channel)
// pairs form the Cd axis: out[(ph,pw),ci][i][j] = sum_{co,dh,dw} w'[(co,dh,dw)][(ph,pw),ci] *
// dz[co][i*m + dh][j*m + dw], m = B/stride, with zero weights where a class has no such tap.
// The packings that suit the 16-row halo-tile kernel (conv_imggrad_halo) are chosen when it is among the autotuner's candidates
// (I2V_IGHALO not 0) and not switched off (I2V_IMG_SPLIT=0: the round-3 packings; read per plan so that tests can compare the two)
static bool img_prefer_16_rows() {
    static const bool no_igh = [] { const char* e = getenv("I2V_IGHALO"); return e && e[0] == '0'; }();
    const char* const es = getenv("I2V_IMG_SPLIT");
    return !no_igh && !(es && es[0] == '0');
}
// `only_ct` >= 0: pack that temporal class ALONE -- grid = its own frames, its own frame taps --, as the frame-skipping case below does
// for the one class that has taps (pack_img decides when a dense temporal stride is split into one launch per class).
static int pack_img_one(Net& n, Node& nd, Node::ImgGrad& ig, const int only_ct) {
    const i2v_conv3d_desc& c = nd.cd;
    const Buffer& sb = n.bufs[n.tens[c.src].buf];
    const Buffer& db = n.bufs[n.tens[c.dst].buf];
    const int st = c.stride, B = st == 1 ? 2 : st, m = B / st;
    const int stt = c.stride_t;
    // temporal classes: one per stride residue (1 for images).  A stem that samples every stt-th frame with a
    // kernel that reaches no other residue (SlowFast's slow pathway: kt = 1) has taps in ONE class only: then
    // only that class is packed (grid = the sampled frames, output stride stt) and the frames in between are
    // zero-filled by a memset instead of being computed as stt - 1 classes of zero weights.
    int with_taps = 0, only = 0;
    for (int ct = 0; ct < stt; ++ct) {
        bool any = false;
        for (int q = 0; q < c.kt; ++q) if (posmod(ct + c.pad_t - q * c.dil_t, stt) == 0) any = true;
        if (any) { with_taps++; only = ct; }
    }
    const bool forced = only_ct >= 0;
    if (forced) only = only_ct;
    const bool sparse = forced || (stt > 1 && with_taps == 1);
    int Bt = sparse ? 1 : stt; const int ct0 = sparse ? only : 0;
    static const bool no_tpair = [] { const char* e = getenv("I2V_TPAIR"); return e && e[0] == '0'; }();
    // (Tried and dropped for the DENSE temporal stride of I3D's stem: two stride periods per grid frame -- 48 of 64 rows over 4 dz frames
    //  instead of 2 x (24 of 32 over 3) -- runs the 64-row tiles and was 23 % SLOWER, 1343 -> 1657 us per launch.)
    int dt_lo = 1 << 30, dt_hi = -(1 << 30), dh_lo = 1 << 30, dh_hi = -(1 << 30), dw_lo = 1 << 30, dw_hi = -(1 << 30);
    for (int ct = ct0; ct < ct0 + Bt; ++ct)
        for (int q = 0; q < c.kt; ++q)
            if (posmod(ct + c.pad_t - q * c.dil_t, stt) == 0) { int d = floordiv(ct + c.pad_t - q * c.dil_t, stt); dt_lo = d < dt_lo ? d : dt_lo; dt_hi = d > dt_hi ? d : dt_hi; }
    for (int ph = 0; ph < B; ++ph)
        for (int r = 0; r < c.kh; ++r)
            if (posmod(ph + c.pad - r, st) == 0) { int d = floordiv(ph + c.pad - r, st); dh_lo = d < dh_lo ? d : dh_lo; dh_hi = d > dh_hi ? d : dh_hi; }
    for (int pw = 0; pw < B; ++pw)
        for (int s = 0; s < c.kw; ++s)
            if (posmod(pw + c.pad - s, st) == 0) { int d = floordiv(pw + c.pad - s, st); dw_lo = d < dw_lo ? d : dw_lo; dw_hi = d > dw_hi ? d : dw_hi; }
    const int TH = dh_hi - dh_lo + 1, TW = dw_hi - dw_lo + 1;
    // Pairs of SAMPLED frames (round 3; the gradient-side twin of pack_fwd's frame pairs): a frame-skipping stem whose kernel spans
    // time (SlowFast's fast stem: every 2nd frame, 5 taps) gives each sampled frame 5 dz frames, two neighbouring sampled frames 6
    // between them -- both as temporal classes of ONE grid frame: 24 of 32 rows over 6 frame taps instead of two launches' worth
    // of 12 of 16 rows over 5.  The classes lie stt frames apart (I2VConvParams::oct).  Zero weights where a class has no tap: same bits.
    // ... unless the 16-row halo-tile kernel can take the UNPAIRED quad-row packing (round 5: a 4 x 4 tap window, the frame-tap planes of
    // 1, 2 or 4 channels -- a whole number of four-chunk groups -- within its 20 planes): 12 of 16 rows over the sampled frame's own 5
    // taps is 17 % less matrix work than 24 of 32 over 6, and that kernel stages a plane once whatever the row count.
    const int TTu = dt_hi - dt_lo + 1, cps_u = TTu % 4 == 0 ? 1 : TTu % 2 == 0 ? 2 : 4;
    const bool unpaired_halo = img_prefer_16_rows() && c.cout % I2V_KC != 0 && TH == 4 && TW >= 2 && TW <= 4 && c.cout % cps_u == 0 && cps_u * TTu <= 20 &&
                               B * B * c.cin <= 16 && m == 1;
    const bool pairs = sparse && !forced && !no_tpair && !unpaired_halo && dt_hi > dt_lo && 2 * B * B * c.cin <= 32 && (sb.T - ct0 + stt - 1) / stt >= 2;
    if (pairs) { Bt = 2; dt_hi += 1; }
    const int TT = dt_hi - dt_lo + 1;
    Packed& P = ig.P;
    // few output channels (SlowFast's fast stem: 8) cannot use the tap-uniform path; instead of the per-row path they take the
    // "quad rows" order (channel, frame tap, row tap, column-tap quad x 4): see pack_fwd
    static const bool no_quad = [] { const char* e = getenv("I2V_QUAD"); return e && e[0] == '0'; }();
    // Round 5, measured and NOT taken (opt-in I2V_IMG_QUAD=1): quad rows for the stride-2 7x7 stems with 64 output channels (ResNet, I3D, SlowFast's
    // slow pathway) too.  Their class-packed gradient has exactly FOUR column taps per row run (TW = 4: one quad, no padding) and issues 16
    // four-byte im2col DMA pieces per wave and chunk beside 16 MFMAs of 32 cycles (PMC: matrix pipe 0.59 busy); as quad rows the same chunk is four
    // 16-byte pieces per wave.  Slower all the same: image gradient 52.0 -> 44.6 TFLOP/s on the I2V stem, 60.7 -> 53.6 on I3D's, 43.7 -> 40.0 on
    // SlowFast's (gpurun_out r5p, same box, alternated): the channel-major K order sweeps one channel's 4 x 4 window per chunk, and the masked
    // fragment reads of MODE 4 cost more than the DMA instructions they save.  Same products either way (the host simulation follows the k-table).
    static const bool img_quad = [] { const char* e = getenv("I2V_IMG_QUAD"); return e && e[0] == '1'; }();
    const bool quad = !no_quad && (c.cout % I2V_KC != 0 || (img_quad && TW == 4)) && TW >= 2 && TW <= 8;
    const int TWq = quad ? (TW + 3) / 4 * 4 : TW;               // column taps per run, padded to whole quads
    P.K = TT * TH * TWq * c.cout; P.Kpad = (int)align_up(P.K, I2V_KC);
    P.Cd = Bt * B * B * c.cin; P.Cdpad = (int)align_up(P.Cd, 128);
    P.tap_uniform = (!quad && c.cout % I2V_KC == 0) ? 1 : 0;
    if (quad) { P.quad = TWq / 4; P.quad_kw = TW; P.quad_dw0 = dw_lo; }
    if (P.tap_uniform) { P.ig_tt = TT; P.ig_th = TH; P.ig_tw = TW; }
    else if (quad && TWq == 4 && TH == 4) { P.ig_tt = TT; P.ig_th = TH; P.ig_tw = TWq; }      // (conv_imggrad_halo, QUAD: one 4 x 4 plane per chunk)
    P.Tg = sparse ? (sb.T - ct0 + stt - 1) / stt : (sb.T + Bt - 1) / Bt; P.Hg = (sb.H + B - 1) / B; P.Wg = (sb.W + B - 1) / B;
    ig.blk = B; ig.sh = m; ig.blkt = Bt; ig.ost = sparse ? stt : Bt; ig.ot0 = ct0; ig.skips = sparse && !forced;
    if (pairs) { P.Tg = (P.Tg + 1) / 2; ig.ost = 2 * stt; ig.st = 2; ig.oct = stt; }
    {   // this launch's part of the node's algorithmic flops: its classes' frame taps over all of them
        int mine = 0, all = 0;
        for (int ct = 0; ct < stt; ++ct)
            for (int q = 0; q < c.kt; ++q)
                if (posmod(ct + c.pad_t - q * c.dil_t, stt) == 0) { ++all; if (!forced || ct == only_ct) ++mine; }
        ig.flop_share = all > 0 ? (double)mine / all : 1.0;
    }
    std::vector<float> wp((size_t)P.Kpad * P.Cdpad, 0.f);
    std::vector<I2VKEntry> kt(P.Kpad, I2VKEntry{0, 0, 0, 0});
    // K order = (16-channel chunk, tap, channel in chunk) when the channel count allows: every 16-row K chunk
    // still has ONE tap (MODE 2), and a block sweeps all taps of 16 channels before moving on, so the taps'
    // overlapping reads of `dz` are a few KB apart instead of a full 64-channel sweep apart (the co-resident
    // blocks' halos then fit the L2).  Otherwise (tap, channel) -- or the quad-row order.
    const int NT = TT * TH * TW;
    auto krow = [&](int tt, int th, int tw, int co) {
        if (quad) return (((co * TT + tt) * TH + th) * TWq) + tw;
        const int tap = (tt * TH + th) * TW + tw;
        return P.tap_uniform ? ((co / I2V_KC) * NT + tap) * I2V_KC + co % I2V_KC : tap * c.cout + co;
    };
    for (int tt = 0; tt < TT; ++tt)
    for (int th = 0; th < TH; ++th)
        for (int tw = 0; tw < TWq; ++tw)
            for (int co = 0; co < c.cout; ++co)
                kt[krow(tt, th, tw, co)] = I2VKEntry{co * db.H * db.W, th + dh_lo, tw + dw_lo, (tw < TW ? 1 : 0) + 2 * (tt + dt_lo)};
    for (int cc = 0; cc < Bt; ++cc)
    for (int q = 0; q < c.kt; ++q) {
        const int ct = pairs ? ct0 : ct0 + cc;          // (pairs: both classes are the ONE residue with taps, a sampled frame apart)
        if (posmod(ct + c.pad_t - q * c.dil_t, stt)) continue;
        const int tt = floordiv(ct + c.pad_t - q * c.dil_t, stt) - dt_lo + (pairs ? cc : 0);
        for (int ph = 0; ph < B; ++ph)
            for (int pw = 0; pw < B; ++pw)
                for (int r = 0; r < c.kh; ++r) {
                    if (posmod(ph + c.pad - r, st)) continue;
                    const int th = floordiv(ph + c.pad - r, st) - dh_lo;
                    for (int s = 0; s < c.kw; ++s) {
                        if (posmod(pw + c.pad - s, st)) continue;
                        const int tw = floordiv(pw + c.pad - s, st) - dw_lo;
                        for (int co = 0; co < c.cout; ++co)
                            for (int ci = 0; ci < c.cin; ++ci)
                                wp[(size_t)krow(tt, th, tw, co) * P.Cdpad + ((cc * B + ph) * B + pw) * c.cin + ci] =
                                    nd.w[((((size_t)co * c.cin + ci) * c.kt + q) * c.kh + r) * c.kw + s];
                    }
                }
    }
    for (const I2VKEntry& e : kt) if (e.valid >> 1) P.has_dt = 1;
    // conv_igvfma_kernel skips the class-row pairs a tap cannot feed under the stride-2 7 x 7 geometry (row class ph owns row tap th iff
    // ph == 1 || th < 3, column class alike): claimed only when EVERY weight outside that pattern is an exact zero in this packing
    if (quad && TWq == 4 && TH == 4 && Bt == 1 && B == 2 && c.cin == 3 && P.Cd == 12) {
        bool ok = true;
        for (int k = 0; k < P.K && ok; ++k) {
            const int tw = k % 4, th = (k / 4) % 4;
            for (int cd = 0; cd < 12 && ok; ++cd) {
                const int cls = cd / 3, ph = cls / 2, pw = cls % 2;
                const bool owned = (ph == 1 || th < 3) && (pw == 1 || tw < 3);
                if (!owned && wp[(size_t)k * P.Cdpad + cd] != 0.f) ok = false;
            }
        }
        P.ig_p77 = ok ? 1 : 0;
        std::vector<float> wc((size_t)P.Kpad * 16, 0.f);             // rows of 16 floats for the vector-FMA kernel's scalar loads
        for (int k = 0; k < P.K; ++k) for (int cd = 0; cd < 12; ++cd) wc[(size_t)k * 16 + cd] = wp[(size_t)k * P.Cdpad + cd];
        if (upload(n, wc, &P.wpc)) return 1;
    }
    if (upload(n, wp, &P.wp)) return 1;
    return upload(n, kt, &P.ktab);
}

// The input gradient of a stem as one class-packed launch -- or, for a DENSE temporal stride (I3D: 5x7x7 / (2,2,2): two temporal classes
// with 3 and 2 frame taps), one launch per temporal class: packed together the classes share the union of their frame taps (3) and a
// 32-row fragment (24 of 32 rows), alone each runs its own taps on 12 of 16 rows -- 5 x 16 instead of 3 x 32 row-taps per pair of
// frames, and the 16-row conv_imggrad_halo keeps six blocks per CU where the 32-row one keeps four.  Only when that kernel is among the
// candidates (tap-uniform packing of a stride-2 stem, I2V_IGHALO not 0): conv_tile prefers the packed form.  A zero weight adds an exact
// +0 to the k-ordered chain and the real taps keep their order: same bits either way.
static int pack_img(Net& n, Node& nd) {
    const i2v_conv3d_desc& c = nd.cd;
    const int stt = c.stride_t, B = c.stride == 1 ? 2 : c.stride;
    int classes_with_taps = 0;
    for (int ct = 0; ct < stt; ++ct) {
        bool any = false;
        for (int q = 0; q < c.kt; ++q) if (posmod(ct + c.pad_t - q * c.dil_t, stt) == 0) any = true;
        classes_with_taps += any ? 1 : 0;
    }
    const bool split = img_prefer_16_rows() && stt > 1 && classes_with_taps == stt && c.cout % I2V_KC == 0 && c.stride == 2 && B * B * c.cin <= 16 &&
                       c.kh <= 8 && c.kw <= 8 && (((c.kh + 1) / 2) * ((c.kw + 1) / 2)) % 4 == 0 && n.bufs[n.tens[c.src].buf].T >= stt;
    nd.imgs.clear();
    if (!split) { nd.imgs.emplace_back(); return pack_img_one(n, nd, nd.imgs.back(), -1); }
    nd.imgs.resize(stt);
    for (int ct = 0; ct < stt; ++ct) if (pack_img_one(n, nd, nd.imgs[ct], ct)) return 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// views
// ---------------------------------------------------------------------------------------------
struct View { float* p; int64_t nstride; int C, H, W; int T = 1; };

static View view_of(Net& n, int t, bool grad) {
    const Tensor& T = n.tens[t];
    const Buffer& B = n.bufs[T.buf];
    size_t off = grad ? B.grad_off : B.act_off;
    View v;
    v.p = n.arena + off + (size_t)T.c_off * B.H * B.W;
    v.nstride = (int64_t)B.C * B.H * B.W;
    v.C = T.C; v.H = B.H; v.W = B.W; v.T = B.T;
    return v;
}

// ---------------------------------------------------------------------------------------------
// C ABI: lifetime
// ---------------------------------------------------------------------------------------------
extern "C" const char* i2v_last_error(void) { return g_err.c_str(); }
extern "C" int i2v_abi_version(void) { return 1; }
extern "C" const char* i2v_backend(void) { return be_name(); }
static long long g_overlap_launches = 0;        // launches issued on a side stream (mark_overlap): a relaxed counter, diagnostics only
extern "C" long long i2v_backend_stat(const char* name) {
    if (name && !strcmp(name, "overlap_launches")) return __atomic_load_n(&g_overlap_launches, __ATOMIC_RELAXED);
    return name ? be_stat(name) : -1;
}

extern "C" int i2v_create(int device, i2v_handle* out) {
    if (!out) return fail("i2v_create: null out");
    if (be_set_device(device)) return fail("i2v_create: cannot select device %d: %s", device,
                                           be_error() ? be_error() : "?");
    *out = new i2v_ctx(); (*out)->device = device;
    (*out)->nets.reserve(I2V_MAX_NETS);     // the table never reallocates: other threads may be executing planned nets while one is added
    return 0;
}

static void free_net(Net* n) {
    if (!n) return;
    for (void* p : n->dev_allocs) be_free(p);
    if (n->arena) be_free(n->arena);
    for (void* e : n->ov_ev) be_event_destroy(e);
    if (n->side) be_stream_destroy(n->side);
    delete n;
}

extern "C" int i2v_destroy(i2v_handle h) {
    if (!h) return 0;
    for (Net* n : h->nets) free_net(n);
    for (auto& t : h->timed) { be_event_destroy(t.start); be_event_destroy(t.stop); }
    delete h;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// C ABI: description
// ---------------------------------------------------------------------------------------------
extern "C" int i2v_net_create(i2v_handle h, int* net) {
    if (!h || !net) return fail("i2v_net_create: null argument");
    std::lock_guard<std::mutex> lock(h->nets_mu);
    for (size_t i = 0; i < h->nets.size(); ++i)       // a long run re-plans whenever its batch grows: reuse the ids it gave back
        if (!h->nets[i]) { h->nets[i] = new Net(); *net = (int)i; return 0; }
    if (h->nets.size() >= I2V_MAX_NETS) return fail("more than %d backbones alive on one handle", I2V_MAX_NETS);
    h->nets.push_back(new Net());
    *net = (int)h->nets.size() - 1;
    return 0;
}

extern "C" int i2v_net_destroy(i2v_handle h, int net) {
    Net* n = get_net(h, net); if (!n) return 1;
    free_net(n);
    std::lock_guard<std::mutex> lock(h->nets_mu);
    h->nets[net] = nullptr;          // the id may be handed out again by i2v_net_create
    return 0;
}

extern "C" int i2v_net_add_buffer3d(i2v_handle h, int net, int C, int T, int H, int W, int* buf) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (n->planned) return fail("net already planned");
    if (C <= 0 || T <= 0 || H <= 0 || W <= 0 || !buf) return fail("bad buffer shape %dx%dx%dx%d", C, T, H, W);
    Buffer b{C, H, W}; b.T = T;
    n->bufs.push_back(b);
    *buf = (int)n->bufs.size() - 1;
    return 0;
}

extern "C" int i2v_net_add_buffer(i2v_handle h, int net, int C, int H, int W, int* buf) {
    return i2v_net_add_buffer3d(h, net, C, 1, H, W, buf);
}

extern "C" int i2v_net_add_tensor(i2v_handle h, int net, int buf, int c_off, int C, int post_relu, int* tensor) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (buf < 0 || buf >= (int)n->bufs.size()) return fail("bad buffer id %d", buf);
    if (c_off < 0 || C <= 0 || c_off + C > n->bufs[buf].C) return fail("tensor view outside buffer");
    n->tens.push_back(Tensor{buf, c_off, C, post_relu != 0});
    *tensor = (int)n->tens.size() - 1;
    return 0;
}

extern "C" int i2v_net_tensor_frames(i2v_handle h, int net, int tensor, int* T) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (tensor < 0 || tensor >= (int)n->tens.size() || !T) return fail("bad tensor id");
    *T = n->bufs[n->tens[tensor].buf].T;
    return 0;
}

extern "C" int i2v_net_set_input(i2v_handle h, int net, int tensor) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (tensor < 0 || tensor >= (int)n->tens.size()) return fail("bad tensor id");
    n->input = tensor;
    n->bufs[n->tens[tensor].buf].is_input = true;
    return 0;
}

extern "C" int i2v_net_set_relu_gain(i2v_handle h, int net, int tensor, float gain) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (n->planned) return fail("net already planned");
    if (tensor < 0 || tensor >= (int)n->tens.size()) return fail("bad tensor id");
    if (!n->tens[tensor].post_relu) return fail("i2v_net_set_relu_gain: tensor %d is not the output of a ReLU", tensor);
    if (!(gain > 0.f) || !isfinite(gain)) return fail("i2v_net_set_relu_gain: the gain must be positive and finite");
    n->tens[tensor].bwd_gain = gain;
    return 0;
}

extern "C" int i2v_net_add_conv3d(i2v_handle h, int net, const i2v_conv3d_desc* d, const float* weight,
                                  const float* scale, const float* shift) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (n->planned) return fail("net already planned");
    if (!d || !weight || !scale || !shift) return fail("i2v_net_add_conv: null argument");
    int nt = (int)n->tens.size();
    if (d->src < 0 || d->src >= nt || d->dst < 0 || d->dst >= nt || d->residual >= nt)
        return fail("i2v_net_add_conv: bad tensor id");
    const Tensor& S = n->tens[d->src]; const Tensor& D = n->tens[d->dst];
    const Buffer& sb = n->bufs[S.buf]; const Buffer& db = n->bufs[D.buf];
    if (S.C != d->cin || D.C != d->cout) return fail("conv channel mismatch");
    if (d->stride < 1 || d->kh < 1 || d->kw < 1 || d->pad < 0 || d->kt < 1 || d->stride_t < 1 || d->pad_t < 0 || d->dil_t < 1)
        return fail("bad conv geometry");
    int Ho = (sb.H + 2 * d->pad - d->kh) / d->stride + 1, Wo = (sb.W + 2 * d->pad - d->kw) / d->stride + 1;
    int To = (sb.T + 2 * d->pad_t - d->dil_t * (d->kt - 1) - 1) / d->stride_t + 1;
    if (Ho != db.H || Wo != db.W) return fail("conv output %dx%d does not match buffer %dx%d", Ho, Wo, db.H, db.W);
    if (To != db.T) return fail("conv output has %d frames per clip, buffer has %d", To, db.T);
    if (d->residual >= 0) {
        const Tensor& R = n->tens[d->residual]; const Buffer& rb = n->bufs[R.buf];
        if (R.C != d->cout || rb.H != db.H || rb.W != db.W || rb.T != db.T) return fail("residual shape mismatch");
    }
    Node nd; nd.type = 0; nd.cd = *d; memset(&nd.pd, 0, sizeof nd.pd);
    size_t per = (size_t)d->cin * d->kt * d->kh * d->kw;
    nd.w.resize((size_t)d->cout * per);
    for (int co = 0; co < d->cout; ++co)
        for (size_t i = 0; i < per; ++i) nd.w[co * per + i] = weight[co * per + i] * scale[co];
    nd.shift.assign(shift, shift + d->cout);
    n->nodes.push_back(std::move(nd));
    return 0;
}

extern "C" int i2v_net_add_conv(i2v_handle h, int net, const i2v_conv_desc* d, const float* weight,
                                const float* scale, const float* shift) {
    if (!d) return fail("i2v_net_add_conv: null argument");
    i2v_conv3d_desc q{d->src, d->dst, d->cin, d->cout, 1, d->kh, d->kw, 1, d->stride, 0, d->pad, 1, d->relu, d->residual};
    return i2v_net_add_conv3d(h, net, &q, weight, scale, shift);
}

extern "C" int i2v_net_add_conv_preact(i2v_handle h, int net, const i2v_conv_desc* d, const float* weight,
                                       const float* scale, const float* shift, const float* pre_scale,
                                       const float* pre_shift) {
    if (!d || !pre_scale || !pre_shift) return fail("i2v_net_add_conv_preact: null argument");
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->residual >= 0)
        return fail("pre-activation is supported on plain 1x1/stride-1 convolutions only");
    if (i2v_net_add_conv(h, net, d, weight, scale, shift)) return 1;
    Net* n = get_net(h, net);
    if (n->tens[d->src].post_relu) return fail("pre-activation input must not be a ReLU output");
    Node& nd = n->nodes.back();
    nd.pre_scale.assign(pre_scale, pre_scale + d->cin);
    nd.pre_shift.assign(pre_shift, pre_shift + d->cin);
    return 0;
}

extern "C" int i2v_net_add_maxpool3d(i2v_handle h, int net, const i2v_pool3d_desc* d) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (n->planned) return fail("net already planned");
    int nt = (int)n->tens.size();
    if (!d || d->src < 0 || d->src >= nt || d->dst < 0 || d->dst >= nt) return fail("bad pool desc");
    if (n->tens[d->src].C != n->tens[d->dst].C) return fail("pool channel mismatch");
    if (d->kt < 1 || d->k < 1 || d->stride_t < 1 || d->stride < 1 || d->pad_t < 0 || d->pad < 0) return fail("bad pool geometry");
    const Buffer& sb = n->bufs[n->tens[d->src].buf]; const Buffer& db = n->bufs[n->tens[d->dst].buf];
    if ((sb.T + 2 * d->pad_t - d->kt) / d->stride_t + 1 != db.T) return fail("pool output frames per clip do not match the buffer");
    {   // the destination plane must be what this window produces (floor, or ceil_mode's one extra row / column)
        const int ho = (sb.H + 2 * d->pad - d->k) / d->stride + 1, hc = (sb.H + 2 * d->pad - d->k + d->stride - 1) / d->stride + 1;
        const int wo = (sb.W + 2 * d->pad - d->k) / d->stride + 1, wc = (sb.W + 2 * d->pad - d->k + d->stride - 1) / d->stride + 1;
        if (sb.H + 2 * d->pad < d->k || sb.W + 2 * d->pad < d->k || db.H < ho || db.H > hc || db.W < wo || db.W > wc)
            return fail("pool output plane %dx%d does not match %dx%d pooled by k=%d stride=%d pad=%d", db.H, db.W, sb.H, sb.W, d->k, d->stride, d->pad);
    }
    Node nd; nd.type = 1; nd.pd = *d; memset(&nd.cd, 0, sizeof nd.cd);
    n->nodes.push_back(std::move(nd));
    return 0;
}

extern "C" int i2v_net_add_maxpool(i2v_handle h, int net, const i2v_pool_desc* d) {
    if (!d) return fail("bad pool desc");
    i2v_pool3d_desc q{d->src, d->dst, 1, d->k, 1, d->stride, 0, d->pad};
    return i2v_net_add_maxpool3d(h, net, &q);
}

extern "C" int i2v_net_add_avgpool(i2v_handle h, int net, const i2v_pool_desc* d) {
    if (i2v_net_add_maxpool(h, net, d)) return 1;
    Net* n = get_net(h, net);
    if (d->pad != 0) return fail("average pooling with padding is not supported");
    n->nodes.back().type = 2;
    return 0;
}

extern "C" int i2v_net_add_attention(i2v_handle h, int net, const i2v_attn_desc* d) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (n->planned) return fail("net already planned");
    const int nt = (int)n->tens.size();
    if (!d) return fail("bad attention desc");
    for (int t : {d->theta, d->phi, d->g, d->dst}) if (t < 0 || t >= nt) return fail("bad attention tensor");
    const Tensor& th = n->tens[d->theta]; const Tensor& ph = n->tens[d->phi]; const Tensor& gg = n->tens[d->g]; const Tensor& ds = n->tens[d->dst];
    if (th.C != ph.C || th.C != gg.C || th.C != ds.C) return fail("attention: theta / phi / g / dst channel counts differ");
    const Buffer& tb = n->bufs[th.buf]; const Buffer& pb = n->bufs[ph.buf]; const Buffer& gb = n->bufs[gg.buf]; const Buffer& db = n->bufs[ds.buf];
    if (tb.T != db.T || tb.H != db.H || tb.W != db.W) return fail("attention: dst must have theta's positions");
    if (pb.T != gb.T || pb.H != gb.H || pb.W != gb.W) return fail("attention: phi and g must have the same positions");
    if (th.post_relu || ph.post_relu || gg.post_relu) return fail("attention: theta / phi / g are linear embeddings (no ReLU)");
    if (!(d->scale > 0.f)) return fail("attention: scale must be positive");
    Node nd; nd.type = 3; nd.ad = *d; memset(&nd.cd, 0, sizeof nd.cd); memset(&nd.pd, 0, sizeof nd.pd);
    n->nodes.push_back(std::move(nd));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------------------------
static void conv_common(I2VConvParams& p, const Packed& P) {
    memset(&p, 0, sizeof p);
    p.wp = P.wp; p.wpc = P.wpc; p.ktab = P.ktab; p.K = P.K; p.Kpad = P.Kpad; p.tap_uniform = P.tap_uniform; p.Cd = P.Cd; p.Cdpad = P.Cdpad;
    p.wp3 = P.wp3; p.bf3 = P.wp3 ? 1 : 0;
    p.add0_stride = 1;
    p.blkt = 1; p.Tg = p.Ts = p.To = p.st = p.ost = 1; p.ot0 = 0; p.oct = 1;
    p.temporal = P.has_dt;      // conv_run adds the frame-mapping half of the condition
    p.quad = P.quad; p.quad_kw = P.quad_kw; p.quad_dw0 = P.quad_dw0;
    p.halo = P.halo;
    p.ig_tt = P.ig_tt; p.ig_th = P.ig_th; p.ig_tw = P.ig_tw; p.ig_p77 = P.ig_p77;
}

static bool overlaps(const Tensor& a, const Tensor& b) {
    return a.buf == b.buf && a.c_off < b.c_off + b.C && b.c_off < a.c_off + a.C;
}

namespace {

struct Planner {
    Net& n; bool dry; size_t off; size_t N;
    std::vector<int> left; std::vector<std::vector<Addend>> pending;
    std::vector<float*> hook_tmp;      // per hook: temp gradient buffer or null (direct)
    std::vector<View> galias; std::vector<char> has_alias;   // residual gradient that is just a view
    std::vector<char> accum;                                 // per buffer: gradient accumulates (dense blocks)
    std::string err;

    float* base() const { return dry ? (float*)nullptr : n.arena; }
    size_t nf(int T) const { return N / (size_t)n.Tin() * (size_t)T; }      // frames of a tensor with T frames per clip
    size_t carve(size_t floats) { size_t o = off; off = align_up(off + floats, 64); return o; }
    float* temp(size_t floats) { return base() + carve(floats); }

    View view(int t, bool grad) {
        if (grad && has_alias[t]) return galias[t];
        const Tensor& T = n.tens[t]; const Buffer& B = n.bufs[T.buf];
        View v; v.p = base() + (grad ? B.grad_off : B.act_off) + (size_t)T.c_off * B.H * B.W;
        v.nstride = (int64_t)B.C * B.H * B.W; v.C = T.C; v.H = B.H; v.W = B.W; v.T = B.T;
        return v;
    }
    void emit(std::vector<Launch>& L, const Launch& l) { if (!dry) L.push_back(l); }
    // gate rows of tensor t (null when its buffer keeps no gates)
    uint32_t* gate_rows(int t, int* stride) const {
        const Tensor& T = n.tens[t]; const Buffer& B = n.bufs[T.buf];
        if (!B.gated) return nullptr;
        *stride = B.gate_words;
        return (uint32_t*)(base() + B.gate_off) + (size_t)T.c_off * B.gate_words;
    }
    // the ReLU gate of tensor t for a launch that finalises its gradient: bits when available, else the fp32 activation
    void set_gate(I2VConvParams& p, int t) {
        int st = 0;
        if (uint32_t* g = gate_rows(t, &st)) { p.gate = g; p.gate_stride = st; p.gate_pix0 = 0; }
        else { View a = view(t, false); p.mask = a.p; p.mask_nstride = a.nstride; }
    }

    void emit_addmask(View out, const std::vector<Addend>& adds, int t) {
        Launch l; memset(&l.am, 0, sizeof l.am); l.kind = L_ADDMASK;
        l.am.out = out.p; l.am.out_nstride = out.nstride;
        for (size_t i = 0; i < adds.size() && i < 3; ++i) { l.am.a[i] = adds[i].p; l.am.a_nstride[i] = adds[i].nstride; }
        if (n.tens[t].post_relu) {
            int st = 0;
            if (uint32_t* g = gate_rows(t, &st)) { l.am.gate = g; l.am.gate_stride = st; }
            else { View a = view(t, false); l.am.mask = a.p; l.am.mask_nstride = a.nstride; }
        }
        l.am.N = 0; l.am.C = out.C; l.am.HW = out.H * out.W; l.T = out.T;
        emit(n.bwd, l);
    }

    // reduce pending list of tensor t to at most `keep` plain/compact addends
    bool squeeze_pending(int t, size_t keep) {
        auto& P = pending[t];
        while (P.size() > keep) {
            // fold the last two plain addends into a temp
            size_t a = P.size() - 1, b = P.size() - 2;
            if (P[a].stride != 1 || P[b].stride != 1) { err = "cannot fold compact addends"; return false; }
            View g = view(t, true);
            View tv = g; tv.p = temp(nf(g.T) * g.C * g.H * g.W); tv.nstride = (int64_t)g.C * g.H * g.W;
            Launch l; memset(&l.am, 0, sizeof l.am); l.kind = L_ADDMASK; l.T = g.T;
            l.am.out = tv.p; l.am.out_nstride = tv.nstride;
            l.am.a[0] = P[a].p; l.am.a_nstride[0] = P[a].nstride;
            l.am.a[1] = P[b].p; l.am.a_nstride[1] = P[b].nstride;
            l.am.C = g.C; l.am.HW = g.H * g.W;
            emit(n.bwd, l);
            P.pop_back(); P.pop_back();
            P.push_back(Addend{tv.p, tv.nstride, 1, g.H, g.W});
        }
        return true;
    }

    void conv_launches(const Node& nd, View dz, View out, bool raw, int t, bool compact) {
        const i2v_conv3d_desc& c = nd.cd;
        for (const Packed& P : nd.bwd) {
            if (compact && (P.ph || P.pw || P.pt)) continue;
            if (P.Hg <= 0 || P.Wg <= 0 || P.Tg <= 0) continue;
            Launch l; l.kind = L_CONV; conv_common(l.conv, P);
            l.node = (int)(&nd - n.nodes.data());
            I2VConvParams& p = l.conv;
            p.src = dz.p; p.src_nstride = dz.nstride; p.Hs = dz.H; p.Ws = dz.W; p.Cs = dz.C;
            p.Hg = P.Hg; p.Wg = P.Wg; p.sh = 1; p.sw = 1;
            p.Tg = P.Tg; p.Ts = dz.T; p.st = 1; l.T = P.Tg;
            p.dst = out.p; p.dst_nstride = out.nstride;
            if (compact) { p.Ho = P.Hg; p.Wo = P.Wg; p.osh = p.osw = 1; p.oh0 = p.ow0 = 0; p.To = P.Tg; }
            else {
                p.Ho = out.H; p.Wo = out.W; p.osh = p.osw = c.stride; p.oh0 = P.ph; p.ow0 = P.pw;
                p.To = out.T; p.ost = c.stride_t; p.ot0 = P.pt;
            }
            if (!raw) {
                for (const Addend& a : pending[t]) {
                    if (a.stride != 1 || p.add0 == nullptr) {
                        if (p.add0 != nullptr) { p.add1 = p.add0; p.add1_nstride = p.add0_nstride; }
                        p.add0 = a.p; p.add0_nstride = a.nstride; p.add0_stride = a.stride; p.add0_H = a.H; p.add0_W = a.W;
                    } else { p.add1 = a.p; p.add1_nstride = a.nstride; }
                }
                if (n.tens[t].post_relu) set_gate(p, t);
            }
            p.pointwise = (c.kt == 1 && c.stride_t == 1 && c.pad_t == 0 && c.kh == 1 && c.kw == 1 && c.stride == 1 &&
                           c.pad == 0 && (dz.H * dz.W) % 4 == 0 && !compact) ? 1 : 0;
            emit(n.bwd, l);
        }
    }
    bool is_hook(int t) const { for (int hk : n.hooks) if (hk == t) return true; return false; }
    static bool has_compact(const std::vector<Addend>& A) { for (auto& a : A) if (a.stride != 1) return true; return false; }

    bool contribute_conv(int t, const Node& nd, View dz) {
        left[t]--;
        View g = view(t, true);
        const i2v_conv3d_desc& c = nd.cd;
        if (left[t] > 0) {
            bool compact = (c.kt == 1 && c.stride_t == 1 && c.pad_t == 0 && c.kh == 1 && c.kw == 1 && c.stride > 1 && c.pad == 0);
            if (compact) {
                const Packed& P = nd.bwd[0];
                View tv; tv.C = g.C; tv.H = P.Hg; tv.W = P.Wg; tv.T = g.T; tv.nstride = (int64_t)g.C * P.Hg * P.Wg;
                tv.p = temp(nf(g.T) * tv.nstride);
                conv_launches(nd, dz, tv, true, t, true);
                pending[t].push_back(Addend{tv.p, tv.nstride, c.stride, P.Hg, P.Wg});
            } else {
                View tv = g; tv.nstride = (int64_t)g.C * g.H * g.W; tv.p = temp(nf(g.T) * tv.nstride);
                conv_launches(nd, dz, tv, true, t, false);
                pending[t].push_back(Addend{tv.p, tv.nstride, 1, g.H, g.W});
            }
            return true;
        }
        // final contributor: at most one compact + one plain, or two plain addends fit the epilogue
        size_t ncompact = 0; for (auto& a : pending[t]) if (a.stride != 1) ncompact++;
        if (ncompact > 1) { err = "more than one strided addend"; return false; }
        if (ncompact == 1) {
            // keep the compact one, fold plain ones down to a single addend
            std::vector<Addend> plain, comp;
            for (auto& a : pending[t]) (a.stride == 1 ? plain : comp).push_back(a);
            pending[t] = plain; if (!squeeze_pending(t, 1)) return false;
            pending[t].push_back(comp[0]);
        } else if (!squeeze_pending(t, 2)) return false;
        conv_launches(nd, dz, g, false, t, false);
        pending[t].clear();
        return true;
    }

    bool contribute_alias(int t, View dz) {
        if (left[t] == 1 && pending[t].empty() && !n.tens[t].post_relu && !is_hook(t)) {
            left[t] = 0; galias[t] = dz; has_alias[t] = 1;     // sole consumer, no gate: alias the view
            return true;
        }
        left[t]--;
        pending[t].push_back(Addend{dz.p, dz.nstride, 1, dz.H, dz.W});
        if (left[t] > 0) return true;
        if (has_compact(pending[t])) { err = "alias finaliser with strided addend"; return false; }
        if (!squeeze_pending(t, 3)) return false;
        emit_addmask(view(t, true), pending[t], t);
        pending[t].clear();
        return true;
    }

    bool run() {
        const int NT = (int)n.tens.size();
        left.assign(NT, 0); pending.assign(NT, {}); galias.assign(NT, View{}); has_alias.assign(NT, 0);
        // Buffers read through a pre-activation conv (DenseNet concatenation buffers) ACCUMULATE their
        // gradient: zeroed at the start of the backward pass, every reader adds into its view.  They stay
        // outside the single-finaliser protocol (`left` / `pending`) of all other tensors.
        accum.assign(n.bufs.size(), 0);
        for (const Node& nd : n.nodes) if (nd.type == 0 && nd.preact()) accum[n.tens[nd.cd.src].buf] = 1;
        for (const Node& nd : n.nodes) {
            if (nd.type == 0) {
                if (accum[n.tens[nd.cd.src].buf] && !nd.preact()) { err = "a dense (accumulating) buffer may only be read by pre-activation convs"; return false; }
                if (!nd.preact()) left[nd.cd.src]++;
                if (nd.cd.residual >= 0) left[nd.cd.residual]++;
            } else if (nd.type == 3) {
                for (int t : {nd.ad.theta, nd.ad.phi, nd.ad.g}) {
                    if (accum[n.tens[t].buf]) { err = "attention over a dense (accumulating) buffer is not supported"; return false; }
                    left[t]++;
                }
            } else {
                if (accum[n.tens[nd.pd.src].buf]) { err = "pooling directly from a dense (accumulating) buffer is not supported"; return false; }
                left[nd.pd.src]++;
            }
        }
        // A ReLU output that is only ever read through a wider concatenation view which is NOT declared post-ReLU
        // (SlowFast: max-pooled slow features ++ ReLU'd lateral features) is gated in place before its producer's
        // input-gradient runs; the covering view's finaliser cannot do it.
        std::vector<char> need_gate(NT, 0);
        for (int t = 0; t < NT; ++t) {
            if (!n.tens[t].post_relu || left[t] > 0 || is_hook(t)) continue;
            for (int u = 0; u < NT; ++u)
                if (u != t && left[u] > 0 && !n.tens[u].post_relu && overlaps(n.tens[u], n.tens[t])) need_gate[t] = 1;
        }
        int img_seen = 0;
        // ---------------- forward ----------------
        for (const Node& nd : n.nodes) {
            Launch l;
            if (nd.type == 0) {
                const i2v_conv3d_desc& c = nd.cd;
                l.kind = L_CONV; conv_common(l.conv, nd.fwd);
                l.node = (int)(&nd - n.nodes.data());
                I2VConvParams& p = l.conv;
                View d = view(c.dst, false);
                const Buffer& sb = n.bufs[n.tens[c.src].buf];
                if (c.src == n.input) { l.src_is_input = true; p.src = nullptr; p.src_nstride = (int64_t)sb.C * sb.H * sb.W; }
                else { View s = view(c.src, false); p.src = s.p; p.src_nstride = s.nstride; }
                p.Hs = sb.H; p.Ws = sb.W; p.Cs = c.cin; p.Hg = d.H; p.Wg = d.W; p.sh = p.sw = c.stride;
                p.dst = d.p; p.dst_nstride = d.nstride; p.Ho = d.H; p.Wo = d.W; p.osh = p.osw = 1;
                p.Tg = p.To = d.T; p.Ts = sb.T; p.st = c.stride_t; l.T = d.T;
                if (nd.fwd.tpair) {              // two output frames per grid frame (pack_fwd): class-packed epilogue, blk = 1
                    p.Tg = (d.T + 1) / 2; l.T = p.Tg; p.st = 2 * c.stride_t; p.ost = 2; p.ot0 = 0; p.blkt = 2; p.blk = 1;
                }
                p.shift = nd.shift_d; p.relu = c.relu;
                if (c.residual >= 0) { View r = view(c.residual, false); p.add0 = r.p; p.add0_nstride = r.nstride; p.add0_stride = 1; }
                p.pointwise = (c.kt == 1 && c.stride_t == 1 && c.pad_t == 0 && c.kh == 1 && c.kw == 1 && c.stride == 1 &&
                               c.pad == 0 && (sb.H * sb.W) % 4 == 0 && c.src != n.input) ? 1 : 0;
                if (nd.preact()) { p.pre_scale = nd.pre_scale_d; p.pre_shift = nd.pre_shift_d; }
                if (c.relu) { int st = 0; if (uint32_t* g = gate_rows(c.dst, &st)) { p.gate_out = g; p.gate_out_stride = st; p.gate_out_pix0 = 0; } }
                if (nd.fwd.quad) l.alg_flops_per_frame = 2.0 * d.H * d.W * c.cout * (double)c.cin * c.kt * c.kh * c.kw * d.T / p.Tg;   // per GRID frame
            } else if (nd.type == 3) {
                // S = scale * theta^T phi  ->  P = softmax rows  ->  y = g P^T   (P stays in the arena for the backward pass)
                View th = view(nd.ad.theta, false), ph = view(nd.ad.phi, false), gv = view(nd.ad.g, false), y = view(nd.ad.dst, false);
                const int M = th.T * th.H * th.W, Nn = ph.T * ph.H * ph.W;
                float* P = base() + nd.p_off;
                Launch a; a.kind = L_AGEMM; memset(&a.ag, 0, sizeof a.ag); a.T = th.T;
                a.ag.form = 1; a.ag.Cc = th.C; a.ag.M = M; a.ag.N = Nn; a.ag.scale = nd.ad.scale;
                a.ag.A = I2VActMat{th.p, th.nstride, th.T, th.H * th.W}; a.ag.B = I2VActMat{ph.p, ph.nstride, ph.T, ph.H * ph.W}; a.ag.D = P;
                emit(n.fwd, a);
                Launch sm; sm.kind = L_SOFTMAX; memset(&sm.sm, 0, sizeof sm.sm); sm.T = th.T;
                sm.sm.X = P; sm.sm.N = Nn; sm.sm.mode = 0; sm.sm_rows_per_clip = M;
                emit(n.fwd, sm);
                l.kind = L_AGEMM; memset(&l.ag, 0, sizeof l.ag); l.T = th.T;
                l.ag.form = 2; l.ag.Cc = th.C; l.ag.M = M; l.ag.N = Nn; l.ag.scale = 1.f;
                l.ag.A = I2VActMat{gv.p, gv.nstride, gv.T, gv.H * gv.W}; l.ag.Din = P;
                l.ag.Cact = y.p; l.ag.C_nstride = y.nstride; l.ag.C_T = y.T; l.ag.C_HW = y.H * y.W;
            } else {
                const i2v_pool3d_desc& q = nd.pd;
                const bool vid = q.kt != 1 || q.stride_t != 1 || q.pad_t != 0;
                if (vid && nd.type == 2) { err = "average pooling over time is not supported"; return false; }
                l.kind = nd.type == 2 ? L_AVGF : vid ? L_POOL3F : L_POOLF; memset(&l.pool, 0, sizeof l.pool);
                View s = view(q.src, false), d = view(q.dst, false);
                if (q.src == n.input) { err = "maxpool directly on the input is not supported"; return false; }
                l.pool.x = s.p; l.pool.x_nstride = s.nstride; l.pool.C = s.C; l.pool.Hs = s.H; l.pool.Ws = s.W;
                l.pool.y = d.p; l.pool.y_nstride = d.nstride; l.pool.Ho = d.H; l.pool.Wo = d.W;
                l.pool.k = q.k; l.pool.stride = q.stride; l.pool.pad = q.pad;
                l.pool.kt = q.kt; l.pool.stride_t = q.stride_t; l.pool.pad_t = q.pad_t; l.pool.Ts = s.T; l.pool.To = d.T;
                l.pool.idx = (uint8_t*)(base() + nd.idx_off);
                l.T = d.T;
            }
            emit(n.fwd, l);
        }
        // ---------------- hooks ----------------
        hook_tmp.assign(n.hooks.size(), nullptr);
        for (size_t hk = 0; hk < n.hooks.size(); ++hk) {
            int t = n.hooks[hk];
            bool consumed = false;
            for (const Node& nd : n.nodes) {
                int srcs[3] = {nd.src0(), nd.type == 0 ? nd.cd.residual : nd.type == 3 ? nd.ad.phi : -1, nd.type == 3 ? nd.ad.g : -1};
                for (int s : srcs) if (s >= 0 && overlaps(n.tens[s], n.tens[t])) consumed = true;
            }
            if (consumed || accum[n.tens[t].buf]) { View g = view(t, true); hook_tmp[hk] = temp(nf(g.T) * g.C * g.H * g.W); }
        }
        // ---------------- backward ----------------
        for (const Node& nd : n.nodes)
            if (nd.type == 0 && nd.cd.src == n.input && !nd.imgs.empty() && nd.imgs[0].skips) {     // a stem gradient that skips frames
                const Buffer& ib = n.bufs[n.tens[n.input].buf];
                Launch l; l.kind = L_MEMSET; l.ms_gx = true; l.ms_floats_per_frame = (size_t)ib.C * ib.H * ib.W; l.T = ib.T;
                emit(n.bwd, l);
                break;
            }
        for (size_t b = 0; b < n.bufs.size(); ++b)
            if (accum[b]) {
                Launch l; l.kind = L_MEMSET;
                l.ms_ptr = base() + n.bufs[b].grad_off; l.ms_floats_per_frame = (size_t)n.bufs[b].C * n.bufs[b].H * n.bufs[b].W;
                l.T = n.bufs[b].T;
                emit(n.bwd, l);
            }
        for (size_t hk = 0; hk < n.hooks.size(); ++hk)          // hook gradients of dense buffers: G += H right away
            if (accum[n.tens[n.hooks[hk]].buf]) {
                const int t = n.hooks[hk];
                View g = view(t, true);
                std::vector<Addend> adds = {Addend{g.p, g.nstride, 1, g.H, g.W},
                                            Addend{hook_tmp[hk], (int64_t)g.C * g.H * g.W, 1, g.H, g.W}};
                emit_addmask(g, adds, t);
            }
        for (int i = (int)n.nodes.size() - 1; i >= 0; --i) {
            const Node& nd = n.nodes[i];
            int dst = nd.dst0();
            // a hook gradient kept in a side buffer joins the gradient of every node output the hooked view COVERS:
            // the hooked tensor itself, or -- a hooked concatenation (SqueezeNet Fire output = expand1x1 ++ expand3x3,
            // TPAMI_attack.py:195-197) -- each branch's channel slice of it
            for (size_t hk = 0; hk < n.hooks.size(); ++hk) {
                const Tensor& HT = n.tens[n.hooks[hk]]; const Tensor& DT = n.tens[dst];
                const bool covers = HT.buf == DT.buf && HT.c_off <= DT.c_off && DT.c_off + DT.C <= HT.c_off + HT.C;
                if (covers && hook_tmp[hk] && !accum[DT.buf]) {
                    View g = view(dst, true);
                    const int64_t hD = (int64_t)HT.C * g.H * g.W;
                    std::vector<Addend> adds = {Addend{g.p, g.nstride, 1, g.H, g.W},
                                                Addend{hook_tmp[hk] + (size_t)(DT.c_off - HT.c_off) * g.H * g.W, hD, 1, g.H, g.W}};
                    emit_addmask(g, adds, dst);
                }
            }
            View dz = view(dst, true);
            if (need_gate[dst]) emit_addmask(dz, {Addend{dz.p, dz.nstride, 1, dz.H, dz.W}}, dst);
            // a backward gain on this node's ReLU (i2v_net_set_relu_gain): G(dst) is complete and gated here -- every consumer and
            // hook has contributed, the finaliser applied the gate -- so the gain is one in-place pass in front of the node's own
            // input-gradient work (the gate is 0 or 1: gain * gate * g, whichever is applied first)
            if (n.tens[dst].bwd_gain != 1.f) {
                if (accum[n.tens[dst].buf] || has_alias[dst]) { err = "a ReLU gain on an accumulating or aliased gradient view is not planned"; return false; }
                Launch l; memset(&l.am, 0, sizeof l.am); l.kind = L_ADDMASK; l.T = dz.T;
                l.am.out = dz.p; l.am.out_nstride = dz.nstride; l.am.a[0] = dz.p; l.am.a_nstride[0] = dz.nstride;
                l.am.C = dz.C; l.am.HW = dz.H * dz.W; l.am.gain = n.tens[dst].bwd_gain;
                emit(n.bwd, l);
            }
            if (nd.type == 0) {
                const i2v_conv3d_desc& c = nd.cd;
                if (c.residual >= 0 && !contribute_alias(c.residual, dz)) return false;
                if (c.src == n.input) {
                    const bool acc_node = img_seen++ > 0;          // (a node's temporal classes write disjoint frames: one flag for all of them)
                    for (const Node::ImgGrad& ig : nd.imgs) {
                    Launch l; l.kind = L_IMGGRAD; conv_common(l.conv, ig.P);
                    l.img_accumulate = acc_node;
                    const Buffer& ib = n.bufs[n.tens[n.input].buf];
                    I2VConvParams& p = l.conv;
                    p.src = dz.p; p.src_nstride = dz.nstride; p.Hs = dz.H; p.Ws = dz.W; p.Cs = dz.C;
                    p.Hg = ig.P.Hg; p.Wg = ig.P.Wg; p.sh = p.sw = ig.sh;
                    p.dst = nullptr; p.dst_nstride = (int64_t)ib.C * ib.H * ib.W; p.Ho = ib.H; p.Wo = ib.W;
                    p.osh = p.osw = ig.blk; p.blk = ig.blk;
                    p.blkt = ig.blkt; p.Tg = ig.P.Tg; p.Ts = dz.T; p.st = ig.st; p.To = ib.T; p.ost = ig.ost; p.ot0 = ig.ot0; p.oct = ig.oct; l.T = ig.P.Tg;
                    l.alg_flops_per_frame = ig.flop_share * 2.0 * dz.T * dz.H * dz.W * c.cout * c.cin * c.kt * c.kh * c.kw / ig.P.Tg;   // per grid frame
                    emit(n.bwd, l);
                    }
                } else if (nd.preact()) {
                    // G(view) += W'^T dz gated by the pre-activation sign; W' carries the BN scale per input channel
                    View g = view(c.src, true), x = view(c.src, false);
                    Launch l; l.kind = L_CONV; conv_common(l.conv, nd.bwd[0]);
                    I2VConvParams& p = l.conv;
                    p.src = dz.p; p.src_nstride = dz.nstride; p.Hs = dz.H; p.Ws = dz.W; p.Cs = dz.C;
                    p.Hg = g.H; p.Wg = g.W; p.sh = p.sw = 1;
                    p.dst = g.p; p.dst_nstride = g.nstride; p.Ho = g.H; p.Wo = g.W; p.osh = p.osw = 1;
                    p.Tg = p.Ts = p.To = g.T; l.T = g.T;
                    p.add1 = g.p; p.add1_nstride = g.nstride;
                    p.mask = x.p; p.mask_nstride = x.nstride; p.gate_scale = nd.pre_scale_d; p.gate_shift = nd.pre_shift_d;
                    p.pointwise = ((dz.H * dz.W) % 4 == 0) ? 1 : 0;
                    emit(n.bwd, l);
                } else if (!contribute_conv(c.src, nd, dz)) return false;
            } else if (nd.type == 3) {
                // dY = dz.  dP = dY^T g;  dg = dY P;  dS = P o (dP - rowsum(dP o P));  dtheta = phi dS^T;  dphi = theta dS
                for (int t : {nd.ad.theta, nd.ad.phi, nd.ad.g})
                    if (left[t] != 1 || !pending[t].empty() || is_hook(t)) { err = "attention operands must have the attention node as their only consumer"; return false; }
                View th = view(nd.ad.theta, false), ph = view(nd.ad.phi, false), gv = view(nd.ad.g, false);
                View dth = view(nd.ad.theta, true), dph = view(nd.ad.phi, true), dgv = view(nd.ad.g, true);
                const int M = th.T * th.H * th.W, Nn = ph.T * ph.H * ph.W;
                float* P = base() + nd.p_off;
                float* dP = temp(nf(th.T) / th.T * (size_t)M * Nn);
                auto act = [](const View& v) { return I2VActMat{v.p, v.nstride, v.T, v.H * v.W}; };
                auto gemm = [&](int form, I2VActMat A, const I2VActMat* B, float* D, const float* Din, const View* out) {
                    Launch l; l.kind = L_AGEMM; memset(&l.ag, 0, sizeof l.ag); l.T = th.T;
                    l.ag.form = form; l.ag.Cc = th.C; l.ag.M = M; l.ag.N = Nn; l.ag.scale = form == 1 ? nd.ad.scale : 1.f; l.ag.A = A;
                    if (B) l.ag.B = *B;
                    l.ag.D = D; l.ag.Din = Din;
                    if (out) { l.ag.Cact = out->p; l.ag.C_nstride = out->nstride; l.ag.C_T = out->T; l.ag.C_HW = out->H * out->W; }
                    // few output tiles per clip under a long reduction (dg, dphi: Cc x N outputs summed over the M positions): cut K.
                    // The cut depends on the clip's own shape only, so a clip's result does not depend on the batch it is in.
                    const int cols = form == 2 ? M : Nn, K = form == 2 ? Nn : M;
                    const int tiles = ((th.C + 63) / 64) * ((cols + 63) / 64);
                    if (form != 1 && tiles <= 128 && K >= 512) {
                        l.ag.ksplit = std::min(8, K / 256);
                        l.ag.part = temp(nf(th.T) / th.T * (size_t)l.ag.ksplit * th.C * cols);
                    }
                    emit(n.bwd, l);
                };
                const I2VActMat gA = act(gv);
                // (the softmax backward works on d(scale * theta^T phi): the scale reaches dtheta / dphi through dS below)
                { Launch l; l.kind = L_AGEMM; memset(&l.ag, 0, sizeof l.ag); l.T = th.T; l.ag.form = 1; l.ag.Cc = th.C; l.ag.M = M; l.ag.N = Nn;
                  l.ag.scale = 1.f; l.ag.A = act(dz); l.ag.B = gA; l.ag.D = dP; emit(n.bwd, l); }
                gemm(3, act(dz), nullptr, nullptr, P, &dgv);
                { Launch l; l.kind = L_SOFTMAX; memset(&l.sm, 0, sizeof l.sm); l.T = th.T; l.sm.X = dP; l.sm.P = P; l.sm.N = Nn; l.sm.mode = 1;
                  l.sm_rows_per_clip = M; emit(n.bwd, l); }
                if (nd.ad.scale != 1.f) { err = "attention: only scale 1 is planned (gluoncv's gaussian non-local block has none)"; return false; }
                gemm(2, act(ph), nullptr, nullptr, dP, &dth);
                gemm(3, act(th), nullptr, nullptr, dP, &dph);
                left[nd.ad.theta] = left[nd.ad.phi] = left[nd.ad.g] = 0;
            } else {
                const i2v_pool3d_desc& q = nd.pd;
                const bool vid = q.kt != 1 || q.stride_t != 1 || q.pad_t != 0;
                left[q.src]--;
                const bool shared = left[q.src] > 0 || !pending[q.src].empty();      // other consumers contribute to this gradient too
                Launch l; l.kind = nd.type == 2 ? L_AVGB : vid ? L_POOL3B : L_POOLB; memset(&l.pool, 0, sizeof l.pool);
                View x = view(q.src, false), gx = view(q.src, true);
                if (shared) {       // (non-local block: x feeds theta, the 1x2x2 max-pool in front of phi / g, and the residual)
                    if (left[q.src] == 0) { err = "a max-pool must not be the last contributor to a shared gradient (order the graph's nodes so that a convolution is)"; return false; }
                    gx.nstride = (int64_t)gx.C * gx.H * gx.W; gx.p = temp(nf(gx.T) * gx.nstride);
                    pending[q.src].push_back(Addend{gx.p, gx.nstride, 1, gx.H, gx.W});
                }
                l.pool.x = x.p; l.pool.x_nstride = x.nstride; l.pool.C = x.C; l.pool.Hs = x.H; l.pool.Ws = x.W;
                l.pool.y = dz.p; l.pool.y_nstride = dz.nstride; l.pool.Ho = dz.H; l.pool.Wo = dz.W;
                l.pool.gx = gx.p; l.pool.gx_nstride = gx.nstride;
                l.pool.k = q.k; l.pool.stride = q.stride; l.pool.pad = q.pad;
                l.pool.mask_relu = (n.tens[q.src].post_relu && !shared) ? 1 : 0;        // (shared: the finaliser applies the gate)
                if (l.pool.mask_relu && nd.type == 1) { View ya = view(q.dst, false); l.pool.yact = ya.p; l.pool.yact_nstride = ya.nstride; }
                l.pool.kt = q.kt; l.pool.stride_t = q.stride_t; l.pool.pad_t = q.pad_t; l.pool.Ts = x.T; l.pool.To = dz.T;
                l.pool.idx = (uint8_t*)(base() + nd.idx_off);
                l.T = dz.T;
                emit(n.bwd, l);
            }
        }
        return true;
    }
};

}  // namespace

static int autotune(Net& n);
static void mark_fusable(Net& n);
static void mark_overlap(Net& n);

extern "C" int i2v_net_plan(i2v_handle h, int net, const int* hook_tensors, int n_hooks, int max_frames) {
    Net* np = get_net(h, net); if (!np) return 1;
    Net& n = *np;
    if (n.planned) return fail("net already planned");
    if (n.input < 0) return fail("input tensor not set");
    if (n_hooks <= 0 || max_frames <= 0) return fail("need >=1 hook and >=1 frame");
    // the split-bf16 loop lives in the EXPERIMENTAL build only (i2v_conv_exp.hip): asking the product library for it is an error, not
    // a silent fp32 run under another name
    if (math_bf16x3() && strncmp(be_name(), "hip", 3) == 0 && be_stat("experimental") != 1)
        return fail("I2V_MATH=bf16x3 needs a library built with -DI2V_EXPERIMENTAL (python __graft_entry__.py --experimental, then I2V_LIB=...)");
    for (int i = 0; i < n_hooks; ++i)
        if (hook_tensors[i] < 0 || hook_tensors[i] >= (int)n.tens.size()) return fail("bad hook tensor");
    n.hooks.assign(hook_tensors, hook_tensors + n_hooks);
    n.maxN = max_frames;
    const size_t N = (size_t)max_frames;
    const int Tin = n.Tin();
    if (max_frames % Tin) return fail("max_frames=%d is not a multiple of the input's %d frames per clip", max_frames, Tin);

    for (Node& nd : n.nodes) {
        if (nd.type != 0) continue;
        if (upload(n, nd.shift, &nd.shift_d)) return 1;
        if (pack_fwd(n, nd)) return 1;
        if (nd.preact()) {          // operand-side affine, padded to Kpad with zeros (relu(0*x+0) = 0 for the K tail)
            std::vector<float> ps(nd.fwd.Kpad > nd.fwd.Cdpad ? nd.fwd.Kpad : nd.fwd.Cdpad, 0.f), pt(ps.size(), 0.f);
            for (int c = 0; c < nd.cd.cin; ++c) { ps[c] = nd.pre_scale[c]; pt[c] = nd.pre_shift[c]; }
            if (upload(n, ps, &nd.pre_scale_d) || upload(n, pt, &nd.pre_shift_d)) return 1;
        }
        if (nd.cd.src == n.input) { if (pack_img(n, nd)) return 1; }
        else if (pack_bwd(n, nd)) return 1;
    }
    size_t off = 64;            // 256 bytes of slack in front of the first tensor (and behind the last, below): the quad-row
                                // staging of conv_igemm (MODE 4) reads a few pixels past either end of a source view
    for (const Node& nd : n.nodes) if (nd.type == 0 && nd.cd.src == n.input && nd.fwd.quad) n.stage_input = true;
    if (n.stage_input) {        // ... which the caller's frame tensor cannot promise: forward() copies it in here first
        const Buffer& ib = n.bufs[n.tens[n.input].buf];
        n.in_stage_off = off; off = align_up(off + N * ib.C * ib.H * ib.W, 64) + 64;
    }
    for (Buffer& b : n.bufs) {
        if (b.is_input) continue;
        size_t sz = N / Tin * b.T * b.C * b.H * b.W;
        b.act_off = off; off = align_up(off + sz, 64);
        b.grad_off = off; off = align_up(off + sz, 64);
    }
    // 1-bit ReLU gates (I2VConvParams::gate*): every buffer holding the output of a ReLU convolution gets one bit per
    // element, rows per channel; the input-gradient pass gates with these instead of re-reading fp32 activations
    const char* gates_env = getenv("I2V_GATES");           // developer knob: I2V_GATES=0 keeps the fp32-activation gates
    const bool use_gates = !(gates_env && gates_env[0] == '0');
    for (const Node& nd : n.nodes)
        if (use_gates && nd.type == 0 && nd.cd.relu && !nd.fwd.tpair) n.bufs[n.tens[nd.cd.dst].buf].gated = true;   // (the class-packed epilogue writes no gate words)
    for (Buffer& b : n.bufs) {
        if (!b.gated) continue;
        const size_t pix = N / Tin * b.T * b.H * b.W;
        if (pix + 64 >= (1ull << 31)) return fail("gate rows of more than 2^31 bits are not supported");
        b.gate_words = (int)((pix + 31) / 32 + 2);
        b.gate_off = off; off = align_up(off + (size_t)b.C * b.gate_words, 64);
    }
    for (Node& nd : n.nodes)
        if (nd.type == 1) {
            const Buffer& db = n.bufs[n.tens[nd.pd.dst].buf];
            nd.idx_off = off; off = align_up(off + (N / Tin * db.T * n.tens[nd.pd.dst].C * db.H * db.W + 3) / 4, 64);
        }
    for (Node& nd : n.nodes)
        if (nd.type == 3) {
            const Buffer& tb = n.bufs[n.tens[nd.ad.theta].buf]; const Buffer& pb = n.bufs[n.tens[nd.ad.phi].buf];
            nd.p_off = off; off = align_up(off + N / Tin * ((size_t)tb.T * tb.H * tb.W) * ((size_t)pb.T * pb.H * pb.W), 64);
        }
    Planner dry{n, true, off, N};
    if (!dry.run()) return fail("plan: %s", dry.err.c_str());
    n.arena_floats = dry.off + 64;
    n.arena = (float*)be_malloc(n.arena_floats * sizeof(float));
    if (!n.arena) return fail("arena allocation of %zu bytes failed", n.arena_floats * sizeof(float));
    CHECK_BE(be_memset0(n.arena, n.arena_floats * sizeof(float), nullptr));
    Planner real{n, false, off, N};
    if (!real.run()) return fail("plan: %s", real.err.c_str());
    n.hook_tmp = real.hook_tmp;
    mark_fusable(n);
    if (autotune(n)) return 1;
    mark_overlap(n);
    // planning works on the null stream (uploads, arena clears, tuning probes); the caller may execute the net on any
    // stream, including non-blocking ones that do not order against it
    CHECK_BE(be_stream_sync(nullptr));
    n.planned = true;
    return 0;
}

extern "C" int i2v_net_fusion_info(i2v_handle h, int net, int32_t out[4]) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (!n->planned || !out) return fail("i2v_net_fusion_info: net not planned");
    out[0] = out[1] = out[2] = out[3] = 0;
    int k = 0;
    for (const std::vector<Launch>* L : {&n->fwd, &n->bwd}) {
        for (const Launch& l : *L) { out[k] += l.fuse_ok ? 1 : 0; out[2 + k] += (l.fuse_ok && l.fuse_b[0]) ? 1 : 0; }
        ++k;
    }
    return 0;
}

extern "C" size_t i2v_net_workspace_bytes(i2v_handle h, int net) {
    Net* n = get_net(h, net); if (!n) return 0;
    return n->arena_floats * sizeof(float) + n->weight_bytes;
}

// ---------------------------------------------------------------------------------------------
// execution
// ---------------------------------------------------------------------------------------------
// A launch's parameters as the kernel sees them: the caller's input / gradient pointers patched in, the dense-epilogue and temporal
// flags derived (shared by the plain launch and the fused pair)
static I2VConvParams conv_prep(const Launch& l, const float* x, float* gx, int accumulate) {
    I2VConvParams p = l.conv;
    if (l.src_is_input) p.src = x;
    if (l.kind == L_IMGGRAD) { p.dst = gx; if (accumulate || l.img_accumulate) { p.add1 = gx; p.add1_nstride = p.dst_nstride; } }
    p.vec_epilogue = (p.blk <= 1 && p.osh == 1 && p.osw == 1 && p.oh0 == 0 && p.ow0 == 0 && p.Hg == p.Ho &&
                      p.Wg == p.Wo && p.Tg == p.To && p.ost == 1 && p.ot0 == 0 &&
                      (p.Ho * p.Wo) % 4 == 0 && p.dst_nstride % 4 == 0 &&
                      (p.add0_stride == 1 || (p.add0_stride == 2 && p.Wo % 4 == 0 && p.add0_W * 2 == p.Wo &&
                                              p.add0_W % 2 == 0 && (p.add0_H * p.add0_W) % 2 == 0)) &&
                      p.add0_nstride % 4 == 0 && p.add1_nstride % 4 == 0 && p.mask_nstride % 4 == 0 &&
                      (((uintptr_t)p.dst | (uintptr_t)p.add0 | (uintptr_t)p.add1 | (uintptr_t)p.mask) & 15) == 0)
                         ? 1 : 0;
    if (!(p.Tg == p.Ts && p.Ts == p.To && p.st == 1 && p.ost == 1 && p.ot0 == 0 && p.blkt == 1)) p.temporal = 1;
    return p;
}

// The fused pair: `a` (3x3) and `b` (the pointwise convolution over a's output) as ONE launch; only for pairs mark_fusable admitted.
// A pair whose source would have to be sliced (>= 2 GiB spans) is not fused (the caller falls back to two launches).
static bool fused_fits(const Launch& a, const Launch& b, int frames) {
    for (const Launch* l : {&a, &b}) {
        const I2VConvParams& p = l->conv;
        const int64_t span = ((int64_t)frames * p.Ts / p.Tg - 1) * p.src_nstride * 4 + (int64_t)p.Cs * p.Hs * p.Ws * 4;
        if (span >= (1ll << 31)) return false;
    }
    return true;
}
static int fused_run(const Launch& a, const Launch& b, int frames, const float* x, int halo, i2v_stream_t s) {
    I2VConvParams pa = conv_prep(a, x, nullptr, 0), pb = conv_prep(b, x, nullptr, 0);
    for (I2VConvParams* p : {&pa, &pb}) {
        p->N = frames;
        p->src_span_bytes = (int32_t)(((int64_t)frames * p->Ts / p->Tg - 1) * p->src_nstride * 4 + (int64_t)p->Cs * p->Hs * p->Ws * 4);
    }
    CHECK_BE(k_conv_fused(pa, pb, halo, s));
    return 0;
}

// The fused fast-pathway block: launches L[li] .. L[li + fb_ok - 1] as ONE kernel.  Returns 0 done, 1 error, 2 not eligible at this frame
// count (the caller runs the separate launches).
static int fast_run(const std::vector<Launch>& L, size_t li, int frames, const float* x, i2v_stream_t s) {
    const int g = L[li].fb_ok;
    I2VConvParams q[4];
    for (int j = 0; j < g; ++j) { q[j] = conv_prep(L[li + j], x, nullptr, 0); q[j].N = frames; }
    const I2VConvParams* c = g == 3 ? &q[2] : g == 4 ? &q[3] : nullptr;
    const I2VConvParams* d = g == 4 ? &q[2] : nullptr;
    if (i2v_fastblock_rows(q[0], q[1], c, d) <= 0) return 2;
    if (k_fastblock(q[0], q[1], c, d, s)) { fail("k_fastblock: %s", be_error() ? be_error() : "backend error"); return 1; }
    return 0;
}

// One convolution launch over `frames` grid frames (= clips * Tg), possibly sliced over whole clips: 32-bit
// buffer offsets keep a launch's source span < 2 GiB
static int conv_run(const Launch& l, int frames, const float* x, float* gx, int accumulate, i2v_stream_t s) {
    I2VConvParams p = conv_prep(l, x, gx, accumulate);
    const int clips = frames / p.Tg;
    const int64_t plane_bytes = (int64_t)p.Cs * p.Hs * p.Ws * 4, stride_bytes = p.src_nstride * 4;
    const int64_t clip_bytes = (int64_t)(p.Ts - 1) * stride_bytes + plane_bytes;        // span of one clip's source frames
    int64_t per = clip_bytes >= (1ll << 31) ? 0 : 1 + ((1ll << 31) - 1 - clip_bytes) / (stride_bytes * p.Ts);
    if (per < 1) return fail("one clip of a convolution input exceeds 2 GiB");
    if (per < clips && (p.gate || p.gate_out)) {        // a slice must start on a 32-bit boundary of the gate rows
        per -= per % 32;
        if (per < 1) return fail("a sliced convolution launch cannot keep its gate rows word-aligned");
    }
    for (int c0 = 0; c0 < clips; c0 += (int)per) {
        I2VConvParams q = p;
        const int nc = clips - c0 < per ? clips - c0 : (int)per;
        q.N = nc * p.Tg;
        q.src += (int64_t)c0 * p.Ts * p.src_nstride; q.dst += (int64_t)c0 * p.To * p.dst_nstride;
        if (q.add0) q.add0 += (int64_t)c0 * p.To * p.add0_nstride;
        if (q.add1) q.add1 += (int64_t)c0 * p.To * p.add1_nstride;
        if (q.mask) q.mask += (int64_t)c0 * p.To * p.mask_nstride;
        q.gate_pix0 = p.gate_pix0 + (int32_t)((int64_t)c0 * p.To * p.Ho * p.Wo);
        q.gate_out_pix0 = p.gate_out_pix0 + (int32_t)((int64_t)c0 * p.To * p.Ho * p.Wo);
        q.src_span_bytes = (int32_t)((int64_t)(nc * p.Ts - 1) * stride_bytes + plane_bytes);
        CHECK_BE(k_conv(q, s));
    }
    return 0;
}

// Fused pairs (k_conv_fused): launch i a 3x3 convolution, launch i + 1 the pointwise convolution that reads its output -- a
// bottleneck's conv2 -> conv3 in the forward list, the input gradients of conv2 -> conv1 in the backward list.  The fused kernel
// never stores the intermediate, so the pair qualifies only if NOTHING else touches that memory: no other launch of either list
// (as source, addend, mask, destination -- which also rules out I2V_GATES=0, whose backward pass reads fp32 activations, and the
// in-place ReLU gain of i2v_net_set_relu_gain), no hook, and not the network input.  Conservative: any overlap of address ranges counts.
static void mark_fusable(Net& n) {
    const int64_t frames = n.maxN, clips = n.maxN / n.Tin();
    typedef std::pair<const float*, const float*> Range;
    // extent of an operand: `fr` frames of stride `ns`, the last one `plane` floats long (a view of a wider buffer ends with its own
    // channels, not with the frame stride); null operands have no extent
    auto rng = [](const void* q, int64_t fr, int64_t ns, int64_t plane) {
        const float* f = (const float*)q;
        return f ? Range(f, f + (fr > 0 ? (fr - 1) * ns : 0) + std::max<int64_t>(plane, 1)) : Range(nullptr, nullptr);
    };
    auto meet = [](const Range& a, const Range& b) { return a.first && b.first && a.first < b.second && b.first < a.second; };
    // every memory range a launch reads or writes, with the launch's OWN frame count; `skip_src` / `skip_dst` leave out the operand
    // that legitimately is the intermediate (b's source, a's destination)
    auto touches = [&](const Launch& c, const Range& w, bool skip_src, bool skip_dst) {
        if (c.kind == L_CONV || c.kind == L_IMGGRAD) {
            const I2VConvParams& q = c.conv;
            const int64_t fs = clips * std::max(1, q.Ts), fo = clips * std::max(1, q.To), dplane = (int64_t)(q.blk > 1 ? q.Cd / (q.blk * q.blk) : q.Cd) * q.Ho * q.Wo;
            return (!skip_src && meet(w, rng(q.src, fs, q.src_nstride, (int64_t)q.Cs * q.Hs * q.Ws))) ||
                   meet(w, rng(q.add0, fo, q.add0_nstride, dplane)) || meet(w, rng(q.add1, fo, q.add1_nstride, dplane)) ||
                   meet(w, rng(q.mask, fo, q.mask_nstride, dplane)) || (!skip_dst && meet(w, rng(q.dst, fo, q.dst_nstride, dplane)));
        }
        if (c.kind == L_ADDMASK) {
            const int64_t fr = clips * std::max(1, c.T), pl = (int64_t)c.am.C * c.am.HW;
            return meet(w, rng(c.am.out, fr, c.am.out_nstride, pl)) || meet(w, rng(c.am.a[0], fr, c.am.a_nstride[0], pl)) ||
                   meet(w, rng(c.am.a[1], fr, c.am.a_nstride[1], pl)) || meet(w, rng(c.am.a[2], fr, c.am.a_nstride[2], pl)) ||
                   meet(w, rng(c.am.mask, fr, c.am.mask_nstride, pl));
        }
        if (c.kind == L_MEMSET) return meet(w, rng(c.ms_ptr, 1, 0, (int64_t)c.ms_floats_per_frame * clips * std::max(1, c.T)));
        if (c.kind == L_AGEMM || c.kind == L_SOFTMAX) return true;     // attention launches address whole matrices: not analysed, such nets are not fused
        const I2VPoolParams& q = c.pool;
        const int64_t fr = clips * std::max(std::max(1, c.T), std::max(q.Ts, q.To)), pin = (int64_t)q.C * q.Hs * q.Ws, pout = (int64_t)q.C * q.Ho * q.Wo;
        return meet(w, rng(q.x, fr, q.x_nstride, pin)) || meet(w, rng(q.gx, fr, q.gx_nstride, pin)) || meet(w, rng(q.y, fr, q.y_nstride, pout)) ||
               meet(w, rng(q.yact, fr, q.yact_nstride, pout));
    };
    std::vector<Range> hooked;
    for (size_t h = 0; h < n.hooks.size(); ++h) {
        View a = view_of(n, n.hooks[h], false), g = view_of(n, n.hooks[h], true);
        hooked.push_back(rng(a.p, frames, a.nstride, (int64_t)a.C * a.H * a.W));
        hooked.push_back(rng(g.p, frames, g.nstride, (int64_t)g.C * g.H * g.W));
    }
    auto dst_of = [&](const Launch& l) {
        return rng(l.conv.dst, clips * std::max(1, l.conv.To), l.conv.dst_nstride, (int64_t)l.conv.Cd * l.conv.Ho * l.conv.Wo);
    };
    // A first bottleneck runs its shortcut convolution between conv2 and conv3 (conv3 adds it): where that launch and the 3x3 are
    // independent of each other, they swap places so that the pair becomes adjacent.
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
        for (size_t i = 0; i + 2 < L->size(); ++i) {
            Launch& a = (*L)[i]; Launch& c = (*L)[i + 1]; Launch& b = (*L)[i + 2];
            if (a.kind != L_CONV || c.kind != L_CONV || b.kind != L_CONV || a.src_is_input || c.src_is_input || !k_conv_fusable(a.conv, b.conv)) continue;
            if (k_conv_fusable(a.conv, c.conv)) continue;
            const I2VConvParams& pa = a.conv; const I2VConvParams& pc = c.conv;
            const bool dep = touches(c, dst_of(a), false, false) || touches(a, dst_of(c), false, false) ||
                             (pa.gate_out && (pa.gate_out == pc.gate || pa.gate_out == pc.gate_out)) || (pc.gate_out && pc.gate_out == pa.gate);
            if (!dep) std::swap(a, c);
        }
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
        for (size_t i = 0; i + 1 < L->size(); ++i) {
            Launch& a = (*L)[i]; Launch& b = (*L)[i + 1];
            a.fuse_ok = 0;
            if (a.kind != L_CONV || b.kind != L_CONV || a.src_is_input || b.src_is_input || a.T != b.T) continue;
            const int ok = k_conv_fusable(a.conv, b.conv);
            if (!ok) continue;
            const Range w = dst_of(a);
            bool other = false;
            for (auto& hk : hooked) other |= meet(hk, w);
            for (std::vector<Launch>* M : {&n.fwd, &n.bwd})
                for (size_t j = 0; j < M->size() && !other; ++j)
                    other |= touches((*M)[j], w, M == L && j == i + 1, M == L && j == i);
            if (!other) a.fuse_ok = ok;
            if (getenv("I2V_FUSE_DEBUG"))
                fprintf(stderr, "[i2v fuse] %s pair %zu: 3x3 Cd=%d K=%d %dx%d -> 1x1 Cd=%d K=%d: eligible=%d other_reader=%d\n", L == &n.fwd ? "fwd" : "bwd", i,
                        a.conv.Cd, a.conv.K, a.conv.Hg, a.conv.Wg, b.conv.Cd, b.conv.K, ok, (int)other);
        }
    // ---- fused fast-pathway blocks (k_fastblock): groups of consecutive launches, found from the graph nodes behind them ----
    {
        const bool fb_off = [] { const char* e = getenv("I2V_FASTBLOCK"); return e && e[0] == '0'; }();      // (read per plan)
        auto cd_of = [&](const Launch& l) -> const i2v_conv3d_desc* { return (l.kind == L_CONV && l.node >= 0 && n.nodes[l.node].type == 0 && !n.nodes[l.node].preact()) ? &n.nodes[l.node].cd : nullptr; };
        auto unit = [](const i2v_conv3d_desc& c) { return c.stride == 1 && c.stride_t == 1 && c.dil_t == 1; };
        auto is_pw = [&](const i2v_conv3d_desc& c) { return unit(c) && c.kt == 1 && c.kh == 1 && c.kw == 1 && c.pad == 0 && c.pad_t == 0; };
        auto is_t = [&](const i2v_conv3d_desc& c) { return unit(c) && c.kh == 1 && c.kw == 1 && c.pad == 0 && (c.kt & 1) && 2 * c.pad_t == c.kt - 1; };      // k x 1 x 1, "same"
        auto is_33 = [&](const i2v_conv3d_desc& c) { return unit(c) && c.kt == 1 && c.kh == 3 && c.kw == 3 && c.pad == 1 && c.pad_t == 0; };
        // nothing but the group's own launches may touch an intermediate, and it must not be hooked
        auto private_to = [&](std::vector<Launch>* L, size_t first, size_t last, const Launch& w) {
            const Range r = dst_of(w);
            for (auto& hk : hooked) if (meet(hk, r)) return false;
            for (std::vector<Launch>* M : {&n.fwd, &n.bwd})
                for (size_t j = 0; j < M->size(); ++j) {
                    if (M == L && j >= first && j <= last) continue;
                    if (touches((*M)[j], r, false, false)) return false;
                }
            return true;
        };
        for (Launch& l : n.fwd) l.fb_ok = 0;
        for (Launch& l : n.bwd) l.fb_ok = 0;
        // backward: [conv3's input gradient, X, conv2's input gradient] with X independent of both (a first block's projection-shortcut
        // gradient sits between them): X moves in front
        for (size_t i = 0; !fb_off && i + 2 < n.bwd.size(); ++i) {
            Launch& a = n.bwd[i]; Launch& x = n.bwd[i + 1]; Launch& b = n.bwd[i + 2];
            const i2v_conv3d_desc* ca = cd_of(a); const i2v_conv3d_desc* cb = cd_of(b);
            if (!ca || !cb || !is_pw(*ca) || !is_33(*cb) || b.conv.src != a.conv.dst || x.kind != L_CONV) continue;
            const I2VConvParams& px = x.conv;
            const bool dep = touches(x, dst_of(a), false, false) || touches(a, dst_of(x), false, false) || touches(x, dst_of(b), false, false) || touches(b, dst_of(x), false, false) ||
                             (px.gate_out && (px.gate_out == a.conv.gate || px.gate_out == b.conv.gate));
            if (!dep) { Launch t = x; n.bwd[i + 1] = n.bwd[i]; n.bwd[i] = t; }
        }
        for (size_t i = 0; !fb_off && i + 1 < n.bwd.size(); ++i) {
            Launch& a = n.bwd[i]; Launch& b = n.bwd[i + 1];
            const i2v_conv3d_desc* ca = cd_of(a); const i2v_conv3d_desc* cb = cd_of(b);
            if (!ca || !cb || !is_pw(*ca) || !is_33(*cb) || a.T != b.T || b.conv.src != a.conv.dst) continue;
            if (i2v_fastblock_rows(a.conv, b.conv, nullptr, nullptr) <= 0) continue;
            if (!private_to(&n.bwd, i, i + 1, a)) continue;
            a.fb_ok = 2;
        }
        for (size_t i = 0; !fb_off && i + 2 < n.fwd.size(); ++i) {
            Launch& a = n.fwd[i]; Launch& b = n.fwd[i + 1];
            const i2v_conv3d_desc* ca = cd_of(a); const i2v_conv3d_desc* cb = cd_of(b);
            if (!ca || !cb || a.src_is_input || !(is_t(*ca) || is_pw(*ca)) || !is_33(*cb) || !ca->relu || !cb->relu || ca->residual >= 0 || cb->residual >= 0 || cb->src != ca->dst) continue;
            Launch& c3 = n.fwd[i + 2];
            const i2v_conv3d_desc* cc = cd_of(c3);
            if (!cc || !is_pw(*cc)) continue;
            if (cc->src == cb->dst && cc->residual >= 0 && cc->relu && a.T == b.T && a.T == c3.T) {               // identity shortcut
                if (i2v_fastblock_rows(a.conv, b.conv, &c3.conv, nullptr) <= 0) continue;
                if (!private_to(&n.fwd, i, i + 2, a) || !private_to(&n.fwd, i, i + 2, b)) continue;
                a.fb_ok = 3;
            } else if (i + 3 < n.fwd.size() && cc->src == ca->src && !cc->relu && cc->residual < 0) {             // projection shortcut, then conv3
                Launch& c4 = n.fwd[i + 3];
                const i2v_conv3d_desc* c4d = cd_of(c4);
                if (!c4d || !is_pw(*c4d) || c4d->src != cb->dst || c4d->residual != cc->dst || !c4d->relu || a.T != b.T || a.T != c3.T || a.T != c4.T) continue;
                if (i2v_fastblock_rows(a.conv, b.conv, &c4.conv, &c3.conv) <= 0) continue;
                if (!private_to(&n.fwd, i, i + 3, a) || !private_to(&n.fwd, i, i + 3, b) || !private_to(&n.fwd, i, i + 3, c3)) continue;
                a.fb_ok = 4;
            }
        }
        // without the autotuner (I2V_AUTOTUNE=0: tests, tools, the host simulation) a group is fused only on request
        const bool fb_force = [] { const char* e = getenv("I2V_FORCE_FASTBLOCK"); return e && e[0] == '1'; }();      // (read per plan: tests compare both)
        for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
            for (Launch& a : *L)
                for (int bk = 0; bk < 4; ++bk) a.fb_b[bk] = (a.fb_ok && fb_force) ? 1 : 0;
        if (getenv("I2V_FUSE_DEBUG"))
            for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
                for (size_t i = 0; i < L->size(); ++i)
                    if ((*L)[i].fb_ok) fprintf(stderr, "[i2v fastblock] %s launch %zu: group of %d (Cs %d -> %d mid channels, %dx%d)\n", L == &n.fwd ? "fwd" : "bwd", i, (*L)[i].fb_ok,
                                               (*L)[i].conv.Cs, (*L)[i].conv.Cd, (*L)[i].conv.Hg, (*L)[i].conv.Wg);
    }
    // without the autotuner (I2V_AUTOTUNE=0: tests, tools) a pair is fused only on request: I2V_FORCE_FUSE = 1 (plain) / 2 (halo where it applies)
    const char* force = getenv("I2V_FORCE_FUSE");
    const int f = force ? atoi(force) : 0;
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
        for (Launch& a : *L)
            for (int b = 0; b < 4; ++b) a.fuse_b[b] = (a.fuse_ok && f) ? ((f == 2 && (a.fuse_ok & 2)) ? 2 : 1) : 0;
}

// Launch overlap (round 6): a single 32-frame clip leaves the 14 x 14 layers with 1.5 tiles per CU, and a launch
// list is a chain -- except where a block has a projection shortcut: forward, the shortcut convolution depends on the block's input
// only (conv3 adds its result); backward, its input gradient depends on the gradient of the block's output only (conv1's input
// gradient adds it).  Such a launch is HOISTED: issued on the net's side stream as soon as the launch it depends on has been issued
// on the main stream, joined (event wait on the main stream) in front of the first launch that touches its operands.  Which launches
// qualify is decided from the address ranges every launch of a list reads and writes -- data operands, 1-bit gate rows, arg-max
// bytes, the caller's gradient tensor as one opaque range; attention launches are barriers -- conservatively (ranges at the planned
// frame count; any overlap counts).  No launch changes, so every result stays bit-identical; the host simulation, which executes
// launches synchronously in ISSUE order, really runs a hoisted launch early and so tests the analysis (tests/test_planner_hostsim.py).
// MEASURED AND NOT TAKEN (tools/overlap_probe.py, profiles/r6_overlap_probe.txt): one 32-frame clip 610 -> 589 frames/s (-3.4 %), four
// clips 761 -> 755 -- every hoisted launch costs two event records and two cross-queue waits, and a queue that waits on another
// queue's signal resumes tens of microseconds late, more than the idle tail it was meant to fill.  OFF by default; active only for
// calls of at most $I2V_OVERLAP_MAX_FRAMES frames (the tests set it) and never while launches are being timed.
static void mark_overlap(Net& n) {
    for (int k = 0; k < 2; ++k) n.ov_at[k].clear();
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd}) for (Launch& l : *L) { l.ov_after = -2; l.ov_join = -1; }
    const char* e = getenv("I2V_OVERLAP_MAX_FRAMES");
    n.ov_max_frames = e ? atoi(e) : 0;
    if (n.ov_max_frames <= 0) return;
    const int64_t clips = n.maxN / n.Tin();
    typedef std::pair<const char*, const char*> Range;
    struct RW { std::vector<Range> r, w; bool barrier = false; };
    static const char gx_tag[16] = {0};          // stands for the caller's gradient tensor (every launch that touches it conflicts with every other)
    const Range GX(gx_tag, gx_tag + 16);
    auto rng = [](const void* q, int64_t fr, int64_t stride_bytes, int64_t plane_bytes) {
        const char* f = (const char*)q;
        return f ? Range(f, f + (fr > 0 ? (fr - 1) * stride_bytes : 0) + std::max<int64_t>(plane_bytes, 1)) : Range(nullptr, nullptr);
    };
    auto rw_of = [&](const Launch& c) {
        RW o;
        auto rd = [&](const Range& x) { if (x.first) o.r.push_back(x); };
        auto wr = [&](const Range& x) { if (x.first) o.w.push_back(x); };
        if (c.kind == L_CONV || c.kind == L_IMGGRAD) {
            const I2VConvParams& q = c.conv;
            const int64_t fs = clips * std::max(1, q.Ts), fo = clips * std::max(1, q.To);
            const int64_t dplane = 4ll * (q.blk > 1 ? q.Cd / (q.blk * q.blk) : q.Cd) * q.Ho * q.Wo;
            if (!c.src_is_input) rd(rng(q.src, fs, 4 * q.src_nstride, 4ll * q.Cs * q.Hs * q.Ws));
            rd(rng(q.add0, fo, 4 * q.add0_nstride, dplane)); rd(rng(q.add1, fo, 4 * q.add1_nstride, dplane)); rd(rng(q.mask, fo, 4 * q.mask_nstride, dplane));
            rd(rng(q.gate, 1, 0, 4ll * q.Cd * q.gate_stride)); wr(rng(q.gate_out, 1, 0, 4ll * q.Cd * q.gate_out_stride));
            if (c.kind == L_IMGGRAD) { wr(GX); rd(GX); } else wr(rng(q.dst, fo, 4 * q.dst_nstride, dplane));
        } else if (c.kind == L_ADDMASK) {
            const int64_t fr = clips * std::max(1, c.T), pl = 4ll * c.am.C * c.am.HW;
            wr(rng(c.am.out, fr, 4 * c.am.out_nstride, pl));
            for (int i = 0; i < 3; ++i) rd(rng(c.am.a[i], fr, 4 * c.am.a_nstride[i], pl));
            rd(rng(c.am.mask, fr, 4 * c.am.mask_nstride, pl)); rd(rng(c.am.gate, 1, 0, 4ll * c.am.C * c.am.gate_stride));
        } else if (c.kind == L_MEMSET) {
            if (c.ms_gx) wr(GX); else wr(rng(c.ms_ptr, 1, 0, 4ll * (int64_t)c.ms_floats_per_frame * clips * std::max(1, c.T)));
        } else if (c.kind == L_AGEMM || c.kind == L_SOFTMAX || c.kind == L_CONVB_UNUSED) {
            o.barrier = true;
        } else {
            const I2VPoolParams& q = c.pool;
            const int64_t fr = clips * std::max(std::max(1, c.T), std::max(q.Ts, q.To)), pin = 4ll * q.C * q.Hs * q.Ws, pout = 4ll * q.C * q.Ho * q.Wo;
            const Range idx = rng(q.idx, 1, 0, fr * (int64_t)q.C * q.Ho * q.Wo);
            const bool fwd = c.kind == L_POOLF || c.kind == L_POOL3F || c.kind == L_AVGF;
            if (fwd) { rd(rng(q.x, fr, 4 * q.x_nstride, pin)); wr(rng(q.y, fr, 4 * q.y_nstride, pout)); wr(idx); }
            else { rd(rng(q.x, fr, 4 * q.x_nstride, pin)); rd(rng(q.y, fr, 4 * q.y_nstride, pout)); rd(rng(q.yact, fr, 4 * q.yact_nstride, pout)); rd(idx);
                   wr(rng(q.gx, fr * std::max(1, q.stride_t), 4 * q.gx_nstride, pin)); }
        }
        return o;
    };
    auto meet = [](const std::vector<Range>& A, const std::vector<Range>& B) {
        for (const Range& a : A) for (const Range& b : B) if (a.first < b.second && b.first < a.second) return true;
        return false;
    };
    auto conflict = [&](const RW& a, const RW& b) { return a.barrier || b.barrier || meet(a.r, b.w) || meet(a.w, b.w) || meet(a.w, b.r); };
    int k = 0;
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd}) {
        const int M = (int)L->size();
        std::vector<RW> rw; rw.reserve(M);
        for (const Launch& l : *L) rw.push_back(rw_of(l));
        std::vector<char> member(M, 0);          // part of a group that may run as ONE kernel (fused pair, fused fast-pathway block): stays in place
        for (int i = 0; i < M; ++i) {
            const Launch& l = (*L)[i];
            for (int j = 0; j < l.fb_ok && i + j < M; ++j) member[i + j] = 1;
            if (l.fuse_ok) { member[i] = 1; if (i + 1 < M) member[i + 1] = 1; }
        }
        n.ov_at[k].assign(M + 1, {});
        int hoisted = 0;
        for (int i = 1; i < M; ++i) {
            Launch& l = (*L)[i];
            if (l.kind != L_CONV || member[i] || l.src_is_input) continue;
            int dep = -1;
            for (int j = i - 1; j >= 0; --j) if (conflict(rw[i], rw[j])) { dep = j; break; }
            if (dep >= i - 1) continue;                                   // depends on its predecessor: nothing to overlap with
            if (dep >= 0 && (*L)[dep].ov_after != -2) continue;            // depends on a hoisted launch: stays on the main stream (which joins in front of it)
            int join = M;
            for (int j = i + 1; j < M; ++j) if (conflict(rw[j], rw[i])) { join = j; break; }
            l.ov_after = dep; l.ov_join = join;
            n.ov_at[k][dep + 1].push_back(i);
            ++hoisted;
            if (getenv("I2V_FUSE_DEBUG"))
                fprintf(stderr, "[i2v overlap] %s launch %d (Cd %d, K %d, %dx%d): side stream after launch %d, joined before launch %d of %d\n", k ? "bwd" : "fwd", i,
                        l.conv.Cd, l.conv.K, l.conv.Hg, l.conv.Wg, dep, join, M);
        }
        if (!hoisted) n.ov_at[k].clear();
        ++k;
    }
    if (n.ov_at[0].empty() && n.ov_at[1].empty()) return;
    n.side = be_stream_create();
    if (!n.side) { n.ov_at[0].clear(); n.ov_at[1].clear(); }
}

// Plan-time autotuning ("measure, don't guess"): every convolution launch of both passes is timed with each
// valid tile configuration on the planned shapes (max_frames) and the fastest is pinned.  Tile choice never
// changes results: each output element is the same k-ordered fmaf chain whatever the tile.
static int autotune(Net& n) {
    const char* off = getenv("I2V_AUTOTUNE");
    if (off && off[0] == '0') return 0;
    const Buffer& ib = n.bufs[n.tens[n.input].buf];
    const size_t img = (size_t)n.maxN * ib.C * ib.H * ib.W;
    float* scratch = (float*)be_malloc(2 * img * sizeof(float));     // stand-ins for the caller's x and gx
    if (!scratch) return fail("autotune scratch allocation failed");
    CHECK_BE(be_memset0(scratch, 2 * img * sizeof(float), nullptr));
    void* e0 = be_event_create(); void* e1 = be_event_create();
    const float* xin = n.stage_input ? n.arena + n.in_stage_off : scratch;      // quad-row stems need slack around their source
    int rc = 0;
    // One tuning per batch bucket (the planned size, half, a quarter, an eighth of its clips): a configuration that wins at 128
    // frames (fewer, larger tiles) loses up to 10 % at 32, where the launch no longer fills the chip -- a caller that plans once for
    // its largest group (image_main.py --group_clips, bench.py's single_clip phase) runs every group size on its own choice.
    const int max_clips = n.maxN / n.Tin();
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
        for (Launch& l : *L) {
            if (l.kind != L_CONV && l.kind != L_IMGGRAD) continue;
            const int planned_cfg = l.conv.cfg;
            for (int b = 3; b >= 0 && !rc; --b) {               // (bucket 0 last: conv.cfg ends as the planned size's choice)
                const int clips_b = max_clips >> b;
                if (clips_b < 1 || (b > 0 && clips_b == (max_clips >> (b - 1)))) continue;
                const int lf = clips_b * l.conv.Tg;             // grid frames of this launch at the bucket's size
                if (lf * l.conv.Hg * l.conv.Wg == 0) continue;
                int cand[16]; I2VConvParams probe = l.conv; probe.N = lf;
                probe.temporal = conv_prep(l, xin, scratch + img, 0).temporal;      // (the frame-map half of `temporal` is derived at run time: candidates must see what the launch will be)
                const int nc = k_conv_candidates(probe, cand);
                if (nc <= 1) { if (nc == 1) l.cfg_b[b] = cand[0] + 1; continue; }
                // one configuration: a warm-up launch, then two timed launches; returns ms per launch.  (Timing short launches until a
                // millisecond had passed -- up to 16 repetitions -- tripled the plan time, 2.0 -> 7.1 s for the bench's three plans, and moved
                // neither the headline nor the single-clip figure beyond box-to-box noise: not kept.)
                auto time_cfg = [&](int cfg_plus_1, float* per_launch) -> int {
                    l.conv.cfg = cfg_plus_1;
                    if (conv_run(l, lf, xin, scratch + img, 0, nullptr)) return 1;                   // warm-up
                    // (the stem gradients -- one to three launches per net, their candidates 5-15 % apart -- are timed over four launches: with two,
                    //  alternated runs of one build disagreed about the I3D stem's kernel, an 8 % difference on 18 % of an ILAF step)
                    const int reps = l.kind == L_IMGGRAD ? 4 : 2;
                    be_event_record(e0, nullptr);
                    for (int r = 0; r < reps; ++r) if (conv_run(l, lf, xin, scratch + img, 0, nullptr)) return 1;
                    be_event_record(e1, nullptr);
                    if (be_stream_sync(nullptr)) return 1;
                    float ms = 0.f; be_event_elapsed_ms(e0, e1, &ms);
                    *per_launch = ms / reps;
                    return 0;
                };
                float best = 1e30f; int best_c = -1;
                for (int ci = 0; ci < nc && !rc; ++ci) {
                    float ms = 0.f;
                    if (time_cfg(cand[ci] + 1, &ms)) { rc = 1; break; }
                    if ((cand[ci] & 2048) && getenv("I2V_FUSE_DEBUG"))
                        fprintf(stderr, "[i2v vfma] %s launch Cd %d K %d (%dx%d) at %d frames: conv_vfma %.1f us, best tile so far %.1f us\n", L == &n.fwd ? "fwd" : "bwd",
                                l.conv.Cd, l.conv.K, l.conv.Hg, l.conv.Wg, lf, 1e3f * ms, 1e3f * best);
                    if (ms < best) { best = ms; best_c = cand[ci]; }
                }
                // second stage: the winner with streaming (non-temporal) epilogue stores, bit 7 -- one more timing per launch
                // instead of doubling the candidate list (only the dense vector epilogue has them; elsewhere the bit is inert)
                static const bool no_nt = [] { const char* e = getenv("I2V_NT"); return e && e[0] == '0'; }();
                if (best_c >= 0 && l.kind == L_CONV && !no_nt && !rc) {
                    float ms = 0.f;
                    if (time_cfg((best_c | 128) + 1, &ms)) { rc = 1; break; }
                    if (ms < 0.98f * best) { best = ms; best_c |= 128; }
                }
                l.cfg_b[b] = best_c >= 0 ? best_c + 1 : 0;
            }
            l.conv.cfg = l.cfg_b[0] ? l.cfg_b[0] : planned_cfg;
            if (rc) break;
        }
    // Fused fast-pathway blocks, per batch bucket: the group's launches on their tuned tiles against the group as one kernel; taken when
    // at least 3 % faster (two timed runs each after a warm-up).  Only where timing means something (the device).
    if (!rc && strncmp(be_name(), "hip", 3) == 0) {
        const bool fb_force = [] { const char* e = getenv("I2V_FORCE_FASTBLOCK"); return e && e[0] == '1'; }();      // (read per plan: tests compare both)
        for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
            for (size_t i = 0; i < L->size() && !rc; ++i) {
                Launch& a = (*L)[i];
                if (!a.fb_ok || i + a.fb_ok > L->size()) continue;
                for (int bk = 3; bk >= 0 && !rc; --bk) {
                    a.fb_b[bk] = 0;
                    const int clips_b = max_clips >> bk;
                    if (clips_b < 1 || (bk > 0 && clips_b == (max_clips >> (bk - 1)))) continue;
                    const int lf = clips_b * a.conv.Tg;
                    auto run_sep = [&]() { for (int j = 0; j < a.fb_ok; ++j) { Launch& m = (*L)[i + j]; if (m.cfg_b[bk]) m.conv.cfg = m.cfg_b[bk]; if (conv_run(m, lf, xin, scratch + img, 0, nullptr)) return 1; } return 0; };
                    auto run_fb = [&]() { return fast_run(*L, i, lf, xin, nullptr); };
                    auto timed = [&](auto&& fn, float* ms) -> int {
                        if (fn()) return 1;
                        be_event_record(e0, nullptr);
                        for (int r = 0; r < 3; ++r) if (fn()) return 1;
                        be_event_record(e1, nullptr);
                        if (be_stream_sync(nullptr)) return 1;
                        be_event_elapsed_ms(e0, e1, ms);
                        return 0;
                    };
                    float t_sep = 0.f, t_fb = 0.f;
                    if (run_fb() != 0) continue;                       // not eligible at this size (or it failed: the separate launches stay)
                    if (timed(run_sep, &t_sep) || timed(run_fb, &t_fb)) { rc = 1; break; }
                    a.fb_b[bk] = (fb_force || t_fb < 0.97f * t_sep) ? 1 : 0;
                    if (getenv("I2V_FUSE_DEBUG"))
                        fprintf(stderr, "[i2v fastblock] %s group %zu (%d launches, %d -> %d mid channels) at %d frames: separate %.1f us, fused %.1f us -> %s\n",
                                L == &n.fwd ? "fwd" : "bwd", i, a.fb_ok, a.conv.Cs, a.conv.Cd, lf, 1e3f * t_sep / 3, 1e3f * t_fb / 3, a.fb_b[bk] ? "fused" : "separate");
                }
                for (int j = 0; j < a.fb_ok; ++j) { Launch& m = (*L)[i + j]; m.conv.cfg = m.cfg_b[0] ? m.cfg_b[0] : m.conv.cfg; }
            }
    }
    // Fused pairs, per batch bucket: the two launches on their tuned tiles against the pair as one kernel (plain / halo staging).
    // (Only where timing means something: the host simulation of the tests has one configuration of everything.)
    // Opt-in (I2V_FUSE=1) since round 5: the fused kernel loses 7-35 % on every pair at the headline size and wins 1-2 % on three pairs at
    // 32 frames (profiles/r4_fuse_pairs.txt) -- not worth up to 9 probe launches per pair and bucket in every plan, nor a plan that
    // depends on the process environment (a profiling tool's per-dispatch overhead flatters the single launch).
    static const bool fuse_on = [] { const char* e = getenv("I2V_FUSE"); return e && e[0] == '1'; }();
    const bool on_device = fuse_on && strncmp(be_name(), "hip", 3) == 0;
    for (std::vector<Launch>* L : {&n.fwd, &n.bwd})
        for (size_t i = 0; on_device && i + 1 < L->size() && !rc; ++i) {
            Launch& a = (*L)[i]; Launch& b = (*L)[i + 1];
            if (!a.fuse_ok) continue;
            for (int bk = 3; bk >= 0 && !rc; --bk) {
                a.fuse_b[bk] = 0;
                const int clips_b = max_clips >> bk;
                if (clips_b < 1 || (bk > 0 && clips_b == (max_clips >> (bk - 1)))) continue;
                const int lf = clips_b * a.conv.Tg;
                if (!fused_fits(a, b, lf)) continue;
                const int ca = a.conv.cfg, cb = b.conv.cfg;
                if (a.cfg_b[bk]) a.conv.cfg = a.cfg_b[bk];
                if (b.cfg_b[bk]) b.conv.cfg = b.cfg_b[bk];
                float best = 1e30f; int best_v = 0;
                for (int v = 0; v <= ((a.fuse_ok & 2) ? 2 : 1) && !rc; ++v) {
                    auto once = [&]() { return v == 0 ? (conv_run(a, lf, xin, scratch + img, 0, nullptr) || conv_run(b, lf, xin, scratch + img, 0, nullptr))
                                                      : fused_run(a, b, lf, xin, v == 2, nullptr); };
                    if (v > 0 && once()) {       // a fused variant that will not launch is simply not a candidate
                        if (getenv("I2V_FUSE_DEBUG")) fprintf(stderr, "[i2v fuse] pair %zu variant %d does not launch: %s\n", i, v, g_err.c_str());
                        g_err.clear(); continue;
                    }
                    if (v == 0) rc |= once();
                    be_event_record(e0, nullptr);
                    for (int r = 0; r < 2 && !rc; ++r) rc |= once();
                    be_event_record(e1, nullptr);
                    if (rc || be_stream_sync(nullptr)) { rc = 1; break; }
                    float ms = 0.f; be_event_elapsed_ms(e0, e1, &ms);
                    // one launch instead of two also saves whatever a tracing tool adds per dispatch (under `rocprofv3 --pmc` that made the
                    // fused kernel "win" pairs it loses by 10-40 % in a plain run): it has to be 3 % faster to be taken
                    if ((v == 0 ? ms : ms * 1.03f) < best) { best = v == 0 ? ms : ms * 1.03f; best_v = v; }
                    if (getenv("I2V_FUSE_DEBUG"))
                        fprintf(stderr, "[i2v fuse] %s pair %zu (Cd %d -> %d, %dx%d) at %d frames: variant %d = %.1f us\n", L == &n.fwd ? "fwd" : "bwd", i,
                                a.conv.Cd, b.conv.Cd, a.conv.Hg, a.conv.Wg, lf, v, ms * 500.f);
                }
                a.fuse_b[bk] = best_v;
                a.conv.cfg = ca; b.conv.cfg = cb;
            }
        }
    be_event_destroy(e0); be_event_destroy(e1); be_free(scratch);
    if (rc) return g_err.empty() ? fail("autotune failed") : 1;
    // the probes scribbled over activations and gradients; start from a clean arena like a fresh plan
    CHECK_BE(be_memset0(n.arena, n.arena_floats * sizeof(float), nullptr));
    return 0;
}

static TimedLaunch* timing_begin(i2v_ctx* h, int kind, double flops, i2v_stream_t s, TimedLaunch* prev) {
    if (!h->timing) return nullptr;
    TimedLaunch* t;
    {
        std::lock_guard<std::mutex> lock(h->timing_mu);
        if (h->timed_used == h->timed.size()) {
            TimedLaunch fresh{be_event_create(), be_event_create(), 0, 0.0, 0, 0, 0, 0, 0, nullptr, 0.0, 1};
            if (!fresh.start || !fresh.stop) return nullptr;
            h->timed.push_back(fresh);
        }
        t = &h->timed[h->timed_used++];
    }
    t->kind = kind; t->flops = flops; t->Cd = t->K = t->HWg = t->frames = t->pw = 0; t->bytes = 0.0; t->count = 1;
    t->chain_from = prev ? prev->stop : nullptr;
    if (!t->chain_from) be_event_record(t->start, s);
    return t;
}

static int run_list(i2v_ctx* h, Net& n, std::vector<Launch>& L, int in_frames, const float* x, float* gx, int accumulate,
                    i2v_stream_t s, bool backward_pass) {
    const int clips = in_frames / n.Tin();
    TimedLaunch* prev_timed = nullptr;                   // the first launch of the list records its own start event
    TimedLaunch* seg = nullptr;                          // segment mode: the open segment
    // launch overlap (mark_overlap): hoisted launches go to the side stream right after the launch they depend on was issued here
    const std::vector<std::vector<int>>& ov_at = n.ov_at[backward_pass ? 1 : 0];
    const bool overlap = !h->timing && n.side && !ov_at.empty() && in_frames <= n.ov_max_frames;
    struct SidePending { int join; void* done; };
    std::vector<SidePending> side_pending;
    if (overlap) n.ov_used = 0;
    auto ov_event = [&]() -> void* {
        if (n.ov_used == n.ov_ev.size()) { void* e = be_event_create(); if (!e) return nullptr; n.ov_ev.push_back(e); }
        return n.ov_ev[n.ov_used++];
    };
    auto issue_side = [&](int p) -> int {                // the hoisted launches that follow main launch p (-1: the start of the list)
        for (int i : ov_at[p + 1]) {
            void* ready = ov_event(); void* done = ov_event();
            if (!ready || !done) return fail("launch overlap: event creation failed");
            CHECK_BE(be_event_record(ready, s)); CHECK_BE(be_stream_wait_event(n.side, ready));
            Launch& m = L[i];
            const int fr = clips * m.T;
            if (fr * m.conv.Hg * m.conv.Wg != 0) {
                const int cb = m.cfg_b[cfg_bucket(clips, n.maxN / n.Tin())];
                if (cb) m.conv.cfg = cb;
                if (conv_run(m, fr, x, gx, accumulate, n.side)) return 1;
                __atomic_fetch_add(&g_overlap_launches, 1, __ATOMIC_RELAXED);
            }
            CHECK_BE(be_event_record(done, n.side));
            side_pending.push_back(SidePending{m.ov_join, done});
        }
        return 0;
    };
    auto join_side = [&](int upto) -> int {              // the main stream waits for every hoisted launch whose first dependent is at or before `upto`
        for (size_t i = 0; i < side_pending.size();) {
            if (side_pending[i].join <= upto) { CHECK_BE(be_stream_wait_event(s, side_pending[i].done)); side_pending[i] = side_pending.back(); side_pending.pop_back(); }
            else ++i;
        }
        return 0;
    };
    if (overlap && issue_side(-1)) return 1;
    for (size_t li = 0; li < L.size(); ++li) {
        Launch& l = L[li];
        const int frames = clips * l.T;                  // frames this launch iterates over
        const size_t li0 = li;
        if (overlap && l.ov_after != -2) continue;       // hoisted: already issued on the side stream
        // this 3x3 launch and the pointwise launch behind it as ONE kernel (mark_fusable / autotune): the next entry is skipped
        const int fuse = (l.kind == L_CONV && l.fuse_ok && li + 1 < L.size() && frames * l.conv.Hg * l.conv.Wg > 0 && fused_fits(l, L[li + 1], frames))
                             ? l.fuse_b[cfg_bucket(clips, n.maxN / n.Tin())] : 0;
        // this launch and the next fb_ok - 1 as ONE fused fast-pathway block (mark_fusable / autotune): those entries are skipped
        const int fb = (l.kind == L_CONV && l.fb_ok && li + l.fb_ok <= L.size() && frames * l.conv.Hg * l.conv.Wg > 0) ? l.fb_b[cfg_bucket(clips, n.maxN / n.Tin())] * l.fb_ok : 0;
        if (overlap && join_side((int)li + (fb ? fb - 1 : fuse ? 1 : 0))) return 1;
        double flops = 0.0;
        if (fb) { for (int j = 0; j < fb; ++j) { const Launch& m = L[li + j]; flops += m.alg_flops_per_frame > 0 ? m.alg_flops_per_frame * frames : 2.0 * frames * m.conv.Hg * m.conv.Wg * (double)m.conv.Cd * m.conv.K; } }
        else if (l.kind == L_CONV && l.alg_flops_per_frame > 0) flops = l.alg_flops_per_frame * frames;     // quad-row packings pad K
        else if (l.kind == L_CONV) flops = 2.0 * frames * l.conv.Hg * l.conv.Wg * ((double)l.conv.Cd * l.conv.K + (fuse ? (double)L[li + 1].conv.Cd * L[li + 1].conv.K : 0.0));
        else if (l.kind == L_IMGGRAD) flops = l.alg_flops_per_frame * frames;
        else if (l.kind == L_AGEMM) flops = 2.0 * clips * (double)l.ag.Cc * l.ag.M * l.ag.N;
        // timing kinds: 0 conv fwd, 1 image gradient, 2 pool fwd, 3 pool bwd, 4 addmask, 5 conv input-gradient
        const int tkind = (l.kind == L_CONV && backward_pass) ? 5 : (l.kind == L_AVGF || l.kind == L_POOL3F) ? 2
                          : (l.kind == L_AVGB || l.kind == L_POOL3B) ? 3 : (l.kind == L_MEMSET || l.kind == L_SOFTMAX) ? 4
                          : l.kind == L_AGEMM ? (backward_pass ? 5 : 0) : (int)l.kind;
        // Segment mode (i2v_timing_enable(h, 2)): consecutive launches of one kind share ONE event pair -- a forward list is three or
        // four segments instead of fifty event records -- and the segment accumulates their flops / bytes / count; the per-launch
        // fields of a dump line and the low-intensity split need mode 1.
        TimedLaunch* tl = nullptr;
        if (h->timing == 2) {
            if (!seg || seg->kind != tkind) {
                if (seg) be_event_record(seg->stop, s);
                seg = timing_begin(h, tkind, 0.0, s, seg);
                if (seg) seg->count = 0;
            }
            if (seg) { seg->flops += flops; seg->count += 1; }
        } else tl = timing_begin(h, tkind, flops, s, prev_timed);
        prev_timed = tl;
        TimedLaunch seg_bytes_sink{};              // (segment mode: the byte count below is added to the open segment)
        TimedLaunch* const bt = tl ? tl : (seg ? &seg_bytes_sink : nullptr);
        if (bt && (l.kind == L_CONV || l.kind == L_IMGGRAD)) {
            TimedLaunch* const tl = bt;
            tl->Cd = l.conv.Cd; tl->K = l.conv.K; tl->HWg = l.conv.Hg * l.conv.Wg; tl->frames = frames; tl->pw = l.conv.pointwise;
            // ALGORITHMIC bytes of the launch: every operand once -- the source view, the packed weights, the output, each
            // epilogue addend, and the ReLU gate (fp32 activation, or 1 bit per element) -- whatever the tiling re-reads
            const I2VConvParams& q = l.conv;
            const double out = (double)frames * q.Hg * q.Wg * q.Cd;
            const double src = (double)clips * q.Ts * q.Cs * q.Hs * q.Ws;
            double b = 4.0 * (src + (double)q.K * q.Cd + out);
            if (q.add0) b += 4.0 * out / (q.add0_stride * q.add0_stride);
            if (q.add1 || l.kind == L_IMGGRAD) b += (q.add1 || accumulate || l.img_accumulate) ? 4.0 * out : 0.0;
            if (q.mask) b += 4.0 * out;
            if (q.gate) b += out / 8.0;
            if (q.gate_out) b += out / 8.0;
            if (fb) {         // a fused fast-pathway block: its source, every member's weights and gate words, the last member's output (+ an identity residual)
                const I2VConvParams& last = L[li + fb - 1].conv;
                const double outl = (double)frames * last.Hg * last.Wg * last.Cd;
                b = 4.0 * (src + outl);
                for (int j = 0; j < fb; ++j) {
                    const I2VConvParams& r = L[li + j].conv;
                    const double o = (double)frames * r.Hg * r.Wg * r.Cd;
                    b += 4.0 * (double)r.K * r.Cd + (r.gate ? o / 8.0 : 0.0) + (r.gate_out ? o / 8.0 : 0.0);
                }
                if (fb == 3) b += 4.0 * outl;
            }
            if (fuse) {       // + the pointwise half: its weights, output and epilogue operands; the intermediate is neither written nor read
                const I2VConvParams& r = L[li + 1].conv;
                const double out2 = (double)frames * r.Hg * r.Wg * r.Cd;
                b += 4.0 * ((double)r.K * r.Cd + out2) - 4.0 * out;
                if (r.add0) b += 4.0 * out2;
                if (r.add1) b += 4.0 * out2;
                if (r.mask) b += 4.0 * out2;
                if (r.gate) b += out2 / 8.0;
                if (r.gate_out) b += out2 / 8.0;
            }
            tl->bytes = b;
        }
        if (seg && !tl) seg->bytes += seg_bytes_sink.bytes;
        if (tl && l.kind == L_AGEMM) {                  // (dump fields: channels, reduction length, output columns, 10 + product form)
            const I2VAttnGemm& q = l.ag;
            tl->Cd = q.Cc; tl->K = q.form == 1 ? q.Cc : (q.form == 2 ? q.N : q.M); tl->HWg = q.form == 2 ? q.M : q.N; tl->frames = frames; tl->pw = 10 + q.form;
            tl->bytes = 4.0 * clips * ((double)q.M * q.N + (double)q.Cc * q.M + (double)q.Cc * q.N);
        }
        struct Stop { TimedLaunch* t; i2v_stream_t s; ~Stop() { if (t) be_event_record(t->stop, s); } } stop{tl, s};
        switch (l.kind) {
            case L_CONV:
            case L_IMGGRAD: {
                if (frames * l.conv.Hg * l.conv.Wg == 0) break;
                if (fuse) { if (fused_run(l, L[li + 1], frames, x, fuse == 2, s)) return 1; ++li; break; }
                if (fb) {
                    const int rc = fast_run(L, li, frames, x, s);
                    if (rc == 1) return 1;
                    if (rc == 0) { li += fb - 1; break; }
                    // (rc == 2: not eligible at this frame count -- the separate launches run, this one now and the others in their turn)
                }
                const int cb = l.cfg_b[cfg_bucket(clips, n.maxN / n.Tin())];
                if (cb) l.conv.cfg = cb;
                if (conv_run(l, frames, x, gx, accumulate, s)) return 1;
            } break;
            case L_POOLF: { I2VPoolParams p = l.pool; p.N = frames; CHECK_BE(k_pool_fwd(p, s)); } break;
            // A 1 x k x k window with temporal stride st over Ts = st*To frames is the image pooling kernel on every
            // st-th frame (frame stride * st); its backward leaves the skipped frames zero.
            case L_POOL3F: {
                I2VPoolParams p = l.pool; p.N = frames;
                if (p.kt == 1 && p.pad_t == 0 && p.Ts == p.stride_t * p.To) { p.x_nstride *= p.stride_t; CHECK_BE(k_pool_fwd(p, s)); }
                else CHECK_BE(k_pool3d_fwd(p, s));
            } break;
            case L_POOL3B: {
                I2VPoolParams p = l.pool; p.N = frames;
                if (p.kt == 1 && p.pad_t == 0 && p.Ts == p.stride_t * p.To && p.gx_nstride == (int64_t)p.C * p.Hs * p.Ws) {
                    CHECK_BE(be_memset0(p.gx, (size_t)frames * p.stride_t * p.gx_nstride * sizeof(float), s));
                    p.x_nstride *= p.stride_t; p.gx_nstride *= p.stride_t;
                    CHECK_BE(k_pool_bwd(p, s));
                } else CHECK_BE(k_pool3d_bwd(p, s));
            } break;
            case L_AVGF: { I2VPoolParams p = l.pool; p.N = frames; CHECK_BE(k_avgpool_fwd(p, s)); } break;
            case L_AVGB: { I2VPoolParams p = l.pool; p.N = frames; CHECK_BE(k_avgpool_bwd(p, s)); } break;
            case L_MEMSET:
                if (!l.ms_gx) CHECK_BE(be_memset0(l.ms_ptr, l.ms_floats_per_frame * frames * sizeof(float), s));
                else if (!accumulate) CHECK_BE(be_memset0(gx, l.ms_floats_per_frame * frames * sizeof(float), s));
                break;
            case L_CONVB_UNUSED: break;
            case L_POOLB: { I2VPoolParams p = l.pool; p.N = frames; CHECK_BE(k_pool_bwd(p, s)); } break;
            case L_ADDMASK: { I2VAddMaskParams p = l.am; p.N = frames; CHECK_BE(k_addmask(p, s)); } break;
            case L_AGEMM: { I2VAttnGemm p = l.ag; p.clips = clips; CHECK_BE(k_attn_gemm(p, s)); } break;
            case L_SOFTMAX: { I2VSoftmaxRows p = l.sm; p.rows = (int64_t)clips * l.sm_rows_per_clip; CHECK_BE(k_softmax_rows(p, s)); } break;
        }
        if (overlap) for (size_t p = li0; p <= li; ++p) if (issue_side((int)p)) return 1;      // (a fused group advanced li past its members)
    }
    if (overlap && join_side((int)L.size())) return 1;   // nothing of this list is still running on the side stream when the caller's next kernel starts
    if (seg) be_event_record(seg->stop, s);
    return 0;
}

extern "C" int i2v_net_forward(i2v_handle h, int net, const float* x, int frames, void* stream) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (!n->planned) return fail("net not planned");
    if (frames <= 0 || frames > n->maxN) return fail("frames=%d outside 1..%d", frames, n->maxN);
    if (frames % n->Tin()) return fail("frames=%d is not a multiple of the input's %d frames per clip", frames, n->Tin());
    if (!x) return fail("null input");
    n->frames = frames;
    if (n->stage_input) {
        // The slack around the source is what conv_tile's quad-row staging (MODE 4) needs.  When the autotuner gave every quad-row
        // launch that reads the input the halo-tile kernel for this batch bucket (configuration bit 10: conv_stem_halo stages whole
        // windows and range-checks every piece) and the caller's frames are 16-byte aligned, nothing reads outside them and the copy --
        // 2 % of an ILAF step on SlowFast -- is skipped.
        static const bool always = [] { const char* e = getenv("I2V_STAGE_COPY"); return e && e[0] == '1'; }();      // (developer knob: A/B)
        bool need = always || ((uintptr_t)x & 15) != 0;
        const int bucket = cfg_bucket(frames / n->Tin(), n->maxN / n->Tin());
        for (const Launch& l : n->fwd)
            if (l.kind == L_CONV && l.src_is_input && l.conv.quad && !(l.cfg_b[bucket] > 0 && ((l.cfg_b[bucket] - 1) & 1024))) need = true;
        if (need) {
            const Buffer& ib = n->bufs[n->tens[n->input].buf];
            const size_t bytes = (size_t)frames * ib.C * ib.H * ib.W * sizeof(float);
            float* staged = n->arena + n->in_stage_off;
            CHECK_BE(be_d2d_2d(staged, bytes, x, bytes, bytes, 1, stream));
            x = staged;
        }
    }
    return run_list(h, *n, n->fwd, frames, x, nullptr, 0, stream, false);
}

extern "C" int i2v_net_backward(i2v_handle h, int net, float* gx, int accumulate, void* stream) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (!n->planned || n->frames <= 0) return fail("backward before forward");
    if (!gx) return fail("null gradient output");
    return run_list(h, *n, n->bwd, n->frames, nullptr, gx, accumulate, stream, true);
}

extern "C" int i2v_timing_enable(i2v_handle h, int enable) {
    if (!h) return fail("null handle");
    h->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0); h->timed_used = 0;
    return 0;
}

// kinds: 0 conv_igemm forward, 1 first-layer image gradient, 2 pool fwd, 3 pool bwd, 4 addmask, 5 conv_igemm dgrad
// fields per kind: 0 ms, 1 algorithmic flops, 2 launches, 3 algorithmic bytes (convolution launches), and the same for the
// LOW-INTENSITY launches alone -- flops/byte below the machine balance 157.3 TFLOP/s / 8 TB/s = 19.7, i.e. the ones the
// HBM roofline bounds --: 4 ms, 5 bytes, 6 launches, 7 flops
#define I2V_TIMING_FIELDS 8
extern "C" int i2v_timing_collect_ex(i2v_handle h, double* out, int n_kinds, int n_fields) {
    if (!h || !out || n_kinds < 6 || n_fields < I2V_TIMING_FIELDS) return fail("i2v_timing_collect_ex: bad argument");
    for (int i = 0; i < n_kinds * n_fields; ++i) out[i] = 0.0;
    if (h->timed_used) CHECK_BE(be_device_sync());
    const char* dump_path = getenv("I2V_TIMING_DUMP");      // debug: one line per launch
    FILE* dump = dump_path ? fopen(dump_path, "a") : nullptr;
    for (size_t i = 0; i < h->timed_used; ++i) {
        const TimedLaunch& t = h->timed[i];
        float ms = 0.f;
        CHECK_BE(be_event_elapsed_ms(t.chain_from ? t.chain_from : t.start, t.stop, &ms));
        if (dump && h->timing != 2) fprintf(dump, "%d %d %d %d %d %d %.4f %.3f %.3f\n", t.kind, t.Cd, t.K, t.HWg, t.frames, t.pw, ms, t.flops * 1e-9, t.bytes * 1e-6);
        if (t.kind < 0 || t.kind >= n_kinds) continue;
        double* o = out + (size_t)t.kind * n_fields;
        o[0] += ms; o[1] += t.flops; o[2] += t.count; o[3] += t.bytes;
        if (h->timing != 2 && t.bytes > 0 && t.flops < 19.7 * t.bytes) { o[4] += ms; o[5] += t.bytes; o[6] += 1; o[7] += t.flops; }
    }
    if (dump) fclose(dump);
    h->timed_used = 0;
    return 0;
}

extern "C" int i2v_timing_collect(i2v_handle h, double* ms_by_kind, double* flops_by_kind, int64_t* launches_by_kind,
                                  int n_kinds) {
    if (!h || !ms_by_kind || !flops_by_kind || !launches_by_kind || n_kinds < 6 || n_kinds > 16) return fail("i2v_timing_collect: bad argument");
    double tmp[16 * I2V_TIMING_FIELDS];
    if (i2v_timing_collect_ex(h, tmp, n_kinds, I2V_TIMING_FIELDS)) return 1;
    for (int i = 0; i < n_kinds; ++i) {
        ms_by_kind[i] = tmp[i * I2V_TIMING_FIELDS]; flops_by_kind[i] = tmp[i * I2V_TIMING_FIELDS + 1];
        launches_by_kind[i] = (int64_t)tmp[i * I2V_TIMING_FIELDS + 2];
    }
    return 0;
}

extern "C" int i2v_net_hook_info(i2v_handle h, int net, int hook, float** act, int64_t* act_stride,
                                 float** grad, int64_t* grad_stride, int64_t* D, int32_t* post_relu) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (!n->planned) return fail("net not planned");
    if (hook < 0 || hook >= (int)n->hooks.size()) return fail("bad hook index");
    int t = n->hooks[hook];
    View a = view_of(*n, t, false), g = view_of(*n, t, true);
    int64_t d = (int64_t)a.C * a.H * a.W;
    if (act) *act = a.p;
    if (act_stride) *act_stride = a.nstride;
    if (grad_stride) *grad_stride = n->hook_tmp[hook] ? d : g.nstride;
    if (grad) *grad = n->hook_tmp[hook] ? n->hook_tmp[hook] : g.p;
    if (D) *D = d;
    if (post_relu) *post_relu = n->tens[t].post_relu ? 1 : 0;
    return 0;
}

extern "C" int i2v_net_read_tensor(i2v_handle h, int net, int tensor, int which, float* out, int frames,
                                   void* stream) {
    Net* n = get_net(h, net); if (!n) return 1;
    if (!n->planned) return fail("net not planned");
    if (tensor < 0 || tensor >= (int)n->tens.size() || tensor == n->input) return fail("bad tensor id");
    View v = view_of(*n, tensor, which != 0);
    if (frames <= 0 || frames > n->maxN / n->Tin() * v.T) return fail("bad frame count");
    // The intermediate of a fused pair (k_conv_fused) is never stored: reading it back would hand out stale arena contents.
    for (const std::vector<Launch>* L : {&n->fwd, &n->bwd})
        for (const Launch& l : *L) {
            if (l.kind != L_CONV || !l.fuse_ok || !(l.fuse_b[0] | l.fuse_b[1] | l.fuse_b[2] | l.fuse_b[3])) continue;
            const float* lo = l.conv.dst; const float* hi = lo + (int64_t)(n->maxN / n->Tin() * std::max(1, l.conv.To) - 1) * l.conv.dst_nstride + (int64_t)l.conv.Cd * l.conv.Ho * l.conv.Wo;
            const float* vlo = v.p; const float* vhi = v.p + (int64_t)(frames - 1) * v.nstride + (int64_t)v.C * v.H * v.W;
            if (vlo < hi && lo < vhi)
                return fail("i2v_net_read_tensor: this tensor is the intermediate of a fused 3x3 -> pointwise pair and is never stored "
                            "(plan without I2V_FUSE / I2V_FORCE_FUSE to read it)");
        }
    // ... and so are the intermediates of a fused fast-pathway block (k_fastblock): every member's output but the last
    for (const std::vector<Launch>* L : {&n->fwd, &n->bwd})
        for (size_t i = 0; i < L->size(); ++i) {
            const Launch& l = (*L)[i];
            if (l.kind != L_CONV || !l.fb_ok || !(l.fb_b[0] | l.fb_b[1] | l.fb_b[2] | l.fb_b[3])) continue;
            for (int j = 0; j + 1 < l.fb_ok && i + j < L->size(); ++j) {
                const I2VConvParams& q = (*L)[i + j].conv;
                const float* lo = q.dst; const float* hi = lo + (int64_t)(n->maxN / n->Tin() * std::max(1, q.To) - 1) * q.dst_nstride + (int64_t)q.Cd * q.Ho * q.Wo;
                const float* vlo = v.p; const float* vhi = v.p + (int64_t)(frames - 1) * v.nstride + (int64_t)v.C * v.H * v.W;
                if (vlo < hi && lo < vhi)
                    return fail("i2v_net_read_tensor: this tensor is an intermediate of a fused fast-pathway block and is never stored "
                                "(plan with I2V_FASTBLOCK=0 to read it)");
            }
        }
    size_t row = (size_t)v.C * v.H * v.W * sizeof(float);
    CHECK_BE(be_d2d_2d(out, row, v.p, (size_t)v.nstride * sizeof(float), row, frames, stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// loop kernels
// ---------------------------------------------------------------------------------------------
extern "C" int i2v_clip_from_u8_f32(const uint8_t* frames, float* video, int b, int t, int hh, int w, void* stream) {
    if (!frames || !video || b <= 0 || t <= 0 || hh <= 0 || w <= 0) return fail("i2v_clip_from_u8_f32: bad argument");
    CHECK_BE(k_clip_from_u8(frames, video, b, t, hh, w, stream));
    return 0;
}

extern "C" int i2v_clip_resize_crop_u8_f32(const uint8_t* frames, float* video, const int32_t* xtab, const int32_t* ytab, int b, int t,
                                           int H, int W, int rh, int rw, int crop_y, int crop_x, int out_h, int out_w, void* stream) {
    if (!frames || !video || !xtab || !ytab || b <= 0 || t <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
        return fail("i2v_clip_resize_crop_u8_f32: bad argument");
    if (crop_y < 0 || crop_x < 0 || crop_y + out_h > rh || crop_x + out_w > rw) return fail("i2v_clip_resize_crop_u8_f32: crop window outside the resized frame");
    CHECK_BE(k_clip_resize_crop(frames, video, xtab, ytab, b, t, H, W, crop_y, crop_x, out_h, out_w, stream));
    return 0;
}

extern "C" int i2v_clip_resample_crop_u8_f32(const uint8_t* frames, float* video, const int32_t* xbounds, const int32_t* xcoef, int kx,
                                             const int32_t* ybounds, const int32_t* ycoef, int ky, int b, int t, int H, int W, int rh, int rw,
                                             int crop_y, int crop_x, int out_h, int out_w, void* stream) {
    if (!frames || !video || !xbounds || !xcoef || !ybounds || !ycoef || kx <= 0 || ky <= 0 || b <= 0 || t <= 0 || H <= 0 || W <= 0 || out_h <= 0 ||
        out_w <= 0) return fail("i2v_clip_resample_crop_u8_f32: bad argument");
    if (crop_y < 0 || crop_x < 0 || crop_y + out_h > rh || crop_x + out_w > rw) return fail("i2v_clip_resample_crop_u8_f32: crop window outside the resized frame");
    CHECK_BE(k_clip_resample_crop(frames, video, xbounds, xcoef, kx, ybounds, ycoef, ky, b, t, H, W, crop_y, crop_x, out_h, out_w, stream));
    return 0;
}

extern "C" int i2v_frames_from_video_f32(const float* video, float* x, float* u, int b, int f, int hh, int w,
                                         void* stream) {
    if (!video || !x || !u || b <= 0 || f <= 0 || hh <= 0 || w <= 0) return fail("i2v_frames_from_video_f32: bad argument");
    CHECK_BE(k_frames_from_video(video, x, u, b, f, hh, w, stream));
    return 0;
}

extern "C" int i2v_compose_f32(const float* u, const float* delta, float* x, int b, int f, int hh, int w,
                               float eps, int video_layout, void* stream) {
    if (!u || !delta || !x || b <= 0 || f <= 0 || hh <= 0 || w <= 0) return fail("i2v_compose_f32: bad argument");
    CHECK_BE(k_compose(u, delta, x, b, f, hh, w, eps, video_layout, stream));
    return 0;
}

extern "C" size_t i2v_cossim_scratch_bytes(int64_t D, int frames) {
    return ((size_t)frames * cos_nblk(D) * 4 + 2) * sizeof(double);
}

extern "C" int i2v_cossim_fwd_bwd_f32(const float* a, int64_t a_stride, const float* b, int64_t b_stride,
                                      int64_t D, int frames, const float* coef_dev, int coef_index,
                                      float coef_host, int mask_relu, int accumulate, float* cos_out,
                                      float* grad, int64_t grad_stride, void* scratch, void* stream) {
    if (!a || !b || !cos_out || !grad || !scratch || D <= 0 || frames <= 0) return fail("i2v_cossim_fwd_bwd_f32: bad argument");
    I2VCosParams p; memset(&p, 0, sizeof p);
    p.a = a; p.a_nstride = a_stride; p.b = b; p.b_nstride = b_stride; p.D = D; p.N = frames;
    p.partial = (float*)scratch; p.nblk = cos_nblk(D); p.cos_out = cos_out; p.grad = grad; p.grad_nstride = grad_stride;
    p.coef_dev = coef_dev; p.coef_index = coef_index; p.coef_host = coef_host; p.mask_relu = mask_relu; p.accumulate = accumulate;
    CHECK_BE(k_cos(p, stream));
    return 0;
}

static void std_params(I2VStdParams& p, const float* a, int64_t a_stride, int64_t D, int frames, void* scratch) {
    memset(&p, 0, sizeof p);
    p.a = a; p.a_nstride = a_stride; p.D = D; p.N = frames; p.nblk = cos_nblk(D);
    // scratch layout: [2] sums, then the per-block partials
    p.sums = (double*)scratch; p.partial = (double*)scratch + 2;
}

extern "C" int i2v_std_reduce_f32(const float* a, int64_t a_stride, int64_t D, int frames, void* scratch, void* stream) {
    if (!a || !scratch || D <= 0 || frames <= 0) return fail("i2v_std_reduce_f32: bad argument");
    I2VStdParams p; std_params(p, a, a_stride, D, frames, scratch);
    CHECK_BE(k_std_reduce(p, stream));
    return 0;
}

extern "C" int i2v_std_grad_f32(const float* a, int64_t a_stride, int64_t D, int frames, int64_t total_count,
                                int mask_relu, int accumulate, float* std_out, float* grad, int64_t grad_stride,
                                void* scratch, void* stream) {
    if (!a || !std_out || !grad || !scratch || D <= 0 || frames <= 0 || total_count < 2) return fail("i2v_std_grad_f32: bad argument");
    I2VStdParams p; std_params(p, a, a_stride, D, frames, scratch);
    p.total_count = (double)total_count; p.std_out = std_out; p.grad = grad; p.grad_nstride = grad_stride;
    p.mask_relu = mask_relu; p.accumulate = accumulate;
    CHECK_BE(k_std_grad(p, stream));
    return 0;
}

extern "C" int i2v_std_fwd_bwd_f32(const float* a, int64_t a_stride, int64_t D, int frames, int mask_relu,
                                   int accumulate, float* std_out, float* grad, int64_t grad_stride,
                                   void* scratch, void* stream) {
    if (i2v_std_reduce_f32(a, a_stride, D, frames, scratch, stream)) return 1;
    return i2v_std_grad_f32(a, a_stride, D, frames, (int64_t)frames * D, mask_relu, accumulate, std_out, grad,
                            grad_stride, scratch, stream);
}

extern "C" int i2v_adam_step_f32(float* delta, float* m, float* v, const float* gx, const float* u,
                                 int64_t frames, int hw, float eps, double lr, double beta1, double beta2,
                                 double adam_eps, int step_t, void* stream) {
    if (!delta || !m || !v || !gx || !u || frames <= 0 || hw <= 0 || step_t < 1) return fail("i2v_adam_step_f32: bad argument");
    // scalar prep in double, as torch/optim/adam.py does on the host for the non-capturable path
    const double bc1 = 1.0 - pow(beta1, step_t), bc2 = 1.0 - pow(beta2, step_t);
    CHECK_BE(k_adam(delta, m, v, gx, u, frames * 3 * (int64_t)hw, hw, eps, (float)(lr / bc1), (float)sqrt(bc2),
                    (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)adam_eps, stream));
    return 0;
}

extern "C" int i2v_sign_step_f32(float* adv, const float* u, const float* grad, int64_t nel, int64_t chan_stride,
                                 float step, float eps, void* stream) {
    if (!adv || !u || !grad || nel <= 0 || chan_stride <= 0) return fail("i2v_sign_step_f32: bad argument");
    CHECK_BE(k_sign_bim(adv, u, grad, nel, chan_stride, step, eps, stream));
    return 0;
}

extern "C" int i2v_sign_step_delta_f32(float* delta, const float* grad, int64_t nel, float step, void* stream) {
    if (!delta || !grad || nel <= 0) return fail("i2v_sign_step_delta_f32: bad argument");
    CHECK_BE(k_sign_delta(delta, grad, nel, step, stream));
    return 0;
}

extern "C" int i2v_sign_step_delta_gx_f32(float* delta, const float* gx, const float* u, int64_t nel, float eps,
                                          float step, void* stream) {
    if (!delta || !gx || !u || nel <= 0) return fail("i2v_sign_step_delta_gx_f32: bad argument");
    CHECK_BE(k_sign_delta_gx(delta, gx, u, nel, eps, step, stream));
    return 0;
}

static void ilaf_params(I2VIlafParams& p, const float* a, int64_t a_stride, const float* ori, const float* adv0,
                        int64_t D, int frames, void* scratch, int frames_per_seg = 0) {
    memset(&p, 0, sizeof p);
    p.a = a; p.a_nstride = a_stride; p.ori = ori; p.adv0 = adv0; p.D = D; p.N = frames; p.nblk = cos_nblk(D);
    p.fps = frames_per_seg;
    const int nseg = frames_per_seg > 0 ? frames / frames_per_seg : 1;
    p.sums = (double*)scratch; p.partial = (double*)scratch + 2 * nseg;  // [nseg][2] sums, then the per-(frame, block) partials
}

extern "C" int i2v_ilaf_reduce_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                                   int frames, void* scratch, void* stream) {
    if (!a || !ori || !adv0 || !scratch || D <= 0 || frames <= 0) return fail("i2v_ilaf_reduce_f32: bad argument");
    I2VIlafParams p; ilaf_params(p, a, a_stride, ori, adv0, D, frames, scratch);
    CHECK_BE(k_ilaf_reduce(p, stream));
    return 0;
}

extern "C" int i2v_ilaf_grad_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                                 int frames, double init_norm, int mask_relu, int accumulate, float* loss_out,
                                 float* grad, int64_t grad_stride, void* scratch, void* stream) {
    if (!a || !ori || !adv0 || !scratch || !loss_out || !grad || D <= 0 || frames <= 0 || !(init_norm > 0.0))
        return fail("i2v_ilaf_grad_f32: bad argument");
    I2VIlafParams p; ilaf_params(p, a, a_stride, ori, adv0, D, frames, scratch);
    p.init_norm = init_norm; p.mask_relu = mask_relu; p.accumulate = accumulate; p.loss_out = loss_out;
    p.grad = grad; p.grad_nstride = grad_stride;
    CHECK_BE(k_ilaf_grad(p, stream));
    return 0;
}

// K independent one-clip problems in one launch (segments of frames_per_seg frames): per-segment sums / losses; the initial
// norms come from device memory (the squared norms an initial `reduce` left), so the loop needs no read-back at all.
extern "C" size_t i2v_ilaf_scratch_bytes(int64_t D, int frames, int frames_per_seg) {
    const int nseg = frames_per_seg > 0 ? frames / frames_per_seg : 1;
    return ((size_t)2 * nseg + (size_t)2 * frames * cos_nblk(D)) * sizeof(double) + 64;
}

extern "C" int i2v_ilaf_reduce_seg_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                                       int frames, int frames_per_seg, void* scratch, void* stream) {
    if (!a || !ori || !adv0 || !scratch || D <= 0 || frames <= 0 || frames_per_seg <= 0 || frames % frames_per_seg)
        return fail("i2v_ilaf_reduce_seg_f32: bad argument");
    I2VIlafParams p; ilaf_params(p, a, a_stride, ori, adv0, D, frames, scratch, frames_per_seg);
    CHECK_BE(k_ilaf_reduce(p, stream));
    return 0;
}

extern "C" int i2v_ilaf_grad_seg_f32(const float* a, int64_t a_stride, const float* ori, const float* adv0, int64_t D,
                                     int frames, int frames_per_seg, const double* init_sq, int mask_relu, int accumulate,
                                     float* loss_out, float* grad, int64_t grad_stride, void* scratch, void* stream) {
    if (!a || !ori || !adv0 || !scratch || !loss_out || !grad || !init_sq || D <= 0 || frames <= 0 || frames_per_seg <= 0 ||
        frames % frames_per_seg)
        return fail("i2v_ilaf_grad_seg_f32: bad argument");
    I2VIlafParams p; ilaf_params(p, a, a_stride, ori, adv0, D, frames, scratch, frames_per_seg);
    p.init_sq = init_sq; p.mask_relu = mask_relu; p.accumulate = accumulate; p.loss_out = loss_out;
    p.grad = grad; p.grad_nstride = grad_stride;
    CHECK_BE(k_ilaf_grad(p, stream));
    return 0;
}

extern "C" int i2v_tap_distance_f32(const float* a, int64_t a_stride, const float* clean, int64_t D, int frames, int frames_per_seg,
                                    double coef, int mask_relu, int accumulate, float* dist_out, float* grad, int64_t grad_stride,
                                    void* scratch, void* stream) {
    if (!a || !clean || !scratch || !dist_out || !grad || D <= 0 || frames <= 0 || frames_per_seg <= 0 || frames % frames_per_seg)
        return fail("i2v_tap_distance_f32: bad argument");
    I2VIlafParams p; ilaf_params(p, a, a_stride, clean, clean, D, frames, scratch, frames_per_seg);
    p.mode = 1; p.coef = coef; p.mask_relu = mask_relu; p.accumulate = accumulate; p.loss_out = dist_out;
    p.grad = grad; p.grad_nstride = grad_stride;
    CHECK_BE(k_ilaf_reduce(p, stream));
    CHECK_BE(k_ilaf_grad(p, stream));
    return 0;
}

extern "C" size_t i2v_head_scratch_bytes(int C, int clips) { return (size_t)2 * C * clips * sizeof(float) + 64; }

static void head_feature(I2VHeadParams& p, const float* a, int64_t a_stride, int C, int HW, int T, int clips, int Ctot, int c_off, void* scratch) {
    memset(&p, 0, sizeof p);
    p.a = a; p.a_nstride = a_stride; p.C = C; p.HW = HW; p.T = T; p.clips = clips; p.Ctot = Ctot; p.c_off = c_off;
    p.pooled = (float*)scratch; p.dpooled = (float*)scratch + (size_t)Ctot * clips;
}

extern "C" int i2v_head_ce_f32(const float* a, int64_t a_stride, int C, int HW, int T, int clips, const float* W, const float* bias,
                               int K, const int32_t* labels, float scale, int mask_relu, int accumulate, float* logits, float* loss_each,
                               float* grad, int64_t grad_stride, void* scratch, void* stream) {
    if (!a || !W || !labels || !logits || !loss_each || !grad || !scratch || C <= 0 || HW <= 0 || T <= 0 || clips <= 0 || K <= 0)
        return fail("i2v_head_ce_f32: bad argument");
    I2VHeadParams p; head_feature(p, a, a_stride, C, HW, T, clips, C, 0, scratch);
    p.K = K; p.W = W; p.bias = bias; p.labels = labels; p.scale = scale; p.logits = logits; p.loss_each = loss_each;
    p.grad = grad; p.grad_nstride = grad_stride; p.mask_relu = mask_relu; p.accumulate = accumulate; p.phase = 7;
    CHECK_BE(k_head_ce(p, stream));
    return 0;
}

// The same head over SEVERAL features (SlowFast pools its two pathways separately and concatenates): pool every feature into
// its columns of the Ctot-wide vector, then one logits / loss call, then every feature's gradient.  scratch >=
// i2v_head_scratch_bytes(Ctot, clips), the same block in all three.
extern "C" int i2v_head_pool_f32(const float* a, int64_t a_stride, int C, int HW, int T, int clips, int Ctot, int c_off, void* scratch,
                                 void* stream) {
    if (!a || !scratch || C <= 0 || HW <= 0 || T <= 0 || clips <= 0 || c_off < 0 || c_off + C > Ctot) return fail("i2v_head_pool_f32: bad argument");
    I2VHeadParams p; head_feature(p, a, a_stride, C, HW, T, clips, Ctot, c_off, scratch); p.phase = 1;
    CHECK_BE(k_head_ce(p, stream));
    return 0;
}

extern "C" int i2v_head_logits_ce_f32(int Ctot, int clips, const float* W, const float* bias, int K, const int32_t* labels, float scale,
                                      float* logits, float* loss_each, void* scratch, void* stream) {
    if (!W || !labels || !logits || !loss_each || !scratch || Ctot <= 0 || clips <= 0 || K <= 0) return fail("i2v_head_logits_ce_f32: bad argument");
    I2VHeadParams p; head_feature(p, nullptr, 0, Ctot, 1, 1, clips, Ctot, 0, scratch);
    p.K = K; p.W = W; p.bias = bias; p.labels = labels; p.scale = scale; p.logits = logits; p.loss_each = loss_each; p.phase = 2;
    CHECK_BE(k_head_ce(p, stream));
    return 0;
}

extern "C" int i2v_head_grad_f32(const float* a, int64_t a_stride, int C, int HW, int T, int clips, int Ctot, int c_off, int mask_relu,
                                 int accumulate, float* grad, int64_t grad_stride, void* scratch, void* stream) {
    if (!a || !grad || !scratch || C <= 0 || HW <= 0 || T <= 0 || clips <= 0 || c_off < 0 || c_off + C > Ctot) return fail("i2v_head_grad_f32: bad argument");
    I2VHeadParams p; head_feature(p, a, a_stride, C, HW, T, clips, Ctot, c_off, scratch);
    p.grad = grad; p.grad_nstride = grad_stride; p.mask_relu = mask_relu; p.accumulate = accumulate; p.phase = 4;
    CHECK_BE(k_head_ce(p, stream));
    return 0;
}

extern "C" int i2v_tt_grad_mix_f32(const float* grads, float* out, const float* kernel, const int32_t* moves, int D, int64_t NC, int T, int HW,
                                  float weight, void* stream) {
    if (!grads || !out || !kernel || !moves || D <= 0 || D > 64 || NC <= 0 || T <= 0 || HW <= 0) return fail("i2v_tt_grad_mix_f32: bad argument");
    const float w1 = (float)(1.0 - (double)weight);           // python: (1 - self.weight) in double, then a float32 tensor scalar
    CHECK_BE(k_tt_grad_mix(grads, out, kernel, (const int*)moves, D, NC, T, HW, w1, weight, stream));
    return 0;
}

extern "C" int i2v_resample_nearest_f32(const float* src, float* dst, int64_t planes, int Hs, int Ws, int Hd, int Wd, const int32_t* map_y,
                                        const int32_t* map_x, void* stream) {
    if (!src || !dst || !map_y || !map_x || planes <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return fail("i2v_resample_nearest_f32: bad argument");
    CHECK_BE(k_resample_nearest(src, dst, planes, Hs, Ws, Hd, Wd, map_y, map_x, stream));
    return 0;
}

extern "C" int i2v_resample_nearest_bwd_f32(const float* g, float* gsrc, int64_t planes, int Hd, int Wd, int Hs, int Ws, const int32_t* ylo,
                                            const int32_t* yhi, const int32_t* xlo, const int32_t* xhi, void* stream) {
    if (!g || !gsrc || !ylo || !yhi || !xlo || !xhi || planes <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0)
        return fail("i2v_resample_nearest_bwd_f32: bad argument");
    CHECK_BE(k_resample_nearest_bwd(g, gsrc, planes, Hd, Wd, Hs, Ws, ylo, yhi, xlo, xhi, stream));
    return 0;
}

extern "C" int i2v_dwconv1d_f32(const float* src, float* dst, int64_t outer, int len, int64_t inner, const float* taps, int k, void* stream) {
    if (!src || !dst || src == dst || !taps || outer <= 0 || len <= 0 || inner <= 0 || k <= 0 || k > 64 || !(k & 1))
        return fail("i2v_dwconv1d_f32: bad argument (odd k <= 64, out of place)");
    CHECK_BE(k_dwconv1d(src, dst, outer, len, inner, taps, k, stream));
    return 0;
}

extern "C" int64_t i2v_grad_post_scratch_bytes(int b, int c, int f, int h, int w, int mode) {
    if (b <= 0 || c <= 0 || f <= 0 || h <= 0 || w <= 0 || mode < 0 || mode > 4) return 0;
    int64_t ge = 0; const int G = k_grad_post_groups(b, c, f, h, w, mode, &ge);
    return (int64_t)(G > 0 ? G : 1) * k_grad_post_splits(ge) * 8 + 64;
}

extern "C" int i2v_grad_post_f32(const float* g, float* momentum, float* out, int b, int c, int f, int h, int w, int frame_major, int mode,
                                 float decay, void* scratch, void* stream) {
    if (!g || !out || g == out || b <= 0 || c <= 0 || f <= 0 || h <= 0 || w <= 0 || mode < 0 || mode > 4 || (mode > 0 && !scratch) ||
        (int64_t)b * c * f * h * w >= (1ll << 31))
        return fail("i2v_grad_post_f32: bad argument (out of place, mode 0..4, < 2^31 elements)");
    CHECK_BE(k_grad_post(g, momentum, out, b, c, f, h, w, frame_major, mode, decay, (double*)scratch, stream));
    return 0;
}

extern "C" int64_t i2v_tap_scratch_bytes(int64_t) { return 1024 * 8 + 64; }

extern "C" int i2v_tap_perts_f32(const float* adv, const float* videos, float* out, int b, int c, int f, int h, int w, void* stream) {
    if (!adv || !videos || !out || b <= 0 || c != 3 || f <= 0 || h <= 0 || w <= 0) return fail("i2v_tap_perts_f32: bad argument (c == 3)");
    CHECK_BE(k_tap_perts(adv, videos, out, b, c, f, h, w, stream));
    return 0;
}

extern "C" int i2v_tap_sign_abs_f32(const float* smooth, float* sign_out, float* reg, int64_t n, void* scratch, void* stream) {
    if (!smooth || !sign_out || !reg || !scratch || n <= 0) return fail("i2v_tap_sign_abs_f32: bad argument");
    CHECK_BE(k_tap_sign_abs(smooth, sign_out, reg, n, (double*)scratch, stream));
    return 0;
}

extern "C" int i2v_tap_grad_f32(const float* gx, const float* boxsign, float* out, int b, int c, int f, int h, int w, float weight, void* stream) {
    if (!gx || !boxsign || !out || gx == out || b <= 0 || c != 3 || f <= 0 || h <= 0 || w <= 0) return fail("i2v_tap_grad_f32: bad argument (c == 3, out of place)");
    CHECK_BE(k_tap_grad(gx, boxsign, out, b, c, f, h, w, weight, stream));
    return 0;
}

extern "C" int i2v_aens_coeffs_f32(const float* prev, float* coeffs, float momentum, int L, void* stream) {
    if (!prev || !coeffs || L <= 0 || L > 64) return fail("i2v_aens_coeffs_f32: bad argument");
    CHECK_BE(k_aens_coeffs(prev, coeffs, momentum, L, stream));
    return 0;
}

extern "C" int i2v_aens_reduce_f32(const float* cos, const float* coeffs, int L, int frames, float* feat_sum,
                                   float* weighted, void* stream) {
    if (!cos || !coeffs || !feat_sum || !weighted || L <= 0 || frames <= 0) return fail("i2v_aens_reduce_f32: bad argument");
    CHECK_BE(k_aens_reduce(cos, coeffs, L, frames, feat_sum, weighted, stream));
    return 0;
}
