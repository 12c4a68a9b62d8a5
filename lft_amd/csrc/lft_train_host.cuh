// lft_train_host.cuh -- host orchestration of the fp32 training step (included inside lft_api.hip's anonymous
// namespace): tape layout, forward with saved activations, backward into one flat gradient buffer.
//
// Counterpart of autograd over reference model/LFT.py:52-83 as driven by train.py:89-107.  Parameter indices follow
// lft_amd/params.py:param_table (the reference's registration order); the flat gradient buffer holds the 78 gradients
// back to back in that order, so a data-parallel job needs ONE all-reduce per step.
#pragma once
#include "lft_train.cuh"

// ---- parameter indices ----
constexpr int P_CONV0 = 0, P_CONV = 1, P_LAYER0 = 4, P_PER_LAYER = 18, P_UP0 = 76, P_UP3 = 77;
enum { S_MLP = 0, S_N1W, S_N1B, S_INPROJ, S_OUT, S_N2W, S_N2B, S_FF1, S_FF2, S_LIN, A_N1W, A_N1B, A_INPROJ, A_OUT, A_N2W, A_N2B, A_FF1, A_FF2 };
inline int pidx(int l, int which) { return P_LAYER0 + P_PER_LAYER * l + which; }

struct ParamInfo { long long numel[LFT_NUM_PARAMS], off[LFT_NUM_PARAMS], total; };
inline ParamInfo param_info(int s) {
    ParamInfo pi;
    int i = 0;
    pi.numel[i++] = 576;
    for (int k = 0; k < 3; ++k) pi.numel[i++] = 64 * 576;
    for (int l = 0; l < kLayers; ++l) {
        const long long sp[10] = {128 * 576, 128, 128, 384 * 128, 128 * 128, 128, 128, 256 * 128, 128 * 256, 64 * 128};
        const long long an[8] = {64, 64, 192 * 64, 64 * 64, 64, 64, 128 * 64, 64 * 128};
        for (long long v : sp) pi.numel[i++] = v;
        for (long long v : an) pi.numel[i++] = v;
    }
    pi.numel[i++] = 64LL * s * s * 64;
    pi.numel[i++] = 576;
    long long o = 0;
    for (int k = 0; k < LFT_NUM_PARAMS; ++k) { pi.off[k] = o; o += pi.numel[k]; }
    pi.total = o;
    return pi;
}

// ---- packed weight views: every matrix the step multiplies by, in both orientations, as k_pack fragment streams ----
enum { VW_CONV_F = 0, VW_CONV_B = 3, VW_LAYER0 = 6, VW_PER_LAYER = 24, VW_UP_F = VW_LAYER0 + 4 * VW_PER_LAYER, VW_UP_B, VW_UPM, VW_UPM_B, VW_COUNT };
enum { MLP_F = 0, MLP_B, SIN_F, SQK_B, SUNUSED, SV_B, SOUT_F, SOUT_B, SFF1_F, SFF1_B, SFF2_F, SFF2_B, SLIN_F, SLIN_B,
       AIN_F, AQK_B, AV_B, AOUT_F, AOUT_B, AFF1_F, AFF1_B, AFF2_F, AFF2_B };
inline int vw(int l, int which) { return VW_LAYER0 + VW_PER_LAYER * l + which; }
struct WView { size_t frag0 = 0; int OT = 0, KS = 0, taps = 0; };
struct WViews { WView v[VW_COUNT]; size_t nfrags = 0; };

// Enumerates the views in storage order.  P may be null (layout only).  out(n), in(k) are the VIEW's output / contraction
// dims; element (n, k, tap) of the view is src[n*ld + k*kmul + kadd + tap*st].
WViews build_views(const float* const* P, int s, std::vector<PackOp>* ops) {
    WViews W;
    auto add = [&](int id, const float* src, int O, int I, int taps, int ld, int kmul, int kadd, int st) {
        WView& v = W.v[id];
        v.frag0 = W.nfrags; v.OT = O / 32; v.KS = I / 16; v.taps = taps;
        W.nfrags += (size_t)taps * v.OT * v.KS;
        if (ops)
            for (int t = 0; t < taps; ++t) {
                ops->push_back(lin_op(src, 0, O, ld, 0, I / 16, 0, 1.0f, kmul, kadd + t * st));
                ops->back().order = 1;
            }
    };
    auto fwd = [&](int id, const float* src, int O, int I, int row0 = 0) { add(id, src, O, I, 1, I, 1, row0 * I, 0); };
    auto bwd = [&](int id, const float* src, int O, int I, int row0 = 0) { add(id, src, I, O, 1, 1, I, row0 * I, 0); };   // W^T of rows row0..row0+O
    auto src = [&](int idx) { return P ? P[idx] : nullptr; };
    for (int i = 0; i < 3; ++i) add(VW_CONV_F + i, src(P_CONV + i), 64, 64, 9, 576, 9, 0, 1);
    for (int i = 0; i < 3; ++i) add(VW_CONV_B + i, src(P_CONV + i), 64, 64, 9, 9, 576, 0, 1);
    for (int l = 0; l < kLayers; ++l) {
        add(vw(l, MLP_F), src(pidx(l, S_MLP)), 128, 64, 9, 576, 9, 0, 1);
        add(vw(l, MLP_B), src(pidx(l, S_MLP)), 64, 128, 9, 9, 576, 0, 1);
        fwd(vw(l, SIN_F), src(pidx(l, S_INPROJ)), 384, 128);
        bwd(vw(l, SQK_B), src(pidx(l, S_INPROJ)), 256, 128, 0);        // (Wq | Wk)^T: d n = [dQ | dK] [Wq ; Wk]
        bwd(vw(l, SV_B), src(pidx(l, S_INPROJ)), 128, 128, 256);
        fwd(vw(l, SOUT_F), src(pidx(l, S_OUT)), 128, 128); bwd(vw(l, SOUT_B), src(pidx(l, S_OUT)), 128, 128);
        fwd(vw(l, SFF1_F), src(pidx(l, S_FF1)), 256, 128); bwd(vw(l, SFF1_B), src(pidx(l, S_FF1)), 256, 128);
        fwd(vw(l, SFF2_F), src(pidx(l, S_FF2)), 128, 256); bwd(vw(l, SFF2_B), src(pidx(l, S_FF2)), 128, 256);
        fwd(vw(l, SLIN_F), src(pidx(l, S_LIN)), 64, 128); bwd(vw(l, SLIN_B), src(pidx(l, S_LIN)), 64, 128);
        fwd(vw(l, AIN_F), src(pidx(l, A_INPROJ)), 192, 64);
        bwd(vw(l, AQK_B), src(pidx(l, A_INPROJ)), 128, 64, 0);
        bwd(vw(l, AV_B), src(pidx(l, A_INPROJ)), 64, 64, 128);
        fwd(vw(l, AOUT_F), src(pidx(l, A_OUT)), 64, 64); bwd(vw(l, AOUT_B), src(pidx(l, A_OUT)), 64, 64);
        fwd(vw(l, AFF1_F), src(pidx(l, A_FF1)), 128, 64); bwd(vw(l, AFF1_B), src(pidx(l, A_FF1)), 128, 64);
        fwd(vw(l, AFF2_F), src(pidx(l, A_FF2)), 64, 128); bwd(vw(l, AFF2_B), src(pidx(l, A_FF2)), 64, 128);
    }
    fwd(VW_UP_F, src(P_UP0), 64 * s * s, 64); bwd(VW_UP_B, src(P_UP0), 64 * s * s, 64);
    {   // overlap-add matrix of the final 3x3 conv (see k_pack / upm_entry): (s+2)^2 rows padded to 32-row tiles, 64 s^2 columns
        const int gp = (s + 2) * (s + 2), gt = (gp + 31) / 32;
        WView& v = W.v[VW_UPM];
        v.frag0 = W.nfrags; v.OT = gt; v.KS = 4 * s * s; v.taps = 1;
        W.nfrags += (size_t)gt * v.KS;
        if (ops) {
            PackOp m = lin_op(src(P_UP3), 0, gp, 0, 0, v.KS, 0, 1.0f);
            m.kind = 1; m.s = s; m.ntiles = gt; m.order = 1;
            ops->push_back(m);
        }
        WView& vb = W.v[VW_UPM_B];                    // its transpose: 64 s^2 rows, 32 gt columns (the padding columns are zero)
        vb.frag0 = W.nfrags; vb.OT = 2 * s * s; vb.KS = 2 * gt; vb.taps = 1;
        W.nfrags += (size_t)vb.OT * vb.KS;
        if (ops) {
            PackOp m = lin_op(src(P_UP3), 0, 64 * s * s, 0, 0, vb.KS, 0, 1.0f);
            m.kind = 2; m.s = s; m.order = 1;
            ops->push_back(m);
        }
    }
    return W;
}

// ---- tape: everything the backward pass re-reads (floats, offsets in floats) ----
struct AngTape { size_t n, qk, v, o, t1, m, hdn, y; };
struct SpaTape { size_t petok, tok, n, qk, v, o, t1, m, hdn, t2, y; };      // qk = [N][256]: Q | K
struct TrainLayout {
    size_t wp;                                   // packed weight views (build_views order), 512 floats per fragment
    size_t pe_ang, pe_spa, x0, c1, c2, c3, feat;
    AngTape ang[kLayers];
    SpaTape spa[kLayers];
    size_t body, act, skip;                      // act = lrelu(U) [N, 64 s^2]; skip = bicubic(lr)
    // backward scratch
    size_t bwd, bwd_floats, gu, stats, part, pgb;       // bwd: arena of the backward pass's gradient tensors (re-used as they die)
    size_t part_floats, total;                   // total in floats
    int rc = 0;                                  // status of the sizing (dry) run of the backward pass: non-zero = the layout is not usable
};
constexpr int kWgChunksMax = 512;                // ... raised up to this for small matrices (wgrad)
constexpr int kWgChunks = 128;                   // token chunks (= workgroups of 4 waves) of a weight-gradient launch
constexpr int kLnBlocks = 512;                   // workgroups (= partial rows) of a LayerNorm backward
constexpr int kTailWaves = 2048;                 // waves of the up-sampler / conv0 weight-gradient kernels
inline int wg_chunks(long long N) { return (int)std::min<long long>(kWgChunks, std::max<long long>(4, N / 512)); }   // workgroups of 4 waves, >= 128 tokens per wave

struct BwdSizes { size_t arena, part; int rc; };   // floats: peak live set of the gradient arena, high-water mark of the partial sums; rc of the dry run
BwdSizes bwd_sizes(const Dims& d);
TrainLayout train_layout(const Dims& d) {
    TrainLayout T;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o += (floats + 63) & ~(size_t)63; return r; };
    const size_t n = (size_t)d.ntok, ss = (size_t)d.s * d.s;
    T.wp = take(build_views(nullptr, d.s, nullptr).nfrags * 768);     // 3 KiB per fragment in the bf16x6 mode (2 KiB in the others)
    T.pe_ang = take((size_t)d.V * 64); T.pe_spa = take((size_t)d.hw * 64);
    T.x0 = take(n * 64); T.c1 = take(n * 64); T.c2 = take(n * 64); T.c3 = take(n * 64); T.feat = take(n * 64);
    for (int l = 0; l < kLayers; ++l) {
        AngTape& a = T.ang[l];
        a.n = take(n * 64); a.qk = take(n * 128); a.v = take(n * 64); a.o = take(n * 64); a.t1 = take(n * 64);
        a.m = take(n * 64); a.hdn = take(n * 128); a.y = take(n * 64);
        SpaTape& sp = T.spa[l];
        sp.petok = take((size_t)d.hw * 128);
        sp.tok = take(n * 128); sp.n = take(n * 128); sp.qk = take(n * 256); sp.v = take(n * 128);
        sp.o = take(n * 128); sp.t1 = take(n * 128); sp.m = take(n * 128); sp.hdn = take(n * 256); sp.t2 = take(n * 128);
        sp.y = take(n * 64);
    }
    T.body = take(n * 64); T.act = take(n * 64 * ss);
    T.skip = take((size_t)d.B * d.A * d.h * d.s * d.A * d.w * d.s);
    // The gradient tensors of the backward pass live in an arena and are released at their last use (train_backward: every
    // kernel of a pass runs on ONE stream, so a buffer may be handed out again as soon as its last reader is enqueued); the
    // arena is as large as the pass's peak live set, found by running the pass's own allocation sequence without launching
    // anything (bwd_sizes).  Round 2 gave every tensor a buffer of its own (40 KB per token) so that the weight-gradient
    // kernels could run on a second stream: that overlap bought 3 % and cost half of the tape.
    const BwdSizes bsz = bwd_sizes(d);
    T.rc = bsz.rc;                               // callers refuse a layout whose sizing run failed (the message is in lft_last_error)
    T.bwd_floats = bsz.arena;
    T.bwd = take(T.bwd_floats);
    T.gu = take(n * 64 * ss);
    T.stats = take(n * 8 * 3);
    // Partial sums (weight gradients per token chunk, LayerNorm rows, tails) are reduced at the end of every LAYER of the
    // backward pass (k_reduce_all) and the region is re-used by the next one: as large as the largest layer's, from the same dry
    // run (round 2 kept all parameters' partials at once: 0.6 GB at B = 8).
    T.part_floats = bsz.part;
    T.part = take(T.part_floats);
    T.pgb = 0;
    T.total = o;
    return T;
}

int run_pack_split(std::vector<PackOp>& ops, float* dst, int expect_frags, hipStream_t st, bool three = false) {
    int total = 0;
    size_t i = 0;
    while (i < ops.size()) {
        PackArgs a{};
        int nf = 0;
        while (i < ops.size() && a.nops < LFT_PACK_MAXOPS) {
            a.op[a.nops] = ops[i];
            a.op[a.nops].frag0 = nf;
            nf += ops[i].ntiles * ops[i].ksteps;
            ++a.nops; ++i;
        }
        if (three) k_pack_split<true><<<nf, 64, 0, st>>>(a, dst + (size_t)total * 768);
        else k_pack_split<false><<<nf, 64, 0, st>>>(a, dst + (size_t)total * 512);
        LFT_LAUNCH_OK("k_pack_split");
        total += nf;
    }
    if (total != expect_frags) return fail(LFT_ERR_ARG, "internal: stream has %d fragments, expected %d", total, expect_frags);
    return 0;
}

// ---- launch helpers ----
struct TrainCtx;
int red_push(const TrainCtx& c, size_t part_off, int nch, int n, int stride, float* dst, int chain);
struct TrainCtx {
    const Dims& d;
    float* tp;                 // tape base
    const TrainLayout& T;
    const WViews& W;
    hipStream_t st;
    int math;                  // LFT_MATH_F32 or LFT_MATH_BF16X3
    RedTab* red = nullptr;     // backward only: pending reductions (k_reduce_all) ...
    size_t* part_used = nullptr;   // ... and the next free float of the partial buffer
    size_t* part_peak = nullptr;   // its high-water mark (the dry run sizes the buffer with it)
    const float* gbase = nullptr;  // flat gradient buffer (segment destinations are offsets into it)
    bool dry = false;              // sizing pass: allocation sequence only, nothing is launched
    const bool* arena_overflow = nullptr;   // backward only: set by BwdArena::get when a request does not fit -- no kernel is launched after that
    float* F(size_t off) const { return tp + off; }
    int launch_ok() const {        // every launcher of the backward pass asks before it enqueues anything
        if (arena_overflow && *arena_overflow) return fail(LFT_ERR_ARG, "internal: backward arena overflow (a request did not fit the sized arena); nothing launched for it");
        return 0;
    }
};

// n floats of the partial-sum buffer (released as a whole by the next k_reduce_all)
int part_take(const TrainCtx& c, size_t n, size_t* off) {
    *off = *c.part_used;
    *c.part_used += n;
    if (c.part_peak && *c.part_used > *c.part_peak) *c.part_peak = *c.part_used;
    if (!c.dry && *c.part_used > c.T.part_floats) return fail(LFT_ERR_ARG, "internal: partial buffer overflow");
    return 0;
}
// Y[N][ldy cols o0..] = act(X W(view)^T) (+R).  ot0 / nOT select a block of the view's output tiles (taps == 1 only).
int run_lin(const TrainCtx& c, int view, int ot0, int nOT, const float* X, int ldx, int flip, int act, const float* R, int ldr,
            float* Y, int ldy, long long N, const float* M = nullptr, int mact = 0) {
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    const WView& v = c.W.v[view];
    if (nOT <= 0) nOT = v.OT;
    if ((v.taps != 1 && (ot0 || nOT != v.OT))) return fail(LFT_ERR_ARG, "run_lin: bad tile block (view %d)", view);
    const int mm = c.math == LFT_MATH_BF16X3 ? 1 : c.math == LFT_MATH_BF16X6 ? 2 : 0;
    LinP p{X, ldx, c.F(c.T.wp) + v.frag0 * (mm == 2 ? 768 : 512), v.OT, v.KS, ot0, R, ldr, Y, ldy, M, ldy, mact, v.taps, flip, act, c.d.h, c.d.w, N, 1};
    const unsigned gx = (unsigned)((N + 127) / 128);
    // output tiles per wave: 4 when that still gives the chip >= 2 waves per SIMD, fewer (more, thinner waves) for small N
    const long long tiles = (N + 31) / 32;
    int nt = 4;
    while (nt > 1 && (nOT % nt || tiles * (nOT / nt) < 2048)) nt >>= 1;
    const dim3 g(gx, (unsigned)(nOT / nt));
    // weights through the LDS ring (k_linr) when every workgroup's output block is one packed group of the view: full groups of
    // four tiles, or the view's last group of two; large N only (below, the coalesced-input form of k_lin is the faster one)
    const bool full4 = nt == 4 && ot0 % 4 == 0 && ot0 + nOT <= (v.OT / 4) * 4;
    const bool last2 = nt == 2 && nOT == 2 && v.OT % 4 == 2 && ot0 == (v.OT / 4) * 4;
    // k_linr's own preconditions, enforced here (a view that misses them takes k_lin): for 3x3 views its chunk stream is
    // addressed as tap * OT * KS + c * NT, i.e. the workgroup's NT tiles must be ALL of the view's output tiles (OT == nt); the
    // generic loop consumes two k-steps per iteration, so the stream length taps * KS must be even
    const bool ring = N > 65536 && (full4 || last2) && (v.taps == 1 || (nOT == v.OT && v.OT == nt)) && (v.taps * v.KS) % 2 == 0;
    const bool tiled = v.taps == 1 && v.KS % 4 == 0 && N <= 65536;       // measured: +7 % at 25.6 k tokens, -4 % at 205 k
#define LFT_LAUNCH_LIN(NTV)                                                                                          \
    do {                                                                                                             \
        if (mm == 1) { if (tiled) k_lin<NTV, 1, true><<<g, 256, 0, c.st>>>(p); else k_lin<NTV, 1, false><<<g, 256, 0, c.st>>>(p); }      \
        else if (mm == 2) { if (tiled) k_lin<NTV, 2, true><<<g, 256, 0, c.st>>>(p); else k_lin<NTV, 2, false><<<g, 256, 0, c.st>>>(p); } \
        else { if (tiled) k_lin<NTV, 0, true><<<g, 256, 0, c.st>>>(p); else k_lin<NTV, 0, false><<<g, 256, 0, c.st>>>(p); }     \
    } while (0)
    if (ring) {
        p.gy = nOT / nt;
        const dim3 g1((unsigned)((gx + 7) / 8 * 8 * p.gy));
        // 3x3 on 32-wide views (a wave's 32 tokens = one image row): the row fragments are loaded once per tap row (k_linr<.., KS>)
        const int ks3 = (v.taps == 9 && c.d.w == 32 && (v.KS == 4 || v.KS == 8)) ? v.KS : 0;
#define LFT_LAUNCH_R(NTV, KSV) do { if (mm == 1) k_linr<NTV, 1, KSV><<<g1, 256, 0, c.st>>>(p); else if (mm == 2) k_linr<NTV, 2, KSV><<<g1, 256, 0, c.st>>>(p); else k_linr<NTV, 0, KSV><<<g1, 256, 0, c.st>>>(p); } while (0)
        if (nt == 4) { if (ks3 == 4) LFT_LAUNCH_R(4, 4); else LFT_LAUNCH_R(4, 0); }     // (no 3x3 view has 128 inputs and 128 outputs)
        else { if (ks3 == 4) LFT_LAUNCH_R(2, 4); else if (ks3 == 8) LFT_LAUNCH_R(2, 8); else LFT_LAUNCH_R(2, 0); }
#undef LFT_LAUNCH_R
    } else if (nt == 4) LFT_LAUNCH_LIN(4); else if (nt == 2) LFT_LAUNCH_LIN(2); else LFT_LAUNCH_LIN(1);
#undef LFT_LAUNCH_LIN
    LFT_LAUNCH_OK(prof_name("k_lin", "k_lin:%d>%d%s%s%s", v.KS * 16, nOT * 32, v.taps == 9 ? " 3x3" : "", R ? " +R" : "", M ? " *M" : ""));
    return 0;
}
// Linear / conv forward through view `view` (all of its output rows, or tiles [ot0, ot0 + nOT)): Y = act(X W^T) (+R)
int lin_fwd(const TrainCtx& c, int view, const float* X, int act, const float* R, float* Y, long long N, int ot0 = 0, int nOT = 0) {
    const WView& v = c.W.v[view];
    const int Co = (nOT > 0 ? nOT : v.OT) * 32;
    return run_lin(c, view, ot0, nOT, X, v.KS * 16, 0, act, R, Co, Y, Co, N);
}
// Input gradient through a transposed view: dX = dY W (+R); 3x3 convs flip their taps
// (M, mact): dX is additionally multiplied by act'(.) read off the saved activation output M (same shape as dX)
int lin_bwd(const TrainCtx& c, int view, const float* dY, const float* R, float* dX, long long N, const float* M = nullptr, int mact = 0) {
    const WView& v = c.W.v[view];
    return run_lin(c, view, 0, 0, dY, v.KS * 16, 1, 0, R, v.OT * 32, dX, v.OT * 32, N, M, mact);
}
// weight gradient of either: dW (+)= dY^T X  (taps = 1 or 9)
int wgrad(const TrainCtx& c, const float* dY, int Co, const float* X, int Ci, int taps, float* dW, int accumulate, long long N) {
    if (Co % 32 || Ci % 64) return fail(LFT_ERR_ARG, "wgrad: Co %d / Ci %d not supported", Co, Ci);
    if (taps == 9 && Ci != 64) return fail(LFT_ERR_ARG, "wgrad: 3x3 with Ci %d not supported", Ci);   // all 3x3 convolutions of the network have Ci = 64
    const long long wsize = (long long)Co * Ci * taps;
    // Token chunks.  A chunk is gridy workgroups (one per output tile x input group x tap row); the launch should fill the chip in
    // WHOLE rounds of resident workgroups (MI355X: 256 CUs x 2 / 3 / 4 workgroups by the variant's registers): at the fixed 128
    // chunks a 64x64 weight put ONE wave on every SIMD, which alternated between waiting for its loads and multiplying (92 us
    // exact, 11 us of matrix work), while 1 024 workgroups on 768 slots ran a second round a third full.
    const int gridy = taps == 9 ? Co / 32 * 3 : Co / 32 * (Ci % 128 == 0 ? Ci / 128 : Ci / 64);
    const int slots = 256 * (taps == 9 ? 2 : Ci % 128 == 0 ? 3 : 4);
    const int base = wg_chunks(N), rounds = std::max(1, (base * gridy + slots - 1) / slots);
    int nch = std::min<long long>(std::min(kWgChunksMax, rounds * slots / gridy), std::max<long long>(base, N / 256));
    // split-bf16 products are 5x cheaper, the loop is bound by its loads and conversions and a workgroup's epilogue (LDS sum of
    // the four waves, partial image) weighs more: measured, more chunks pay only where the launch filled under a quarter of the chip
    if (c.math == LFT_MATH_BF16X3 && base * gridy * 4 > slots) nch = base;       // (bf16x6: measured, no difference either way)
    nch = std::max(nch, 1);
    long long len = (N + nch - 1) / nch;
    len = (len + 63) & ~63LL;
    if (((len >> 2) + 2 * c.d.w + 48) * (long long)std::max(Co, Ci) * 4 >= (1LL << 32))     // k_wgrad addresses a wave's tokens with 32-bit byte offsets
        return fail(LFT_ERR_SHAPE, "wgrad: %lld tokens per wave exceed the 32-bit offset range", len >> 2);
    size_t poff;
    int rc;
    if ((rc = part_take(c, (size_t)nch * wsize, &poff))) return rc;
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    hipStream_t ws = c.st;
    WgP p{dY, Co, X, Ci, c.F(c.T.part) + poff, wsize, Ci * taps, taps, 1, Co, Ci, taps, c.d.h, c.d.w, N, len, 1, nch, 1};
    const int mm = c.math == LFT_MATH_BF16X3 ? 1 : c.math == LFT_MATH_BF16X6 ? 2 : 0;
#define LFT_LAUNCH_WG(NIV, TXV)                                                                                        \
    do {                                                                                                               \
        const size_t lds = (size_t)3 * TXV * NIV * 16 * 64 * sizeof(float);                                            \
        const dim3 g((unsigned)((nch + 7) / 8 * 8 * p.gy * TXV));                                                       \
        if (mm == 1) { if ((rc = allow_lds(k_wgrad<NIV, 1, TXV>, lds, "k_wgrad"))) return rc; k_wgrad<NIV, 1, TXV><<<g, 256, lds, ws>>>(p); }   \
        else if (mm == 2) { if ((rc = allow_lds(k_wgrad<NIV, 2, TXV>, lds, "k_wgrad"))) return rc; k_wgrad<NIV, 2, TXV><<<g, 256, lds, ws>>>(p); }   \
        else { if ((rc = allow_lds(k_wgrad<NIV, 0, TXV>, lds, "k_wgrad"))) return rc; k_wgrad<NIV, 0, TXV><<<g, 256, lds, ws>>>(p); }    \
    } while (0)
    if (taps == 9) {
        p.igroups = 1; p.gy = Co / 32;
        LFT_LAUNCH_WG(2, 3);
    } else if (Ci % 128 == 0) {
        p.igroups = Ci / 128; p.gy = Co / 32 * p.igroups;
        LFT_LAUNCH_WG(4, 1);
    } else {
        p.igroups = Ci / 64; p.gy = Co / 32 * p.igroups;
        LFT_LAUNCH_WG(2, 1);
    }
#undef LFT_LAUNCH_WG
    LFT_LAUNCH_OK(prof_name("k_wgrad", "k_wgrad:%dx%d%s", Co, Ci, taps == 9 ? " 3x3" : ""));
    return red_push(c, poff, nch, (int)wsize, (int)wsize, dW, accumulate);
}
int red_push(const TrainCtx& c, size_t part_off, int nch, int n, int stride, float* dst, int chain) {
    RedTab& t = *c.red;
    if (chain) {                                      // second partial set for the most recent segment with this destination
        for (int i = t.nseg - 1; i >= 0; --i)
            if (t.s[i].dst_off == dst - c.gbase) { t.s[i].part2_off = (long long)part_off; t.s[i].nch2 = nch; return 0; }
        return fail(LFT_ERR_ARG, "internal: chained reduction without a first segment");
    }
    if (t.nseg >= kRedMax) return fail(LFT_ERR_ARG, "internal: reduction table full");
    RedSeg& sg = t.s[t.nseg++];
    sg.part_off = (long long)part_off; sg.part2_off = 0; sg.dst_off = dst - c.gbase;
    sg.nch = nch; sg.nch2 = 0; sg.n = n; sg.stride = stride; sg.blk0 = t.nblk;
    t.nblk += (n + 63) / 64;
    return 0;
}
int ln_fwd(const TrainCtx& c, int C, const float* X, const float* pe, int mode, const float* g, const float* b, float* Y, long long N) {
    if (C == 64) k_ln_fwd<64><<<blocks_for(N, 16), 256, 0, c.st>>>(X, pe, mode, g, b, Y, N, c.d.hw, c.d.V);
    else k_ln_fwd<128><<<blocks_for(N, 16), 256, 0, c.st>>>(X, pe, mode, g, b, Y, N, c.d.hw, c.d.V);
    LFT_LAUNCH_OK("k_ln_fwd");
    return 0;
}
// out = (add ? add : 0) + dLN/du ; dgamma, dbeta written
int ln_bwd(const TrainCtx& c, int C, const float* X, const float* pe, int mode, const float* g, const float* dY, const float* add,
           float* out, float* dgamma, float* dbeta, long long N) {
    const int nb = (int)std::min<long long>(kLnBlocks, (N + 15) / 16);
    size_t poff;
    int rc;
    if ((rc = part_take(c, (size_t)nb * 2 * C, &poff))) return rc;
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    float* pgb = c.F(c.T.part) + poff;
    if (C == 64) k_ln_bwd<64><<<nb, 256, 0, c.st>>>(X, pe, mode, g, dY, add, out, pgb, N, c.d.hw, c.d.V);
    else k_ln_bwd<128><<<nb, 256, 0, c.st>>>(X, pe, mode, g, dY, add, out, pgb, N, c.d.hw, c.d.V);
    LFT_LAUNCH_OK("k_ln_bwd");
    // partial rows are [dgamma(C) | dbeta(C)] and the two gradients are neighbours in the flat buffer (norm.weight, norm.bias)
    if (dbeta != dgamma + C) return fail(LFT_ERR_ARG, "internal: LayerNorm gradients are not adjacent");
    return red_push(c, poff, nb, 2 * C, 2 * C, dgamma, 0);
}
int act_bwd(const TrainCtx& c, const float* g, const float* y, float* out, long long n, int mode) {
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    k_act_bwd<<<blocks_for(n / 4, 256), 256, 0, c.st>>>(g, y, out, n / 4, mode);
    LFT_LAUNCH_OK("k_act_bwd");
    return 0;
}
int add3(const TrainCtx& c, float* out, const float* a, const float* b, long long n) {
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    k_add3<<<blocks_for(n / 4, 256), 256, 0, c.st>>>(out, a, b, n / 4);
    LFT_LAUNCH_OK("k_add3");
    return 0;
}
int add_to(const TrainCtx& c, float* a, const float* b, long long n) {
    k_add<<<blocks_for(n / 4, 256), 256, 0, c.st>>>(a, b, n / 4);
    LFT_LAUNCH_OK("k_add");
    return 0;
}
template <bool BWD>
int ang_attn(const TrainCtx& c, const float* QK, const float* Vv, float* O, const float* dO, float* dQK, float* dV) {
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    const int V = c.d.V, npix = c.d.B * c.d.hw;
    int rc;
    if (V <= 32) {
        const size_t lds = BWD ? (size_t)(4 * 8 * kAngHS<32> + 8 * (32 * 3 + 1)) * 4 : (size_t)(2 * 8 * kAngHS<32>) * 4;
        k_ang_attn<32, BWD><<<npix, 256, lds, c.st>>>(QK, Vv, O, dO, dQK, dV, V, c.d.hw);
    } else {
        const size_t lds = BWD ? (size_t)(4 * 8 * kAngHS<128> + 8 * (128 * 3 + 1)) * 4 : (size_t)(2 * 8 * kAngHS<128>) * 4;
        if ((rc = allow_lds(k_ang_attn<128, BWD>, lds, "k_ang_attn"))) return rc;
        k_ang_attn<128, BWD><<<npix, 1024, lds, c.st>>>(QK, Vv, O, dO, dQK, dV, V, c.d.hw);
    }
    LFT_LAUNCH_OK(prof_name("k_ang_attn", "k_ang_attn:%s", BWD ? "backward" : "forward"));
    return 0;
}

template <int MODE>
int win_attn(const TrainCtx& c, const float* Q, const float* K, const float* V, float* O, const float* dO, float* dQ, float* dK, float* dV) {   // Q | K and dQ | dK: [N][256]
    if (c.dry) return 0;
    if (int ok_ = c.launch_ok()) return ok_;
    const Dims& d = c.d;
    const unsigned tiles = (unsigned)(((d.w + kWaTX - 1) / kWaTX) * ((d.h + kWaTY - 1) / kWaTY) * d.B * d.V);
    int rc;
    if ((rc = allow_lds(k_win_attn_lds<MODE>, kWaLds, "k_win_attn_lds"))) return rc;
    k_win_attn_lds<MODE><<<dim3(tiles, 4), 256, kWaLds, c.st>>>(Q, K, V, O, dO, dQ, dK, dV, c.F(c.T.stats), d.h, d.w, 256);
    LFT_LAUNCH_OK(prof_name("k_win_attn_lds", "k_win_attn_lds:%s", MODE == 0 ? "forward" : MODE == 1 ? "bwd A (dQ)" : "bwd B (dK dV)"));
    return 0;
}

// ---------------------------------------------------------------------------- forward with tape
int train_forward(const float* const* P, const float* lr, float* out, float* tape, const Dims& d, int math, hipStream_t st) {
    const TrainLayout T = train_layout(d);
    if (T.rc) return T.rc;
    std::vector<PackOp> ops;
    const WViews WV = build_views(P, d.s, &ops);
    const TrainCtx c{d, tape, T, WV, st, math};
    const long long N = d.ntok;
    const int nimg = d.B * d.V;
    int rc;
#define TRY(x) do { if ((rc = (x))) return rc; } while (0)
    if (math == LFT_MATH_BF16X3 || math == LFT_MATH_BF16X6) TRY(run_pack_split(ops, c.F(T.wp), (int)WV.nfrags, st, math == LFT_MATH_BF16X6));
    else TRY(run_pack<float>(ops, c.F(T.wp), (int)WV.nfrags, st));   // both orientations of every matrix, for this step's weights
    k_pe_plain<<<blocks_for(std::max(d.V, d.hw) * 64, 256), 256, 0, st>>>(c.F(T.pe_ang), c.F(T.pe_spa), d.V, d.h, d.w);
    LFT_LAUNCH_OK("k_pe_plain");
    // conv_init0, conv_init + residual (LFT.py:65-66)
    k_conv0<float><<<dim3((unsigned)((d.hw + kConv0Tok - 1) / kConv0Tok), (unsigned)nimg), 256, 0, st>>>(lr, P[P_CONV0], c.F(T.x0), d.B, d.A, d.h, d.w);
    LFT_LAUNCH_OK("k_conv0");
    TRY(lin_fwd(c, VW_CONV_F + 0, c.F(T.x0), 2, nullptr, c.F(T.c1), N));
    TRY(lin_fwd(c, VW_CONV_F + 1, c.F(T.c1), 2, nullptr, c.F(T.c2), N));
    TRY(lin_fwd(c, VW_CONV_F + 2, c.F(T.c2), 2, nullptr, c.F(T.c3), N));
    LFT_HIP_OK(hipMemcpyAsync(c.F(T.feat), c.F(T.c3), (size_t)N * 64 * 4, hipMemcpyDeviceToDevice, st));
    TRY(add_to(c, c.F(T.feat), c.F(T.x0), N * 64));
    const float* x = c.F(T.feat);
    for (int l = 0; l < kLayers; ++l) {
        // ---- AngTrans (LFT.py:225-238) ----
        const AngTape& a = T.ang[l];
        TRY(ln_fwd(c, 64, x, c.F(T.pe_ang), 1, P[pidx(l, A_N1W)], P[pidx(l, A_N1B)], c.F(a.n), N));
        TRY(lin_fwd(c, vw(l, AIN_F), c.F(a.n), 0, nullptr, c.F(a.qk), N, 0, 4));                 // Q | K from the normed tokens
        TRY(lin_fwd(c, vw(l, AIN_F), x, 0, nullptr, c.F(a.v), N, 4, 2));               // V from the raw tokens
        TRY(ang_attn<false>(c, c.F(a.qk), c.F(a.v), c.F(a.o), nullptr, nullptr, nullptr));
        TRY(lin_fwd(c, vw(l, AOUT_F), c.F(a.o), 0, x, c.F(a.t1), N));
        TRY(ln_fwd(c, 64, c.F(a.t1), nullptr, 0, P[pidx(l, A_N2W)], P[pidx(l, A_N2B)], c.F(a.m), N));
        TRY(lin_fwd(c, vw(l, AFF1_F), c.F(a.m), 1, nullptr, c.F(a.hdn), N));
        TRY(lin_fwd(c, vw(l, AFF2_F), c.F(a.hdn), 0, c.F(a.t1), c.F(a.y), N));
        x = c.F(a.y);
        // ---- SpaTrans (LFT.py:176-191) ----
        const SpaTape& sp = T.spa[l];
        TRY(lin_fwd(c, vw(l, MLP_F), x, 0, nullptr, c.F(sp.tok), N));
        TRY(lin_fwd(c, vw(l, MLP_F), c.F(T.pe_spa), 0, nullptr, c.F(sp.petok), d.hw));   // LFT.py:180
        TRY(ln_fwd(c, 128, c.F(sp.tok), c.F(sp.petok), 2, P[pidx(l, S_N1W)], P[pidx(l, S_N1B)], c.F(sp.n), N));
        TRY(lin_fwd(c, vw(l, SIN_F), c.F(sp.n), 0, nullptr, c.F(sp.qk), N, 0, 8));                 // Q | K in one pass over n
        TRY(lin_fwd(c, vw(l, SIN_F), c.F(sp.tok), 0, nullptr, c.F(sp.v), N, 8, 4));
        TRY(win_attn<0>(c, c.F(sp.qk), c.F(sp.qk) + 128, c.F(sp.v), c.F(sp.o), nullptr, nullptr, nullptr, nullptr));
        TRY(lin_fwd(c, vw(l, SOUT_F), c.F(sp.o), 0, c.F(sp.tok), c.F(sp.t1), N));
        TRY(ln_fwd(c, 128, c.F(sp.t1), nullptr, 0, P[pidx(l, S_N2W)], P[pidx(l, S_N2B)], c.F(sp.m), N));
        TRY(lin_fwd(c, vw(l, SFF1_F), c.F(sp.m), 1, nullptr, c.F(sp.hdn), N));
        TRY(lin_fwd(c, vw(l, SFF2_F), c.F(sp.hdn), 0, c.F(sp.t1), c.F(sp.t2), N));
        const bool last = l == kLayers - 1;                                               // + global skip, LFT.py:76
        TRY(lin_fwd(c, vw(l, SLIN_F), c.F(sp.t2), 0, last ? c.F(T.feat) : nullptr, last ? c.F(T.body) : c.F(sp.y), N));
        x = last ? c.F(T.body) : c.F(sp.y);
    }
    // ---- up-sampler + bicubic skip (LFT.py:79-81) ----
    TRY(lin_fwd(c, VW_UP_F, c.F(T.body), 2, nullptr, c.F(T.act), N));
    // final 3x3 conv over the mosaic + bicubic skip: overlap-add footprints G = M lrelu(U) (backward scratch gu holds them), then gather
    const int gt = (d.gp + 31) / 32;
    TRY(lin_fwd(c, VW_UPM, c.F(T.act), 0, nullptr, c.F(T.gu), N, 0, gt));
    launch_assemble(lr, c.F(T.gu), out, d.B, d.A, d.h, d.w, d.s, st, 32 * gt);
    LFT_LAUNCH_OK("k_assemble_t");
    return 0;
}

// ---------------------------------------------------------------------------- backward
// Gradient buckets, in the order the backward pass finishes them: contiguous ranges of the flat gradient buffer
// (state_dict order: conv_init0, conv_init, altblock.0 .. 3, upsampling), so a data-parallel job can all-reduce bucket b
// while the kernels of bucket b+1 run.
static_assert(LFT_GRAD_BUCKETS == 3, "bucket ranges below");
void grad_bucket_range(int s, int bucket, size_t* first, size_t* count) {
    const ParamInfo pi = param_info(s);
    const size_t cut1 = pi.off[4 + 18 * 2], cut0 = pi.off[4];        // altblock.2 / altblock.0 start
    if (bucket == 0) { *first = cut1; *count = pi.total - cut1; }        // altblock.2, altblock.3, upsampling: ready after layer 2
    else if (bucket == 1) { *first = cut0; *count = cut1 - cut0; }      // altblock.0, altblock.1
    else { *first = 0; *count = cut0; }                                  // conv_init0, conv_init
}
typedef int (*BucketFn)(void* user, int bucket, size_t first_float, size_t n_floats);

// First-fit arena over the tape's backward region (floats).  Host-side bookkeeping only: the kernels of a pass run on one stream,
// so a range may be handed out again as soon as the last kernel reading it has been ENQUEUED.
struct BwdArena {
    float* base = nullptr;
    size_t cap = 0, peak = 0;
    bool overflow = false;                           // a request went past `cap`: the pointer handed out is the arena's base (valid
                                                     // memory) and TrainCtx::launch_ok() stops every launcher before it enqueues
    std::vector<std::pair<size_t, size_t>> used;     // (offset, floats), sorted by offset
    float* get(size_t n) {
        n = (n + 63) & ~(size_t)63;
        size_t off = 0, i = 0;
        for (; i < used.size(); ++i) {
            if (used[i].first - off >= n) break;
            off = used[i].first + used[i].second;
        }
        used.insert(used.begin() + (long)i, std::make_pair(off, n));
        peak = std::max(peak, off + n);
        if (off + n > cap) { overflow = true; return base; }
        return base + off;
    }
    void put(const float* p) {
        const size_t off = (size_t)(p - base);
        for (size_t i = 0; i < used.size(); ++i)
            if (used[i].first == off) { used.erase(used.begin() + (long)i); return; }
    }
};

// dry: the allocation sequence only (P, lr, tape, dout, G may be null) -- returns the arena's peak through *peak_out.
// One block of the pass on its own (lft_train_block_backward): its incoming gradient comes from the caller, its outgoing gradient
// goes to the caller, only its own parameter gradients are produced.  The tape must hold a full forward.
struct BlockSel { int block, layer; const float* d_out; float* d_in; };
int train_backward(const float* const* P, const float* lr, float* tape, const float* dout, float* G, const Dims& d, int math,
                   hipStream_t st, BucketFn on_bucket = nullptr, void* user = nullptr,
                   bool dry = false, size_t* peak_out = nullptr, size_t* part_peak_out = nullptr, const BlockSel* sel = nullptr) {
    TrainLayout Tdry{};
    const TrainLayout T = dry ? Tdry : train_layout(d);
    if (T.rc) return T.rc;
    const WViews WV = build_views(nullptr, d.s, nullptr);            // packed by this step's lft_train_forward
    RedTab red{};                                                    // every partial-sum producer registers a segment here
    size_t part_used = 0, part_peak = 0;
    static float dummy_base[64];
    TrainCtx c{d, dry ? dummy_base : tape, T, WV, st, math, &red, &part_used, &part_peak, G};
    c.dry = dry;
    const ParamInfo pi = param_info(d.s);
    const long long N = d.ntok;
    const int ss = d.s * d.s, nimg = d.B * d.V;
    int rc;
    auto g = [&](int idx) { return G + pi.off[idx]; };
    BwdArena A;
    A.base = c.F(T.bwd);
    A.cap = dry ? (size_t)-1 : T.bwd_floats;
    c.arena_overflow = &A.overflow;
    auto nb = [&](int width) { return A.get((size_t)N * width); };   // [N][width] gradient buffer; A.put() at its last use
    // End of a gradient bucket: the partial sums registered since the last bucket are reduced (one table-driven launch) and
    // their region is free again, then the caller is told -- everything enqueued before the callback belongs to the bucket,
    // nothing after it touches the bucket's range of G.
    int red_done = 0;
    float* dM = nullptr;
    auto flush = [&]() -> int {                                      // reduce the partial sums registered so far; their region is free again
        if (dry) { part_used = 0; return 0; }
        RedTab sub{};
        for (int i = red_done; i < red.nseg; ++i) {
            sub.s[sub.nseg] = red.s[i];
            sub.s[sub.nseg].blk0 = sub.nblk;
            sub.nblk += (red.s[i].n + 63) / 64;
            ++sub.nseg;
        }
        red_done = red.nseg;
        if (sub.nseg) {
            k_reduce_all<<<sub.nblk, 256, 0, st>>>(sub, c.F(T.part), G);
            LFT_LAUNCH_OK("k_reduce_all");
        }
        part_used = 0;                                               // the next layer's partial sums overwrite these (stream order)
        return 0;
    };
    auto end_bucket = [&](int bucket) -> int {
        if (int frc = flush()) return frc;
        if (dry) return 0;
        if (bucket == 0) {                                           // upsampling.3.weight: the reduced dM folded onto the 3x3 taps
            k_upm_fold<<<3, 256, 0, st>>>(dM, g(P_UP3), d.s);
            LFT_LAUNCH_OK("k_upm_fold");
        }
        if (on_bucket) {
            size_t first, count;
            grad_bucket_range(d.s, bucket, &first, &count);
            if (on_bucket(user, bucket, first, count) != 0)              // the caller asked to stop: enqueue nothing further
                return fail(LFT_ERR_CALLBACK, "on_bucket asked to stop at bucket %d", bucket);
        }
        return 0;
    };
#define TRY(x) do { if ((rc = (x))) return rc; } while (0)
    // Block selection (sel != null): `want(block, layer)` says whether a section runs; a selected section takes its incoming gradient
    // from the caller and hands its result back through `give` (a device copy on the pass's stream), then the partial sums are reduced.
    auto want = [&](int block, int layer) { return !sel || (sel->block == block && (block == LFT_BLOCK_UPSAMPLE || block == LFT_BLOCK_INIT || sel->layer == layer)); };
    auto give = [&](const float* src, size_t floats) -> int {
        if (!sel || !sel->d_in) return 0;
        LFT_HIP_OK(hipMemcpyAsync(sel->d_in, src, floats * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    };
    float* gskip = nullptr;
    const float* dy = sel ? sel->d_out : nullptr;
    if (want(LFT_BLOCK_UPSAMPLE, 0)) {
    // ---- up-sampler tail (the transpose of its forward: gather, GEMM with the overlap-add matrix) ----
    float* gu = c.F(T.gu);
    const int gt = (d.gp + 31) / 32;
    float* dG = nb(32 * gt);
    if (!dry) {
        k_up_gather_bwd<<<blocks_for(N * 32 * gt, 256), 256, 0, st>>>(dout, dG, d.B, d.A, d.h, d.w, d.s, 32 * gt);
        LFT_LAUNCH_OK("k_up_gather_bwd");
    }
    dM = A.get((size_t)32 * gt * 64 * ss);                           // [32 gt][64 s^2], folded onto upsampling.3.weight at the end of bucket 0
    TRY(wgrad(c, dG, 32 * gt, c.F(T.act), 64 * ss, 1, dM, 0, N));
    TRY(run_lin(c, VW_UPM_B, 0, 0, dG, 32 * gt, 0, 0, nullptr, 0, gu, 64 * ss, N, c.F(T.act), 2));   // dU = (M^T dG) * lrelu'(U)
    A.put(dG);
    TRY(wgrad(c, gu, 64 * ss, c.F(T.body), 64, 1, g(P_UP0), 0, N));
    gskip = nb(64);                                                  // d body = d y3 = d feat (global skip): lives to the end of the pass
    TRY(lin_bwd(c, VW_UP_B, gu, nullptr, gskip, N));
    dy = gskip;
    if (sel) {                                                       // the block on its own: fold dM now (the full pass does it at the end of bucket 0)
        TRY(flush());
        k_upm_fold<<<3, 256, 0, st>>>(dM, g(P_UP3), d.s);
        LFT_LAUNCH_OK("k_upm_fold");
        TRY(give(gskip, (size_t)N * 64));
    }
    }
    for (int l = kLayers - 1; l >= 0; --l) {
        // ================= SpaTrans backward: dy [N,64] -> dx =================
        float* dx = nullptr;
        if (want(LFT_BLOCK_SPA, l)) {
            const SpaTape& sp = T.spa[l];
            const float* xin = c.F(T.ang[l].y);
            float* gin = g(pidx(l, S_INPROJ));
            TRY(wgrad(c, dy, 64, c.F(sp.t2), 128, 1, g(pidx(l, S_LIN)), 0, N));
            float* dt2 = nb(128);
            TRY(lin_bwd(c, vw(l, SLIN_B), dy, nullptr, dt2, N));
            if (dy != gskip) A.put(dy);
            TRY(wgrad(c, dt2, 128, c.F(sp.hdn), 256, 1, g(pidx(l, S_FF2)), 0, N));
            float* dhz = nb(256);
            TRY(lin_bwd(c, vw(l, SFF2_B), dt2, nullptr, dhz, N, c.F(sp.hdn), 1));           // d (W1 m) = d hdn * relu'()
            TRY(wgrad(c, dhz, 256, c.F(sp.m), 128, 1, g(pidx(l, S_FF1)), 0, N));
            float* dm = nb(128);
            TRY(lin_bwd(c, vw(l, SFF1_B), dhz, nullptr, dm, N));
            A.put(dhz);
            float* dt1 = nb(128);
            TRY(ln_bwd(c, 128, c.F(sp.t1), nullptr, 0, P ? P[pidx(l, S_N2W)] : nullptr, dm, dt2, dt1, g(pidx(l, S_N2W)), g(pidx(l, S_N2B)), N));
            A.put(dm); A.put(dt2);
            TRY(wgrad(c, dt1, 128, c.F(sp.o), 128, 1, g(pidx(l, S_OUT)), 0, N));
            float* dO = nb(128);
            TRY(lin_bwd(c, vw(l, SOUT_B), dt1, nullptr, dO, N));
            float* dqk = nb(256);
            float* dV = nb(128);
            TRY(win_attn<1>(c, c.F(sp.qk), c.F(sp.qk) + 128, c.F(sp.v), nullptr, dO, dqk, nullptr, nullptr));        // dQ (+ row stats)
            TRY(win_attn<2>(c, c.F(sp.qk), c.F(sp.qk) + 128, c.F(sp.v), nullptr, dO, nullptr, dqk + 128, dV));      // dK, dV
            A.put(dO);
            TRY(wgrad(c, dV, 128, c.F(sp.tok), 128, 1, gin + 256 * 128, 0, N));
            float* dtokA = nb(128);
            TRY(lin_bwd(c, vw(l, SV_B), dV, dt1, dtokA, N));                                 // d tok = d t1 + dV Wv ...
            A.put(dV); A.put(dt1);
            TRY(wgrad(c, dqk, 256, c.F(sp.n), 128, 1, gin, 0, N));                           // rows 0..255 of in_proj: Wq, Wk
            float* dn = nb(128);
            TRY(lin_bwd(c, vw(l, SQK_B), dqk, nullptr, dn, N));
            A.put(dqk);
            float* du = nb(128);
            TRY(ln_bwd(c, 128, c.F(sp.tok), c.F(sp.petok), 2, P ? P[pidx(l, S_N1W)] : nullptr, dn, nullptr, du, g(pidx(l, S_N1W)), g(pidx(l, S_N1B)), N));  // d(tok+pe)
            A.put(dn);
            float* dpe = A.get((size_t)d.hw * 128);
            if (!dry) {
                k_sum_images<<<blocks_for((long long)d.hw * 128, 256), 256, 0, st>>>(du, nimg, (long long)d.hw * 128, dpe);
                LFT_LAUNCH_OK("k_sum_images");
            }
            float* dtok = nb(128);
            TRY(add3(c, dtok, dtokA, du, N * 128));                                          // ... + d(tok+pe)
            A.put(dtokA); A.put(du);
            TRY(wgrad(c, dtok, 128, xin, 64, 9, g(pidx(l, S_MLP)), 0, N));
            TRY(wgrad(c, dpe, 128, c.F(T.pe_spa), 64, 9, g(pidx(l, S_MLP)), 1, d.hw));      // the position tokens are MLP(unfold(PE)) too (LFT.py:180)
            A.put(dpe);
            dx = nb(64);
            TRY(lin_bwd(c, vw(l, MLP_B), dtok, nullptr, dx, N));
            A.put(dtok);
            if (sel) { TRY(flush()); TRY(give(dx, (size_t)N * 64)); }
        }
        // ================= AngTrans backward: dx -> dy of the layer below =================
        if (want(LFT_BLOCK_ANG, l)) {
            if (sel) {                                               // the block on its own: its incoming gradient is the caller's (copied: the section releases dx)
                dx = nb(64);
                LFT_HIP_OK(hipMemcpyAsync(dx, sel->d_out, (size_t)N * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            const AngTape& a = T.ang[l];
            const float* xin = l == 0 ? c.F(T.feat) : c.F(T.spa[l - 1].y);
            float* gin = g(pidx(l, A_INPROJ));
            TRY(wgrad(c, dx, 64, c.F(a.hdn), 128, 1, g(pidx(l, A_FF2)), 0, N));
            float* dhz = nb(128);
            TRY(lin_bwd(c, vw(l, AFF2_B), dx, nullptr, dhz, N, c.F(a.hdn), 1));             // d (W1 m) = d hdn * relu'()
            TRY(wgrad(c, dhz, 128, c.F(a.m), 64, 1, g(pidx(l, A_FF1)), 0, N));
            float* dm = nb(64);
            TRY(lin_bwd(c, vw(l, AFF1_B), dhz, nullptr, dm, N));
            A.put(dhz);
            float* dt1 = nb(64);
            TRY(ln_bwd(c, 64, c.F(a.t1), nullptr, 0, P ? P[pidx(l, A_N2W)] : nullptr, dm, dx, dt1, g(pidx(l, A_N2W)), g(pidx(l, A_N2B)), N));
            A.put(dm); A.put(dx);
            TRY(wgrad(c, dt1, 64, c.F(a.o), 64, 1, g(pidx(l, A_OUT)), 0, N));
            float* dO = nb(64);
            TRY(lin_bwd(c, vw(l, AOUT_B), dt1, nullptr, dO, N));
            float* dqk = nb(128);
            float* dV = nb(64);
            TRY(ang_attn<true>(c, c.F(a.qk), c.F(a.v), nullptr, dO, dqk, dV));
            A.put(dO);
            TRY(wgrad(c, dV, 64, xin, 64, 1, gin + 128 * 64, 0, N));
            float* dxa = nb(64);
            TRY(lin_bwd(c, vw(l, AV_B), dV, dt1, dxa, N));                                   // d x = d t1 + dV Wv ...
            A.put(dV); A.put(dt1);
            TRY(wgrad(c, dqk, 128, c.F(a.n), 64, 1, gin, 0, N));
            float* dn = nb(64);
            TRY(lin_bwd(c, vw(l, AQK_B), dqk, nullptr, dn, N));
            A.put(dqk);
            float* dxo = nb(64);
            TRY(ln_bwd(c, 64, xin, c.F(T.pe_ang), 1, P ? P[pidx(l, A_N1W)] : nullptr, dn, dxa, dxo, g(pidx(l, A_N1W)), g(pidx(l, A_N1B)), N));   // ... + d LN(x + PE)
            A.put(dn); A.put(dxa);
            dy = dxo;
            if (sel) { TRY(flush()); TRY(give(dxo, (size_t)N * 64)); }
        }
        if (sel) continue;
        if (l == 2) { TRY(end_bucket(0)); A.put(dM); }
        else if (l == 0) TRY(end_bucket(1));
        else TRY(flush());
    }
    if (!want(LFT_BLOCK_INIT, 0)) {                                  // a single block other than the feature extractor: done
        if (peak_out) *peak_out = A.peak;
        if (part_peak_out) *part_peak_out = part_peak;
        return 0;
    }
    // ---- initial feature extractor ----
    float* dfeat = nb(64);
    if (sel) {                                                       // the block on its own: d feat is the caller's
        LFT_HIP_OK(hipMemcpyAsync(dfeat, sel->d_out, (size_t)N * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        TRY(add3(c, dfeat, dy, gskip, N * 64));
        A.put(dy); A.put(gskip);
    }
    float* dz3 = nb(64);
    TRY(act_bwd(c, dfeat, c.F(T.c3), dz3, N * 64, 2));
    TRY(wgrad(c, dz3, 64, c.F(T.c2), 64, 9, g(P_CONV + 2), 0, N));
    float* dz2 = nb(64);
    TRY(lin_bwd(c, VW_CONV_B + 2, dz3, nullptr, dz2, N, c.F(T.c2), 2));                    // d z2 = d c2 * lrelu'()
    A.put(dz3);
    TRY(wgrad(c, dz2, 64, c.F(T.c1), 64, 9, g(P_CONV + 1), 0, N));
    float* dz1 = nb(64);
    TRY(lin_bwd(c, VW_CONV_B + 1, dz2, nullptr, dz1, N, c.F(T.c1), 2));                    // d z1 = d c1 * lrelu'()
    A.put(dz2);
    TRY(wgrad(c, dz1, 64, c.F(T.x0), 64, 9, g(P_CONV + 0), 0, N));
    float* dx0 = nb(64);
    TRY(lin_bwd(c, VW_CONV_B + 0, dz1, dfeat, dx0, N));                                    // d x0 = d feat + conv path
    A.put(dz1); A.put(dfeat);
    size_t poff0;
    TRY(part_take(c, (size_t)kTailWaves * 576, &poff0));
    if (peak_out) *peak_out = A.peak;
    if (part_peak_out) *part_peak_out = part_peak;
    if (dry) return 0;
    if (A.peak > T.bwd_floats) return fail(LFT_ERR_ARG, "internal: backward arena overflow (%zu > %zu)", A.peak, T.bwd_floats);
    {
        const long long per = (N + kTailWaves - 1) / kTailWaves;
        const size_t poff = poff0;
        k_conv0_wgrad<<<kTailWaves / 4, 256, 0, st>>>(dx0, lr, c.F(T.part) + poff, d.B, d.A, d.h, d.w, per);
        LFT_LAUNCH_OK("k_conv0_wgrad");
        TRY(red_push(c, poff, kTailWaves, 576, 576, g(P_CONV0), 0));
    }
    TRY(end_bucket(2));
#undef TRY
    return 0;
}
BwdSizes bwd_sizes(const Dims& d) {
    // the pass's own allocation sequence, nothing launched; cached for the last shape (every entry point computes the layout)
    static thread_local Dims last{};
    static thread_local BwdSizes last_sz{};
    if (last_sz.arena && last.B == d.B && last.A == d.A && last.h == d.h && last.w == d.w && last.s == d.s) return last_sz;
    BwdSizes sz{};
    sz.rc = train_backward(nullptr, nullptr, nullptr, nullptr, nullptr, d, LFT_MATH_F32, nullptr, nullptr, nullptr, true, &sz.arena, &sz.part);
    if (sz.rc) { sz.arena = sz.part = 0; return sz; }             // not cached: the next call reports the failure again
    last = d; last_sz = sz;
    return sz;
}
