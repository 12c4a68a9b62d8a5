// lft_api.hip -- C ABI of liblft_hip.so (see include/lft_hip.h): buffer layouts, weight packing plan,
// kernel launches.  Host-side code only enqueues work on the caller's stream.
//
// The library is built from this file as TWO translation units with different compiler flags (lft_amd/_lib.py):
//   LFT_TU == 1   inference, scene tiling, metrics, debug entry points   -- with -fno-slp-vectorize: hipcc's SLP pass packs
//                 neighbouring scalar f32 operations into v_pk_*_f32, which issue slower beside MFMAs than the two scalar
//                 instructions (+1.4 % on the bench without it; the kernels pack explicitly, f32x2, where that was measured to pay)
//   LFT_TU == 2   the fp32 training step -- default flags: the SLP pass also decides which of its multiply-adds are
//                 contracted, and the step's last-bit behaviour is pinned by the gradient fixtures
//   LFT_TU == 0   everything in one unit (tools, resource reports).
// Every kernel and helper is local to its unit (anonymous namespace); the units share only the error buffer.
#include "../../include/lft_hip.h"
#include "../../include/lft_hip_test.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <set>
#include <string>
#include <algorithm>
#include <utility>
#include <vector>

#ifndef LFT_TU
#define LFT_TU 0
#endif

namespace {
#include "lft_kernels_a.cuh"
#include "lft_kernels_b.cuh"
#include "lft_metrics.cuh"
#include "lft_train.cuh"     // training kernels; the fp32 inference path shares their LDS-tiled window attention
}  // namespace

#if LFT_TU == 2
extern thread_local char lft_g_err[512];
#else
__attribute__((visibility("hidden"))) thread_local char lft_g_err[512] = "";
#endif
#define g_err lft_g_err

namespace {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define LFT_HIP_OK(expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail((int)e_, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
// Optional per-kernel timing (lft_forward_profiled): an event is recorded on the launch stream after each kernel.
struct Profiler {
    bool on = false;
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;
    std::vector<const char*> names;
};
thread_local Profiler g_prof;
// Launch name with the call's shape appended ("k_lin:128>256", profiled runs only); interned, so the pointer stays valid.
inline const char* prof_name(const char* plain, const char* fmt, ...) {
    if (!g_prof.on) return plain;
    static thread_local std::set<std::string> names;
    char buf[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return names.insert(buf).first->c_str();
}
inline void prof_mark(const char* name) {
    if (!g_prof.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g_prof.st);
    g_prof.ev.push_back(e);
    g_prof.names.push_back(name);
}

#define LFT_LAUNCH_OK(name)                                                                  \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) return fail((int)e_, "launch %s: %s", name, hipGetErrorString(e_)); \
        prof_mark(name);                                                                     \
    } while (0)

constexpr int kLayers = 4;   // reference LFT.py:15

// Run `expr` with T bound to the element type of precision `prec` (validated by make_dims beforehand).
#define LFT_BY_PREC(prec, ...)                                                                \
    ((prec) == LFT_PREC_F32    ? ([&] { using T = float; return __VA_ARGS__; })()            \
     : (prec) == LFT_PREC_BF16 ? ([&] { using T = bf16_t; return __VA_ARGS__; })()           \
                               : ([&] { using T = f16_t; return __VA_ARGS__; })())
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Dims {
    int B, A, V, h, w, hw, s, gp, gt, nchunk;
    long long ntok;
};

int make_dims(int B, int A, int h, int w, int s, int prec, Dims* d) {
    if (prec != LFT_PREC_F32 && prec != LFT_PREC_BF16 && prec != LFT_PREC_F16)
        return fail(LFT_ERR_ARG, "prec must be LFT_PREC_F32, LFT_PREC_BF16 or LFT_PREC_F16, got %d", prec);
    if (B < 1 || A < 1 || h < 1 || w < 1) return fail(LFT_ERR_SHAPE, "B, A, h, w must be positive (B=%d A=%d h=%d w=%d)", B, A, h, w);
    if (s != 2 && s != 4) return fail(LFT_ERR_SHAPE, "scale factor must be 2 or 4, got %d", s);
    if (A * A > 128) return fail(LFT_ERR_UNSUPPORTED, "angRes %d (A*A=%d views > 128) is not implemented in this build", A, A * A);
    if ((long long)B * A * A * h * w > (1LL << 27)) return fail(LFT_ERR_SHAPE, "too many tokens");
    d->B = B; d->A = A; d->V = A * A; d->h = h; d->w = w; d->hw = h * w; d->s = s;
    d->gp = (s + 2) * (s + 2); d->gt = (d->gp + 31) / 32; d->nchunk = 2 * s * s;
    d->ntok = (long long)B * d->V * d->hw;
    return 0;
}

// Fragment counts per stream
constexpr int kFragsConv = 72, kFragsAng = 64, kFragsSpa1 = 240, kFragsSpa1NoQ = 208, kFragsSpa2 = 176;
inline int frags_up(const Dims& d) { return d.nchunk * (4 + 2 * d.gt); }

struct PackedLayout {
    size_t conv0_w, ln_ang[kLayers], ln_spa[kLayers], ang_pe, petok[kLayers], spa_pe_img;
    size_t s_conv[3], s_ang[kLayers], s_spa1[kLayers], s_spa2[kLayers], s_up, total;
};

PackedLayout packed_layout(const Dims& d, int prec) {
    const size_t esz = prec == LFT_PREC_F32 ? 4 : 2, fragb = 64 * 8 * esz;
    PackedLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    L.conv0_w = take(576 * 4);
    for (int l = 0; l < kLayers; ++l) { L.ln_ang[l] = take(256 * 4); L.ln_spa[l] = take(512 * 4); }
    L.ang_pe = take((size_t)((d.V + 31) / 32) * 2048 * 4);                       // lane-major, per 32-view tile
    for (int l = 0; l < kLayers; ++l) L.petok[l] = take((size_t)((d.hw + 32 * kNwSpa1 - 1) / (32 * kNwSpa1)) * kNwSpa1 * 4096 * esz);   // lane-major, per 32-token tile
    L.spa_pe_img = take((size_t)d.hw * 64 * esz);
    for (int i = 0; i < 3; ++i) L.s_conv[i] = take(kFragsConv * fragb);
    for (int l = 0; l < kLayers; ++l) {
        L.s_ang[l] = take(kFragsAng * fragb);
        L.s_spa1[l] = take(kFragsSpa1 * fragb);
        L.s_spa2[l] = take(kFragsSpa2 * fragb);
    }
    L.s_up = take((size_t)frags_up(d) * fragb);
    L.total = o;
    return L;
}

struct WorkLayout {
    size_t x0, feat, xa, xb, tok, q, k, v, o, g, status, total;     // status: the sticky flag word (lft_status_read), last 256 bytes
};
WorkLayout work_layout(const Dims& d, int prec) {
    const size_t esz = prec == LFT_PREC_F32 ? 4 : 2;
    WorkLayout W;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    const size_t n = (size_t)d.ntok;
    W.x0 = take(n * 64 * esz); W.feat = take(n * 64 * esz); W.xa = take(n * 64 * esz); W.xb = take(n * 64 * esz);
    W.tok = take(n * 128 * esz); W.q = take(n * 128 * esz); W.k = take(n * 128 * esz); W.v = take(n * 128 * esz);
    W.o = take(n * 128 * esz);
    W.g = take(n * d.gp * 4);
    W.status = take(256);
    W.total = o;
    return W;
}

// Dynamic LDS sizes (bytes) and the opt-in above the 64 KiB default (a workgroup may use all 160 KiB of a CU).
template <typename T> size_t lds_conv64(int w) { return WRing<T, kConv64Chunk, kNwConv>::LDS_BYTES + ConvIn<T, kNwConv>::bytes(w) + kNwConv * TileIO<2, T>::BYTES + kConvZeroRow; }
constexpr size_t kLdsParams = 1024;   // 256 LayerNorm floats
template <typename T, int CH = kSpaChunk> size_t lds_spa1(int w) {      // the tile I/O scratch aliases the conv input tile, which must be large enough for it
    return WRing<T, CH, kNwSpa1>::LDS_BYTES + std::max<size_t>(ConvIn<T, kNwSpa1>::bytes(w), (size_t)kNwSpa1 * TileIO<4, T>::BYTES) + kLdsParams + kConvZeroRow;
}
template <typename T> size_t lds_spa2() { return WRing<T, kSpaChunk, kNwSpa2>::LDS_BYTES + kLdsParams + kNwSpa2 * TileIO<4, T>::BYTES; }
template <typename T> size_t lds_up() { return WRing<T, kUpChunk, kNwUp>::LDS_BYTES + kNwUp * TileIO<2, T>::BYTES; }
template <typename T> size_t lds_ang() { return (size_t)kFragsAng * 1024 * FragInfo<T>::PIECES + kLdsParams + 4 * TileIO<2, T>::BYTES; }
constexpr size_t kMaxLds = 160 * 1024;
template <typename K> int allow_lds(K kernel, size_t bytes, const char* name) {
    if (bytes > kMaxLds) return fail(LFT_ERR_SHAPE, "%s needs %zu B of LDS (> 160 KiB): view width too large for this build", name, bytes);
    if (bytes <= 64 * 1024) return 0;
    // The attribute is sticky per kernel: set it once per (kernel, size) so that steady-state forwards -- and a
    // stream capture of them -- consist of kernel launches only.
    // The attribute belongs to the (device, kernel) pair: a thread that drives two GPUs must set it on each.
    struct Done { int dev; const void* fn; size_t bytes; };
    static thread_local std::vector<Done> done;
    const void* fn = reinterpret_cast<const void*>(kernel);
    int dev = 0;
    LFT_HIP_OK(hipGetDevice(&dev));
    for (const auto& d : done)
        if (d.dev == dev && d.fn == fn && d.bytes >= bytes) return 0;
    LFT_HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.push_back(Done{dev, fn, bytes});
    return 0;
}

template <typename P> P* at(const void* base, size_t off) { return reinterpret_cast<P*>(const_cast<char*>(static_cast<const char*>(base)) + off); }
inline unsigned blocks_for(long long items, int per_block) { return (unsigned)((items + per_block - 1) / per_block); }

// ---------------------------------------------------------------------------- packing plan
PackOp lin_op(const float* src, int row0, int nrows, int ld, int k0, int ksteps, int kmap, float scale, int kmul = 1, int kadd = 0) {
    PackOp p{};
    p.src = src; p.kind = 0; p.row0 = row0; p.nrows = nrows; p.ntiles = (nrows + 31) / 32; p.ld = ld;
    p.kmul = kmul; p.kadd = kadd; p.k0 = k0; p.ksteps = ksteps; p.kmap = kmap; p.scale = scale; p.s = 0; p.frag0 = 0;
    return p;
}
void conv_ops(std::vector<PackOp>& v, const float* w, int nout) {   // [nout][64][3][3] == [nout][576], k index c*9 + tap
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < 4; ++ks) v.push_back(lin_op(w, 0, nout, 576, 16 * ks, 1, 0, 1.0f, 9, tap));
}

template <typename T>
int run_pack(std::vector<PackOp>& ops, T* dst, int expect_frags, hipStream_t st) {
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
        k_pack<T><<<nf, 64, 0, st>>>(a, dst + (size_t)total * 512);
        LFT_LAUNCH_OK("k_pack");
        total += nf;
    }
    if (total != expect_frags) return fail(LFT_ERR_ARG, "internal: stream has %d fragments, expected %d", total, expect_frags);
    return 0;
}

#ifndef LFT_TOKLM
#define LFT_TOKLM 1
#endif
#ifndef LFT_YLM
#define LFT_YLM 1
#endif
// Lane-major hand-off of the spatial tokens from k_spa1 to part B: every workgroup tile of k_spa1 must be full, and the bf16
// consumer (k_spa_b, 8 x 4 blocks) additionally needs a tile to be 32 columns of ONE image row.
template <typename T> bool tok_lane_major(const Dims& d) {
    return LFT_TOKLM && d.hw % (32 * kNwSpa1) == 0 && (sizeof(T) == 4 || d.w % 32 == 0);
}
// k_spa1 launch with the ring chunk size that fits best: 16-fragment chunks if two workgroups then still share a CU
// (<= 80 KiB each) or if they are the only ones fitting at all... else 8-fragment chunks (wide views, fp32).
template <typename T, bool PE_ONLY>
int launch_spa1(unsigned nwg, const T* in, const T* ws, const float* ln, const T* petok, T* tok, T* q, T* k, T* v, T* pe_out,
                int nimg, const Dims& d, hipStream_t st, unsigned* status) {
#ifndef LFT_SPA1_EXTRA_LDS
#define LFT_SPA1_EXTRA_LDS 0
#endif
    const size_t l16 = lds_spa1<T, 16>(d.w) + LFT_SPA1_EXTRA_LDS, l8 = lds_spa1<T, 8>(d.w) + LFT_SPA1_EXTRA_LDS;
    const size_t share = kMaxLds / LFT_SPA_OCC;                           // LDS per workgroup if LFT_SPA_OCC of them share a CU
    const bool use8 = (l16 > share && l8 <= share) || l16 > kMaxLds;
    int rc;
    const bool lm = !PE_ONLY && tok_lane_major<T>(d);      // hand the token tile to part B in lane-major tile form
    // 16-bit with the lane-major hand-off: part B (k_spa_b) computes Q from the token tile itself; k_spa1 then skips that projection
#define LFT_LAUNCH_SPA1(CHV, LMV, LDSV)                                                                                     \
    do {                                                                                                                    \
        constexpr bool WQ = !(LMV && sizeof(T) == 2);                                                                       \
        if ((rc = allow_lds(k_spa1<T, PE_ONLY, CHV, LMV, WQ>, LDSV, "k_spa1"))) return rc;                                  \
        k_spa1<T, PE_ONLY, CHV, LMV, WQ><<<nwg, 64 * kNwSpa1, LDSV, st>>>(in, ws, ln, petok, tok, q, k, v, pe_out, nimg, d.h, d.w, status); \
    } while (0)
    if (use8) { if (lm) LFT_LAUNCH_SPA1(8, true, l8); else LFT_LAUNCH_SPA1(8, false, l8); }
    else { if (lm) LFT_LAUNCH_SPA1(16, true, l16); else LFT_LAUNCH_SPA1(16, false, l16); }
#undef LFT_LAUNCH_SPA1
    return 0;
}

template <typename T>
int pack_impl(const float* const* P, void* packed, const Dims& d, int prec, hipStream_t st) {
    const PackedLayout L = packed_layout(d, prec);
    const float kAngScale = 0.35355339059327373f * LFT_LOG2E;   // 1/sqrt(8) (head_dim 8), exp2 softmax
    const float kSpaScale = 0.25f * LFT_LOG2E;                   // 1/sqrt(16)
    int rc;
    k_copy_f32<<<3, 256, 0, st>>>(P[0], at<float>(packed, L.conv0_w), 576);
    LFT_LAUNCH_OK("k_copy_f32");
    for (int i = 0; i < 3; ++i) {
        std::vector<PackOp> ops;
        conv_ops(ops, P[1 + i], 64);
        if ((rc = run_pack<T>(ops, at<T>(packed, L.s_conv[i]), kFragsConv, st))) return rc;
    }
    k_pe_tables<T><<<blocks_for(std::max<long long>((long long)((d.V + 31) / 32) * 2048, (long long)d.hw * 64), 256), 256, 0, st>>>(
        at<float>(packed, L.ang_pe), at<T>(packed, L.spa_pe_img), d.V, d.h, d.w);
    LFT_LAUNCH_OK("k_pe_tables");
    for (int l = 0; l < kLayers; ++l) {
        const float* const* q = P + 4 + 18 * l;
        float* lnS = at<float>(packed, L.ln_spa[l]);
        float* lnA = at<float>(packed, L.ln_ang[l]);
        const int srcS[4] = {1, 2, 5, 6}, srcA[4] = {10, 11, 14, 15};
        for (int j = 0; j < 4; ++j) {
            k_copy_f32<<<1, 256, 0, st>>>(q[srcS[j]], lnS + 128 * j, 128);
            k_copy_f32<<<1, 256, 0, st>>>(q[srcA[j]], lnA + 64 * j, 64);
        }
        LFT_LAUNCH_OK("k_copy_f32");
        {   // angular stream
            std::vector<PackOp> ops;
            ops.push_back(lin_op(q[12], 0, 64, 64, 0, 4, 1, kAngScale));
            ops.push_back(lin_op(q[12], 64, 64, 64, 0, 4, 1, 1.0f));
            ops.push_back(lin_op(q[12], 128, 64, 64, 0, 4, 1, 1.0f));
            ops.push_back(lin_op(q[13], 0, 64, 64, 0, 4, 1, 1.0f));
            ops.push_back(lin_op(q[16], 0, 128, 64, 0, 4, 1, 1.0f));
            ops.push_back(lin_op(q[17], 0, 64, 128, 0, 8, 1, 1.0f));
            if ((rc = run_pack<T>(ops, at<T>(packed, L.s_ang[l]), kFragsAng, st))) return rc;
        }
        {   // spatial part 1: token embedding conv + in_proj
            std::vector<PackOp> ops;
            conv_ops(ops, q[0], 128);
            ops.push_back(lin_op(q[3], 256, 128, 128, 0, 8, 1, 1.0f));          // Wv (consumed first)
            ops.push_back(lin_op(q[3], 128, 128, 128, 0, 8, 1, 1.0f));          // Wk
            ops.push_back(lin_op(q[3], 0, 128, 128, 0, 8, 1, kSpaScale));      // Wq: last -- k_spa1's ring ends before it when part B computes Q itself (kFragsSpa1NoQ)
            if ((rc = run_pack<T>(ops, at<T>(packed, L.s_spa1[l]), kFragsSpa1, st))) return rc;
        }
        {   // spatial part 2: out_proj, FFN in 4 chunks, 1x1x1 conv.  out_proj's operand: bf16 -- the attention accumulators
            // inside k_spa_b (acc order); fp32 -- the attention output read back from memory by k_spa2 (natural k)
            std::vector<PackOp> ops;
            // 16-bit (k_spa_b): element (h, j) of the attention fragment is the head's channel label 8 h + j of the V tile in LDS.
            // Row-major K / V: labels are the channels themselves -- natural packing.  Lane-major Q / K / V (tok_lane_major): k_spa1
            // wrote each head's 16 channels in acc order, so label 8 h + j is channel acc(h, j) -- acc-order packing.
            ops.push_back(lin_op(q[4], 0, 128, 128, 0, 8, (sizeof(T) == 2 && tok_lane_major<T>(d)) ? 1 : 0, 1.0f));
            for (int c = 0; c < 4; ++c) {
                ops.push_back(lin_op(q[7], 64 * c, 64, 128, 0, 8, 1, 1.0f));
                ops.push_back(lin_op(q[8], 0, 128, 256, 64 * c, 4, 1, 1.0f));
            }
            ops.push_back(lin_op(q[9], 0, 64, 128, 0, 8, 1, 1.0f));
            if ((rc = run_pack<T>(ops, at<T>(packed, L.s_spa2[l]), kFragsSpa2, st))) return rc;
        }
        // embedded spatial position tokens of this layer (reference LFT.py:180), [h*w][128] in the activation type
        if ((rc = launch_spa1<T, true>((unsigned)((d.hw + 32 * kNwSpa1 - 1) / (32 * kNwSpa1)), at<T>(packed, L.spa_pe_img), at<T>(packed, L.s_spa1[l]), nullptr, nullptr,
                                       nullptr, nullptr, nullptr, nullptr, at<T>(packed, L.petok[l]), 1, d, st, nullptr))) return rc;
        LFT_LAUNCH_OK("k_spa1<pe>");
    }
    {   // up-sampler: per 32-row chunk of the 1x1 conv, followed by the matching columns of the overlap-add matrix
        std::vector<PackOp> ops;
        for (int c = 0; c < d.nchunk; ++c) {
            ops.push_back(lin_op(P[76], 32 * c, 32, 64, 0, 4, 1, 1.0f));
            PackOp m = lin_op(P[77], 0, d.gp, 0, 32 * c, 2, 1, 1.0f);
            m.kind = 1; m.s = d.s; m.ntiles = d.gt;
            ops.push_back(m);
        }
        if ((rc = run_pack<T>(ops, at<T>(packed, L.s_up), frags_up(d), st))) return rc;
    }
    return 0;
}

void launch_assemble(const float* lr, const float* g, float* out, int B, int A, int h, int w, int s, hipStream_t st, int gld = 0, unsigned* status = nullptr) {
    const dim3 tg((unsigned)((A * w + 7) / 8) * (unsigned)((A * h + 7) / 8) * (unsigned)B);    // 8 x 8 LR mosaic pixels per workgroup; tile order: k_assemble_t
    if (!gld) gld = (s + 2) * (s + 2);
    if (s == 2) k_assemble_t<2><<<tg, 256, 0, st>>>(lr, g, out, B, A, h, w, gld, status);
    else k_assemble_t<4><<<tg, 256, 0, st>>>(lr, g, out, B, A, h, w, gld, status);
}

// ---------------------------------------------------------------------------- stages
template <typename T>
int init_features(const void* packed, const PackedLayout& L, const float* lr, T* x0, T* ta, T* tb, T* feat, const Dims& d, hipStream_t st) {
    const int nimg = d.B * d.V, nwg = nimg * ((d.hw + 32 * kNwConv - 1) / (32 * kNwConv));
    const size_t lds = lds_conv64<T>(d.w);
    int rc;
    if ((rc = allow_lds(k_conv64<T, false>, lds, "k_conv64"))) return rc;
    if ((rc = allow_lds(k_conv64<T, true>, lds, "k_conv64"))) return rc;
    k_conv0<T><<<dim3((unsigned)((d.hw + kConv0Tok - 1) / kConv0Tok), (unsigned)nimg), 256, 0, st>>>(lr, at<float>(packed, L.conv0_w), x0, d.B, d.A, d.h, d.w);
    LFT_LAUNCH_OK("k_conv0");
    k_conv64<T, false><<<nwg, 64 * kNwConv, lds, st>>>(x0, ta, nullptr, at<T>(packed, L.s_conv[0]), nimg, d.h, d.w);
    LFT_LAUNCH_OK("k_conv64");
    k_conv64<T, false><<<nwg, 64 * kNwConv, lds, st>>>(ta, tb, nullptr, at<T>(packed, L.s_conv[1]), nimg, d.h, d.w);
    LFT_LAUNCH_OK("k_conv64");
    k_conv64<T, true><<<nwg, 64 * kNwConv, lds, st>>>(tb, feat, x0, at<T>(packed, L.s_conv[2]), nimg, d.h, d.w);
    LFT_LAUNCH_OK("k_conv64");
    return 0;
}
template <typename T, int CT>
int ang_multi(const void* packed, const PackedLayout& L, int l, const T* in, T* out, const Dims& d, hipStream_t st, unsigned* status) {
    constexpr bool WLDS = sizeof(T) == 2;                                   // fp32 weights (128 KiB) stay in L2
    constexpr int NG = (sizeof(T) == 2 && CT <= 3) ? 2 : 1;                 // positions per workgroup (they share the LDS weights)
    constexpr size_t FB = 1024 * FragInfo<T>::PIECES;
    const size_t lds = (WLDS ? 64 * FB : 0) + 1024 + (size_t)NG * CT * 8 * FB + (size_t)NG * CT * TileIO<2, T>::BYTES;
    const int npix = d.B * d.hw;
    int rc;
    const unsigned grid = std::min<unsigned>(blocks_for(npix, NG), 256u * (unsigned)std::max<size_t>(1, kMaxLds / lds));
    // score registers of the last key tile that can hold a view: register i covers rows acc_row(i, 0) and acc_row(i, 1) = +4
    const int rows_last = d.V - 32 * (CT - 1);                              // 1 .. 32 by the choice of CT
#define LFT_LAUNCH_ANGM(LL)                                                                                             \
    do {                                                                                                                \
        if ((rc = allow_lds(k_ang_multi<T, CT, WLDS, NG, LL>, lds, "k_ang_multi"))) return rc;                            \
        k_ang_multi<T, CT, WLDS, NG, LL><<<grid, 64 * CT * NG, lds, st>>>(in, out, at<T>(packed, L.s_ang[l]),              \
                                                                         at<float>(packed, L.ln_ang[l]), at<float>(packed, L.ang_pe), d.V, d.hw, npix, status); \
    } while (0)
    if (rows_last <= 17) LFT_LAUNCH_ANGM(9);          // e.g. 9 x 9 = 81 views: 17 rows in the third tile
    else if (rows_last <= 25) LFT_LAUNCH_ANGM(13);
    else LFT_LAUNCH_ANGM(16);
#undef LFT_LAUNCH_ANGM
    LFT_LAUNCH_OK("k_ang");
    return 0;
}
template <typename T>
int ang_block(const void* packed, const PackedLayout& L, int l, const T* in, T* out, const Dims& d, hipStream_t st, unsigned* status = nullptr) {
    if (d.V > 96) return ang_multi<T, 4>(packed, L, l, in, out, d, st, status);
    if (d.V > 64) return ang_multi<T, 3>(packed, L, l, in, out, d, st, status);      // 9x9 = 81 views
    if (d.V > 32) return ang_multi<T, 2>(packed, L, l, in, out, d, st, status);
    const int npix = d.B * d.hw;
    const size_t lds = lds_ang<T>();
    int rc;
    const unsigned grid = std::min<unsigned>(blocks_for(npix, 4), 256u * (unsigned)std::max<size_t>(1, kMaxLds / lds));
    if (d.V <= 25) {                                                  // 5 x 5 and smaller: score rows 25..31 are never a view
        if ((rc = allow_lds(k_ang<T, 13>, lds, "k_ang"))) return rc;
        k_ang<T, 13><<<grid, 256, lds, st>>>(in, out, at<T>(packed, L.s_ang[l]), at<float>(packed, L.ln_ang[l]),
                                             at<float>(packed, L.ang_pe), d.V, d.hw, npix, status);
    } else {
        if ((rc = allow_lds(k_ang<T, 16>, lds, "k_ang"))) return rc;
        k_ang<T, 16><<<grid, 256, lds, st>>>(in, out, at<T>(packed, L.s_ang[l]), at<float>(packed, L.ln_ang[l]),
                                             at<float>(packed, L.ang_pe), d.V, d.hw, npix, status);
    }
    LFT_LAUNCH_OK("k_ang");
    return 0;
}
// SpaTrans = part A (k_spa1: token embedding, LayerNorm, Q / K / V) + part B (attention and the per-token tail).
template <typename T>
int spa_part_a(const void* packed, const PackedLayout& L, int l, const T* in, void* ws, const WorkLayout& W, const Dims& d, hipStream_t st) {
    const int nimg = d.B * d.V, nwg = nimg * ((d.hw + 32 * kNwSpa1 - 1) / (32 * kNwSpa1));
    int rc;
    if ((rc = launch_spa1<T, false>((unsigned)nwg, in, at<T>(packed, L.s_spa1[l]), at<float>(packed, L.ln_spa[l]), at<T>(packed, L.petok[l]),
                                    at<T>(ws, W.tok), at<T>(ws, W.q), at<T>(ws, W.k), at<T>(ws, W.v), nullptr, nimg, d, st, at<unsigned>(ws, W.status)))) return rc;
    LFT_LAUNCH_OK("k_spa1");
    return 0;
}
template <typename T>
int spa_part_b(const void* packed, const PackedLayout& L, int l, const T* skip, T* out, void* ws, const WorkLayout& W,
               const Dims& d, hipStream_t st, bool out_lm = false) {      // out_lm: lane-major output tiles, only for the up-sampler
    const int nimg = d.B * d.V;
    T *tok = at<T>(ws, W.tok), *q = at<T>(ws, W.q), *k = at<T>(ws, W.k), *v = at<T>(ws, W.v), *o = at<T>(ws, W.o);
    const float* ln = at<float>(packed, L.ln_spa[l]);
    int rc;
    if constexpr (sizeof(T) == 2) {
        // bf16: windowed attention + out_proj + FFN + 1x1x1 conv in ONE kernel (the attention output stays in registers)
        const unsigned ntile = (unsigned)(nimg * ((d.h + kAttTY - 1) / kAttTY) * ((d.w + kAttTX - 1) / kAttTX));
#ifndef LFT_SPAB_EXTRA_LDS
#define LFT_SPAB_EXTRA_LDS 0                 // experiments: pad the request so that fewer workgroups share a CU
#endif
        const size_t lds = kSpaBLds + LFT_SPAB_EXTRA_LDS;
#define LFT_LAUNCH_SPAB(SKV, LMV, YLV)                                                                                      \
    do {                                                                                                                    \
        if ((rc = allow_lds(k_spa_b<T, SKV, LMV, YLV>, lds, "k_spa_b"))) return rc;                                            \
        k_spa_b<T, SKV, LMV, YLV><<<ntile, 256, lds, st>>>(tok, q, k, v, at<T>(packed, L.s_spa2[l]), ln, skip, out, d.h, d.w, at<unsigned>(ws, W.status), \
                                                           at<T>(packed, L.s_spa1[l]) + (size_t)kFragsSpa1NoQ * 512, at<T>(packed, L.petok[l])); \
    } while (0)
        const bool tlm = tok_lane_major<T>(d);
        if (out_lm && !(skip && tlm)) return fail(LFT_ERR_ARG, "internal: lane-major output needs the skip variant and lane-major tokens");
        if (skip) { if (out_lm) LFT_LAUNCH_SPAB(true, true, true); else if (tlm) LFT_LAUNCH_SPAB(true, true, false); else LFT_LAUNCH_SPAB(true, false, false); }
        else { if (tlm) LFT_LAUNCH_SPAB(false, true, false); else LFT_LAUNCH_SPAB(false, false, false); }
#undef LFT_LAUNCH_SPAB
        LFT_LAUNCH_OK("k_spa_b");
        return 0;
    } else {
        // fp32: the LDS-tiled window attention of the training step (8 x 16 query tile x head pair per workgroup), Q pre-scaled
        const unsigned tiles = (unsigned)(((d.w + kWaTX - 1) / kWaTX) * ((d.h + kWaTY - 1) / kWaTY) * nimg);
        if ((rc = allow_lds(k_win_attn_lds<0, true>, kWaLds, "k_win_attn_lds"))) return rc;
        k_win_attn_lds<0, true><<<dim3(tiles, 4), 256, kWaLds, st>>>(reinterpret_cast<const float*>(q), reinterpret_cast<const float*>(k),
                                                                     reinterpret_cast<const float*>(v), reinterpret_cast<float*>(o),
                                                                     nullptr, nullptr, nullptr, nullptr, nullptr, d.h, d.w, 128);
    LFT_LAUNCH_OK("k_spa_attn");
    const unsigned nb = blocks_for(d.ntok, 32 * kNwSpa2);
    const bool lm = tok_lane_major<T>(d);      // must match launch_spa1's choice
#define LFT_LAUNCH_SPA2(SKV, LMV, YLV)                                                                                      \
    do {                                                                                                                    \
        if ((rc = allow_lds(k_spa2<T, SKV, LMV, YLV>, lds_spa2<T>(), "k_spa2"))) return rc;                                 \
        k_spa2<T, SKV, LMV, YLV><<<nb, 64 * kNwSpa2, lds_spa2<T>(), st>>>(tok, o, at<T>(packed, L.s_spa2[l]), ln, skip, out, d.ntok, at<unsigned>(ws, W.status)); \
    } while (0)
    if (out_lm && !(skip && lm)) return fail(LFT_ERR_ARG, "internal: lane-major output needs the skip variant and full tiles");
    if (skip) { if (out_lm) LFT_LAUNCH_SPA2(true, true, true); else if (lm) LFT_LAUNCH_SPA2(true, true, false); else LFT_LAUNCH_SPA2(true, false, false); }
    else { if (lm) LFT_LAUNCH_SPA2(false, true, false); else LFT_LAUNCH_SPA2(false, false, false); }
#undef LFT_LAUNCH_SPA2
    LFT_LAUNCH_OK("k_spa2");
    return 0;
    }
}
template <typename T>
int spa_block(const void* packed, const PackedLayout& L, int l, const T* in, const T* skip, T* out, void* ws, const WorkLayout& W,
              const Dims& d, hipStream_t st, bool out_lm = false) {
    int rc;
    if ((rc = spa_part_a<T>(packed, L, l, in, ws, W, d, st))) return rc;
    return spa_part_b<T>(packed, L, l, skip, out, ws, W, d, st, out_lm);
}
template <typename T>
int upsample(const void* packed, const PackedLayout& L, const T* body, const float* lr, float* out, void* ws, const WorkLayout& W,
             const Dims& d, hipStream_t st, bool in_lm = false) {
    float* g = at<float>(ws, W.g);
    const unsigned nb = blocks_for(d.ntok, 32 * kNwUp);
    int rc;
#define LFT_LAUNCH_UP(GTV, LMV)                                                                                              \
    do {                                                                                                                    \
        if ((rc = allow_lds(k_up<T, GTV, LMV>, lds_up<T>(), "k_up"))) return rc;                                            \
        k_up<T, GTV, LMV><<<nb, 64 * kNwUp, lds_up<T>(), st>>>(body, at<T>(packed, L.s_up), g, d.ntok, d.nchunk, d.gp);     \
    } while (0)
    if (d.gt == 1) { if (in_lm) LFT_LAUNCH_UP(1, true); else LFT_LAUNCH_UP(1, false); }
    else { if (in_lm) LFT_LAUNCH_UP(2, true); else LFT_LAUNCH_UP(2, false); }
#undef LFT_LAUNCH_UP
    LFT_LAUNCH_OK("k_up");
    launch_assemble(lr, g, out, d.B, d.A, d.h, d.w, d.s, st, 0, at<unsigned>(ws, W.status));
    LFT_LAUNCH_OK("k_assemble");
    return 0;
}

template <typename T>
int forward_impl(const void* packed, const float* lr, float* out, void* ws, const Dims& d, int prec, hipStream_t st) {
    const PackedLayout L = packed_layout(d, prec);
    const WorkLayout W = work_layout(d, prec);
    T *x0 = at<T>(ws, W.x0), *feat = at<T>(ws, W.feat), *xa = at<T>(ws, W.xa), *xb = at<T>(ws, W.xb);
    int rc;
    if ((rc = init_features<T>(packed, L, lr, x0, xa, xb, feat, d, st))) return rc;
    const T* cur = feat;
    for (int l = 0; l < kLayers; ++l) {                  // angular first, then spatial (reference LFT.py:249-250)
        if ((rc = ang_block<T>(packed, L, l, cur, xa, d, st, at<unsigned>(ws, W.status)))) return rc;
        const bool last = l == kLayers - 1;                  // its output only feeds the up-sampler: same 32-token tiling, lane-major tiles
        if ((rc = spa_block<T>(packed, L, l, xa, last ? feat : nullptr, xb, ws, W, d, st, last && LFT_YLM && tok_lane_major<T>(d)))) return rc;
        cur = xb;
    }
    return upsample<T>(packed, L, xb, lr, out, ws, W, d, st, LFT_YLM && tok_lane_major<T>(d));
}

// Mean duration of ONE kernel of the forward, launched `reps` times back to back between two HIP events on `stream` (no
// event between launches, unlike lft_forward_profiled).  Inputs are whatever a previous lft_forward left in the
// workspace.  Synchronises the stream.  kernel: "k_conv64", "k_ang", "k_spa1", "k_spa_b" (bf16) / "k_spa2" ... see below.
template <typename T>
int kernel_time_impl(const char* name, const void* packed, void* ws, const Dims& d, int prec, int reps, hipStream_t st, float* ms_out) {
    const PackedLayout L = packed_layout(d, prec);
    const WorkLayout W = work_layout(d, prec);
    T *x0 = at<T>(ws, W.x0), *feat = at<T>(ws, W.feat), *xa = at<T>(ws, W.xa), *xb = at<T>(ws, W.xb);
    const std::string k(name);
    auto once = [&]() -> int {
        if (k == "k_ang") return ang_block<T>(packed, L, 1, xb, xa, d, st, at<unsigned>(ws, W.status));
        if (k == "k_spa1") return spa_part_a<T>(packed, L, 1, xa, ws, W, d, st);
        if (k == "k_spa_b" || k == "k_spa_attn+k_spa2") return spa_part_b<T>(packed, L, 1, nullptr, xb, ws, W, d, st);
        if (k == "k_conv64") {
            const int nimg = d.B * d.V, nwg = nimg * ((d.hw + 32 * kNwConv - 1) / (32 * kNwConv));
            const size_t lds = lds_conv64<T>(d.w);
            int rc;
            if ((rc = allow_lds(k_conv64<T, false>, lds, "k_conv64"))) return rc;
            k_conv64<T, false><<<nwg, 64 * kNwConv, lds, st>>>(x0, feat, nullptr, at<T>(packed, L.s_conv[0]), nimg, d.h, d.w);
            LFT_LAUNCH_OK("k_conv64");
            return 0;
        }
        return fail(LFT_ERR_ARG, "lft_kernel_time: unknown kernel %s", name);
    };
    int rc;
    if ((rc = once())) return rc;                                     // warm (attributes, caches)
    hipEvent_t e0, e1;
    LFT_HIP_OK(hipEventCreate(&e0));
    LFT_HIP_OK(hipEventCreate(&e1));
    LFT_HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < reps && !rc; ++i) rc = once();
    LFT_HIP_OK(hipEventRecord(e1, st));
    LFT_HIP_OK(hipEventSynchronize(e1));
    float ms = 0.0f;
    LFT_HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    *ms_out = ms / (float)reps;
    return 0;
}
#include "lft_train_host.cuh"

}  // namespace

// ================================================================================ C ABI
extern "C" {

#if LFT_TU != 2
int lft_version(void) { return LFT_ABI_VERSION; }
const char* lft_last_error(void) { return g_err; }

int lft_packed_bytes(int A, int h, int w, int s, int prec, size_t* out_bytes) {
    Dims d; int rc;
    if (!out_bytes) return fail(LFT_ERR_ARG, "out_bytes is null");
    if ((rc = make_dims(1, A, h, w, s, prec, &d))) return rc;
    *out_bytes = packed_layout(d, prec).total;
    return 0;
}
int lft_workspace_bytes(int B, int A, int h, int w, int s, int prec, size_t* out_bytes) {
    Dims d; int rc;
    if (!out_bytes) return fail(LFT_ERR_ARG, "out_bytes is null");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    *out_bytes = work_layout(d, prec).total;
    return 0;
}

int lft_pack_weights(const float* const* params, int nparams, void* packed, int A, int h, int w, int s, int prec, void* stream) {
    Dims d; int rc;
    if (!params || !packed) return fail(LFT_ERR_ARG, "null pointer");
    if (nparams != LFT_NUM_PARAMS) return fail(LFT_ERR_ARG, "expected %d parameter tensors, got %d", LFT_NUM_PARAMS, nparams);
    for (int i = 0; i < nparams; ++i)
        if (!params[i]) return fail(LFT_ERR_ARG, "parameter %d is null", i);
    if ((rc = make_dims(1, A, h, w, s, prec, &d))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, pack_impl<T>(params, packed, d, prec, st));
}

int lft_forward(const void* packed, const float* lr, float* out, void* workspace, int B, int A, int h, int w, int s, int prec, void* stream) {
    Dims d; int rc;
    if (!packed || !lr || !out || !workspace) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, forward_impl<T>(packed, lr, out, workspace, d, prec, st));
}

int lft_status_reset(void* workspace, int B, int A, int h, int w, int s, int prec, void* stream) {
    Dims d; int rc;
    if (!workspace) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    LFT_HIP_OK(hipMemsetAsync(at<char>(workspace, work_layout(d, prec).status), 0, 256, static_cast<hipStream_t>(stream)));
    return 0;
}
int lft_status_read(const void* workspace, int B, int A, int h, int w, int s, int prec, void* stream, unsigned* host_flags) {
    Dims d; int rc;
    if (!workspace) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned flags = 0;
    LFT_HIP_OK(hipMemcpyAsync(&flags, at<char>(workspace, work_layout(d, prec).status), sizeof(flags), hipMemcpyDeviceToHost, st));
    LFT_HIP_OK(hipStreamSynchronize(st));
    if (host_flags) *host_flags = flags;
    if (flags == 0) return 0;
    return fail(LFT_STATUS_NONFINITE, "non-finite activations or outputs since the last lft_status_reset (flags 0x%x)%s", flags,
                prec == LFT_PREC_F16 ? ": an activation left the fp16 range (|x| > 65504) or the input holds inf / NaN -- use LFT_PREC_BF16 or LFT_PREC_F32 for these weights"
                                     : ": the input or the weights hold inf / NaN, or an activation overflowed");
}

int lft_forward_profiled(const void* packed, const float* lr, float* out, void* workspace, int B, int A, int h, int w, int s, int prec,
                         void* stream, int max_records, float* ms_out, const char** names_out, int* n_out) {
    if (!ms_out || !names_out || !n_out) return fail(LFT_ERR_ARG, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    g_prof.on = true; g_prof.st = st; g_prof.ev.clear(); g_prof.names.clear();
    prof_mark("start");
    int rc = lft_forward(packed, lr, out, workspace, B, A, h, w, s, prec, stream);
    g_prof.on = false;
    hipError_t e = hipStreamSynchronize(st);
    int n = 0;
    for (size_t i = 1; i < g_prof.ev.size() && n < max_records; ++i, ++n) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, g_prof.ev[i - 1], g_prof.ev[i]);
        ms_out[n] = ms;
        names_out[n] = g_prof.names[i];
    }
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.clear(); g_prof.names.clear();
    *n_out = n;
    if (rc) return rc;
    if (e != hipSuccess) return fail((int)e, "hipStreamSynchronize: %s", hipGetErrorString(e));
    return 0;
}

int lft_kernel_time(const char* kernel, const void* packed, void* workspace, int B, int A, int h, int w, int s, int prec, int reps,
                    void* stream, float* ms_out) {
    Dims d; int rc;
    if (!kernel || !packed || !workspace || !ms_out || reps < 1) return fail(LFT_ERR_ARG, "bad argument");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, kernel_time_impl<T>(kernel, packed, workspace, d, prec, reps, st, ms_out));
}

int lft_bicubic_fwd(const float* lr, float* out, int B, int A, int h, int w, int s, void* stream) {
    Dims d; int rc;
    if (!lr || !out) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    const dim3 grid((unsigned)((A * w * s + 255) / 256), (unsigned)(A * h * s), (unsigned)B);
    k_bicubic<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(lr, out, B, A, h, w, s);
    LFT_LAUNCH_OK("k_bicubic");
    return 0;
}

int lft_init_features_fwd(const void* packed, const float* lr, void* act_out, void* workspace, int B, int A, int h, int w, int s,
                          int prec, void* stream) {
    Dims d; int rc;
    if (!packed || !lr || !act_out || !workspace) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    const PackedLayout L = packed_layout(d, prec);
    const WorkLayout W = work_layout(d, prec);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, init_features<T>(packed, L, lr, at<T>(workspace, W.x0), at<T>(workspace, W.xa), at<T>(workspace, W.xb),
                                              static_cast<T*>(act_out), d, st));
}

int lft_ang_block_fwd(const void* packed, int layer, const void* act_in, void* act_out, int B, int A, int h, int w, int s, int prec,
                      void* stream) {
    Dims d; int rc;
    if (!packed || !act_in || !act_out) return fail(LFT_ERR_ARG, "null pointer");
    if (layer < 0 || layer >= kLayers) return fail(LFT_ERR_ARG, "layer %d out of range", layer);
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    const PackedLayout L = packed_layout(d, prec);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, ang_block<T>(packed, L, layer, static_cast<const T*>(act_in), static_cast<T*>(act_out), d, st));
}

int lft_spa_block_fwd(const void* packed, int layer, const void* act_in, const void* skip, void* act_out, void* workspace, int B, int A,
                      int h, int w, int s, int prec, void* stream) {
    Dims d; int rc;
    if (!packed || !act_in || !act_out || !workspace) return fail(LFT_ERR_ARG, "null pointer");
    if (layer < 0 || layer >= kLayers) return fail(LFT_ERR_ARG, "layer %d out of range", layer);
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    const PackedLayout L = packed_layout(d, prec);
    const WorkLayout W = work_layout(d, prec);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, spa_block<T>(packed, L, layer, static_cast<const T*>(act_in), static_cast<const T*>(skip),
                                          static_cast<T*>(act_out), workspace, W, d, st));
}

int lft_upsample_fwd(const void* packed, const void* act_in, const float* lr, float* out, void* workspace, int B, int A, int h, int w,
                     int s, int prec, void* stream) {
    Dims d; int rc;
    if (!packed || !act_in || !lr || !out || !workspace) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    const PackedLayout L = packed_layout(d, prec);
    const WorkLayout W = work_layout(d, prec);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return LFT_BY_PREC(prec, upsample<T>(packed, L, static_cast<const T*>(act_in), lr, out, workspace, W, d, st));
}

// Debug aid (tools/stress_conv.py): one k_conv64 launch. which = 0..2 selects the weight stream, with_res the variant,
// extra_lds pads the dynamic LDS request (e.g. to force one workgroup per CU).
int lft_debug_conv64(const void* packed, int which, int with_res, const void* in, const void* res, void* out, int B, int A, int h,
                     int w, int s, int prec, int extra_lds, void* stream) {
    Dims d; int rc;
    if (!packed || !in || !out || which < 0 || which > 2) return fail(LFT_ERR_ARG, "bad argument");
    if ((rc = make_dims(B, A, h, w, s, prec, &d))) return rc;
    const PackedLayout L = packed_layout(d, prec);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nimg = d.B * d.V, nwg = nimg * ((d.hw + 32 * kNwConv - 1) / (32 * kNwConv));
    rc = LFT_BY_PREC(prec, [&]() -> int {
        const size_t lds = lds_conv64<T>(d.w) + extra_lds;
        int r;
        if ((r = allow_lds(k_conv64<T, false>, lds, "k_conv64"))) return r;
        if ((r = allow_lds(k_conv64<T, true>, lds, "k_conv64"))) return r;
        if (with_res) k_conv64<T, true><<<nwg, 64 * kNwConv, lds, st>>>((const T*)in, (T*)out, (const T*)res, at<T>(packed, L.s_conv[which]), nimg, d.h, d.w);
        else k_conv64<T, false><<<nwg, 64 * kNwConv, lds, st>>>((const T*)in, (T*)out, nullptr, at<T>(packed, L.s_conv[which]), nimg, d.h, d.w);
        return 0;
    }());
    if (rc) return rc;
    LFT_LAUNCH_OK("k_conv64");
    return 0;
}

#if defined(LFT_EXPERIMENT) && !defined(LFT_ISA_MARKS)
// Diagnostic build only (lft_experiment.cuh): copy the stamp buffer to the host (synchronises).
int lft_debug_read_stamps(unsigned long long* host_out, int n) {
    LFT_HIP_OK(hipDeviceSynchronize());
    LFT_HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lft_stamps), sizeof(unsigned long long) * (size_t)n));
    return 0;
}
int lft_debug_clear_stamps(void) {
    static unsigned long long zeros[4096 * 32];
    LFT_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_lft_stamps), zeros, sizeof(zeros)));
    return 0;
}
#endif

static int scene_counts(int h0, int w0, int patch, int stride, int* nu, int* nv) {
    if (h0 < 1 || w0 < 1 || patch < 1 || stride < 1 || stride > patch) return fail(LFT_ERR_SHAPE, "bad scene tiling (h0=%d w0=%d patch=%d stride=%d)", h0, w0, patch, stride);
    const int bdr = (patch - stride) / 2, h = h0 + 2 * bdr, w = w0 + 2 * bdr;
    if (h < patch || w < patch) return fail(LFT_ERR_SHAPE, "view %dx%d is smaller than one patch after extension", h0, w0);
    *nu = (h - patch) / stride + ((h - patch) % stride ? 2 : 1);
    *nv = (w - patch) / stride + ((w - patch) % stride ? 2 : 1);
    return 0;
}
int lft_scene_counts(int h0, int w0, int patch, int stride, int* num_u, int* num_v) {
    if (!num_u || !num_v) return fail(LFT_ERR_ARG, "null pointer");
    return scene_counts(h0, w0, patch, stride, num_u, num_v);
}
int lft_scene_divide(const float* scene, float* patches, int A, int h0, int w0, int patch, int stride, void* stream) {
    int nu, nv, rc;
    if (!scene || !patches || A < 1) return fail(LFT_ERR_ARG, "bad argument");
    if ((rc = scene_counts(h0, w0, patch, stride, &nu, &nv))) return rc;
    const dim3 grid((unsigned)((A * patch + 255) / 256), (unsigned)(A * patch), (unsigned)(nu * nv));
    k_scene_divide<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(scene, patches, A, h0, w0, patch, stride, nv);
    LFT_LAUNCH_OK("k_scene_divide");
    return 0;
}
int lft_scene_integrate(const float* sr_patches, float* sr_scene, int A, int h0, int w0, int patch, int stride, int s, void* stream) {
    int nu, nv, rc;
    if (!sr_patches || !sr_scene || A < 1 || s < 1) return fail(LFT_ERR_ARG, "bad argument");
    if ((rc = scene_counts(h0, w0, patch, stride, &nu, &nv))) return rc;
    const dim3 grid((unsigned)((A * w0 * s + 255) / 256), (unsigned)(A * h0 * s));
    k_scene_integrate<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(sr_patches, sr_scene, A, patch * s, stride * s, h0 * s, w0 * s, nv);
    LFT_LAUNCH_OK("k_scene_integrate");
    return 0;
}

int lft_mfma_selftest(const float* Am, const float* Bm, const float* W2, float* C, float* D, int prec, void* stream) {
    if (!Am || !Bm || !W2 || !C || !D) return fail(LFT_ERR_ARG, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (prec == LFT_PREC_F32) k_selftest<float><<<1, 64, 0, st>>>(Am, Bm, W2, C, D);
    else if (prec == LFT_PREC_BF16) k_selftest<bf16_t><<<1, 64, 0, st>>>(Am, Bm, W2, C, D);
    else if (prec == LFT_PREC_F16) k_selftest<f16_t><<<1, 64, 0, st>>>(Am, Bm, W2, C, D);
    else return fail(LFT_ERR_ARG, "bad prec %d", prec);
    LFT_LAUNCH_OK("k_selftest");
    return 0;
}

// ================================================================================ metrics
int lft_view_metrics_scratch_bytes(int B, int A, int h, int w, size_t* out_bytes) {
    if (!out_bytes || B < 1 || A < 1 || h < 1 || w < 1) return fail(LFT_ERR_ARG, "bad argument");
    const size_t ntiles = (size_t)((h + kMetTile - 1) / kMetTile) * ((w + kMetTile - 1) / kMetTile);
    *out_bytes = (size_t)B * A * A * ntiles * 3 * sizeof(double);
    return 0;
}
int lft_view_metrics(const float* label, const float* out, int B, int A, int h, int w, float ssim_range, float* psnr, float* ssim,
                     void* scratch, void* stream) {
    if (!label || !out || !psnr || !ssim || !scratch || B < 1 || A < 1) return fail(LFT_ERR_ARG, "bad argument");
    if (h < 11 || w < 11) return fail(LFT_ERR_SHAPE, "views of %dx%d are smaller than the 11x11 SSIM window", h, w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntiles = ((h + kMetTile - 1) / kMetTile) * ((w + kMetTile - 1) / kMetTile), nviews = B * A * A;
    const double C1 = (0.01 * ssim_range) * (0.01 * ssim_range), C2 = (0.03 * ssim_range) * (0.03 * ssim_range);
    k_view_metrics<<<dim3((unsigned)ntiles, (unsigned)nviews), 256, 0, st>>>(label, out, static_cast<double*>(scratch), A, h, w, C1, C2);
    LFT_LAUNCH_OK("k_view_metrics");
    k_view_metrics_final<<<blocks_for(nviews, 64), 64, 0, st>>>(static_cast<const double*>(scratch), nviews, ntiles, h, w, psnr, ssim);
    LFT_LAUNCH_OK("k_view_metrics_final");
    return 0;
}

#endif  // LFT_TU != 2
#if LFT_TU != 1
// ================================================================================ training (fp32)
int lft_train_tape_bytes(int B, int A, int h, int w, int s, size_t* out_bytes) {
    Dims d; int rc;
    if (!out_bytes) return fail(LFT_ERR_ARG, "out_bytes is null");
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    const TrainLayout T = train_layout(d);
    if (T.rc) return T.rc;                                      // the sizing run of the backward pass failed: no tape size to report
    *out_bytes = T.total * sizeof(float);
    return 0;
}
int lft_train_grad_floats(int s, size_t* out_floats) {
    if (!out_floats) return fail(LFT_ERR_ARG, "out_floats is null");
    if (s != 2 && s != 4) return fail(LFT_ERR_SHAPE, "scale factor must be 2 or 4, got %d", s);
    *out_floats = (size_t)param_info(s).total;
    return 0;
}
int lft_train_tape_offset(const char* name, int B, int A, int h, int w, int s, size_t* out_float_offset) {
    Dims d; int rc;
    if (!name || !out_float_offset) return fail(LFT_ERR_ARG, "null pointer");
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    const TrainLayout T = train_layout(d);
    if (T.rc) return T.rc;
    const std::string n(name);
    auto layer = [&](char c) { return c - '0'; };
    if (n == "x0") *out_float_offset = T.x0; else if (n == "feat") *out_float_offset = T.feat;
    else if (n == "c1") *out_float_offset = T.c1; else if (n == "c2") *out_float_offset = T.c2; else if (n == "c3") *out_float_offset = T.c3;
    else if (n == "body") *out_float_offset = T.body; else if (n == "act") *out_float_offset = T.act;
    else if (n == "skip") *out_float_offset = T.skip;
    else if (n.size() > 5 && n.compare(0, 3, "ang") == 0 && n[3] >= '0' && n[3] < '0' + kLayers && n[4] == '.') {
        const AngTape& a = T.ang[layer(n[3])];
        const std::string f = n.substr(5);
        if (f == "n") *out_float_offset = a.n; else if (f == "qk") *out_float_offset = a.qk; else if (f == "v") *out_float_offset = a.v;
        else if (f == "o") *out_float_offset = a.o; else if (f == "t1") *out_float_offset = a.t1; else if (f == "m") *out_float_offset = a.m;
        else if (f == "hdn") *out_float_offset = a.hdn; else if (f == "y") *out_float_offset = a.y;
        else return fail(LFT_ERR_ARG, "unknown tape field %s", name);
    } else if (n.size() > 5 && n.compare(0, 3, "spa") == 0 && n[3] >= '0' && n[3] < '0' + kLayers && n[4] == '.') {
        const SpaTape& a = T.spa[layer(n[3])];
        const std::string f = n.substr(5);
        if (f == "tok") *out_float_offset = a.tok; else if (f == "n") *out_float_offset = a.n; else if (f == "qk") *out_float_offset = a.qk;
        else if (f == "v") *out_float_offset = a.v; else if (f == "o") *out_float_offset = a.o;
        else if (f == "t1") *out_float_offset = a.t1; else if (f == "m") *out_float_offset = a.m; else if (f == "hdn") *out_float_offset = a.hdn;
        else if (f == "t2") *out_float_offset = a.t2; else if (f == "y") *out_float_offset = a.y; else if (f == "petok") *out_float_offset = a.petok;
        else return fail(LFT_ERR_ARG, "unknown tape field %s", name);
    } else return fail(LFT_ERR_ARG, "unknown tape field %s", name);
    return 0;
}
int lft_train_forward(const float* const* params, int nparams, const float* lr, float* out, void* tape,
                      int B, int A, int h, int w, int s, int math, void* stream) {
    Dims d; int rc;
    if (!params || !lr || !out || !tape) return fail(LFT_ERR_ARG, "null pointer");
    if (nparams != LFT_NUM_PARAMS) return fail(LFT_ERR_ARG, "expected %d parameter tensors, got %d", LFT_NUM_PARAMS, nparams);
    for (int i = 0; i < nparams; ++i) if (!params[i]) return fail(LFT_ERR_ARG, "parameter %d is null", i);
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    if (math != LFT_MATH_F32 && math != LFT_MATH_BF16X3 && math != LFT_MATH_BF16X6) return fail(LFT_ERR_ARG, "math must be LFT_MATH_F32, LFT_MATH_BF16X3 or LFT_MATH_BF16X6, got %d", math);
    return train_forward(params, lr, out, static_cast<float*>(tape), d, math, static_cast<hipStream_t>(stream));
}
int lft_train_backward(const float* const* params, int nparams, const float* lr, void* tape, const float* dout, float* grads,
                       int B, int A, int h, int w, int s, int math, void* stream) {
    Dims d; int rc;
    if (!params || !lr || !tape || !dout || !grads) return fail(LFT_ERR_ARG, "null pointer");
    if (nparams != LFT_NUM_PARAMS) return fail(LFT_ERR_ARG, "expected %d parameter tensors, got %d", LFT_NUM_PARAMS, nparams);
    for (int i = 0; i < nparams; ++i) if (!params[i]) return fail(LFT_ERR_ARG, "parameter %d is null", i);
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    if (math != LFT_MATH_F32 && math != LFT_MATH_BF16X3 && math != LFT_MATH_BF16X6) return fail(LFT_ERR_ARG, "math must be LFT_MATH_F32, LFT_MATH_BF16X3 or LFT_MATH_BF16X6, got %d", math);
    return train_backward(params, lr, static_cast<float*>(tape), dout, grads, d, math, static_cast<hipStream_t>(stream));
}
int lft_train_backward_buckets(const float* const* params, int nparams, const float* lr, void* tape, const float* dout, float* grads,
                               int B, int A, int h, int w, int s, int math, void* stream,
                               lft_bucket_fn on_bucket, void* user) {
    Dims d; int rc;
    if (!params || !lr || !tape || !dout || !grads || !on_bucket) return fail(LFT_ERR_ARG, "null pointer");
    if (nparams != LFT_NUM_PARAMS) return fail(LFT_ERR_ARG, "expected %d parameter tensors, got %d", LFT_NUM_PARAMS, nparams);
    for (int i = 0; i < nparams; ++i) if (!params[i]) return fail(LFT_ERR_ARG, "parameter %d is null", i);
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    if (math != LFT_MATH_F32 && math != LFT_MATH_BF16X3 && math != LFT_MATH_BF16X6) return fail(LFT_ERR_ARG, "math must be LFT_MATH_F32, LFT_MATH_BF16X3 or LFT_MATH_BF16X6, got %d", math);
    return train_backward(params, lr, static_cast<float*>(tape), dout, grads, d, math, static_cast<hipStream_t>(stream), on_bucket, user);
}
int lft_train_block_backward(const float* const* params, int nparams, const float* lr, void* tape, int block, int layer,
                             const float* d_out, float* d_in, float* grads,
                             int B, int A, int h, int w, int s, int math, void* stream) {
    Dims d; int rc;
    if (!params || !lr || !tape || !d_out || !grads) return fail(LFT_ERR_ARG, "null pointer");
    if (block < LFT_BLOCK_UPSAMPLE || block > LFT_BLOCK_INIT) return fail(LFT_ERR_ARG, "block must be LFT_BLOCK_UPSAMPLE .. LFT_BLOCK_INIT, got %d", block);
    if ((block == LFT_BLOCK_SPA || block == LFT_BLOCK_ANG) && (layer < 0 || layer >= kLayers)) return fail(LFT_ERR_ARG, "layer %d out of range", layer);
    if (block != LFT_BLOCK_INIT && !d_in) return fail(LFT_ERR_ARG, "d_in is null");
    if (nparams != LFT_NUM_PARAMS) return fail(LFT_ERR_ARG, "expected %d parameter tensors, got %d", LFT_NUM_PARAMS, nparams);
    for (int i = 0; i < nparams; ++i) if (!params[i]) return fail(LFT_ERR_ARG, "parameter %d is null", i);
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    if (math != LFT_MATH_F32 && math != LFT_MATH_BF16X3 && math != LFT_MATH_BF16X6) return fail(LFT_ERR_ARG, "math must be LFT_MATH_F32, LFT_MATH_BF16X3 or LFT_MATH_BF16X6, got %d", math);
    const BlockSel sel{block, layer, d_out, d_in};
    return train_backward(params, lr, static_cast<float*>(tape), block == LFT_BLOCK_UPSAMPLE ? d_out : nullptr, grads, d, math,
                          static_cast<hipStream_t>(stream), nullptr, nullptr, false, nullptr, nullptr, &sel);
}
int lft_train_step_profiled(const float* const* params, int nparams, const float* lr, float* out, void* tape, const float* dout, float* grads,
                            int B, int A, int h, int w, int s, int math, void* stream, int max_records, float* ms_out, const char** names_out, int* n_out) {
    Dims d; int rc;
    if (!params || !lr || !out || !tape || !dout || !grads || !ms_out || !names_out || !n_out) return fail(LFT_ERR_ARG, "null pointer");
    if (nparams != LFT_NUM_PARAMS) return fail(LFT_ERR_ARG, "expected %d parameter tensors, got %d", LFT_NUM_PARAMS, nparams);
    if ((rc = make_dims(B, A, h, w, s, LFT_PREC_F32, &d))) return rc;
    if (math != LFT_MATH_F32 && math != LFT_MATH_BF16X3 && math != LFT_MATH_BF16X6) return fail(LFT_ERR_ARG, "math must be LFT_MATH_F32, LFT_MATH_BF16X3 or LFT_MATH_BF16X6, got %d", math);
    hipStream_t st = static_cast<hipStream_t>(stream);
    g_prof.on = true; g_prof.st = st; g_prof.ev.clear(); g_prof.names.clear();
    prof_mark("start");
    rc = train_forward(params, lr, out, static_cast<float*>(tape), d, math, st);
    if (!rc) rc = train_backward(params, lr, static_cast<float*>(tape), dout, grads, d, math, st, nullptr);    // one stream: every kernel between two events
    g_prof.on = false;
    hipError_t e = hipStreamSynchronize(st);
    int n = 0;
    for (size_t i = 1; i < g_prof.ev.size() && n < max_records; ++i, ++n) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, g_prof.ev[i - 1], g_prof.ev[i]);
        ms_out[n] = ms;
        names_out[n] = g_prof.names[i];
    }
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.clear(); g_prof.names.clear();
    *n_out = n;
    if (rc) return rc;
    if (e != hipSuccess) return fail((int)e, "hipStreamSynchronize: %s", hipGetErrorString(e));
    return 0;
}
int lft_train_grad_bucket(int s, int bucket, size_t* first_float, size_t* n_floats) {
    if (!first_float || !n_floats) return fail(LFT_ERR_ARG, "null pointer");
    if (s != 2 && s != 4) return fail(LFT_ERR_SHAPE, "scale factor must be 2 or 4, got %d", s);
    if (bucket < 0 || bucket >= LFT_GRAD_BUCKETS) return fail(LFT_ERR_ARG, "bucket %d out of range (0..%d)", bucket, LFT_GRAD_BUCKETS - 1);
    grad_bucket_range(s, bucket, first_float, n_floats);
    return 0;
}
int lft_l1_loss(const float* sr, const float* hr, long long n, float* dsr, float gscale, float* loss, float* scratch1024, void* stream) {
    if (!sr || !hr || !loss || !scratch1024 || n < 1) return fail(LFT_ERR_ARG, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = (int)std::min<long long>(1024, (n + 255) / 256);
    k_l1_partial<<<nb, 256, 0, st>>>(sr, hr, dsr, gscale, n, scratch1024);
    LFT_LAUNCH_OK("k_l1_partial");
    k_l1_final<<<1, 64, 0, st>>>(scratch1024, nb, 1.0f / (float)n, loss);
    LFT_LAUNCH_OK("k_l1_final");
    return 0;
}
int lft_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                  int step, float gscale, float weight_decay, void* stream) {
    if (!p || !g || !m || !v || n < 1 || step < 1 || weight_decay < 0.0f) return fail(LFT_ERR_ARG, "bad argument");
    const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step)), bc2 = (float)(1.0 - std::pow((double)beta2, (double)step));   // in double, as torch.optim.Adam
    k_adam<<<blocks_for(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2, gscale, weight_decay);
    LFT_LAUNCH_OK("k_adam");
    return 0;
}

#endif  // LFT_TU != 1
}  // extern "C"
