// lft_train.cuh -- fp32 training kernels (forward with a saved-activation tape + backward) for the LFT network.
//
// The reference trains in fp32 through PyTorch autograd (train.py:74-107: net(data) -> L1Loss -> loss.backward() ->
// Adam).  This file is the MI355X counterpart of what autograd records and replays for model/LFT.py:52-83:
// every linear map of the network as an exact-fp32 MFMA product in the "token on lane" layout of lft_common.cuh,
//   k_lin    Y = act(X W^T) (+R)        any of the network's Linear / 1x1 / 3x3-per-view convolutions, and -- with
//                                       transposed weight strides and flipped taps -- their input gradients
//   k_wgrad  dW = dY^T X                split over token chunks (deterministic two-stage reduction)
// plus the small VALU kernels around them (LayerNorm, both attentions, activations, up-sampler tail).
// Weights change every step: at the start of a step both orientations (W for the forward, W^T for the input gradients)
// are re-packed into fragment order by k_pack (9 MB, microseconds) so that every weight read is a coalesced 1 KiB piece.
// All tensors are fp32 channels-last [token][channel]; token = ((b*V + v)*h + y)*w + x.
#pragma once
#include "lft_common.cuh"
#include "lft_kernels_b.cuh"   // asm-issued loads with counted waits (ld16_async_ofs, wait_vm_q)

// ------------------------------------------------------------------------------------------
// Split-bf16 products ("bf16x3").  The fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) is 16x slower than the
// bf16 one, so an fp32 operand x is split into two bf16 numbers x = hi + lo (+ O(2^-17 x)) and a product is
//   a * b ~= ah*bh + ah*bl + al*bh            (3 MFMAs, fp32 accumulate; the dropped al*bl is O(2^-18 a b))
// i.e. fp32 operands with ~2^-16 relative error per product at 3/16 of the fp32-MFMA cost.
// Packed weights in this mode: fragment f = [1 KiB hi piece][1 KiB lo piece] (same 2 KiB as an fp32 fragment).
// ------------------------------------------------------------------------------------------
struct Frag2 { Frag<bf16_t> hi, lo; };
LFT_DEV Frag2 split_frag(const Frag<float>& f) {
    Frag2 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? f.lo[j] : f.hi[j - 4];
        const bf16_t h = (bf16_t)x;
        r.hi.v[j] = h;
        r.lo.v[j] = (bf16_t)(x - (float)h);
    }
    return r;
}
LFT_DEV void mma3(const Frag2& a, const Frag2& b, f32x16& c) {
    mma(a.lo, b.hi, c);
    mma(a.hi, b.lo, c);
    mma(a.hi, b.hi, c);
}
// "bf16x6" (round 4): an fp32 number IS the sum of three bf16 numbers, x = a + b + c exactly (8 + 8 + 8 significant bits; each
// remainder is exact in fp32), so a product is the sum of nine bf16 products of which the three smallest (b c', c b', c c':
// <= 2^-23 |x y| together) are dropped:
//   x * y ~= a a' + (a b' + b a') + (a c' + c a' + b b')        6 MFMAs, fp32 accumulate, smallest terms first
// -- fp32-class products (the fp32 MFMA rounds its product to 2^-24) at 6/16 of the fp32-MFMA cost.  Used by the weight-gradient
// kernel, whose operands are activations split in registers (no packed weights, no layout change).
struct Frag3 { Frag<bf16_t> a, b, c; };
LFT_DEV void split3(float x, bf16_t& a, bf16_t& b, bf16_t& c) {
    a = (bf16_t)x;
    const float r1 = x - (float)a;
    b = (bf16_t)r1;
    c = (bf16_t)(r1 - (float)b);
}
LFT_DEV Frag3 split3_frag(const Frag<float>& f) {
    Frag3 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bf16_t a, b, c;
        split3(j < 4 ? f.lo[j] : f.hi[j - 4], a, b, c);
        r.a.v[j] = a; r.b.v[j] = b; r.c.v[j] = c;
    }
    return r;
}
LFT_DEV void mma6(const Frag3& x, const Frag3& y, f32x16& c) {
    mma(x.c, y.a, c);
    mma(x.a, y.c, c);
    mma(x.b, y.b, c);
    mma(x.b, y.a, c);
    mma(x.a, y.b, c);
    mma(x.a, y.a, c);
}
LFT_DEV Frag2 load_wfrag2(const float* __restrict__ stream, int f, int lane) {
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)f * 2048 + lane * 16;
    Frag2 r;
    r.hi.v = __builtin_bit_cast(bf16x8, load_raw16(base));
    r.lo.v = __builtin_bit_cast(bf16x8, load_raw16(base + 1024));
    return r;
}
// bf16x6: packed weight fragment f = [1 KiB a][1 KiB b][1 KiB c] (3 KiB; the fp32 and split-bf16 fragments are 2 KiB)
constexpr int kFragBytes2 = 2048, kFragBytes3 = 3072;
LFT_DEV Frag3 load_wfrag3(const float* __restrict__ stream, int f, int lane) {
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)f * kFragBytes3 + lane * 16;
    Frag3 r;
    r.a.v = __builtin_bit_cast(bf16x8, load_raw16(base));
    r.b.v = __builtin_bit_cast(bf16x8, load_raw16(base + 1024));
    r.c.v = __builtin_bit_cast(bf16x8, load_raw16(base + 2048));
    return r;
}
// k_pack's twin for the split modes: same PackOp description (natural k order), writes hi / lo pieces (THREE = false) or the
// three exact bf16 parts a / b / c (THREE = true, 3 KiB per fragment).
template <bool THREE>
__global__ __launch_bounds__(64) void k_pack_split(PackArgs args, float* __restrict__ dst) {
    const int f = blockIdx.x, lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    int oi = 0;
    for (int i = 0; i < args.nops; ++i)
        if (f >= args.op[i].frag0) oi = i;
    const PackOp& op = args.op[oi];
    int nt, ks;
    wfrag_coords(op.ntiles, op.ksteps, f - op.frag0, op.order, nt, ks);
    const int n = 32 * nt + r;
    bf16_t* d = reinterpret_cast<bf16_t*>(dst) + (size_t)f * (THREE ? 1536 : 1024);   // 2 KiB = 1024 bf16 (3 KiB = 1536) per fragment
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = op.k0 + 16 * ks + 8 * h + j;
        float v = 0.0f;
        if (n < op.nrows) {
            if (op.kind == 0) v = op.scale * op.src[(size_t)(op.row0 + n) * op.ld + kk * op.kmul + op.kadd];
            else if (op.kind == 1) v = upm_entry(op.src, n, kk, op.s);
            else v = kk < (op.s + 2) * (op.s + 2) ? upm_entry(op.src, kk, n, op.s) : 0.0f;
        }
        if constexpr (THREE) {
            bf16_t a, b, c;
            split3(v, a, b, c);
            d[lane * 8 + j] = a; d[512 + lane * 8 + j] = b; d[1024 + lane * 8 + j] = c;
        } else {
            const bf16_t hi = (bf16_t)v;
            d[lane * 8 + j] = hi;
            d[512 + lane * 8 + j] = (bf16_t)(v - (float)hi);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic linear / per-view 3x3 convolution on MFMA.
//   Y[t][o] = act( sum_tap sum_i W(o, i, tap) * X[shift_tap(t)][i] ) (+ R[t][o])
// taps == 1: plain Linear.  taps == 9: tap -> (dy, dx) = (tap/3 - 1, tap%3 - 1), the source token is
// (y + dy, x + dx) of the same view image (zero outside: per-view zero padding, reference LFT.py:24,28,167);
// flip negates the offset, which together with the transposed packing turns the kernel into the convolution's input gradient.
// A wave owns 32 tokens x NT*32 output channels (blockIdx.y selects the channel group).
// ------------------------------------------------------------------------------------------
struct LinP {
    const float* X; int ldx;
    const float* Wp; int OT, KS, ot0; // packed fragments of the whole view (k_pack order 1, natural k order): frag (tap, ot, ks) at tap*OT*KS + wfrag_index(OT, KS, ot, ks); ot0 = first output tile of this call
    const float* R; int ldr;          // optional: accumulator initialised with R (residual, or Y itself to accumulate)
    float* Y; int ldy;
    const float* M; int ldm, mact;    // optional backward-of-activation epilogue: Y = acc * act'(M), M = the saved activation OUTPUT (mact 1 relu, 2 lrelu)
    int taps, flip, act;              // act: 0 none, 1 relu, 2 leaky relu 0.2
    int h, w;
    long long N;
    int gy;                           // k_linr: output groups (workgroups per 128 tokens)
};

// TILED (plain Linears at small batch): the 32 input rows of a wave, consecutive in memory, are fetched in coalesced
// 64-channel chunks through the scratch instead of one 32-byte piece per lane and k-step -- fewer, fuller memory
// requests when there are too few waves to hide latency; at large batch the direct form's higher occupancy wins.
template <int NT, int MM, bool TILED>      // MM: 0 exact fp32 MFMA, 1 split-bf16 (3 products), 2 bf16x6 (6 products, fp32-class)
__global__ __launch_bounds__(256, 3) void k_lin(const LinP p) {
    constexpr bool M3 = MM == 1;
    constexpr int SCR = TileIO<(NT > 2 ? NT : 2), float>::BYTES;          // output tile or a 64-channel input chunk
    __shared__ __attribute__((aligned(16))) char scr_all[4 * SCR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, kh = lane >> 5;
    char* scr = scr_all + wave * SCR;
    const long long t0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (t0 >= p.N) return;
    const int nvalid = (int)min((long long)32, p.N - t0);
    const int otl = blockIdx.y * NT, o0 = otl * 32, ot0 = p.ot0 + otl;     // otl: tile inside this call's output block (Y / R / M columns), ot0: tile of the view
    const long long t = min(t0 + r, p.N - 1);
    const int hw = p.h * p.w;
    const int pix = (int)(t % hw), y = pix / p.w, x = pix - y * p.w;
    f32x16 acc[NT];
    if (p.R) load_tile<NT, float>(p.R + t0 * p.ldr + o0, nvalid, lane, acc, scr, (size_t)p.ldr * 4);
    else zero_acc<NT>(acc);
    if constexpr (TILED) {
        // plain Linear: the wave's 32 input rows are consecutive in memory -> coalesced 64-channel chunks through the scratch
        for (int k0 = 0; k0 < p.KS; k0 += 4) {
            Frag<float> b[4];
            load_tile_frags_s<4, float>(p.X + t0 * p.ldx + 16 * k0, (size_t)p.ldx * 4, nvalid, lane, b, scr);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (MM == 2) {
                    const Frag3 b3 = split3_frag(b[ks]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma6(load_wfrag3(p.Wp, wfrag_index(p.OT, p.KS, ot0 + nt, k0 + ks), lane), b3, acc[nt]);
                } else if constexpr (M3) {
                    const Frag2 b2 = split_frag(b[ks]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma3(load_wfrag2(p.Wp, wfrag_index(p.OT, p.KS, ot0 + nt, k0 + ks), lane), b2, acc[nt]);
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma(load_wfrag(p.Wp, wfrag_index(p.OT, p.KS, ot0 + nt, k0 + ks), lane), b[ks], acc[nt]);
                }
            }
        }
    } else
    for (int tap = 0; tap < p.taps; ++tap) {
        int dy = 0, dx = 0;
        if (p.taps == 9) { dy = tap / 3 - 1; dx = tap % 3 - 1; if (p.flip) { dy = -dy; dx = -dx; } }
        const bool ok = (t0 + r < p.N) && (y + dy >= 0) && (y + dy < p.h) && (x + dx >= 0) && (x + dx < p.w);
        const float* row = p.X + (ok ? (t + dy * p.w + dx) : t) * p.ldx + 8 * kh;
        const int fbase = tap * p.OT * p.KS;
        for (int ks = 0; ks < p.KS; ++ks) {
            const Frag<float> b = load_row8(row + 16 * ks, ok, 0.0f);
            if constexpr (MM == 2) {
                const Frag3 b3 = split3_frag(b);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma6(load_wfrag3(p.Wp, fbase + wfrag_index(p.OT, p.KS, ot0 + nt, ks), lane), b3, acc[nt]);
            } else if constexpr (M3) {
                const Frag2 b2 = split_frag(b);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma3(load_wfrag2(p.Wp, fbase + wfrag_index(p.OT, p.KS, ot0 + nt, ks), lane), b2, acc[nt]);
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma(load_wfrag(p.Wp, fbase + wfrag_index(p.OT, p.KS, ot0 + nt, ks), lane), b, acc[nt]);
            }
        }
    }
    if (p.act) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[nt][i];
                acc[nt][i] = v > 0.0f ? v : (p.act == 1 ? 0.0f : 0.2f * v);
            }
    }
    if (p.M) {                                                           // dZ = dY * act'(Z), sign(Z) read off the saved act(Z)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {                                // one 32-channel tile at a time: 16 live registers, not 16 NT
            f32x16 mk[1];
            load_tile<1, float>(p.M + t0 * p.ldm + o0 + 32 * nt, nvalid, lane, mk, scr, (size_t)p.ldm * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = mk[0][i] > 0.0f ? acc[nt][i] : (p.mact == 1 ? 0.0f : 0.2f * acc[nt][i]);
        }
    }
    store_tile<NT, float>(p.Y + t0 * p.ldy + o0, nvalid, lane, acc, scr, (size_t)p.ldy * 4);
}

// ------------------------------------------------------------------------------------------
// k_lin with the weights through LDS.  The four waves of a workgroup (32 tokens each) multiply by the SAME NT x KS (x taps)
// fragments; read per wave from L2 they are 4 KB per token against 1 KB of activations in and 1 KB out (a knock-out build
// without the weight reads ran the family 20 % faster, one without the MFMAs not at all).  Here the group's fragments -- one
// contiguous stream in consumption order (k_pack order 1) -- pass once per workgroup through the 3-slot LDS ring of the inference
// kernels (as WRing: one chunk = one k-step's NT fragments, LDS-DMA two k-steps ahead, one barrier per k-step), and the activation
// rows of the NEXT k-step are in flight while this one is multiplied.
// Ordering (vmcnt retires in issue order).  The rows are ordinary loads: hipcc waits for them itself, so whatever it does with
// the registers is safe (rows loaded from inline asm were COPIED in front of the asm wait that was to guard them).  Rows of
// k-step s are issued in iteration s - 1, i.e. after the DMA of chunk s (behind barrier s - 2) and before that of chunk s + 1:
// the compiler's wait for them -- pinned in front of barrier s -- therefore also covers this wave's pieces of chunk s, and
// the ring needs no wait of its own.
// LDS: ring slots 0, 1 | slot 2 overlaid by the waves' tile scratch -- the scratch is used before barrier 0 (residual tile)
// and after the last k-step (activation-derivative tile, output tile); slot 2 is first written by the DMA behind barrier 0.
// Requires: the call's output block is one packed group (NT == its tile count, first tile a multiple of 4); taps == 9: OT == NT.
// ------------------------------------------------------------------------------------------
// One token to the left / right inside a wave's 32-token row (lanes 0-31 and 32-63 hold the two k-halves of the same tokens):
// DIR = +1: lane r takes lane r + 1 (v_mov_dpp wave_shl:1), DIR = -1: lane r - 1 (wave_shr:1); the row's first / last token takes 0
// (the lane the shift would pull across the two halves, or from outside the wave).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int DIR> LFT_DEV u32x4_t lane_shift1_u(u32x4_t u, bool edge) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = __builtin_amdgcn_update_dpp(0, (int)u[i], DIR > 0 ? 0x130 : 0x138, 0xf, 0xf, true);
        u[i] = edge ? 0u : (unsigned)s;
    }
    return u;
}
template <int DIR> LFT_DEV bf16x8 lane_shift1(bf16x8 v, bool edge) { return __builtin_bit_cast(bf16x8, lane_shift1_u<DIR>(__builtin_bit_cast(u32x4_t, v), edge)); }
template <int DIR> LFT_DEV f32x4 lane_shift1(f32x4 v, bool edge) { return __builtin_bit_cast(f32x4, lane_shift1_u<DIR>(__builtin_bit_cast(u32x4_t, v), edge)); }

// KS3 > 0 (= KS, 4 or 8): a per-view 3x3 convolution on 32-wide views, where a wave's 32 tokens are ONE image row.  The KS row
// fragments of an input row (dy) are loaded ONCE and serve its three taps: the neighbours to the left and right are the same
// registers one lane over (lane_shift1), the image's left / right border is the shift's zero.  Input rows come from L2 three
// times instead of nine and the split into bf16 pairs is done once per row fragment instead of once per tap -- in the SAME step
// order (tap row, tap column, k-step) and with the same operand values as the generic form, so the results are bit-identical.
template <int NT, int MM, int KS3 = 0>     // MM as k_lin
__global__ __launch_bounds__(256, 3) void k_linr(const LinP p) {
    constexpr bool M3 = MM == 1;
    constexpr int FB = MM == 2 ? kFragBytes3 : kFragBytes2;               // bytes per packed fragment
    constexpr int CHUNK = NT * FB, NPIECE = CHUNK / 1024;                 // bytes / 1 KiB pieces of weights per k-step
    // pieces per wave and chunk: NPIECE / 4, except bf16x6 with two output tiles (6 pieces): waves 0, 1 move two, waves 2, 3 one
    constexpr bool EVEN = NPIECE % 4 == 0;
    constexpr int PPW = EVEN ? NPIECE / 4 : 2;
    constexpr int SCR = TileIO<NT, float>::BYTES;
    static_assert(NT == 2 || NT == 4, "two or four output tiles");
    __shared__ __attribute__((aligned(16))) char lds[2 * CHUNK + (4 * SCR > CHUNK ? 4 * SCR : CHUNK)];
    const int lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* scr = lds + 2 * CHUNK + wave * SCR;
    // 1-D grid: the p.gy workgroups (output groups) that read the same 128 tokens get ids 8 apart = the same XCD at about the
    // same time: the second reader of the rows hits in that XCD's L2 (as grid (tokens, group) they were a launch apart)
    const int grp8 = (int)blockIdx.x / (8 * p.gy), rem = (int)blockIdx.x % (8 * p.gy);
    const long long tile = (long long)grp8 * 8 + (rem & 7);
    const int by = rem >> 3;
    const long long t0w = (tile * 4 + wave) * 32;
    if (tile * 128 >= p.N) return;                                       // (whole workgroup: the last group of 8 may be short)
    const bool active = t0w < p.N;                                       // a wave past the end still runs the ring protocol (barriers, DMA share)
    const long long t0 = active ? t0w : 0;
    const int nvalid = active ? (int)min((long long)32, p.N - t0) : 0;
    const int otl = by * NT, o0 = otl * 32, ot0 = p.ot0 + otl;
    const long long t = min(t0 + r, p.N - 1);
    const int hw = p.h * p.w;
    const int pix = (int)(t % hw), y = pix / p.w, x = pix - y * p.w;
    const int S = p.taps * p.KS;                                         // k-steps = ring chunks
    // the ring: chunk c (the NT fragments of k-step c) in slot c % 3; every wave moves PPW pieces of it.  The DMA is issued from
    // inline asm: hipcc does not know it, so its waits for the row loads are computed without it (stricter, never unsafe) and
    // it does not guard the ring's LDS reads with a vmcnt(0) of its own (it did, with the builtin form)
    const int my_first = EVEN ? wave * PPW : (wave < 2 ? 2 * wave : 2 + wave), my_count = EVEN ? PPW : (wave < 2 ? 2 : 1);   // wave-uniform
    const char* wsrc = reinterpret_cast<const char*>(p.Wp) + (size_t)wfrag_index(p.OT, p.KS, ot0, 0) * FB + my_first * 1024 + lane * 16;
    auto issue_w = [&](int c, int slot) {
        if (c >= S) return;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (EVEN || i < my_count) glds16_asm(wsrc + (size_t)c * CHUNK + i * 1024, lds + slot * CHUNK + (my_first + i) * 1024);
    };
    issue_w(0, 0);
    issue_w(1, 1);
    f32x16 acc[NT];
    if (p.R) load_tile<NT, float>(p.R + t0 * p.ldr + o0, nvalid, lane, acc, scr, (size_t)p.ldr * 4);
    else zero_acc<NT>(acc);
    if constexpr (KS3 > 0) {
        const bool lane_ok = t0w + r < p.N, e_lo = r == 0, e_hi = r == 31;
        int cs = 0, slot = 0;
        for (int dyi = 0; dyi < 3; ++dyi) {
            const int dy = p.flip ? 1 - dyi : dyi - 1;
            const bool okr = lane_ok && (y + dy >= 0) && (y + dy < p.h);
            const float* row = p.X + (okr ? t + dy * p.w : t) * p.ldx + 8 * kh;
            Frag<float> xr[KS3];
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) xr[ks] = load_row8(row + 16 * ks, true, 0.0f);
            Frag<float> xf[MM == 0 ? KS3 : 1];
            Frag2 x2[M3 ? KS3 : 1];
            Frag3 x3[MM == 2 ? KS3 : 1];
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
                const Frag<float> z = okr ? xr[ks] : frag_zero(0.0f);
                if constexpr (MM == 2) x3[ks] = split3_frag(z); else if constexpr (M3) x2[ks] = split_frag(z); else xf[ks] = z;
            }
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int dx = p.flip ? 1 - dxi : dxi - 1;               // uniform
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) {
                    // this wave's pieces of chunk cs have landed (only those of chunk cs + 1 may still be in flight: the counted wait
                    // of WRing), then the barrier publishes the chunk and retires chunk cs - 1
                    wait_vmcnt(cs + 1 < S ? my_count : 0);
                    wg_barrier_keep_vm();
                    issue_w(cs + 2, slot == 0 ? 2 : slot - 1);
                    const char* base = lds + slot * CHUNK + lane * 16;
                    Frag2 b2;
                    Frag3 b3;
                    Frag<float> b;
                    if constexpr (MM == 2) {
                        b3 = x3[ks];
                        if (dx > 0) { b3.a.v = lane_shift1<1>(x3[ks].a.v, e_hi); b3.b.v = lane_shift1<1>(x3[ks].b.v, e_hi); b3.c.v = lane_shift1<1>(x3[ks].c.v, e_hi); }
                        else if (dx < 0) { b3.a.v = lane_shift1<-1>(x3[ks].a.v, e_lo); b3.b.v = lane_shift1<-1>(x3[ks].b.v, e_lo); b3.c.v = lane_shift1<-1>(x3[ks].c.v, e_lo); }
                    } else if constexpr (M3) {
                        b2 = x2[ks];
                        if (dx > 0) { b2.hi.v = lane_shift1<1>(x2[ks].hi.v, e_hi); b2.lo.v = lane_shift1<1>(x2[ks].lo.v, e_hi); }
                        else if (dx < 0) { b2.hi.v = lane_shift1<-1>(x2[ks].hi.v, e_lo); b2.lo.v = lane_shift1<-1>(x2[ks].lo.v, e_lo); }
                    } else {
                        b = xf[ks];
                        if (dx > 0) { b.lo = lane_shift1<1>(xf[ks].lo, e_hi); b.hi = lane_shift1<1>(xf[ks].hi, e_hi); }
                        else if (dx < 0) { b.lo = lane_shift1<-1>(xf[ks].lo, e_lo); b.hi = lane_shift1<-1>(xf[ks].hi, e_lo); }
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (MM == 2) {
                            Frag3 w3;
                            w3.a.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * FB));
                            w3.b.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * FB + 1024));
                            w3.c.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * FB + 2048));
                            mma6(w3, b3, acc[nt]);
                        } else if constexpr (M3) {
                            Frag2 w2;
                            w2.hi.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * 2048));
                            w2.lo.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * 2048 + 1024));
                            mma3(w2, b2, acc[nt]);
                        } else {
                            Frag<float> wf;
                            wf.lo = __builtin_bit_cast(f32x4, load_raw16(base + nt * 2048));
                            wf.hi = __builtin_bit_cast(f32x4, load_raw16(base + nt * 2048 + 1024));
                            mma(wf, b, acc[nt]);
                        }
                    }
                    ++cs;
                    slot = slot == 2 ? 0 : slot + 1;
                }
            }
        }
    } else {
    // cursor of the next row load: (tap, k-step); past the last k-step it stays there (one redundant, cached reload)
    int ctap = 0, cks = 0;
    auto load_x = [&](bool& ok) -> Frag<float> {                         // raw rows; zeroed by `ok` where they are consumed (a select here would pull the wait up to here)
        int dy = 0, dx = 0;
        if (p.taps == 9) { dy = ctap / 3 - 1; dx = ctap % 3 - 1; if (p.flip) { dy = -dy; dx = -dx; } }
        ok = (t0w + r < p.N) && (y + dy >= 0) && (y + dy < p.h) && (x + dx >= 0) && (x + dx < p.w);
        const float* row = p.X + (ok ? (t + dy * p.w + dx) : t) * p.ldx + 8 * kh + 16 * cks;
        if (cks + 1 < p.KS) ++cks; else if (ctap + 1 < p.taps) { cks = 0; ++ctap; }
        return load_row8(row, true, 0.0f);
    };
    int cs = 0, slot = 0;                                                // current k-step and its ring slot
    auto step = [&](const Frag<float>& xraw, bool ok) {
        const Frag<float> xs = ok ? xraw : frag_zero(0.0f);
        Frag2 b2;
        Frag3 b3;
        if constexpr (MM == 2) {
            b3 = split3_frag(xs);
            asm volatile("" :: "v"(b3.a.v), "v"(b3.b.v), "v"(b3.c.v));
        } else if constexpr (M3) {
            b2 = split_frag(xs);                                         // (the compiler's wait for the rows sits here, in front of the barrier)
            asm volatile("" :: "v"(b2.hi.v), "v"(b2.lo.v));
        } else {
            asm volatile("" :: "v"(xs.lo), "v"(xs.hi));
        }
        wg_barrier_keep_vm();                                            // chunk cs published by every wave, chunk cs - 1 retired
        issue_w(cs + 2, slot == 0 ? 2 : slot - 1);                       // into the slot chunk cs - 1 vacated
        const char* base = lds + slot * CHUNK + lane * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (MM == 2) {
                Frag3 w3;
                w3.a.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * FB));
                w3.b.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * FB + 1024));
                w3.c.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * FB + 2048));
                mma6(w3, b3, acc[nt]);
            } else if constexpr (M3) {
                Frag2 w2;
                w2.hi.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * 2048));
                w2.lo.v = __builtin_bit_cast(bf16x8, load_raw16(base + nt * 2048 + 1024));
                mma3(w2, b2, acc[nt]);
            } else {
                Frag<float> wf;
                wf.lo = __builtin_bit_cast(f32x4, load_raw16(base + nt * 2048));
                wf.hi = __builtin_bit_cast(f32x4, load_raw16(base + nt * 2048 + 1024));
                mma(wf, xs, acc[nt]);
            }
        }
        ++cs;
        slot = slot == 2 ? 0 : slot + 1;
    };
    bool oka, okb;
    Frag<float> xa = load_x(oka), xb;
    for (int s = 0; s < S; s += 2) {                                    // S is even (KS is); two register sets, no copy between them
        xb = load_x(okb);                                                // rows of k-step s + 1: in flight under this step's MFMAs
        step(xa, oka);
        xa = load_x(oka);
        step(xb, okb);
    }
    }
    wg_barrier_keep_vm();                                                // every wave is done with the ring: slot 2 is scratch again
    if (p.act) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[nt][i];
                acc[nt][i] = v > 0.0f ? v : (p.act == 1 ? 0.0f : 0.2f * v);
            }
    }
    if (p.M) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 mk[1];
            load_tile<1, float>(p.M + t0 * p.ldm + o0 + 32 * nt, nvalid, lane, mk, scr, (size_t)p.ldm * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = mk[0][i] > 0.0f ? acc[nt][i] : (p.mact == 1 ? 0.0f : 0.2f * acc[nt][i]);
        }
    }
    store_tile<NT, float>(p.Y + t0 * p.ldy + o0, nvalid, lane, acc, scr, (size_t)p.ldy * 4);
}

// ------------------------------------------------------------------------------------------
// Weight gradient: part[chunk][o*so + i*si + tap*st] = sum_{t in chunk} dY[t][o] * X[shift_tap(t)][i]
// MFMA with k = token: A[m = o][k] = dY[t0 + k][o], B[k][n = i] = X[shift(t0 + k)][i]; both operands are read
// straight from the channels-last tensors (lanes = consecutive channels of one token: 128-byte segments).
// A wave owns one 32-row block of dY channels, one tap, NI 32-column blocks of X channels and one token chunk.
// grid: 1-D over (chunk, (o tile, i group), tap row), see the kernel.   k_reduce_all then sums the chunks in a fixed order.
// ------------------------------------------------------------------------------------------
struct WgP {
    const float* dY; int ldy;
    const float* X; int ldx;
    float* part; long long wsize;     // floats per chunk image of the weight
    int so, si, st;
    int Co, Ci, taps;
    int h, w;
    long long N, chunk_len;           // chunk_len % 64 == 0: four waves x 16-token k-steps
    int igroups;
    int nch, gy;                      // token chunks; workgroups per chunk and tap row (output tiles x input groups)
};

// TX = 3: the wave owns a whole ROW of taps (dy fixed by blockIdx.z, dx = -1, 0, +1): the dY fragment is loaded (and
// split) once for three products and the three shifted X rows are neighbours in memory.
// The 4 waves of a workgroup split the chunk's tokens and add their accumulators through LDS, in a fixed order:
// one partial image per workgroup (a quarter of the partial-sum traffic for the same number of waves in flight).
template <int NI, int MM, int TX>      // MM: 0 exact fp32 MFMA, 1 split-bf16 (3 products), 2 bf16x6 (6 products, fp32-class)
__global__ __launch_bounds__(256, TX == 3 ? 2 : 3) void k_wgrad(const WgP p) {
    constexpr bool M3 = MM == 1;
    extern __shared__ __attribute__((aligned(16))) float wred[];        // [3 waves][TX * NI tiles][16][64]
    const int lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // 1-D grid: the gy (x 3 tap rows) workgroups that read the SAME token chunk get workgroup ids 8 apart -- ids are dealt round
    // robin to the 8 XCDs, so they run on one XCD at about the same time and the chunk's second .. last reader hits in its L2
    // (as grid (chunk, tile, tap row) they were a whole launch apart: every tile and tap row re-read its operands from HBM)
    const int per = p.gy * (TX == 3 ? 3 : 1);
    const int grp = (int)blockIdx.x / (8 * per), rem = (int)blockIdx.x % (8 * per);
    const int chunk = grp * 8 + (rem & 7), sub_id = rem >> 3;
    if (chunk >= p.nch) return;                                         // (whole workgroup: the last group of 8 may be short)
    const int by = sub_id % p.gy, bz = sub_id / p.gy;
    const int ot = by / p.igroups, ig = by % p.igroups;
    const int o0 = ot * 32, i0 = ig * NI * 32;
    const long long sub = p.chunk_len >> 2;                              // chunk_len % 64 == 0
    const long long ta = (long long)chunk * p.chunk_len + wave * sub, tb = min(ta + sub, p.N);
    const int hw = p.h * p.w;
    const int dy = TX == 3 ? bz - 1 : 0;
    f32x16 acc[TX][NI];
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) zero_acc<NI>(acc[tx]);
    // Addressing: wave-uniform anchors (scalar registers) + one 32-bit byte offset per lane, advanced by 16 tokens per step; the
    // lane's image position (ys, xs) is advanced the same way -- the 64-bit token arithmetic (a division per fragment element:
    // ~1 000 of the loop's 1 600 instructions around 12 MFMAs) happens once per wave.  Lanes outside the chunk or the image read
    // the anchor's own row and are zeroed.
    const int nsub = (int)max(0LL, tb - ta);
    const int pre = TX == 3 ? p.w + 1 : 0;                              // the X anchor lies `pre` tokens before the wave's first token
    const char* ya = reinterpret_cast<const char*>(p.dY) + (ta * p.ldy + o0) * 4;
    const char* xa = reinterpret_cast<const char*>(p.X) + ((ta - pre) * p.ldx + i0) * 4;
    const unsigned safe_x = ((unsigned)pre * p.ldx + r) * 4u, safe_y = r * 4u;
    unsigned vy = ((unsigned)(8 * kh) * p.ldy + r) * 4u, vx = ((unsigned)(8 * kh + pre) * p.ldx + r) * 4u;
    int rl = 8 * kh, ys = 0, xs = 0;
    if constexpr (TX == 3) { const int pix = (int)((ta + 8 * kh) % hw); ys = pix / p.w; xs = pix - ys * p.w; }
    // 3x3 on 32-wide views: a 16-token step lies in one half of ONE image row, every step is full (N, the chunk and the wave's
    // share are multiples of 16).  A lane's three tap fragments X[t + j - 1], X[t + j], X[t + j + 1] (j = 0..7) are ten values:
    // eight centre ones, a left and a right neighbour -- 10 loads instead of 24, and in split mode each value is split ONCE and
    // the three fragments are two packings of the same bf16 pairs (even pairs (c0,c1).. for the centre tap, odd pairs
    // (L,c0),(c1,c2)..,(c7,R) for the other two).  Same values, same MFMA order as the general form below: bit-identical.
    bool fast3 = false;
    if constexpr (TX == 3) fast3 = p.w == 32 && nsub % 16 == 0 && ta % 16 == 0;
    if (fast3) {
        if constexpr (TX == 3) {
            const int pix0 = (int)(ta % hw);
            int yw = pix0 / 32, xb = pix0 % 32;                             // wave-uniform: image row and half (0 / 16) of the step
            for (int step = 0; step * 16 < nsub; ++step) {
                const bool okrow = (yw + dy >= 0) && (yw + dy < p.h);       // uniform
                const bool okl = okrow && (xb + 8 * kh > 0), okr = okrow && (xb + 8 * kh + 8 < 32);
                Frag<float> a;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float av = *reinterpret_cast<const float*>(ya + vy + (unsigned)(j * p.ldy) * 4u);
                    if (j < 4) a.lo[j] = av; else a.hi[j - 4] = av;
                }
                float v[NI][10];                                            // [0] = left neighbour, [1..8] = centre, [9] = right neighbour
#pragma unroll
                for (int e = 0; e < 10; ++e) {
                    const bool ok = e == 0 ? okl : e == 9 ? okr : okrow;
                    const float* xr = reinterpret_cast<const float*>(xa + (ok ? vx + (unsigned)((e - 1 + dy * 32) * p.ldx) * 4u : safe_x));
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) { const float t = xr[32 * ni]; v[ni][e] = ok ? t : 0.0f; }
                }
                vy += 16u * p.ldy * 4u; vx += 16u * p.ldx * 4u;
                xb += 16;
                if (xb == 32) { xb = 0; if (++yw == p.h) yw = 0; }
                if constexpr (MM == 2) {
                    const Frag3 a3 = split3_frag(a);
                    bf16_t sa[NI][10], sb[NI][10], sc[NI][10];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int e = 0; e < 10; ++e) split3(v[ni][e], sa[ni][e], sb[ni][e], sc[ni][e]);
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            Frag3 b3;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { b3.a.v[j] = sa[ni][j + tx]; b3.b.v[j] = sb[ni][j + tx]; b3.c.v[j] = sc[ni][j + tx]; }
                            mma6(a3, b3, acc[tx][ni]);
                        }
                } else if constexpr (M3) {
                    const Frag2 a2 = split_frag(a);
                    bf16_t hi[NI][10], lo[NI][10];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int e = 0; e < 10; ++e) { hi[ni][e] = (bf16_t)v[ni][e]; lo[ni][e] = (bf16_t)(v[ni][e] - (float)hi[ni][e]); }
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            Frag2 b2;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { b2.hi.v[j] = hi[ni][j + tx]; b2.lo.v[j] = lo[ni][j + tx]; }
                            mma3(a2, b2, acc[tx][ni]);
                        }
                } else {
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            Frag<float> b;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { if (j < 4) b.lo[j] = v[ni][j + tx]; else b.hi[j - 4] = v[ni][j + tx]; }
                            mma(a, b, acc[tx][ni]);
                        }
                }
            }
        }
    } else
    for (int step = 0; step * 16 < nsub; ++step) {
        Frag<float> a, b[TX][NI];
        int y = ys, x = xs;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = rl + j < nsub;
            const bool okrow = in && (y + dy >= 0) && (y + dy < p.h);
            const float av = *reinterpret_cast<const float*>(ya + (in ? vy + (unsigned)(j * p.ldy) * 4u : safe_y));
            if (j < 4) a.lo[j] = in ? av : 0.0f; else a.hi[j - 4] = in ? av : 0.0f;
#pragma unroll
            for (int tx = 0; tx < TX; ++tx) {
                const int dx = TX == 3 ? tx - 1 : 0;
                const bool ok = TX == 3 ? okrow && (x + dx >= 0) && (x + dx < p.w) : in;
                const float* xr = reinterpret_cast<const float*>(xa + (ok ? vx + (unsigned)((j + dy * p.w + dx) * p.ldx) * 4u : safe_x));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float bv = xr[32 * ni];
                    const float bz = TX == 3 ? (ok ? bv : 0.0f) : bv;   // plain: a row outside the chunk meets a zeroed dY element (the anchor row read instead is finite data)
                    if (j < 4) b[tx][ni].lo[j] = bz; else b[tx][ni].hi[j - 4] = bz;
                }
            }
            if constexpr (TX == 3) { if (++x == p.w) { x = 0; if (++y == p.h) y = 0; } }
        }
        rl += 16; vy += 16u * p.ldy * 4u; vx += 16u * p.ldx * 4u;
        if constexpr (TX == 3) {
            xs += 16;
            while (xs >= p.w) { xs -= p.w; if (++ys == p.h) ys = 0; }
        }
        if constexpr (MM == 2) {
            const Frag3 a3 = split3_frag(a);
#pragma unroll
            for (int tx = 0; tx < TX; ++tx)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma6(a3, split3_frag(b[tx][ni]), acc[tx][ni]);
        } else if constexpr (M3) {
            const Frag2 a2 = split_frag(a);
#pragma unroll
            for (int tx = 0; tx < TX; ++tx)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma3(a2, split_frag(b[tx][ni]), acc[tx][ni]);
        } else {
#pragma unroll
            for (int tx = 0; tx < TX; ++tx)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma(a, b[tx][ni], acc[tx][ni]);
        }
    }
    if (wave) {
#pragma unroll
        for (int tx = 0; tx < TX; ++tx)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) wred[(((wave - 1) * TX * NI + tx * NI + ni) * 16 + i) * 64 + lane] = acc[tx][ni][i];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int w2 = 0; w2 < 3; ++w2)
#pragma unroll
        for (int tx = 0; tx < TX; ++tx)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tx][ni][i] += wred[((w2 * TX * NI + tx * NI + ni) * 16 + i) * 64 + lane];
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
        const int tap = TX == 3 ? bz * 3 + tx : 0;
        float* dst = p.part + (long long)chunk * p.wsize + (size_t)tap * p.st;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                dst[(size_t)(o0 + acc_row(i, kh)) * p.so + (size_t)(i0 + 32 * ni + r) * p.si] = acc[tx][ni][i];
    }
}

// All partial-sum reductions of one backward pass in ONE launch: every producer (weight-gradient workgroups, LayerNorm
// backward workgroups, the two tail kernels) leaves its partials in its own region of the partial buffer, and a table of
// segments tells this kernel what to sum where.  dst[i] = sum_c part[off + c*stride + i] (+ a second chained partial
// set: the position-token contribution to the embedding weight).  Fixed summation order: deterministic.
struct RedSeg { long long part_off, part2_off, dst_off; int nch, nch2, n, stride, blk0; };
constexpr int kRedMax = 80;
struct RedTab { int nseg, nblk; RedSeg s[kRedMax]; };
__global__ __launch_bounds__(256) void k_reduce_all(const RedTab tab, const float* __restrict__ part, float* __restrict__ grads) {
    __shared__ float red[4][64];
    int lo = 0, hi = tab.nseg - 1;                                      // last segment whose first block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab.s[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const RedSeg& sg = tab.s[lo];
    const int xl = threadIdx.x & 63, yl = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - sg.blk0) * 64 + xl;
    float s0 = 0.0f, s1 = 0.0f;
    if (i < sg.n) {
        const float* p1 = part + sg.part_off + i;
        int c = yl;
        for (; c + 28 < sg.nch; c += 32) {                              // eight loads in flight; each accumulator sees its chunks in the same order as before
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p1[(long long)(c + 4 * k) * sg.stride];
#pragma unroll
            for (int k = 0; k < 8; k += 2) { s0 += v[k]; s1 += v[k + 1]; }
        }
        for (; c + 4 < sg.nch; c += 8) { s0 += p1[(long long)c * sg.stride]; s1 += p1[(long long)(c + 4) * sg.stride]; }
        if (c < sg.nch) s0 += p1[(long long)c * sg.stride];
        const float* p2 = part + sg.part2_off + i;
        for (c = yl; c < sg.nch2; c += 4) s1 += p2[(long long)c * sg.stride];
    }
    red[yl][xl] = s0 + s1;
    __syncthreads();
    if (yl == 0 && i < sg.n) grads[sg.dst_off + i] = (red[0][xl] + red[1][xl]) + (red[2][xl] + red[3][xl]);
}

// ------------------------------------------------------------------------------------------
// Plain position tables (reference LFT.py:86-115): angular [V][64], spatial [h*w][64].
// ------------------------------------------------------------------------------------------
__global__ void k_pe_plain(float* __restrict__ ang, float* __restrict__ spa, int V, int h, int w) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < V * 64) ang[idx] = pe_value(idx >> 6, idx & 63);
    if (idx < h * w * 64) {
        const int pp = idx >> 6, c = idx & 63;
        spa[idx] = (pe_value(pp / w, c) + pe_value(pp % w, c)) / 2.0f;
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm over C channels (C = 64 or 128), input u = X (+ pe): pe_mode 0 none, 1 angular pe[v][c] with
// v = (t / hw) % V, 2 spatial-token pe[t % hw][c].  16 lanes per token, C/16 channels per lane.
// Backward: du = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dY * gamma; out = (add ? add : 0) + du;
// per-block partial sums of dgamma = sum dY * xhat and dbeta = sum dY go to pgb[block][2C].
// ------------------------------------------------------------------------------------------
template <int C>
LFT_DEV void ln_load(const float* __restrict__ X, const float* __restrict__ pe, int pe_mode, long long t, int hw, int V, int sub,
                     float (&u)[C / 16]) {
    constexpr int PER = C / 16;
    const float* row = X + t * C + sub * PER;
#pragma unroll
    for (int g = 0; g < PER / 4; ++g) {
        const f32x4 v = load4(row + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[4 * g + j] = v[j];
    }
    if (pe_mode) {
        const long long prow = pe_mode == 1 ? (t / hw) % V : t % hw;
        const float* pr = pe + prow * C + sub * PER;
#pragma unroll
        for (int g = 0; g < PER / 4; ++g) {
            const f32x4 v = load4(pr + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) u[4 * g + j] += v[j];
        }
    }
}
LFT_DEV float sum16(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}
template <int C>
LFT_DEV void ln_stats(const float (&u)[C / 16], float& mean, float& rstd) {
    constexpr int PER = C / 16;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += u[i];
    mean = sum16(s) * (1.0f / C);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = u[i] - mean; q += d * d; }
    rstd = 1.0f / sqrtf(sum16(q) * (1.0f / C) + LFT_LN_EPS);
}
template <int C>
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ X, const float* __restrict__ pe, int pe_mode,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float* __restrict__ Y, long long N, int hw, int V) {
    constexpr int PER = C / 16;
    const int sub = threadIdx.x & 15;
    const long long t = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long tc = min(t, N - 1);
    float u[PER], mean, rstd;
    ln_load<C>(X, pe, pe_mode, tc, hw, V, sub, u);
    ln_stats<C>(u, mean, rstd);
    if (t >= N) return;
#pragma unroll
    for (int g = 0; g < PER / 4; ++g) {
        const f32x4 gm = load4(gamma + sub * PER + 4 * g), bt = load4(beta + sub * PER + 4 * g);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (u[4 * g + j] - mean) * rstd * gm[j] + bt[j];
        store4(Y + t * C + sub * PER + 4 * g, o);
    }
}
template <int C>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ X, const float* __restrict__ pe, int pe_mode,
                                                const float* __restrict__ gamma, const float* dY,
                                                const float* add, float* out,      // out may alias add or dY (same element, same thread)
                                                float* __restrict__ pgb, long long N, int hw, int V) {
    constexpr int PER = C / 16;
    __shared__ float red[16][2 * C];
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
    float gm[PER], dg[PER], db[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { gm[i] = gamma[sub * PER + i]; dg[i] = 0.0f; db[i] = 0.0f; }
    // the 16 lanes of a token share t, so the sum16 exchanges always see a full group
    for (long long t = (long long)blockIdx.x * 16 + slot; t < N; t += (long long)gridDim.x * 16) {
        float u[PER], mean, rstd, g[PER], xh[PER], dyv[PER];
        ln_load<C>(X, pe, pe_mode, t, hw, V, sub, u);
        ln_stats<C>(u, mean, rstd);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int gq = 0; gq < PER / 4; ++gq) {
            const f32x4 d = load4(dY + t * C + sub * PER + 4 * gq);
#pragma unroll
            for (int j = 0; j < 4; ++j) dyv[4 * gq + j] = d[j];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            xh[i] = (u[i] - mean) * rstd;
            g[i] = dyv[i] * gm[i];
            s1 += g[i]; s2 += g[i] * xh[i];
            dg[i] += dyv[i] * xh[i]; db[i] += dyv[i];
        }
        s1 = sum16(s1) * (1.0f / C); s2 = sum16(s2) * (1.0f / C);
#pragma unroll
        for (int gq = 0; gq < PER / 4; ++gq) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (g[4 * gq + j] - s1 - xh[4 * gq + j] * s2);
            if (add) {
                const f32x4 a = load4(add + t * C + sub * PER + 4 * gq);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += a[j];
            }
            store4(out + t * C + sub * PER + 4 * gq, o);
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) { red[slot][sub * PER + i] = dg[i]; red[slot][C + sub * PER + i] = db[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][i];
        pgb[(size_t)blockIdx.x * 2 * C + i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Elementwise helpers.
// ------------------------------------------------------------------------------------------
// out = g * act'(y): relu (mode 1): y > 0; leaky relu 0.2 (mode 2): y > 0 ? 1 : 0.2  (y = act(z) has the sign of z)
__global__ void k_act_bwd(const float* g, const float* __restrict__ y, float* out, long long n4, int mode) {   // out may alias g
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 gv = load4(g + 4 * i), yv = load4(y + 4 * i);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = yv[j] > 0.0f ? gv[j] : (mode == 1 ? 0.0f : 0.2f * gv[j]);
    store4(out + 4 * i, o);
}
__global__ void k_add3(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, long long n4) {   // out = a + b
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 av = load4(a + 4 * i), bv = load4(b + 4 * i);
    store4(out + 4 * i, f32x4{av[0] + bv[0], av[1] + bv[1], av[2] + bv[2], av[3] + bv[3]});
}
__global__ void k_add(float* __restrict__ a, const float* __restrict__ b, long long n4) {       // a += b
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 av = load4(a + 4 * i), bv = load4(b + 4 * i);
    store4(a + 4 * i, f32x4{av[0] + bv[0], av[1] + bv[1], av[2] + bv[2], av[3] + bv[3]});
}
// out[i] = sum_img src[img][i], i < n (n = hw * C): gradient of the broadcast position tokens
__global__ void k_sum_images(const float* __restrict__ src, int nimg, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    int im = 0;
    for (; im + 8 <= nimg; im += 8) {                                   // eight loads in flight, added in image order (the sum's order is unchanged)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(long long)(im + k) * n + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; im < nimg; ++im) s += src[(long long)im * n + i];
    out[i] = s;
}

// ------------------------------------------------------------------------------------------
// Angular attention (reference LFT.py:230-233): per pixel, sequence = the V views, 8 heads x 8 channels, no mask.
// QK [N][128] (Q | K), Vv / O [N][64].  One workgroup per pixel (b, p); thread = (head, view), VP = padded views.
// Backward is the usual two-pass form: per query (m, 1/l, D = sum_j P dP) and dQ; then per key dK, dV.
// ------------------------------------------------------------------------------------------
LFT_DEV void load8(const float* p, float (&o)[8]) {
    const f32x4 a = load4(p), b = load4(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
}
LFT_DEV void store8(float* p, const float (&o)[8]) {
    store4(p, f32x4{o[0], o[1], o[2], o[3]});
    store4(p + 4, f32x4{o[4], o[5], o[6], o[7]});
}
LFT_DEV float dot8(const float (&a)[8], const float* b) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += a[c] * b[c];
    return s;
}
template <int VP> constexpr int kAngHS = VP * 8 + 8;     // floats per head of k_ang_attn's LDS tiles
template <int VP, bool BWD>
__global__ __launch_bounds__(8 * VP) void k_ang_attn(const float* __restrict__ QK, const float* __restrict__ Vv,
                                                     float* __restrict__ O, const float* __restrict__ dO,
                                                     float* __restrict__ dQK, float* __restrict__ dV, int V, int hw) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // thread = (view i, head hd) with the HEAD fastest: the 8 lanes of a view read its row's 8 head pieces = 256 contiguous
    // bytes (view-fastest, every lane of a wave sat on a different row, 512 KB apart: 64 cache lines per load instruction).
    // LDS tiles [8 heads][HS]: HS = VP * 8 + 8 floats, so that the 8 heads a wave reads together fall on disjoint banks.
    constexpr int HS = kAngHS<VP>;
    float* Ks = sm;                         // [8][VP][8] (+ 8 floats per head)
    float* Vs = Ks + 8 * HS;
    float* Qs = Vs + 8 * HS;                // BWD only
    float* Ds = Qs + 8 * HS;                // BWD only: dO
    float* St = Ds + 8 * HS;                // BWD only: [8][SS] with rows of VP x 3 (m, 1/l, D) + 1 float: the 8 heads a wave reads together sit on 8 banks (at 3 VP floats all on one)
    constexpr int SS = VP * 3 + 1;
    const int hd = threadIdx.x & 7, i = threadIdx.x >> 3;
    const int b = blockIdx.x / hw, pix = blockIdx.x % hw;
    const bool act = i < V;
    const long long row = ((long long)b * V + min(i, V - 1)) * hw + pix;
    const float scale = 0.35355339059327373f;            // 1 / sqrt(8)
    float q[8], tmp[8], dov[8];
    load8(QK + row * 128 + hd * 8, q);
    load8(QK + row * 128 + 64 + hd * 8, tmp);
    store8(Ks + hd * HS + i * 8, tmp);
    load8(Vv + row * 64 + hd * 8, tmp);
    store8(Vs + hd * HS + i * 8, tmp);
    if (BWD) {
        store8(Qs + hd * HS + i * 8, q);
        load8(dO + row * 64 + hd * 8, dov);
        store8(Ds + hd * HS + i * 8, dov);
    }
    __syncthreads();
    // scores once, kept in registers (a second pass that recomputed the dot products was a quarter of the kernel's vector work);
    // the loop over the VP possible keys is unrolled so that the array is indexed by constants, keys >= V are skipped uniformly
    float sc[VP <= 32 ? VP : 1];
    float m = -INFINITY;
    if constexpr (VP <= 32) {
#pragma unroll
        for (int j = 0; j < VP; ++j)
            if (j < V) { sc[j] = dot8(q, Ks + hd * HS + j * 8); m = fmaxf(m, scale * sc[j]); }   // the raw dot product: the expressions below are the two-pass form's, bit for bit
    } else {
        for (int j = 0; j < V; ++j) m = fmaxf(m, scale * dot8(q, Ks + hd * HS + j * 8));
    }
    float l = 0.0f;
    if (!BWD) {
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (VP <= 32) {
#pragma unroll
            for (int j = 0; j < VP; ++j)
                if (j < V) {
                    const float pj = expf(scale * sc[j] - m);
                    l += pj;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] += pj * Vs[hd * HS + j * 8 + c];
                }
        } else {
            for (int j = 0; j < V; ++j) {
                const float pj = expf(scale * dot8(q, Ks + hd * HS + j * 8) - m);
                l += pj;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] += pj * Vs[hd * HS + j * 8 + c];
            }
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] *= inv;
        if (act) store8(O + row * 64 + hd * 8, o);
        return;
    }
    // ---- backward, pass A (per query) ----
    float D = 0.0f, av[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto passA = [&](int j, float pj) {
        const float* kj = Ks + hd * HS + j * 8;
        const float dp = dot8(dov, Vs + hd * HS + j * 8);
        l += pj; D += pj * dp;
#pragma unroll
        for (int c = 0; c < 8; ++c) { av[c] += pj * dp * kj[c]; bv[c] += pj * kj[c]; }
    };
    if constexpr (VP <= 32) {
#pragma unroll
        for (int j = 0; j < VP; ++j)
            if (j < V) passA(j, expf(scale * sc[j] - m));
    } else {
        for (int j = 0; j < V; ++j) passA(j, expf(scale * dot8(q, Ks + hd * HS + j * 8) - m));
    }
    const float inv = 1.0f / l;
    D *= inv;
    float dq[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dq[c] = scale * inv * (av[c] - D * bv[c]);
    if (act) store8(dQK + row * 128 + hd * 8, dq);
    St[hd * SS + i * 3 + 0] = m; St[hd * SS + i * 3 + 1] = inv; St[hd * SS + i * 3 + 2] = D;
    __syncthreads();
    // ---- pass B (per key j = this thread's view) ----
    float kj[8], vj[8], dk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 8; ++c) { kj[c] = Ks[hd * HS + i * 8 + c]; vj[c] = Vs[hd * HS + i * 8 + c]; }
    for (int qi = 0; qi < V; ++qi) {
        const float* qq = Qs + hd * HS + qi * 8;
        const float* dd = Ds + hd * HS + qi * 8;
        const float pij = expf(scale * dot8(kj, qq) - St[hd * SS + qi * 3]) * St[hd * SS + qi * 3 + 1];
        const float ds = pij * (dot8(vj, dd) - St[hd * SS + qi * 3 + 2]);
#pragma unroll
        for (int c = 0; c < 8; ++c) { dk[c] += ds * qq[c]; dv[c] += pij * dd[c]; }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) dk[c] *= scale;
    if (act) {
        store8(dQK + row * 128 + 64 + hd * 8, dk);
        store8(dV + row * 64 + hd * 8, dv);
    }
}

// ------------------------------------------------------------------------------------------
// Spatial windowed attention (reference LFT.py:147-162,183-187): 8 heads x 16, clamped 5x5 window with the
// reference's min(h, x+3) column bound; a query with an empty window (h < w) outputs 0 and passes no gradient.
// Q, K, Vv, O: [N][128].
//   MODE 0: forward.   MODE 1: backward pass A (dQ and per-(token, head) stats m, 1/l, D).
//   MODE 2: backward pass B, gather form: key j collects from the <= 25 queries i = j - (dy, dx) that see it.
// ------------------------------------------------------------------------------------------
LFT_DEV float dot16(const float (&a)[16], const float (&b)[16]) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += a[c] * b[c];
    return s;
}
LFT_DEV void ld16(const float* p, float (&o)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = load4(p + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * g + j] = v[j];
    }
}
LFT_DEV void st16(float* p, const float (&o)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) store4(p + 4 * g, f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]});
}
// ------------------------------------------------------------------------------------------
// The three window-attention passes, LDS-tiled.
// A workgroup owns an 8 x 16 tile of queries of one view image and one PAIR of heads (32 channels); the (8+4) x (16+4)
// halo tile of the two tensors a pass gathers from (K,V for MODE 0/1; Q,dO (+ per-row stats) for MODE 2) is staged in
// LDS once -- every row is used by up to 25 queries x 2 heads -- with rows padded to 144 B so that the 16-byte reads of
// 8 neighbouring lanes fall on distinct banks.  Thread = (query, head of the pair).  Tokens outside the image are
// zero-filled; taps outside the window get weight 0 (MODE 0/1: -inf score; MODE 2: 1/l = 0 in the zero-filled stats).
// ------------------------------------------------------------------------------------------
constexpr int kWaTY = 8, kWaTX = 16, kWaHR = kWaTY + 4, kWaHC = kWaTX + 4, kWaSlots = kWaHR * kWaHC, kWaRow = 36;   // floats per staged row
constexpr int kWaTile = kWaSlots * kWaRow;                     // floats per staged tensor
constexpr size_t kWaLds = (size_t)(2 * kWaTile + kWaSlots * 6) * sizeof(float);

// Both halo tiles of a pass: ALL of a thread's 16-byte pieces (8 per tensor) are requested before the first one is written to
// LDS -- as a load / wait / write loop the staging was 16 serialised memory round trips per workgroup, most of the kernel's time.
LFT_DEV void wa_stage2(const float* __restrict__ srcA, int ldA, float* ldsA, const float* __restrict__ srcB, int ldB, float* ldsB,
                       long long img0, int y0, int x0, int hp, int h, int w) {
    constexpr int NIT = (kWaSlots * 8 + 255) / 256;
    f32x4 va[NIT], vb[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = min((int)threadIdx.x + 256 * i, kWaSlots * 8 - 1);
        const int slot = idx >> 3, piece = idx & 7;
        const int gy = y0 - 2 + slot / kWaHC, gx = x0 - 2 + slot % kWaHC;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        const size_t t = (size_t)(in ? img0 + gy * w + gx : img0);
        va[i] = load4(srcA + t * ldA + hp * 32 + piece * 4);
        vb[i] = load4(srcB + t * ldB + hp * 32 + piece * 4);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = (int)threadIdx.x + 256 * i;
        const int slot = idx >> 3, piece = idx & 7;
        const int gy = y0 - 2 + slot / kWaHC, gx = x0 - 2 + slot % kWaHC;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        if (idx < kWaSlots * 8) {
            *reinterpret_cast<f32x4*>(ldsA + slot * kWaRow + piece * 4) = in ? va[i] : f32x4{0, 0, 0, 0};
            *reinterpret_cast<f32x4*>(ldsB + slot * kWaRow + piece * 4) = in ? vb[i] : f32x4{0, 0, 0, 0};
        }
    }
}
// position (0..31) of lane l (0..31) such that each hardware lane group of ds_read_b128 covers 16 consecutive positions
LFT_DEV int ldsb128_pos(int l) {
    return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
}
LFT_DEV void lds16(const float* p, float (&o)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * g + j] = v[j];
    }
}
template <int MODE, bool PRESCALED = false>   // PRESCALED: Q already carries 1/sqrt(16) * log2(e) (the inference path folds it into Wq)
__global__ __launch_bounds__(256, 2) void k_win_attn_lds(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ Vv, float* __restrict__ O,
                                                         const float* __restrict__ dO, float* __restrict__ dQ, float* __restrict__ dK,
                                                         float* __restrict__ dV, float* __restrict__ stats, int h, int w, int ldq) {
    // Q, K, dQ, dK rows are ldq floats apart (they are the two halves of one [N][256] tensor); V, O, dO, dV rows 128.
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* tA = wsm;                        // K (MODE 0/1) or Q (MODE 2)
    float* tB = wsm + kWaTile;              // V (MODE 0/1) or dO (MODE 2)
    float* tS = wsm + 2 * kWaTile;          // MODE 2: stats of the halo queries, [slot][2 heads][3]
    const int tiles_x = (w + kWaTX - 1) / kWaTX, tiles_y = (h + kWaTY - 1) / kWaTY;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, im = bid / (tiles_x * tiles_y);
    const int hp = blockIdx.y;
    const int y0 = ty * kWaTY, x0 = tx * kWaTX;
    const long long img0 = (long long)im * h * w;
    // Thread -> query.  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
    // same + 32, MI355X_MICROARCH.md, LDS): the 144-byte row stride makes 16 CONSECUTIVE queries of one image row conflict-free, so
    // each hardware group must be such a run -- lane l of a 32-lane half takes position ldsb128_pos(l) of the half's two rows.  (Lanes
    // in plain order put half of two different rows into one group: 2-way conflicts on every tap read, SQ_LDS_BANK_CONFLICT twice
    // the kernel's busy cycles in the round-3 counters.)
    const int hl = threadIdx.x >> 7, qi = (threadIdx.x & 96) + ldsb128_pos(threadIdx.x & 31), qy = qi >> 4, qx = qi & 15;
    const int y = y0 + qy, x = x0 + qx;
    const bool valid = y < h && x < w;
    const long long tok = img0 + min(y, h - 1) * w + min(x, w - 1);
    const size_t off = (size_t)tok * 128 + hp * 32 + hl * 16, offq = (size_t)tok * ldq + hp * 32 + hl * 16;
    const float scale = 0.25f, scale2 = PRESCALED ? 1.0f : 0.25f * LFT_LOG2E;
    // this thread's own rows: requested before the staging so that they arrive under it
    float own_a[16], own_b[16];
    if (MODE == 0 || MODE == 1) { ld16(Q + offq, own_a); if (MODE == 1) ld16(dO + off, own_b); }
    if (MODE == 2) {
        constexpr int NS = (kWaSlots * 6 + 255) / 256;
        float sv[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = min((int)threadIdx.x + 256 * i, kWaSlots * 6 - 1);
            const int slot = idx / 6, e = idx % 6;
            const int gy = y0 - 2 + slot / kWaHC, gx = x0 - 2 + slot % kWaHC;
            const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
            sv[i] = stats[((size_t)(in ? img0 + gy * w + gx : img0) * 8 + hp * 2) * 3 + e];
        }
        wa_stage2(Q, ldq, tA, dO, 128, tB, img0, y0, x0, hp, h, w);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = (int)threadIdx.x + 256 * i;
            const int slot = idx / 6;
            const int gy = y0 - 2 + slot / kWaHC, gx = x0 - 2 + slot % kWaHC;
            const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
            if (idx < kWaSlots * 6) tS[idx] = in ? sv[i] : 0.0f;
        }
    } else {
        wa_stage2(K, ldq, tA, Vv, 128, tB, img0, y0, x0, hp, h, w);
    }
    __syncthreads();
    const float* bA = tA + (qy * kWaHC + qx) * kWaRow + hl * 16;          // tap (ty, tx) adds (ty * kWaHC + tx) * kWaRow
    const float* bB = tB + (qy * kWaHC + qx) * kWaRow + hl * 16;
    if (MODE == 0 || MODE == 1) {
        const int wy0 = max(0, y - 2), wy1 = min(h, y + 3), wx0 = max(0, x - 2), wx1 = min(min(h, x + 3), w);   // LFT.py:150-160 (sic)
        float kv[16], vv[16], sc[25];
        const float (&q)[16] = own_a;
        const float (&dov)[16] = own_b;
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const int ky = y - 2 + t / 5, kx = x - 2 + t % 5;
            const bool ok = ky >= wy0 && ky < wy1 && kx >= wx0 && kx < wx1;
            lds16(bA + ((t / 5) * kWaHC + t % 5) * kWaRow, kv);
            sc[t] = ok ? scale2 * dot16(q, kv) : -INFINITY;
            m = fmaxf(m, sc[t]);
        }
        if (m == -INFINITY) m = 0.0f;                                      // empty window (h < w): all weights 0
        float l = 0.0f, D = 0.0f, o[16], a2[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { o[c] = 0.0f; a2[c] = 0.0f; }
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const float pj = fast_exp2(sc[t] - m);
            l += pj;
            lds16(bB + ((t / 5) * kWaHC + t % 5) * kWaRow, vv);
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 16; ++c) o[c] += pj * vv[c];
            } else {
                lds16(bA + ((t / 5) * kWaHC + t % 5) * kWaRow, kv);
                const float dp = dot16(dov, vv);
                D += pj * dp;
#pragma unroll
                for (int c = 0; c < 16; ++c) { o[c] += pj * dp * kv[c]; a2[c] += pj * kv[c]; }
            }
        }
        if (!valid) return;
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) o[c] *= inv;
            st16(O + off, o);
        } else {
            D *= inv;
#pragma unroll
            for (int c = 0; c < 16; ++c) o[c] = scale * inv * (o[c] - D * a2[c]);
            st16(dQ + offq, o);
            float* s3 = stats + ((size_t)tok * 8 + hp * 2 + hl) * 3;
            s3[0] = m; s3[1] = inv; s3[2] = D;                             // m in the log2 domain
        }
    } else {
        float kj[16], vj[16], qv[16], dv16[16], dk[16], dv[16];
        ld16(K + offq, kj);                                              // (pass B: the staging's 16 pieces + stats leave no registers to request these earlier)
        ld16(Vv + off, vj);
#pragma unroll
        for (int c = 0; c < 16; ++c) { dk[c] = 0.0f; dv[c] = 0.0f; }
        const float* bS = tS + (qy * kWaHC + qx) * 6 + hl * 3;
#pragma unroll 1
        for (int wy = 0; wy < 5; ++wy) {                                  // a rolled loop over the window's rows: fully unrolled, the scheduler hoists the LDS reads of all 25 taps and spills
#pragma unroll
            for (int wx = 0; wx < 5; ++wx) {
                const int so = wy * kWaHC + wx;
                lds16(bA + so * kWaRow, qv);
                lds16(bB + so * kWaRow, dv16);
                const float pij = fast_exp2(scale2 * dot16(qv, kj) - bS[so * 6]) * bS[so * 6 + 1];     // 1/l = 0 outside the image
                const float ds = pij * (dot16(dv16, vj) - bS[so * 6 + 2]);
#pragma unroll
                for (int c = 0; c < 16; ++c) { dk[c] += ds * qv[c]; dv[c] += pij * dv16[c]; }
            }
        }
        if (!valid) return;
        const float seen = x < h ? 1.0f : 0.0f;                            // keys with x >= h are in nobody's window (the column bound uses h)
#pragma unroll
        for (int c = 0; c < 16; ++c) { dk[c] *= scale * seen; dv[c] *= seen; }
        st16(dK + offq, dk);
        st16(dV + off, dv);
    }
}

// ------------------------------------------------------------------------------------------
// Up-sampler tail (reference LFT.py:41-43,80): F = PixelShuffle(lrelu(U)) is never built; Aact = lrelu(U) stays
// in token layout [N][64*s*s] (channel c*s*s + i*s + j <-> HR sub-pixel (i, j), LFT.py:41).  Forward: the mosaic-level
// 3x3 conv 64 -> 1 is the overlap-add GEMM of the inference path (G = M Aact through k_lin with the packed overlap-add
// matrix, then k_assemble_t); backward reads Aact through the index map.  Mosaic HR pixel (Ym, Xm) <-> LR mosaic pixel (Ym/s, Xm/s) =
// (a1*h + y, a2*w + x), sub-pixel (Ym % s, Xm % s).  Zero padding only at the mosaic border.
// ------------------------------------------------------------------------------------------
LFT_DEV long long up_token(int b, int Ym, int Xm, int A, int h, int w, int s, int& sub) {
    const int ly = Ym / s, lx = Xm / s;
    sub = (Ym - ly * s) * s + (Xm - lx * s);
    const int a1 = ly / h, a2 = lx / w;
    return (((long long)b * A * A + a1 * A + a2) * h + (ly - a1 * h)) * w + (lx - a2 * w);
}
// Backward of the tail, mirroring its forward (G = M Aact, out = gather of footprints):
//   dG[t][n]   = dout at the HR pixel that footprint entry n = (I+1)(S+2)+(J+1) of token t lands on (0 outside the mosaic,
//                0 in the padding columns n >= (S+2)^2)                                              -- k_up_gather_bwd
//   dAact      = M^T dG, times lrelu'  (k_lin with the transposed packed matrix and the activation epilogue)
//   dM         = dG^T Aact             (k_wgrad), folded back onto the 64 x 3 x 3 weight by k_upm_fold:
//                M[n(I,J)][c*ss + i*S + j] = w3[c][i-I+1][j-J+1]  =>  dw3[c][ty][tx] = sum_{i,j} dM[n(i-ty+1, j-tx+1)][c*ss + i*S + j]
__global__ __launch_bounds__(256) void k_up_gather_bwd(const float* __restrict__ dout, float* __restrict__ dG, int B, int A, int h, int w,
                                                       int s, int gld) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int V = A * A;
    const long long N = (long long)B * V * h * w;
    if (idx >= N * gld) return;
    const long long t = idx / gld;
    const int n = (int)(idx - t * gld), gp = (s + 2) * (s + 2);
    float v = 0.0f;
    if (n < gp) {
        const int I = n / (s + 2) - 1, J = n % (s + 2) - 1;
        const int x = (int)(t % w), y = (int)((t / w) % h), vv = (int)((t / ((long long)w * h)) % V), b = (int)(t / ((long long)w * h * V));
        const int Y = ((vv / A) * h + y) * s + I, X = ((vv % A) * w + x) * s + J;
        const int HH = A * h * s, WW = A * w * s;
        if (Y >= 0 && Y < HH && X >= 0 && X < WW) v = dout[((long long)b * HH + Y) * WW + X];
    }
    dG[idx] = v;
}
__global__ void k_upm_fold(const float* __restrict__ dM, float* __restrict__ dw3, int s) {       // dM: [32 gt][64 s^2]
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 576) return;
    const int c = idx / 9, ty = (idx % 9) / 3, tx = idx % 3, ss = s * s;
    float acc = 0.0f;
    for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) {
            const int I = i - (ty - 1), J = j - (tx - 1);                        // always inside [-1, s]
            acc += dM[(size_t)((I + 1) * (s + 2) + (J + 1)) * 64 * ss + c * ss + i * s + j];
        }
    dw3[idx] = acc;
}

// conv_init0 weight gradient (reference LFT.py:24): dW0[c][tap] = sum_t dX0[t][c] * lr[view pixel (y+dy, x+dx)].
// One wave per token range, lane = channel.
__global__ __launch_bounds__(256) void k_conv0_wgrad(const float* __restrict__ dX0, const float* __restrict__ lr,
                                                     float* __restrict__ part, int B, int A, int h, int w, long long toks_per_wave) {
    const int c = threadIdx.x & 63, V = A * A;
    const long long wv = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long N = (long long)B * V * h * w;
    float dw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[t] = 0.0f;
    const long long t0 = wv * toks_per_wave, t1 = min(t0 + toks_per_wave, N);
    // the token's position (x, y, view, batch) once per wave, then by increments (four 64-bit divisions per token before)
    int x = 0, y = 0, v = 0, b = 0;
    if (t0 < t1) { x = (int)(t0 % w); y = (int)((t0 / w) % h); v = (int)((t0 / ((long long)w * h)) % V); b = (int)(t0 / ((long long)w * h * V)); }
    for (long long t = t0; t < t1; ++t) {
        const float* img = lr + (size_t)b * (A * h) * (A * w) + (size_t)((v / A) * h) * (A * w) + (v % A) * w;
        const float g = dX0[t * 64 + c];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const float pv = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[yy * (A * w) + xx] : 0.0f;
            dw[tap] += g * pv;
        }
        if (++x == w) { x = 0; if (++y == h) { y = 0; if (++v == V) { v = 0; ++b; } } }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) part[wv * 576 + c * 9 + t] = dw[t];
}

// L1 loss (reference LFT.py:269-277) and its gradient: loss = mean |sr - hr|; dsr = sign(sr - hr) * gscale.
__global__ __launch_bounds__(256) void k_l1_partial(const float* __restrict__ sr, const float* __restrict__ hr, float* __restrict__ dsr,
                                                    float gscale, long long n, float* __restrict__ part) {
    __shared__ float red[256];
    float s = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = sr[i] - hr[i];
        s += fabsf(d);
        if (dsr) dsr[i] = d > 0.0f ? gscale : (d < 0.0f ? -gscale : 0.0f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_l1_final(const float* __restrict__ part, int nb, float inv_n, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.0f;
        for (int i = 0; i < nb; ++i) s += part[i];
        *loss = s * inv_n;
    }
}

// Adam (torch.optim.Adam defaults used by the reference, train.py:77-83: betas, eps, weight_decay 0), one flat
// fp32 buffer of all parameters: p -= lr_t * m_hat / (sqrt(v_hat) + eps).  bc1 = 1 - b1^t, bc2 = 1 - b2^t.
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       long long n, float lr, float b1, float b2, float eps, float bc1, float bc2, float gscale, float wd) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale + wd * p[i];                        // torch.optim.Adam weight_decay: L2 term added to the gradient
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
