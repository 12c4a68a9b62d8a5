// lft_kernels_a.cuh -- weight packing, position tables, initial convolutions, angular Transformer.
#pragma once
#include "lft_common.cuh"

// ------------------------------------------------------------------------------------------
// Weight packing.  A stream is a sequence of fragments in exactly the order a kernel consumes
// them.  One PackOp describes ntiles x ksteps fragments (nt-major) cut from a row-major fp32 matrix:
//   frag(nt, ks)[lane = 32h + r][j] = scale * src[(row0 + 32 nt + r) * ld + (k0 + 16 ks + kk(h, j)) * kmul + kadd]
// kmap 0: natural kk = 8h + j (operand loaded from memory); kmap 1: acc order (see lft_common.cuh).
// kind 1 (UPM) builds the "overlap-add" matrix of the final 3x3 convolution instead (see k_up); kind 2 its transpose.
// ------------------------------------------------------------------------------------------
struct PackOp {
    const float* src;
    int kind, row0, nrows, ntiles, ld, kmul, kadd, k0, ksteps, kmap, frag0, s;
    float scale;
    int order;     // 0: nt-major (frag = nt * ksteps + ks); 1: groups of 4 output tiles, inside a group ks-major (wfrag_index): the
                   // fragments one workgroup of the training GEMM consumes are then ONE contiguous stream in consumption order
};
// Fragment (ot, ks) of a matrix of OT output tiles packed with order 1: groups of four tiles (the last one may hold fewer),
// group g at 4 g KS, inside it k-step-major.
__host__ __device__ __forceinline__ int wfrag_index(int OT, int KS, int ot, int ks) {
    const int g = ot >> 2, G = OT - 4 * g < 4 ? OT - 4 * g : 4;
    return 4 * g * KS + ks * G + (ot & 3);
}
__host__ __device__ __forceinline__ void wfrag_coords(int OT, int KS, int lf, int order, int& nt, int& ks) {
    if (!order) { nt = lf / KS; ks = lf % KS; return; }
    const int gl = (OT - 1) >> 2, g = lf / (4 * KS) < gl ? lf / (4 * KS) : gl, G = OT - 4 * g < 4 ? OT - 4 * g : 4, rem = lf - 4 * g * KS;
    ks = rem / G; nt = 4 * g + rem % G;
}
constexpr int LFT_PACK_MAXOPS = 40;
struct PackArgs {
    int nops;
    PackOp op[LFT_PACK_MAXOPS];
};

// Row n = (I+1)*(s+2) + (J+1), I,J in [-1, s]: HR position relative to the s x s block of one LR pixel.
// Column kk = c*s*s + i*s + j: PixelShuffle source channel (reference LFT.py:41).  Entry = W3[c][i-I+1][j-J+1]
// when that tap exists: out[Y] = sum_dy W3[dy+1] F[Y+dy]  (reference LFT.py:43, cross-correlation, pad 1).
LFT_DEV float upm_entry(const float* __restrict__ w3, int n, int kk, int s) {
    const int I = n / (s + 2) - 1, J = n % (s + 2) - 1;
    const int c = kk / (s * s), i = (kk / s) % s, j = kk % s;
    const int dy = i - I, dx = j - J;
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return 0.0f;
    return w3[c * 9 + (dy + 1) * 3 + (dx + 1)];
}

template <typename T>
__global__ __launch_bounds__(64) void k_pack(PackArgs args, T* __restrict__ dst) {
    const int f = blockIdx.x, lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    int oi = 0;
    for (int i = 0; i < args.nops; ++i)
        if (f >= args.op[i].frag0) oi = i;
    const PackOp& op = args.op[oi];
    int nt, ks;
    wfrag_coords(op.ntiles, op.ksteps, f - op.frag0, op.order, nt, ks);
    const int n = 32 * nt + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = op.k0 + 16 * ks + (op.kmap ? 8 * (j >> 2) + 4 * h + (j & 3) : 8 * h + j);
        float v = 0.0f;
        if (n < op.nrows) {
            if (op.kind == 0) v = op.scale * op.src[(size_t)(op.row0 + n) * op.ld + kk * op.kmul + op.kadd];
            else if (op.kind == 1) v = upm_entry(op.src, n, kk, op.s);
            else v = kk < (op.s + 2) * (op.s + 2) ? upm_entry(op.src, kk, n, op.s) : 0.0f;   // kind 2: the transposed matrix (rows = PixelShuffle channels)
        }
        if constexpr (sizeof(T) == 4) dst[(((size_t)f * 2 + (j >> 2)) * 64 + lane) * 4 + (j & 3)] = (T)v;   // two 1 KiB pieces
        else dst[((size_t)f * 64 + lane) * 8 + j] = (T)v;
    }
}

// Sinusoid position tables (reference LFT.py:91-115): channel c < 32 -> sin(l / T^(2c/64)),
// c >= 32 -> cos(l / T^(2(c-32)/64)).  Angular table [V][64] fp32; spatial image [h*w][64] = (PE_h[y]+PE_w[x])/2.
LFT_DEV float pe_value(int l, int c) {
    const int m = c & 31;
    const float g = (float)pow(10000.0, (double)(2 * m) / 64.0);
    const float p = (float)l / g;
    return (c < 32) ? (float)sin((double)p) : (float)cos((double)p);
}
// Angular table: lane-major fp32 (see load_lane_major_raw): [view tile][k = 0..3][lane][8].
template <typename T>
__global__ void k_pe_tables(float* __restrict__ ang_pe, T* __restrict__ spa_img, int V, int h, int w) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ntile = (V + 31) / 32;
    if (idx < ntile * 2048) {
        const int e = idx & 7, lane = (idx >> 3) & 63, k = (idx >> 9) & 3, ct = idx >> 11;
        const int view = ct * 32 + (lane & 31), nt = k >> 1, g = 2 * (k & 1) + (e >> 2);
        const int ch = 32 * nt + 8 * g + 4 * (lane >> 5) + (e & 3);
        ang_pe[idx] = view < V ? pe_value(view, ch) : 0.0f;
    }
    if (idx < h * w * 64) {
        const int p = idx >> 6, c = idx & 63;
        spa_img[idx] = (T)((pe_value(p / w, c) + pe_value(p % w, c)) / 2.0f);
    }
}
__global__ void k_copy_f32(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// conv_init0: 1 -> 64 channels, per-view 3x3, zero pad per view (reference LFT.py:23-25,65).
// Reads the LR mosaic [B,1,A*h,A*w] (view (a1,a2) is the block at rows a1*h.., cols a2*w.. -- LFT.py:58),
// writes channels-last tokens [B,V,h,w,64].  One thread = 8 channels of one token.
// ------------------------------------------------------------------------------------------
constexpr int kConv0Tok = 128;          // tokens per workgroup
template <typename T>
__global__ __launch_bounds__(256) void k_conv0(const float* __restrict__ lr, const float* __restrict__ w0,
                                               T* __restrict__ out, int B, int A, int h, int w) {
    // grid: x = kConv0Tok-token groups of one view image, y = image (b, v).  32-bit index math only.
    // The 64 x 9 weights, 72 of them per thread: the kernel was bound by these LDS reads (72 ds_read_b32 per thread, 2-way
    // bank conflicts).  9 taps padded to 12 floats per channel = three 16-byte reads; channel groups 100 floats apart put the
    // eight distinct addresses of a 16-lane group on disjoint banks.
    __shared__ __attribute__((aligned(16))) float wl[8 * 100];
    const int hw = h * w, im = blockIdx.y, V = A * A;
    const int cg = threadIdx.x & 7;
    const int b = im / V, v = im - b * V, a1 = v / A, a2 = v - a1 * A;
    const float* img = lr + (size_t)b * (A * h) * (A * w) + (size_t)(a1 * h) * (A * w) + a2 * w;
    // kConv0Tok tokens per workgroup, 32 per pass: the weight staging and the barrier above are paid once per 128 tokens
    // (3 200 workgroups of 32 tokens spent most of their time in that prologue).
    // The kernel is a latency chain, not a throughput problem (13 MB out, 60 MFLOP): round 4 issues the taps of ALL passes before
    // the first use -- one memory round trip per workgroup instead of one per pass (four, each behind the previous pass's compute).
    constexpr int NP = kConv0Tok / 32;
    float val[NP][9];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int p = min(blockIdx.x * kConv0Tok + pass * 32 + (int)(threadIdx.x >> 3), hw - 1);
        const int y = p / w, x = p - y * w;
        // branch-free: all nine taps are loaded from clamped (readable) positions, then selected -- a conditional load is a
        // masked branch per tap, nine dependent round trips instead of nine loads in flight.
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            val[pass][t] = img[min(max(yy, 0), h - 1) * (A * w) + min(max(xx, 0), w - 1)];
        }
    }
    {   // all three loads of a thread in flight before the first LDS write (as a load / wait / write loop: three serialised round trips)
        float wv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) wv[k] = w0[min((int)threadIdx.x + 256 * k, 575)];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = (int)threadIdx.x + 256 * k, ch = i / 9, t = i - ch * 9;
            if (i < 576) wl[(ch >> 3) * 100 + (ch & 7) * 12 + t] = wv[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        // The empty asm names a pass's nine values at once: it keeps hipcc from sinking the loads back under their conditions.
        asm volatile("" : "+v"(val[pass][0]), "+v"(val[pass][1]), "+v"(val[pass][2]), "+v"(val[pass][3]), "+v"(val[pass][4]), "+v"(val[pass][5]),
                          "+v"(val[pass][6]), "+v"(val[pass][7]), "+v"(val[pass][8]));
        const int p = blockIdx.x * kConv0Tok + pass * 32 + (threadIdx.x >> 3);
        if (p >= hw) return;
        const int y = p / w, x = p - y * w;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            val[pass][t] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? val[pass][t] : 0.0f;
        }
        f32x4 o[2];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4* wq = reinterpret_cast<const f32x4*>(wl + cg * 100 + c * 12);
            const f32x4 w03 = wq[0], w47 = wq[1], w8 = wq[2];
            float a = 0.0f;                                              // same order of additions as before: bit-identical results
#pragma unroll
            for (int t = 0; t < 4; ++t) a += w03[t] * val[pass][t];
#pragma unroll
            for (int t = 0; t < 4; ++t) a += w47[t] * val[pass][4 + t];
            a += w8[0] * val[pass][8];
            o[c >> 2][c & 3] = a;
        }
        T* row = out + ((size_t)im * hw + p) * 64 + cg * 8;
        if constexpr (sizeof(T) == 2) {                              // one 16-byte store per thread: a wave writes 1 KiB of whole lines
            typename H16<T>::v8 v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (T)o[c >> 2][c & 3];
            store_raw16(reinterpret_cast<char*>(row), __builtin_bit_cast(raw16, v));
        } else {
            store4(row, o[0]);
            store4(row + 4, o[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Per-view 3x3 convolution, 64 -> 32*NT channels, as an implicit GEMM.  A workgroup owns 128
// consecutive tokens of one view image (4 waves x 32 tokens); K = 9 taps x 64 channels = 36 k-steps.
//  * input: the 128 tokens plus a halo of w+1 tokens on either side are brought into LDS ONCE (each token row is
//    re-read by up to 9 taps) by LDS-DMA: no registers, no round trip before the data is needed, and it
//    shares one wait with the first weight chunks.  LDS-DMA writes lane-linear 1 KiB pieces, so rows are
//    unpadded (128 B bf16 / 256 B fp32) and bank conflicts are removed by an XOR swizzle applied on the
//    per-lane SOURCE address and again on the read (16-byte piece c of slot s lives at piece c ^ f(s)):
//    conflict-free for every tile alignment (checked exhaustively offline; unswizzled it is 8- / 16-way).
//    Tokens outside the image are fetched from a clamped address and never used: the lane predicate zeroes
//    them, as it does at the left/right image border (per-view zero padding, reference LFT.py:24,27,167).
//  * weights: the fragment stream ((tap*4 + ks) * NT + nt) arrives through the workgroup's LDS ring.
// ------------------------------------------------------------------------------------------
template <typename T, int NW = 4> struct ConvIn {     // NW waves x 32 tokens per workgroup tile
    static constexpr int ROW_BYTES = 64 * (int)sizeof(T);
    static constexpr int PPR = ROW_BYTES / 16;                       // 16-byte pieces per token row (8 / 16)
    static constexpr int SLOTS_PER_DMA = 1024 / ROW_BYTES;           // token rows per 1 KiB LDS-DMA piece (8 / 4)
    static __host__ __device__ int slots(int w) { return 32 * NW + 2 + 2 * w; }
    static __host__ __device__ int dma_pieces(int w) { return (slots(w) + SLOTS_PER_DMA - 1) / SLOTS_PER_DMA; }
    static __host__ __device__ int bytes(int w) { return dma_pieces(w) * 1024; }
    static __device__ __forceinline__ int swz(int slot) { return sizeof(T) == 2 ? ((slot >> 1) & 7) : (slot & 15); }
};

// Issues the DMA only; the caller waits (vmcnt) and publishes (barrier) before the first read.
template <typename T, int NW = 4>
LFT_DEV void stage_conv_input(const T* __restrict__ img, int p0, int hw, int w, char* lds_in) {
    using CI = ConvIn<T, NW>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npieces = CI::dma_pieces(w);
    LFT_NOTE_ASM_("DMA", kNoteConvIn, 0);
    for (int piece = wave; piece < npieces; piece += NW) {
        const int slot = piece * CI::SLOTS_PER_DMA + lane / CI::PPR, cpos = lane % CI::PPR;
        const int q = min(max(p0 - w - 1 + slot, 0), hw - 1);                       // clamped: out-of-image rows are masked at use
        const char* src = reinterpret_cast<const char*>(img) + ((size_t)q * CI::ROW_BYTES + ((cpos ^ CI::swz(slot)) * 16));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_in + piece * 1024), 16, 0, 0);
    }
}
LFT_DEV void wait_staged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // own DMA pieces and early loads landed

// tl = token index inside the workgroup tile (0..127); (y, x) its image coordinates.
// Taps that fall outside the image (per-view zero padding) read a row of zeros (`zero_row`, kConvZeroRow bytes the caller
// cleared before publishing the tile) instead of being masked in registers: one select on the row address per tap
// instead of one per fragment register (16 v_cndmask per tap).
constexpr int kConvZeroRow = 256;
LFT_DEV void clear_zero_row(char* zero_row) {                          // call before the barrier that publishes the input tile
    if (threadIdx.x < kConvZeroRow / 16) store_raw16(zero_row + threadIdx.x * 16, raw16{0u, 0u, 0u, 0u});
}
template <int NT, typename T, int NW, typename Ring>
LFT_DEV void conv3x3_tile(const char* lds_in, const char* zero_row, int tl, int y, int x, bool ok, int h, int w, int hh,
                          Ring& ring, f32x16 (&acc)[NT]) {
    using CI = ConvIn<T, NW>;
    static_assert(CI::ROW_BYTES <= kConvZeroRow, "zero row too short");
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int yy = y + dy, xx = x + dx;
        const bool inb = ok && yy >= 0 && yy < h && xx >= 0 && xx < w;
        const int slot = tl + w + 1 + dy * w + dx;
        const char* row = inb ? lds_in + slot * CI::ROW_BYTES : zero_row;
        const int sw = inb ? CI::swz(slot) : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Frag<T> b;
            if constexpr (sizeof(T) == 2) {
                b.v = __builtin_bit_cast(typename H16<T>::v8, load_raw16(row + (((2 * ks + hh) ^ sw) * 16)));
            } else {
                b.lo = __builtin_bit_cast(f32x4, load_raw16(row + (((4 * ks + 2 * hh) ^ sw) * 16)));
                b.hi = __builtin_bit_cast(f32x4, load_raw16(row + (((4 * ks + 2 * hh + 1) ^ sw) * 16)));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mma(ring.next(), b, acc[nt]);
        }
    }
}

// conv_init[i]: 64 -> 64 + LeakyReLU(0.2); the last one adds conv_init0's output (reference LFT.py:26-33,66).
// A workgroup = NW waves x 32 consecutive tokens of one view image; all its waves share one weight ring, so the
// packed weights are streamed into the CU once per 32*NW tokens.
#ifndef LFT_NW_CONV
#define LFT_NW_CONV 4
#endif
constexpr int kNwConv = LFT_NW_CONV;
#ifndef LFT_CONV64_CHUNK
#define LFT_CONV64_CHUNK (LFT_NW_CONV == 4 ? 12 : LFT_NW_CONV)
#endif
constexpr int kConv64Chunk = LFT_CONV64_CHUNK;   // 72 fragments; NW = 4: 6 chunks of 12, 3-slot ring = 36 KiB (bf16) so two workgroups share a CU
template <typename T, bool RES, int NW = kNwConv>
__global__ __launch_bounds__(64 * NW) void k_conv64(const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ res,
                                                   const T* __restrict__ wstream, int nimg, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TT = 32 * NW;                                       // tokens per workgroup tile
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5, wave = threadIdx.x >> 6;
    const int hw = h * w, tpi = (hw + TT - 1) / TT;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);                  // neighbouring tiles (shared halo rows) on one XCD
    const int im = bid / tpi, p0 = (bid % tpi) * TT;
    const int tl = wave * 32 + r, p = p0 + tl;
    const bool ok = p < hw;
    const int t0 = p0 + wave * 32, nvalid = max(0, min(32, hw - t0));                 // this wave's 32 consecutive tokens
    const size_t tile_off = ((size_t)im * hw + min(t0, hw - 1)) * 64;
    char* lds_in = smem + WRing<T, kConv64Chunk, NW>::LDS_BYTES;
    char* scr = lds_in + ConvIn<T, NW>::bytes(w) + wave * TileIO<2, T>::BYTES;        // wave-private tile I/O scratch
    char* zero_row = lds_in + ConvIn<T, NW>::bytes(w) + NW * TileIO<2, T>::BYTES;
    f32x16 rr[2];
    if (RES) load_tile<2, T>(res + tile_off, nvalid, lane, rr, scr);                  // first: its latency hides under the tile
    WRing<T, kConv64Chunk, NW> ring;
    ring.init(wstream, smem, 72);
    stage_conv_input<T, NW>(in + (size_t)im * hw * 64, p0, hw, w, lds_in);
    clear_zero_row(zero_row);
    wait_staged();
    __syncthreads();                                                                 // ... and everybody else's
    LFT_NOTE_ASM_("USE", kNoteConvIn, 0);
    f32x16 acc[2];
    zero_acc<2>(acc);
    conv3x3_tile<2, T, NW>(lds_in, zero_row, tl, p / w, p % w, ok, h, w, hh, ring, acc);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = lrelu02_fast(acc[nt][i]);
    if (RES) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] += rr[nt];
    }
    store_tile<2, T>(out + tile_off, nvalid, lane, acc, scr);
}

// ------------------------------------------------------------------------------------------
// Angular Transformer block, fully fused (reference AngTrans.forward, LFT.py:225-238).
// One wave = one spatial position (b, y, x); its V <= 32 views sit on the 32 MFMA columns.
//   n = LN(x + PE_v); Q = n Wq^T, K = n Wk^T (PE'd, normalised) ; V = x Wv^T (raw tokens, LFT.py:230-232)
//   per head (8 x 8 channels): S^T[kv, q] = K Q^T  -> softmax over kv -> O^T += V^T P^T
//   t = x + O Wo^T ;  t += W2 relu(W1 LN'(t))
// Everything stays in registers: Q^T/K^T accumulators are re-used as MFMA operands; V is produced
// with the operand roles swapped (tokens as A, weight as B) so that it lands view-on-row, which is
// what the P.V product needs as its A operand.  Heads are separated by zeroing half of a k-step
// (QK^T: 16 k values = 2 heads) or whole fragments by lane (P.V: 32 output rows = 4 heads).
// The softmax scale 1/sqrt(8) and log2(e) are folded into the packed Wq; exp2 is used.
// Stream: Wq[2x4] Wk[2x4] Wv[2x4] Wo[2x4] W1[4x4] W2[2x8]  (64 fragments) -- small enough to live
// in LDS for the whole kernel: a persistent workgroup DMAs it once and then walks over positions.
// ------------------------------------------------------------------------------------------
template <int NT_OUT, int KS, typename T>
LFT_DEV void linear_lds(const char* wl, int f0, int lane, const Frag<T> (&x)[KS], f32x16 (&y)[NT_OUT]) {
#pragma unroll
    for (int nt = 0; nt < NT_OUT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            mma(frag_from_pieces(wl + (f0 + nt * KS + ks) * 1024 * FragInfo<T>::PIECES, lane, T()), x[ks], y[nt]);
}

// NLIVE: accumulator registers 0 .. NLIVE-1 of a score tile can hold an existing view (register i <-> view rows
// acc_row(i, 0) and acc_row(i, 1)); for V <= 25 (5 x 5) registers 13..15 are rows 25-27 / 29-31, never a view: the
// softmax skips them at compile time (the kernel is bound by vector-instruction issue, not by the matrix pipe).
template <typename T, int NLIVE = 16>
__global__ __launch_bounds__(256, 2) void k_ang(const T* __restrict__ X, T* __restrict__ Y, const T* __restrict__ ws,
                                             const float* __restrict__ ln, const float* __restrict__ pe,
                                             int V, int hw, int npix, unsigned* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FB = 1024 * FragInfo<T>::PIECES;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned bad = 0;                                                  // non-finite activation seen (layernorm_acc)
    {
        const char* g = reinterpret_cast<const char*>(ws);
        LFT_NOTE_ASM_("DMA", kNoteAngW, 0);
#pragma unroll
        for (int i = 0; i < 16 * FragInfo<T>::PIECES; ++i) {
            const int piece = wave * 16 * FragInfo<T>::PIECES + i;
            glds_piece(g + piece * 1024, smem + piece * 1024, lane);
        }
    }
    float* lds_ln = reinterpret_cast<float*>(smem + 64 * FB);
    LFT_STAMP(0);
    const raw16 lnv = params_load(ln, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own LDS-DMA pieces landed, then publish (see WRing::next)
    params_store(lds_ln, 256, lnv);
    __syncthreads();
    LFT_NOTE_ASM_("USE", kNoteAngW, 0);
    LFT_STAMP(1);
    [[maybe_unused]] int stamp_it = 0;                                     // diagnostic build: second tile's stamps go to slots 7..11
    const bool ok = r < V;
    f32x16 negmask;                                                                       // 0 for key rows < V, -inf beyond
#pragma unroll
    for (int i = 0; i < 16; ++i) negmask[i] = acc_row(i, hh) >= V ? -INFINITY : 0.0f;
    char* scr = reinterpret_cast<char*>(lds_ln) + 1024 + wave * TileIO<2, T>::BYTES;        // wave-private tile I/O scratch
    const size_t vstride = (size_t)hw * 64 * sizeof(T);                                   // one view to the next, same position
    for (int pix = blockIdx.x * 4 + wave; pix < npix; pix += gridDim.x * 4) {
        asm volatile("" ::: "memory");    // keep the (loop-invariant) LDS weight reads inside the loop: hoisted, they cost 256+ VGPRs
        const int b = pix / hw, p = pix % hw;
        const size_t off0 = (((size_t)b * V) * hw + p) * 64;                              // view 0 of this position

        f32x16 x[2], n[2];
        load_tile<2, T>(X + off0, V, lane, x, scr, vstride);                              // rows = views: 8 lanes x 16 B per row
        LFT_STAMP_IT(2, 5 * stamp_it);
        {
            typename RawPiece<float>::type pr[8];
            load_lane_major_raw<2, float>(pe, lane, pr);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) n[nt] = x[nt];
            add_acc_raw<2, float>(n, pr, ok);
        }
        layernorm_acc<2, sizeof(T) == 2>(n, lds_ln, lds_ln + 64, hh, bad);
        Frag<T> nf[4], xf[4];
        acc_frags<2, T>(n, nf);
        acc_frags<2, T>(x, xf);

        f32x16 q[2], k[2], v[2], o[2];
        zero_acc<2>(q); zero_acc<2>(k); zero_acc<2>(v);
        linear_lds<2, 4, T>(smem, 0, lane, nf, q);
        linear_lds<2, 4, T>(smem, 8, lane, nf, k);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)           // V[view, ch] = sum_k x[view, k] Wv[ch, k]: tokens are the A operand
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mma(xf[ks], frag_from_pieces(smem + (16 + nt * 4 + ks) * FB, lane, T()), v[nt]);

        LFT_STAMP_IT(3, 5 * stamp_it);
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) {
            const int nt = hd >> 2, s = (hd >> 1) & 1, half = hd & 1;
            f32x16 S = negmask;                                    // rows of non-existent views start at -inf: no per-head masking
            mma(frag_half(acc_to_frag(k[nt], s, T()), half), acc_to_frag(q[nt], s, T()), S);   // S^T[kv, q]
            float m = S[0];
#pragma unroll
            for (int i = 1; i < NLIVE; ++i) m = fmaxf(m, S[i]);
            m = xhalf_max(m);
            // the kernel is bound by vector-instruction issue: subtract and sum as register pairs (v_pk_add_f32)
            const f32x2 mm = {m, m};
            f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i + 1 < NLIVE; i += 2) {
                f32x2 d = f32x2{S[i], S[i + 1]} - mm;
                d[0] = fast_exp2(d[0]); d[1] = fast_exp2(d[1]);
                S[i] = d[0]; S[i + 1] = d[1];
                sum2 += d;
            }
            float sum = sum2[0] + sum2[1];
            if constexpr (NLIVE & 1) { S[NLIVE - 1] = fast_exp2(S[NLIVE - 1] - m); sum += S[NLIVE - 1]; }
#pragma unroll
            for (int i = NLIVE; i < 16; ++i) S[i] = 0.0f;    // rows that are no view for any lane: probability 0 without computing it
            const float inv = sizeof(T) == 2 ? fast_rcp(xhalf_sum(sum)) : 1.0f / xhalf_sum(sum);
            // O^T of this head in a fresh accumulator: with V's 32 channels (4 heads) as the A operand every row of the product
            // is computed, the head's own 8 rows = registers 4*(hd&3)..+3 are kept (normalised) -- cheaper than zeroing the other
            // heads' channels of V per head (8 v_cndmask per head in a kernel bound by vector-instruction issue)
            f32x16 oh;
#pragma unroll
            for (int i = 0; i < 16; ++i) oh[i] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) mma(acc_to_frag(v[nt], s2, T()), acc_to_frag(S, s2, T()), oh);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[nt][4 * (hd & 3) + i] = oh[4 * (hd & 3) + i] * inv;
        }

        LFT_STAMP_IT(4, 5 * stamp_it);
        Frag<T> of[4];
        acc_frags<2, T>(o, of);
        linear_lds<2, 4, T>(smem, 24, lane, of, x);              // t = x + O Wo^T
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) n[nt] = x[nt];
        layernorm_acc<2, sizeof(T) == 2>(n, lds_ln + 128, lds_ln + 192, hh, bad);
        acc_frags<2, T>(n, nf);
        f32x16 hid[4];
        zero_acc<4>(hid);
        linear_lds<4, 4, T>(smem, 32, lane, nf, hid);
        Frag<T> hf[8];
        acc_frags_relu<4>(hid, hf);
        linear_lds<2, 8, T>(smem, 48, lane, hf, x);
        LFT_STAMP_IT(5, 5 * stamp_it);
        store_tile<2, T>(Y + off0, V, lane, x, scr, vstride);
        LFT_STAMP_IT(6, 5 * stamp_it);
        stamp_it = 1;
    }
    publish_status(status, bad);
}

// ------------------------------------------------------------------------------------------
// Angular block for more than 32 views (e.g. 9x9 = 81 views, BASELINE configs[4]).  A workgroup of CT waves
// owns one spatial position; wave c holds views 32c .. 32c+31 on its MFMA columns and does everything that is
// per-token (LN, projections, out_proj, FFN) exactly as k_ang.  For the attention each wave publishes its K and V
// operand fragments (8 per wave) in LDS; after a barrier every wave runs its 32 queries against all CT key tiles:
// per head CT score tiles (softmax over up to 32*CT keys: registers + one lane^32 exchange) and 2*CT P.V MFMAs.
// WLDS: weights LDS-resident (bf16); the fp32 parity build reads them from L2 (128 KiB would not leave room).
// ------------------------------------------------------------------------------------------
// NG positions share one workgroup (NG * CT waves) and therefore one LDS copy of the weights: more waves per CU for the
// same LDS (9x9, bf16: 2 x 3 waves and 113 KiB instead of 3 waves and 89 KiB).  All groups run the loop in lock-step.
// LASTLIVE: score registers of the LAST key tile that can hold an existing view (as NLIVE in k_ang): 9 x 9 views on three
// tiles leave 17 rows in the last one -- registers 9 .. 15 are never a view and are skipped at compile time.
template <typename T, int CT, bool WLDS, int NG, int LASTLIVE = 16>
__global__ __launch_bounds__(64 * CT * NG) void k_ang_multi(const T* __restrict__ X, T* __restrict__ Y, const T* __restrict__ ws,
                                                            const float* __restrict__ ln, const float* __restrict__ pe,
                                                            int V, int hw, int npix, unsigned* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FB = 1024 * FragInfo<T>::PIECES;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    unsigned bad = 0;                                                  // non-finite activation seen (layernorm_acc)
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave_all / CT, wave = wave_all % CT;             // position group, column tile inside the position
    char* lds_w = smem;                                            // 64 weight fragments when WLDS
    float* lds_ln = reinterpret_cast<float*>(smem + (WLDS ? 64 * FB : 0));
    char* lds_kv = reinterpret_cast<char*>(lds_ln) + 1024 + (size_t)grp * CT * 8 * FB;   // per group [CT][8] fragments: K (nt, s) then V (nt, s)
    char* scr = reinterpret_cast<char*>(lds_ln) + 1024 + (size_t)NG * CT * 8 * FB + wave_all * TileIO<2, T>::BYTES;   // wave-private tile I/O scratch
    if (WLDS) {
        const char* g = reinterpret_cast<const char*>(ws);
        for (int piece = wave_all; piece < 64 * FragInfo<T>::PIECES; piece += CT * NG) glds_piece(g + piece * 1024, lds_w + piece * 1024, lane);
    }
    if (threadIdx.x < 64) {
        const int i = threadIdx.x * 4;
        store_raw16(reinterpret_cast<char*>(lds_ln + i), load_raw16(reinterpret_cast<const char*>(ln + i)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto wfrag = [&](int f) -> Frag<T> {
        if (WLDS) return frag_from_pieces(lds_w + f * FB, lane, T());
        return load_wfrag(ws, f, lane);
    };
    const int view = wave * 32 + r;
    const bool ok = view < V;
    const int nrows = max(0, min(32, V - wave * 32));                // views of this column tile
    const size_t vstride = (size_t)hw * 64 * sizeof(T);              // one view to the next, same position
    // 0 / -inf start value of the LAST key tile's scores (the host picks CT = ceil(V / 32): every earlier tile is full)
    f32x16 negmask;
#pragma unroll
    for (int i = 0; i < 16; ++i) negmask[i] = (32 * (CT - 1) + acc_row(i, hh) >= V) ? -INFINITY : 0.0f;
    for (int base = blockIdx.x * NG; base < npix; base += gridDim.x * NG) {
        asm volatile("" ::: "memory");
        const bool active = base + grp < npix;                        // a trailing group repeats the last position and stores nothing
        const int pix = min(base + grp, npix - 1);
        const int b = pix / hw, p = pix % hw;
        const size_t off = (((size_t)b * V + min(wave * 32, V - 1)) * hw + p) * 64;   // first view of this column tile
        f32x16 x[2], n[2];
        if constexpr (sizeof(T) == 2) load_tile<2, T>(X + off, nrows, lane, x, scr, vstride);   // rows = views, coalesced through the scratch
        else load_acc<2, T>(X + (((size_t)b * V + min(view, V - 1)) * hw + p) * 64, ok, hh, x);   // fp32 build: already register-bound, keep the direct form
        {
            typename RawPiece<float>::type pr[8];
            load_lane_major_raw<2, float>(pe + (size_t)wave * 2048, lane, pr);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) n[nt] = x[nt];
            add_acc_raw<2, float>(n, pr, ok);
        }
        layernorm_acc<2, sizeof(T) == 2>(n, lds_ln, lds_ln + 64, hh, bad);
        Frag<T> nf[4], xf[4];
        acc_frags<2, T>(n, nf);
        acc_frags<2, T>(x, xf);
        f32x16 q[2], o[2];
        zero_acc<2>(q);
        {
            f32x16 k[2], v[2];
            zero_acc<2>(k); zero_acc<2>(v);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    mma(wfrag(nt * 4 + ks), nf[ks], q[nt]);
                    mma(wfrag(8 + nt * 4 + ks), nf[ks], k[nt]);
                    mma(xf[ks], wfrag(16 + nt * 4 + ks), v[nt]);      // V[view, ch]: tokens are the A operand
                }
            __syncthreads();                                            // previous position's fragments fully consumed
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const Frag<T> kf = acc_to_frag(k[nt], s2, T()), vf = acc_to_frag(v[nt], s2, T());
                    char* dk = lds_kv + ((wave * 8 + nt * 2 + s2) * FB);
                    char* dv = lds_kv + ((wave * 8 + 4 + nt * 2 + s2) * FB);
                    if constexpr (sizeof(T) == 4) {
                        store_raw16(dk + lane * 16, __builtin_bit_cast(raw16, kf.lo)); store_raw16(dk + 1024 + lane * 16, __builtin_bit_cast(raw16, kf.hi));
                        store_raw16(dv + lane * 16, __builtin_bit_cast(raw16, vf.lo)); store_raw16(dv + 1024 + lane * 16, __builtin_bit_cast(raw16, vf.hi));
                    } else {
                        store_raw16(dk + lane * 16, __builtin_bit_cast(raw16, kf.v));
                        store_raw16(dv + lane * 16, __builtin_bit_cast(raw16, vf.v));
                    }
                }
            __syncthreads();                                            // all K / V fragments of this position published
        }
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) {
            const int nt = hd >> 2, s = (hd >> 1) & 1, half = hd & 1;
            const Frag<T> qf = acc_to_frag(q[nt], s, T());
            f32x16 S[CT];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int live = j == CT - 1 ? LASTLIVE : 16;
                if (j == CT - 1) S[j] = negmask;
                else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) S[j][i] = 0.0f;
                }
                mma(frag_half(frag_from_pieces(lds_kv + (j * 8 + nt * 2 + s) * FB, lane, T()), half), qf, S[j]);   // S^T[kv, q]
#pragma unroll
                for (int i = 0; i < live; ++i) m = fmaxf(m, S[j][i]);
            }
            m = xhalf_max(m);
            const f32x2 mm = {m, m};                               // register pairs: v_pk_add_f32 (as in k_ang)
            f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int live = j == CT - 1 ? LASTLIVE : 16;
#pragma unroll
                for (int i = 0; i + 1 < live; i += 2) {
                    f32x2 d = f32x2{S[j][i], S[j][i + 1]} - mm;
                    d[0] = fast_exp2(d[0]); d[1] = fast_exp2(d[1]);
                    S[j][i] = d[0]; S[j][i + 1] = d[1];
                    sum2 += d;
                }
                if (live & 1) { S[j][live - 1] = fast_exp2(S[j][live - 1] - m); sum2[0] += S[j][live - 1]; }
#pragma unroll
                for (int i = live; i < 16; ++i) S[j][i] = 0.0f;     // rows that are no view for any lane
            }
            const float inv = sizeof(T) == 2 ? fast_rcp(xhalf_sum(sum2[0] + sum2[1])) : 1.0f / xhalf_sum(sum2[0] + sum2[1]);
            // O^T of this head in a fresh accumulator (all 32 channel rows of the product are computed, the head's own 8 rows =
            // registers 4*(hd&3)..+3 are kept): cheaper than zeroing the other heads' channels of V per head and fragment
            f32x16 oh;
#pragma unroll
            for (int i = 0; i < 16; ++i) oh[i] = 0.0f;
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    mma(frag_from_pieces(lds_kv + (j * 8 + 4 + nt * 2 + s2) * FB, lane, T()), acc_to_frag(S[j], s2, T()), oh);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[nt][4 * (hd & 3) + i] = oh[4 * (hd & 3) + i] * inv;
        }
        Frag<T> of[4];
        acc_frags<2, T>(o, of);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mma(wfrag(24 + nt * 4 + ks), of[ks], x[nt]);         // t = x + O Wo^T
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) n[nt] = x[nt];
        layernorm_acc<2>(n, lds_ln + 128, lds_ln + 192, hh, bad);
        acc_frags<2, T>(n, nf);
        f32x16 hid[4];
        zero_acc<4>(hid);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) mma(wfrag(32 + nt * 4 + ks), nf[ks], hid[nt]);
        Frag<T> hf[8];
        acc_frags_relu<4>(hid, hf);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) mma(wfrag(48 + nt * 8 + ks), hf[ks], x[nt]);
        if constexpr (sizeof(T) == 2) store_tile<2, T>(Y + off, active ? nrows : 0, lane, x, scr, vstride);
        else store_acc<2, T>(Y + (((size_t)b * V + min(view, V - 1)) * hw + p) * 64, ok && active, hh, x);
    }
    publish_status(status, bad);
}
