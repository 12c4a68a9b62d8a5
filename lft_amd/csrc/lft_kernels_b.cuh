// lft_kernels_b.cuh -- spatial Transformer (token embedding + QKV, windowed attention, MLP tail),
// fused up-sampler and the final assemble (overlap-add + bicubic skip).
#pragma once
#include "lft_common.cuh"
#include "lft_kernels_a.cuh"

// ------------------------------------------------------------------------------------------
// SpaTrans part 1 (reference LFT.py:164-169, 179-186): per 32-token tile of one view image
//   tok = conv3x3(x; MLP.weight as [128,64,3,3])            (unfold + Linear 576->128)
//   n   = LN(tok + PEtok[p])                                 (PEtok = same embedding of the position image, cached)
//   Q = n Wq^T (pre-scaled by 1/4 * log2 e), K = n Wk^T, V = tok Wv^T
// Stream: conv[36 x 4] Wv[4x8] Wk[4x8] Wq[4x8]  (240 fragments).
// PE_ONLY: embed the position image itself and write the tokens (pack-time precompute).
// WITH_Q = false (16-bit path with lane-major hand-off): Q is NOT produced here -- k_spa_b computes it from the token tile it
// loads anyway (one tensor less written and read back: -52 MB per layer at B = 4); the ring then ends after Wk (208 fragments).
// ------------------------------------------------------------------------------------------
#ifndef LFT_UP_CHUNK
#define LFT_UP_CHUNK 8
#endif
constexpr int kUpChunk = LFT_UP_CHUNK;    // k_up uses few registers: a smaller ring lets more workgroups share a CU
#ifndef LFT_SPA_CHUNK
#define LFT_SPA_CHUNK 16
#endif
#ifndef LFT_SPA_OCC
#define LFT_SPA_OCC 2
#endif
constexpr int kSpaChunk = LFT_SPA_CHUNK;   // fragments per ring chunk: one conv tap (4 k-steps x 4 row tiles), half an in_proj matrix
// Waves per workgroup (x 32 tokens each).  All waves of a workgroup share one weight ring: the packed weights are
// streamed into the CU once per 32*NW tokens, and one ring barrier serves NW waves.
#ifndef LFT_NW_SPA1
#define LFT_NW_SPA1 4
#endif
#ifndef LFT_NW_SPA2
#define LFT_NW_SPA2 4
#endif
#ifndef LFT_NW_UP
#define LFT_NW_UP 8      // k_up is light on registers: eight waves share one weight stream (48 -> 42 us at B = 4)
#endif
constexpr int kNwSpa1 = LFT_NW_SPA1, kNwSpa2 = LFT_NW_SPA2, kNwUp = LFT_NW_UP;
template <typename T, bool PE_ONLY, int CH = kSpaChunk, bool TOKLM = false, bool WITH_Q = true, int NW = kNwSpa1>   // TOKLM: the token tile goes to part B in lane-major tile format
__global__ __launch_bounds__(64 * NW, LFT_SPA_OCC) void k_spa1(const T* __restrict__ X, const T* __restrict__ ws,
                                              const float* __restrict__ ln, const T* __restrict__ petok,
                                              T* __restrict__ TOK, T* __restrict__ Q, T* __restrict__ K, T* __restrict__ Vv,
                                              T* __restrict__ pe_out, int nimg, int h, int w, unsigned* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5, wave = threadIdx.x >> 6;
    constexpr int TT = 32 * NW;                                       // tokens per workgroup tile
    const int hw = h * w, tpi = (hw + TT - 1) / TT;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);                  // neighbouring tiles (shared halo rows) on one XCD
    const int im = bid / tpi, p0 = (bid % tpi) * TT;
    const int tl = wave * 32 + r, p = p0 + tl;
    const bool ok = p < hw;
    LFT_STAMP(0);
    typename RawPiece<T>::type pe_raw[16];                                    // position tokens of this lane's token, kept packed
    if (!PE_ONLY) load_lane_major_raw<4, T>(petok + (size_t)((p0 >> 5) + wave) * 4096, lane, pe_raw);   // early, 8 coalesced loads
    char* lds_in = smem + WRing<T, CH, NW>::LDS_BYTES;
    float* lds_ln = reinterpret_cast<float*>(lds_in + max(ConvIn<T, NW>::bytes(w), NW * TileIO<4, T>::BYTES));   // the tile I/O scratch aliases the conv input
    char* zero_row = reinterpret_cast<char*>(lds_ln) + 1024;           // behind the 256 LayerNorm floats
    raw16 lnv = raw16{0u, 0u, 0u, 0u};
    if (!PE_ONLY) lnv = params_load(ln, 256);                         // norm.{weight,bias}; ln is null in the pack-time PE_ONLY launch
    WRing<T, CH, NW> ring;
    ring.init(ws, smem, PE_ONLY ? 144 : (WITH_Q ? 240 : 208));
    stage_conv_input<T, NW>(X + (size_t)im * hw * 64, p0, hw, w, lds_in);
    clear_zero_row(zero_row);
    LFT_STAMP(12);
    wait_staged();
    LFT_STAMP(13);
    if (!PE_ONLY) params_store(lds_ln, 256, lnv);
    __syncthreads();                                                  // input tile, first weight chunks and LN parameters published
    LFT_NOTE_ASM_("USE", kNoteConvIn, 0);
    LFT_STAMP(1);
    f32x16 t[4];
    zero_acc<4>(t);
    conv3x3_tile<4, T, NW>(lds_in, zero_row, tl, p / w, p % w, ok, h, w, hh, ring, t);
    LFT_STAMP(2);
    if (PE_ONLY) {
        store_lane_major<4, T>(pe_out + (size_t)((p0 >> 5) + wave) * 4096, lane, t);   // lane-major table, one 32-token tile per wave
        return;
    }
    // Tile I/O scratch aliases the (now dead) conv input tile: every wave must be done reading it first.
    __syncthreads();
    LFT_STAMP(3);
    const int t0 = p0 + wave * 32, nvalid = max(0, min(32, hw - t0));
    const size_t tile_off = ((size_t)im * hw + min(t0, hw - 1)) * 128;
    char* scr = lds_in + wave * TileIO<4, T>::BYTES;
    if constexpr (TOKLM) store_tile_lm<4, T>(TOK + ((size_t)im * hw + t0) * 128, lane, t);   // hw % (32 NW) == 0: every tile is full
    else store_tile<4, T>(TOK + tile_off, nvalid, lane, t, scr);
    LFT_STAMP(4);
    Frag<T> nf[8];
    // Each projection is produced and stored in two 64-channel halves: 32 accumulator registers instead of 64
    // keep the kernel inside 256 VGPRs (2 waves/SIMD) without scratch spills -- a spill reload forces
    // s_waitcnt vmcnt(0), which drains the weight DMA and every output store in flight.
    // bf16 + lane-major hand-off: Q, K and V leave in the same lane-major tile form as the tokens (16-byte stores straight
    // from the accumulators, no LDS transposition): k_spa_b gathers its K / V halo tiles from that form by LDS-DMA -- whole
    // 128-byte lines per 8 tokens, where a row-major [token][128] tensor gives half a line per token and head pair.
    constexpr bool QKVLM = TOKLM && sizeof(T) == 2;
    const size_t lm_off = ((size_t)im * hw + t0) * 128;               // this wave's 32-token tile in a lane-major [tile][k-step][lane][8] tensor
    acc_frags<4, T>(t, nf);                    // V = tok Wv^T first (raw tokens, reference LFT.py:185); tok is then normalised in place
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 a[2];
        zero_acc<2>(a);
        linear_ring<2, 8, T>(ring, nf, a);
        if constexpr (QKVLM) store_tile_lm<2, T>(Vv + lm_off + half * 2048, lane, a);
        else store_tile<2, T, 128>(Vv + tile_off + 64 * half, nvalid, lane, a, scr);
    }
    LFT_STAMP(6);
    add_acc_raw<4, T>(t, pe_raw, ok);
    unsigned bad = 0;
    layernorm_acc<4, sizeof(T) == 2>(t, lds_ln, lds_ln + 128, hh, bad);        // also the overflow detector for the token embedding
    acc_frags<4, T>(t, nf);
    LFT_STAMP(7);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 a[2];
        zero_acc<2>(a);
        linear_ring<2, 8, T>(ring, nf, a);
        if constexpr (QKVLM) store_tile_lm<2, T>(K + lm_off + half * 2048, lane, a);
        else store_tile<2, T, 128>(K + tile_off + 64 * half, nvalid, lane, a, scr);
    }
    LFT_STAMP(9);
    if constexpr (WITH_Q) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 a[2];
            zero_acc<2>(a);
            linear_ring<2, 8, T>(ring, nf, a);
            if constexpr (QKVLM) store_tile_lm<2, T>(Q + lm_off + half * 2048, lane, a);
            else store_tile<2, T, 128>(Q + tile_off + 64 * half, nvalid, lane, a, scr);
        }
    }
    LFT_STAMP(11);
    publish_status(status, bad);
}

// ------------------------------------------------------------------------------------------
// SpaTrans windowed attention (reference LFT.py:147-162 mask + :183-187 attention): keys = clamped 5x5 window around
// the query.  The reference bounds the window columns by min(h, x+3) (LFT.py:155, "h" where "w" is meant) and slicing
// clips at w; reproduced as-is: for h < w some queries see no key at all and get a zero attention output (what the
// reference gives under torch >= 2.5).  Q is pre-scaled by scale*log2(e); softmax uses exp2.
// bf16: fused with the per-token tail in k_spa_b below; fp32: k_win_attn_lds (lft_train.cuh) followed by k_spa2.
// Geometry of the MFMA attention: a workgroup = a 4 x 32 tile of queries of one view image, wave c owning the 8 x 4
// block of columns 8c .. 8c+7 (32 queries on its 32 MFMA columns).  The block's clamped 5x5 windows live inside a
// 12 x 8 neighbourhood = 96 keys = three 32-key tiles; the workgroup's keys are the 8 x 36 halo tile around it.
// ------------------------------------------------------------------------------------------
constexpr int kAttTY = 4, kAttTX = 32, kAttHR = kAttTY + 4, kAttHC = kAttTX + 4;
// quarter swizzle of the K / V halo tiles in LDS (k_spa_b): halo (row, column) -> XOR mask of the 16-byte quarter index
LFT_DEV int att_swz(int row, int col) { return (row & 1) | (((col >> 2) & 1) << 1); }
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// SpaTrans part 2 (reference LFT.py:187-189, 171-174): per 32-token tile
//   t  = tok + O Wo^T ;  t2 = t + W2 relu(W1 LN'(t)) ;  y = Wl t2  (Conv3d 1x1x1 128->64)  [+ global skip, LFT.py:76]
// FFN hidden width 256 is processed in four 64-wide chunks so the hidden activations never leave registers.
// Stream: Wo[4x8, natural k] {W1c[2x8] W2c[4x4]} x4  Wl[2x8]  (176 fragments).
// ------------------------------------------------------------------------------------------
template <typename T, bool SKIP, bool TOKLM = false, bool YLM = false, int NW = kNwSpa2>   // YLM: output tile in lane-major form (consumer: k_up)
__global__ __launch_bounds__(64 * NW, LFT_SPA_OCC) void k_spa2(const T* __restrict__ TOK, const T* __restrict__ O, const T* __restrict__ ws,
                                              const float* __restrict__ ln, const T* __restrict__ skip, T* __restrict__ Y,
                                              long long ntok, unsigned* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const long long t0 = ((long long)blockIdx.x * NW + wave) * 32;                    // this wave's 32 consecutive tokens
    const int nvalid = (int)max(0LL, min(32LL, ntok - t0));
    const long long tb = min(t0, ntok - 1);
    char* scr = smem + WRing<T, kSpaChunk, NW>::LDS_BYTES + 1024 + wave * TileIO<4, T>::BYTES;   // wave-private tile I/O scratch
    WRing<T, kSpaChunk, NW> ring;
    LFT_STAMP(16);
    ring.init(ws, smem, 176);                 // first: the weight DMA is in flight while the activation tiles are fetched
    f32x16 t[4], n[4];
    if constexpr (TOKLM) load_tile_lm<4, T>(TOK + tb * 128, lane, t);       // written by k_spa1 in the same 32-token tiling
    else load_tile<4, T>(TOK + tb * 128, nvalid, lane, t, scr);
    Frag<T> f[8];
    load_tile_frags<8, T>(O + tb * 128, nvalid, lane, f, scr);
    f32x16 sk[2];
    if (SKIP) load_tile<2, T>(skip + tb * 64, nvalid, lane, sk, scr);
    float* lds_ln = reinterpret_cast<float*>(smem + WRing<T, kSpaChunk, NW>::LDS_BYTES);
    params_store(lds_ln, 256, params_load(ln + 256, 256));            // feed_forward.0.{weight,bias}; the tile loads above were waited for anyway; published by the first ring barrier
    LFT_STAMP(17);
    linear_ring<4, 8, T>(ring, f, t);
    LFT_STAMP(18);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) n[nt] = t[nt];
    unsigned bad = 0;
    layernorm_acc<4>(n, lds_ln, lds_ln + 128, hh, bad);
    acc_frags<4, T>(n, f);
    LFT_STAMP(19);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f32x16 hid[2];
        zero_acc<2>(hid);
        linear_ring<2, 8, T>(ring, f, hid);
        Frag<T> hf[4];
        acc_frags_relu<2>(hid, hf);
        linear_ring<4, 4, T>(ring, hf, t);
    }
    LFT_STAMP(20);
    acc_frags<4, T>(t, f);
    f32x16 y[2];
    zero_acc<2>(y);
    linear_ring<2, 8, T>(ring, f, y);
    if (SKIP) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) y[nt] += sk[nt];
    }
    LFT_STAMP(21);
    if constexpr (YLM) store_tile_lm<2, T>(Y + tb * 64, lane, y);
    else store_tile<2, T>(Y + tb * 64, nvalid, lane, y, scr);
    LFT_STAMP(22);
    publish_status(status, bad);
}

// ------------------------------------------------------------------------------------------
// SpaTrans part B, bf16 (reference LFT.py:183-189, 171-174): windowed attention AND the per-token tail in one kernel.
// The O^T accumulators of the MFMA attention (channel on the register, query on the lane) are -- converted to bf16 --
// exactly the B operand of out_proj in "acc order", so the attention output never goes to memory (k_spa_attn_mfma wrote
// 26 MB per launch at B = 4 and k_spa2 read them back through an LDS transposition).  A workgroup = a 4 x 32 tile of
// queries of one view image, wave c owning the 8 x 4 block of columns 8c .. 8c+7, as in k_spa_attn_mfma; the same 32
// tokens then go through  t = tok + O Wo^T ; t += W2 relu(W1 LN'(t)) ; y = Wl t (+ global skip)  as in k_spa2.
// Stream: Wo[4x8, ACC order] {W1c[2x8] W2c[4x4]} x4  Wl[2x8]  (176 fragments) through an 8-fragment-chunk ring, so that
// ring + K/V halo tiles stay below 80 KiB and two workgroups share a CU.  TOK / skip / Y are row-major [token][channel];
// a wave's 8 x 4 block is two 4 x 4 blocks side by side (BlkRows: token 16 b + 4 py + px).
// ------------------------------------------------------------------------------------------
#ifndef LFT_SPAB_CHUNK
#define LFT_SPAB_CHUNK 8
#endif
constexpr int kSpaBChunk = LFT_SPAB_CHUNK;                     // fragments per ring chunk in phase B
constexpr int kAdTile = kAttHR * kAttHC * 64;                 // one tensor's halo tile in LDS: 8 x 36 tokens x 64 B
constexpr int kAdPerWave = 2 * kAdTile / 1024 / 4;            // LDS-DMA pieces per wave and head pair (9)
static_assert(2 * kAdTile == 4 * kAdPerWave * 1024, "the K and V tiles must split into whole pieces over 4 waves");
constexpr int kSpaBLds = 4 * kAdTile + 2048;                  // two K+V buffers + both LayerNorms' parameters (FFN: first KiB; norm: second)
// Phase B re-uses the two K / V buffers (36 KiB each): ring slots 0 .. kSpaBSlotsA-1 in buffer A, the others in buffer B from
// its start; the tile I/O scratch sits 16 KiB into buffer B -- behind the slots there (8-fragment chunks), or on top of the
// LAST slot (16-fragment chunks), which is not filled before every wave has passed the first ring barrier (TOK tile loaded)
// and has had its last chunk consumed well before the final store.
constexpr int kSpaBSlotsA = (2 * kAdTile) / (kSpaBChunk * 1024);
constexpr int kSpaBSlotsB = kSpaBChunk == 8 ? 2 : 2 * kAdTile / (kSpaBChunk * 1024);
constexpr int kSpaBSlots = kSpaBSlotsA + kSpaBSlotsB;
constexpr int kSpaBScratchOfs = 16384;
static_assert(kSpaBChunk == 8 || kSpaBChunk == 16, "phase-B ring chunk: 8 or 16 fragments");
static_assert(kSpaBScratchOfs + 4 * TileIO<4, bf16_t>::BYTES <= 2 * kAdTile, "scratch must fit buffer B");
static_assert(kSpaBChunk == 8 ? kSpaBSlotsB * kSpaBChunk * 1024 <= kSpaBScratchOfs : (kSpaBSlotsB - 1) * kSpaBChunk * 1024 <= kSpaBScratchOfs,
              "scratch may only overlap the last ring slot");

// Two 16-byte query fragments of one head pair, loaded by inline asm so that hipcc neither counts them nor drains the
// LDS-DMA in flight when they are used (it waits vmcnt(0) for ordinary loads while a global_load_lds is outstanding).
// They complete with the counted wait at the top of their head pair's iteration (vmcnt retires in issue order).
template <typename T> LFT_DEV void q_load_async(const T* p0, const T* p1, raw16& a, raw16& b) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "=&v"(a), "=&v"(b) : "v"(p0), "v"(p1) : "memory");
}
// One 16-byte load per lane that hipcc neither counts nor waits for (see q_load_async); the caller's counted wait names the
// destination registers before their first use.
LFT_DEV void ld16_async(const void* p, raw16& d) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// The same with a compile-time byte offset folded into the instruction (13-bit signed immediate: 0 .. 4095): one address
// register pair serves four 1 KiB-spaced pieces instead of one 64-bit add per load.
template <int OFS> LFT_DEV void ld16_async_ofs(const void* p, raw16& d) {
    static_assert(OFS >= 0 && OFS < 4096, "global_load immediate offset");
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFS) : "memory");
}
// eight pieces 1 KiB apart starting at p (two address pairs)
LFT_DEV void ld16_async_x8(const char* p, raw16 (&d)[8]) {
    const char* p2 = p + 4096;
    ld16_async_ofs<0>(p, d[0]); ld16_async_ofs<1024>(p, d[1]); ld16_async_ofs<2048>(p, d[2]); ld16_async_ofs<3072>(p, d[3]);
    ld16_async_ofs<0>(p2, d[4]); ld16_async_ofs<1024>(p2, d[5]); ld16_async_ofs<2048>(p2, d[6]); ld16_async_ofs<3072>(p2, d[7]);
}
// s_waitcnt vmcnt(N), then pin: no use of the eight registers moves above the wait (call once per group of eight; only the
// first call of a sequence needs the real count, the others pass the same N -- a second identical wait costs nothing).
template <int N> LFT_DEV void wait_vm_8(raw16 (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> LFT_DEV void wait_vm_only() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> LFT_DEV void wait_vm_q(raw16& a, raw16& b) {          // s_waitcnt vmcnt(N); names the asm-loaded registers so no use moves above it
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Q^T rows 32 nt .. 32 nt + 31 (one head pair) of a wave's 32 tokens: eight Wq fragments (already in registers) against the
// normalised tokens; the two 16-row halves of the accumulator are the head pair's query fragments (acc order, as K).
template <typename T> LFT_DEV void q_head_pair(const raw16 (&wr)[8], const Frag<T> (&nf)[8], Frag<T>& q0, Frag<T>& q1) {
    f32x16 q;
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        Frag<T> wf;
        wf.v = __builtin_bit_cast(typename H16<T>::v8, wr[ks]);
        mma(wf, nf[ks], q);
    }
    q0 = acc_to_frag(q, 0, T());
    q1 = acc_to_frag(q, 1, T());
}

template <typename T, bool SKIP, bool TOKLM = false, bool YLM = false>   // TOKLM: k_spa1 wrote the tokens as lane-major 32-token tiles (w % 32 == 0: a tile = 32 columns of one image row); YLM: write the output so (consumer: k_up)
__global__ __launch_bounds__(256, 2) void k_spa_b(const T* __restrict__ TOK, const T* __restrict__ Q, const T* __restrict__ K,
                                                  const T* __restrict__ Vv, const T* __restrict__ ws, const float* __restrict__ ln,
                                                  const T* __restrict__ skip, T* __restrict__ Y, int h, int w, unsigned* __restrict__ status,
                                                  const T* __restrict__ wq, const T* __restrict__ petok) {   // wq / petok: TOKLM only (Q is computed here)
    static_assert(sizeof(T) == 2, "k_spa_b is the 16-bit (bf16 / f16) part B; fp32 uses k_win_attn_lds + k_spa2");
    typedef typename H16<T>::v8 V8;
    using Ring = WRingPipe<T, kSpaBChunk, 4, kSpaBSlots, 176>;        // weight fragments double-buffered in registers across the chunk barrier
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufA = smem;                                          // each buffer: K tile, then V tile, of one head pair
    char* const bufB = smem + 2 * kAdTile;
    float* lds_ln = reinterpret_cast<float*>(smem + 4 * kAdTile);
    const int tiles_x = (w + kAttTX - 1) / kAttTX, tiles_y = (h + kAttTY - 1) / kAttTY;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);                  // vertically adjacent tiles share 4 of their 8 halo rows
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, im = bid / (tiles_x * tiles_y);
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y0 = ty * kAttTY, x0 = tx * kAttTX, bxl = 8 * wave;          // block origin: (y0, x0 + bxl)
    const long long img0 = (long long)im * h * w;
    LFT_STAMP(16);
    // This lane's query.  The wave's 8 x 4 block is two 4 x 4 blocks side by side (columns bxl .. bxl+3 and bxl+4 .. bxl+7); token
    // (= MFMA column) r = 16 b + 4 py + px: tokens 0..15 are block A, 16..31 block B, each row-major.  With that order the two
    // 16-lane rows of a 32-column fragment's half ARE the two blocks, which is what lets v_permlane16_swap regroup a fragment
    // into the score tiles' 16-column operands (frag_to_blocks).
    const int qpx = 4 * (r >> 4) + (r & 3), qpy = (r >> 2) & 3;      // position inside the 8 x 4 block
    const int qy = y0 + qpy, qx = x0 + bxl + qpx;
    const long long qtok = img0 + min(qy, h - 1) * w + min(qx, w - 1);
    // row-major Q: [token][128], head hd = channels 16 hd .. (natural order); lane-major: piece [k-step hd][32 hh + column] of
    // the tile of image row qy (acc order -- K / V come in the same order, so the dot products agree, and the channel order
    // of V^T's rows, i.e. of the attention output, is undone by packing Wo in natural k order: lft_api.hip)
    const T* qptr = TOKLM ? Q + (img0 + (long long)min(qy, h - 1) * w + x0) * 128 + (32 * hh + bxl + qpx) * 8
                               : Q + qtok * 128 + 8 * hh;
    constexpr int kQHead = TOKLM ? 512 : 16, kQPair = 2 * kQHead;     // element stride from one head / head pair to the next
    // K / V halo tiles (8 x 36 tokens x 2 heads = 64 B per token, unpadded) come in by LDS-DMA: no staging registers, no
    // LDS store instructions.  36 one-KiB pieces per head pair: waves 0,1 fetch K, waves 2,3 fetch V, 9 pieces each.  Piece
    // u covers 16 tokens; lane l moves 16-byte unit 64 u + l = (token slot, quarter).  Tokens outside the image are
    // fetched from the clamped position: finite values that the -inf bias (K) / zero probability (V) keep out of the result.
    int dofs[kAdPerWave];                                             // byte offsets of this lane's units from the image's first token, head pair 0
    {
        // unit u = (16 (9 (wave & 1) + i) + lane / 4, lane & 3): the slot advances by 16 per piece -- one division for the
        // first slot, then row / column by carry (the kernel is bound by vector-instruction issue, prologue included)
        const int s0 = (wave & 1) * kAdPerWave * 16 + (lane >> 2), piece = lane & 3;
        int row = s0 / kAttHC, col = s0 % kAttHC;
#pragma unroll
        for (int i = 0; i < kAdPerWave; ++i) {
            const int gy = min(max(y0 - 2 + row, 0), h - 1), gx = min(max(x0 - 2 + col, 0), w - 1);
            const int tk = gy * w + gx;
            // Bank swizzle: the four 16-byte quarters of a token's 64 bytes are stored XOR-permuted by (halo row parity, bit 2 of the
            // halo column).  A score tile's K read (16 tokens = 2 rows x 8 columns, one quarter) and a V^T transposing read (8 tokens of
            // one row, two quarters) then touch all 64 banks once per 32-lane half; unswizzled (64-byte token stride) both are 4-way.
            const int pq = piece ^ att_swz(row, col);
            const int pconst = TOKLM ? ((pq >> 1) * 512 + (pq & 1) * 256) * 2 : pq * 16;               // lane-major: piece = (head, half)
            dofs[i] = (TOKLM ? ((tk & ~31) << 8) + ((tk & 31) << 4) : tk * 256) + pconst;
            col += 16;
            if (col >= kAttHC) { col -= kAttHC; row += 1; }
        }
    }
    constexpr int kDmaPair = TOKLM ? 2048 : 64;                       // bytes from one head pair to the next
    // Issued from inline asm (glds16_asm): the compiler must not know about the pieces in flight, or it drains them in
    // front of the next LDS read.
    const char* const dsrc = reinterpret_cast<const char*>((wave < 2 ? K : Vv) + img0 * 128);
    const int ddst = (wave < 2 ? 0 : kAdTile) + (wave & 1) * kAdPerWave * 1024;       // this wave's part of a buffer
    auto stage = [&](int hg, char* buf) {
        LFT_DMA_NOTE("DMA", kNoteKV, hg & 1);
#pragma unroll
        for (int i = 0; i < kAdPerWave; ++i) glds16_asm(dsrc + dofs[i] + hg * kDmaPair, buf + ddst + i * 1024);
    };
    // Pipeline (VM operations retire in issue order), row-major form:  D0 D1 Q0 | it0: wait(0) .. Q1 D2 | it1: wait(9) .. Q2 D3 |
    // it2: wait(9) .. Q3 R0-3 | it3: wait(8) .. R4 | phase B.   Dn = the 9 DMA pieces of head pair n, Qn = its two query
    // loads, Rc = ring chunk c (2 pieces per wave).  The wait at the top of iteration n leaves only the group issued last
    // in flight, so Dn and Qn have landed; a buffer is re-filled right after the barrier that ends its head pair.  Q0 is
    // issued last in the prologue: between an asm load and its wait the compiler must have no reason to touch the
    // destination registers (it does not know they are pending).
    // Lane-major form (TOKLM): there are no query loads -- Q is COMPUTED in the prologue (below) -- and the iterations' waits
    // are the same counts without them:  [tok pe Wq01 D0 D1] wait(34) LN | wait(26) Q0 Wq2 | wait(26) Q1 Wq3 | wait(8) Q2 |
    // wait(0) Q3 | it0 .. D2 | it1: wait(9) .. D3 | it2: wait(9) .. R0-3 | it3: wait(8) ..
    // Score tiles: per block b (columns bxl + 4 b ..) the 8 x 8 key neighbourhood = halo rows 0..7, halo columns bxl + 4 b .. + 7,
    // cut into four 16-key tiles (tile t = halo rows 2 t, 2 t + 1, key kk -> row kk / 8, column kk % 8).  S^T[key, query] of tile
    // (b, t): lane 16 g + qi holds query qi of block b against keys 4 g .. 4 g + 3 (register e), i.e. halo row 2 t + (g >> 1),
    // neighbourhood columns 4 (g & 1) + e.  The 0 / -inf bias (window, image border, the `min(h, x+3)` quirk of LFT.py:155) is the
    // product of a row mask and a column mask of the lane's query: one bit-field extract + AND per element.
    const int g4 = lane >> 4, qi = lane & 15;
    f32x4 bias[2][4];
    int kb[2], vb[2];
    auto setup_tiles = [&]() __attribute__((always_inline)) {
    {
        const int wy0 = max(0, qy - 2), wy1 = min(h, qy + 3), cy0 = y0 - 2;
        const int ya = min(max(wy0 - cy0, 0), 8), yb = min(max(wy1 - cy0, 0), 8);
        const unsigned ymask = ((1u << yb) - 1u) & ~((1u << ya) - 1u);     // halo rows inside the window (empty when yb <= ya)
        const int gy = g4 >> 1, gx4 = 4 * (g4 & 1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // the lane's query in block b: same row, column (lane & 3) of the block (the lane's OWN token r is one of the two)
            const int bqx = x0 + bxl + 4 * b + (lane & 3), cx0 = x0 + bxl + 4 * b - 2;
            const int wx0 = max(0, bqx - 2), wx1 = min(min(h, bqx + 3), w);                   // reference LFT.py:155 (sic)
            const int xa = min(max(wx0 - cx0, 0), 8), xb = min(max(wx1 - cx0, 0), 8);
            const unsigned xs = (((1u << xb) - 1u) & ~((1u << xa) - 1u)) >> gx4;                // bit e: neighbourhood column gx4 + e
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned inval = ~(xs & (unsigned)__builtin_amdgcn_sbfe((int)ymask, 2 * t + gy, 1));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    bias[b][t][e] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_sbfe((int)inval, e, 1) & 0xff800000u);   // 0 inside the window, -inf outside
            }
        }
    }
    // K rows as the A operand of v_mfma_f32_16x16x16: lane 16 g + qi reads labels 4 g .. 4 g + 3 (8 bytes) of key qi of the tile;
    // V^T by transposing reads: lane 16 g + 4 q + p supplies the address of key 4 g + q, channels 4 p .. 4 p + 3.  One base each;
    // block, tile and head are compile-time offsets folded into the instructions.
    // With the quarter swizzle (att_swz) the label quarter 2 hl + gq of a token sits in stored quarter (2 hl + gq) ^ s(row, col);
    // s's column bit is (b ^ lane bit), so there are two bases, selected at compile time by hl ^ b.
    {
        const int ktok = ((qi >> 3) * kAttHC + bxl + (qi & 7)) * 64, ks0 = (qi >> 3) & 1, ks1 = (qi >> 2) & 1;
        const int vtok = ((g4 >> 1) * kAttHC + bxl + 4 * (g4 & 1) + (qi >> 2)) * 64, vs0 = (g4 >> 1) & 1, vs1 = g4 & 1, vp = qi & 3;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            kb[x] = ktok + 32 * (x ^ ks1) + 16 * ((g4 >> 1) ^ ks0) + 8 * (g4 & 1);
            vb[x] = kAdTile + vtok + 32 * (x ^ vs1) + 16 * ((vp >> 1) ^ vs0) + 8 * (vp & 1);
        }
    }
    };
    unsigned bad = 0;                                                 // non-finite activation seen (layernorm_acc; published at the end)
    Frag<T> qfr[8];                                                   // TOKLM: this block's queries, all 8 heads, acc order
    raw16 tokr[8];                                                    // TOKLM: this lane's token pieces, kept packed for the residual of phase B
    raw16 lnv;                                                        // feed_forward.0.{weight,bias}, to LDS after the first wait
    if constexpr (TOKLM) {
        // Q = LN(tok + PEtok) Wq^T for this wave's 32 tokens (reference LFT.py:181-184), from the token tile k_spa1 wrote
        // (16-bit) -- k_spa1 no longer writes Q and this kernel no longer reads it: one [N, 128] tensor less each way.  All
        // loads are inline-asm loads (the compiler neither counts nor drains them) under counted waits; Wq streams through
        // two sets of eight fragment registers, 16 KB from L2 per wave and set.
        raw16 lnv1, per[8], wqa[8], wqb[8];
        const int ty_ = min(qy, h - 1);
        const int lane_piece = (32 * hh + bxl + qpx) * 8;                // this lane's 16-byte piece inside a k-step of a lane-major tile
        const char* tsrc = reinterpret_cast<const char*>(TOK + (img0 + (long long)ty_ * w + x0) * 128 + lane_piece);
        const char* psrc = reinterpret_cast<const char*>(petok + ((long long)ty_ * w + x0) * 128 + lane_piece);
        const char* wsrc = reinterpret_cast<const char*>(wq) + lane * 16;
        const int pi = min((int)threadIdx.x * 4, 252);                   // as params_load: threads beyond 64 re-read the last piece
        ld16_async(ln + pi, lnv1);                                       // norm.{weight,bias}
        ld16_async(ln + 256 + pi, lnv);
        ld16_async_x8(tsrc, tokr);
        ld16_async_x8(psrc, per);
        ld16_async_x8(wsrc, wqa);                                         // head pair 0: fragments (nt = 0, ks = 0..7)
        ld16_async_x8(wsrc + 8 * 1024, wqb);                              // head pair 1
        stage(0, bufA);
        stage(1, bufB);
#ifdef LFT_SPAB_EARLY_SETUP
        // EXPERIMENT (round 4, not adopted): the pure vector work that needs none of the data in flight (window bias tiles, LDS read
        // bases: ~300 instructions) here, under the first memory round trip, instead of behind the Q projection.  With 52 asm loads
        // pending and 32 more live registers hipcc spilled (132 B of scratch) and MOVED asm-loaded registers before their counted wait:
        // tools/asm_load_hazards.py reports 57 reads of in-flight registers for this variant.  The block therefore stays behind Q.
        setup_tiles();
#endif
        // (No LFT_STAMP between here and the last Wq wait: a stamp is compiler-visible code with live registers of its own -- in the
        // diagnostic build it made hipcc spill and MOVE asm-loaded registers that were still in flight, and the kernel faulted on a
        // garbage address.  Run tools/asm_load_hazards.py on a diagnostic listing before launching it.)
        // in flight: 2 + 16 + 16 + 18 = 52.  Tokens, position tokens and LayerNorm parameters first:
        wait_vm_8<16 + 2 * kAdPerWave>(tokr);
        wait_vm_8<16 + 2 * kAdPerWave>(per);
        asm volatile("" : "+v"(lnv), "+v"(lnv1));
        params_store(lds_ln, 256, lnv);
        params_store(lds_ln + 256, 256, lnv1);
        wg_barrier_keep_vm();                                            // both parameter sets published
        f32x16 n[4];
#pragma unroll
        for (int kidx = 0; kidx < 8; ++kidx) {
            const V8 vt = __builtin_bit_cast(V8, tokr[kidx]), vp = __builtin_bit_cast(V8, per[kidx]);
#pragma unroll
            for (int j = 0; j < 8; ++j) n[kidx >> 1][8 * (kidx & 1) + j] = (float)vt[j] + (float)vp[j];
        }
        // (Round 4 tried a third fragment set in the dead position-token registers, requested here so that head pair 2's Wq arrives
        // under the LayerNorm instead of costing an L2 round trip in front of Q2: 32 more live registers across the LayerNorm -> 148 B
        // of scratch and asm-loaded registers moved while in flight (tools/asm_load_hazards.py: 280 reads).  Not adopted.)
        layernorm_acc<4, true>(n, lds_ln + 256, lds_ln + 256 + 128, hh, bad);
        Frag<T> nf[8];
        acc_frags<4, T>(n, nf);
        wait_vm_8<8 + 2 * kAdPerWave>(wqa);                              // younger: Wq set b, D0, D1
        q_head_pair<T>(wqa, nf, qfr[0], qfr[1]);
        ld16_async_x8(wsrc + 16 * 1024, wqa);                             // head pair 2
        wait_vm_8<2 * kAdPerWave + 8>(wqb);                              // younger: D0, D1, the new set a
        q_head_pair<T>(wqb, nf, qfr[2], qfr[3]);
        ld16_async_x8(wsrc + 24 * 1024, wqb);                             // head pair 3
        wait_vm_8<8>(wqa);                                               // D0 and D1 are older: landed as well
        q_head_pair<T>(wqa, nf, qfr[4], qfr[5]);
        wait_vm_8<0>(wqb);
        q_head_pair<T>(wqb, nf, qfr[6], qfr[7]);
        // each head's query fragment -> the two blocks' 16-column operands, in place (2 swaps per head)
#pragma unroll
        for (int hd = 0; hd < 8; ++hd) {
            raw16 f = __builtin_bit_cast(raw16, qfr[hd].v);
            frag_to_blocks(f);
            qfr[hd].v = __builtin_bit_cast(V8, f);
        }
    } else {
        lnv = params_load(ln + 256, 256);
        stage(0, bufA);
        stage(1, bufB);
#ifdef LFT_SPAB_EARLY_SETUP
        setup_tiles();
#endif
    }
    Ring ring;
    ring.setup(ws, smem, kSpaBSlotsA, 2 * kAdTile - kSpaBSlotsA * kSpaBChunk * 1024);
    // this wave's block of tokens in memory (clamped origin: readable even when the block lies outside the image)
    BlkRows rows;
    rows.nrow = max(0, min(4, h - y0)); rows.ncol = max(0, min(8, w - (x0 + bxl)));
    if (rows.ncol == 0) rows.nrow = 0;
    const long long tok0 = img0 + (long long)min(y0, h - 1) * w + min(x0 + bxl, w - 1);
#ifndef LFT_SPAB_EARLY_SETUP
    setup_tiles();
#endif
    LFT_STAMP(17);
    raw16 qa, qb;
    if constexpr (!TOKLM) q_load_async(qptr, qptr + kQHead, qa, qb);
    Frag<T> of[8];                                                    // attention output of all 8 heads: out_proj's B operand, acc order
    // Wave priority: the attention phase is bound by vector-instruction issue, the tail behind it by the matrix pipe, and the two waves of
    // a SIMD are usually in different phases.  Raised priority for the attention phase lets its instructions issue ahead of the other
    // wave's MFMA chain (which keeps the matrix pipe busy with what it has in flight): k_spa_b -1.8 .. -2.7 % in serial traces on three
    // boxes (51.2 -> 50.3, 53.0 -> 51.6 us), +0.7 % on the bench within one run (9 720 -> 9 786, four rounds each; gpurun_out/r5c).
    // Priority 1 does nearly the same (-2.1 %); raising it for the prologue as well loses the gain (53.4 us), for the tail instead
    // changes nothing, for the LayerNorm of the tail as well nothing.  The same device in k_ang's attention phase (+3 %), k_spa1's
    // LayerNorm (+2.4 %) and k_up's activation (0) did not pay and is not in those kernels.
    __builtin_amdgcn_s_setprio(3);
#pragma unroll                        // of[] needs compile-time indices (a runtime index would put it in scratch)
    for (int hg = 0; hg < 4; ++hg) {
        char* const buf = (hg & 1) ? bufB : bufA;
        if constexpr (TOKLM) {
            if (hg == 0) wait_vm_only<0>();                                    // (drained by the last query tile already)
            else if (hg < 3) wait_vm_only<kAdPerWave>();
            else wait_vm_only<kSpaBSlotsA * Ring::PIECES_PER_WAVE>();
        } else {
            if (hg == 0) { wait_vm_q<0>(qa, qb); params_store(lds_ln, 256, lnv); }
            else if (hg < 3) wait_vm_q<kAdPerWave>(qa, qb);
            else wait_vm_q<kSpaBSlotsA * Ring::PIECES_PER_WAVE>(qa, qb);
        }
        wg_barrier_keep_vm();                                              // everybody's pieces of this head pair have landed
        LFT_NOTE_ASM_("USE", kNoteKV, hg & 1);
        LFT_STAMP(18 + 2 * hg);
        raw16 qf[2];                                                       // per head: registers 0, 1 = block A's operand, 2, 3 = block B's
        if constexpr (TOKLM) { qf[0] = __builtin_bit_cast(raw16, qfr[2 * hg].v); qf[1] = __builtin_bit_cast(raw16, qfr[2 * hg + 1].v); }
        else { qf[0] = qa; qf[1] = qb; frag_to_blocks(qf[0]); frag_to_blocks(qf[1]); }
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            f32x4 S[2][4];
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x2 kf = *reinterpret_cast<const u32x2*>(buf + kb[hl ^ b] + (2 * t * kAttHC + 4 * b) * 64);
                    S[b][t] = mfma16k16(kf, u32x2{qf[hl][2 * b], qf[hl][2 * b + 1]}, bias[b][t], T());   // S^T[key, q] + mask bias
#pragma unroll
                    for (int e = 0; e < 4; ++e) { if (b == 0) m0 = fmaxf(m0, S[b][t][e]); else m1 = fmaxf(m1, S[b][t][e]); }
                }
            xrow_combine2(m0, m1, [](float a, float b) { return max_fast(a, b); });
            m0 = fmaxf(m0, -1.0e30f); m1 = fmaxf(m1, -1.0e30f);              // empty window: keep exp2(-inf - m) = 0, not NaN
            // this phase is bound by vector-instruction issue: subtract and sum as register pairs (v_pk_add_f32)
            float sum[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float m = b ? m1 : m0;
                const f32x2 mm = {m, m};
                f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        f32x2 d = f32x2{S[b][t][e], S[b][t][e + 1]} - mm;
                        d[0] = fast_exp2(d[0]); d[1] = fast_exp2(d[1]);
                        S[b][t][e] = d[0]; S[b][t][e + 1] = d[1];
#ifdef LFT_SPAB_VALU_SUM
                        sum2 += d;
#endif
                    }
                sum[b] = sum2[0] + sum2[1];
            }
#ifdef LFT_SPAB_VALU_SUM
            xrow_combine2(sum[0], sum[1], [](float a, float b) { return a + b; });
#endif
            typedef typename H16<T>::v4 V4;
            u32x2 ob[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#ifdef LFT_SPAB_VALU_SUM
                const float inv = sum[b] > 0.0f ? fast_rcp(sum[b]) : 0.0f;     // empty window (h < w quirk): 0, as the pinned reference
#else
                f32x4 osum[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#endif
                f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {                                  // O^T[d, q] += V^T P^T over key tiles 2 u, 2 u + 1 (k = 8 g + j: tile 2 u + (j >> 2), key 4 g + (j & 3))
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(buf + vb[hl ^ b] + (2 * (2 * u) * kAttHC + 4 * b) * 64));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(buf + vb[hl ^ b] + (2 * (2 * u + 1) * kAttHC + 4 * b) * 64));
                    const V8 vf = __builtin_bit_cast(V8, (short __attribute__((ext_vector_type(8)))){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
                    V8 pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pf[e] = (T)S[b][2 * u][e]; pf[4 + e] = (T)S[b][2 * u + 1][e]; }
                    o = mfma16k32(vf, pf, o);
#ifndef LFT_SPAB_VALU_SUM
                    {   // the softmax denominators from the matrix pipe: with an all-ones A operand every row of the product is the column
                        // sum of P^T, i.e. each lane gets its query's sum over the tile pair's 32 keys with no cross-lane step (-16 packed
                        // adds and one 7-instruction exchange per head for 4 small MFMAs; the sum is that of the ROUNDED probabilities, so
                        // the weights that multiply V add up to 1 exactly).  -DLFT_SPAB_VALU_SUM: the vector-unit form.
                        V8 ones;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
                        osum[b] = mfma16k32(ones, pf, osum[b]);
                    }
#endif
                }
#ifndef LFT_SPAB_VALU_SUM
                const float inv = osum[b][0] > 0.0f ? fast_rcp(osum[b][0]) : 0.0f;
#endif
                V4 oc;
#pragma unroll
                for (int e = 0; e < 4; ++e) oc[e] = (T)(o[e] * inv);
                ob[b] = __builtin_bit_cast(u32x2, oc);
            }
            // the two blocks' O^T tiles (row 4 g + e = label, column = query) -> ONE 32-column fragment: out_proj's k-step 2 hg + hl,
            // element (h, j) = the head's label 8 h + j (the order V's channels have in the LDS tile)
            raw16 fo = raw16{ob[0][0], ob[0][1], ob[1][0], ob[1][1]};
            frag_to_blocks(fo);                                                // the same row exchange, read the other way round
            of[2 * hg + hl].v = __builtin_bit_cast(V8, fo);
            // (No scheduling fence between the two heads of a pair any more: with 32 score registers per head instead of 48 the
            // scheduler may overlap one head's MFMAs with the other's softmax without spilling -- measured +1 % on the bench.
            // -DLFT_SPAB_HEAD_FENCE restores it for experiments.)
#ifdef LFT_SPAB_HEAD_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        LFT_STAMP(19 + 2 * hg);
        LFT_NOTE_ASM_("DONE", kNoteKV, hg & 1);
        wg_barrier_keep_vm();                                              // every wave is done reading this buffer
        if constexpr (!TOKLM) {
            if (hg < 3) q_load_async(qptr + kQPair * (hg + 1), qptr + kQPair * (hg + 1) + kQHead, qa, qb);
        }
        if (hg < 2) stage(hg + 2, buf);
        else if (hg == 2) {                                                // buffer A now belongs to the weight ring
            LFT_NOTE_ASM_("ALIAS_LT", kNoteWRingPipe, kSpaBSlotsA);        // (for the static check: ring slots below this one lie in K/V buffer 0,
#pragma unroll                                                             //  the others in buffer 1)
            for (int c = 0; c < kSpaBSlotsA; ++c) ring.issue(c);
        } else {                                                           // ... and so does buffer B
#pragma unroll
            for (int c = kSpaBSlotsA; c < kSpaBSlots; ++c) ring.issue(c);
        }
    }
    char* scr = bufB + kSpaBScratchOfs + wave * TileIO<4, T>::BYTES;
    f32x16 t[4], n[4];
    if constexpr (TOKLM) {
        // Row rr of this wave's 8 x 4 block = lanes 8 wave .. 8 wave + 7 of the lane-major tile of image row y0 + rr: the 16-byte
        // piece [k-step][32 hh + column] holds exactly registers 8k .. 8k+7 of this lane's accumulator tile (load_tile_lm).
        // The pieces were loaded for the queries in the prologue and kept packed (32 registers) through the attention phase:
        // a second read of the token tile would come from beyond the L2 again (26 MB per launch by the PMC counters).
#pragma unroll
        for (int kidx = 0; kidx < 8; ++kidx) {
            const V8 v = __builtin_bit_cast(V8, tokr[kidx]);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[kidx >> 1][8 * (kidx & 1) + j] = (float)v[j];
        }
    } else {
        BlkRows rt = rows; rt.img_row_bytes = w * 256; rt.tok_bytes = 256;
        load_tile_map<4, T>(TOK + tok0 * 128, rt, lane, t, scr);
    }
    LFT_STAMP(26);
    __builtin_amdgcn_s_setprio(0);                                    // (see the attention loop)
    ring.start();
    linear_ring_at<0, 4, 8>(ring, of, t);                             // t = tok + O Wo^T
    LFT_STAMP(27);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) n[nt] = t[nt];
    layernorm_acc<4, true>(n, lds_ln, lds_ln + 128, hh, bad);                  // t carries whatever overflowed in tok / Q / K / V / the attention
    Frag<T> f[8];
    acc_frags<4, T>(n, f);
    auto ffn_chunk = [&](auto cc) __attribute__((always_inline)) {                                   // hidden units 64 c .. 64 c + 63: stream positions 32 + 32 c ..
        constexpr int C = decltype(cc)::value;
        f32x16 hid[2];
        zero_acc<2>(hid);
        linear_ring_at<32 + 32 * C, 2, 8>(ring, f, hid);
        Frag<T> hf[4];
        acc_frags_relu<2>(hid, hf);
        linear_ring_at<32 + 32 * C + 16, 4, 4>(ring, hf, t);
    };
    ffn_chunk(std::integral_constant<int, 0>());
    ffn_chunk(std::integral_constant<int, 1>());
    ffn_chunk(std::integral_constant<int, 2>());
    ffn_chunk(std::integral_constant<int, 3>());
    LFT_STAMP(28);
    acc_frags<4, T>(t, f);
    f32x16 y[2];
    zero_acc<2>(y);
    linear_ring_at<160, 2, 8>(ring, f, y);
    LFT_STAMP(29);
    BlkRows ry = rows; ry.img_row_bytes = w * 128; ry.tok_bytes = 128;
    if (SKIP) {
        f32x16 sk[2];
        load_tile_map<2, T>(skip + tok0 * 64, ry, lane, sk, scr);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) y[nt] += sk[nt];
    }
    if constexpr (YLM) {                                              // the mirror image of the TOKLM load: four 16-byte stores per lane, no LDS
        if (qy < h) {
            T* dst = Y + (img0 + (long long)qy * w + x0) * 64 + (32 * hh + bxl + qpx) * 8;
#pragma unroll
            for (int kidx = 0; kidx < 4; ++kidx) {
                V8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (T)y[kidx >> 1][8 * (kidx & 1) + j];
                store_raw16(reinterpret_cast<char*>(dst + kidx * 512), __builtin_bit_cast(raw16, v));
            }
        }
    } else {
        store_tile_map<2, T>(Y + tok0 * 64, ry, lane, y, scr);
    }
    LFT_STAMP(30);
    publish_status(status, bad);
}

// ------------------------------------------------------------------------------------------
// Up-sampler, fused per LR token (reference LFT.py:39-44): never materialises the [B,64,A*h*s,A*w*s] map.
//   U = Wu x (64 s^2 values) -> LeakyReLU(0.2) -> these are the 64 x s x s HR features of this LR pixel
//   G = M lrelu(U): the token's contribution to the final 3x3 convolution on the (s+2)x(s+2) HR
//       neighbourhood of its s x s block ("overlap-add"; M built from upsampling.3.weight by k_pack/UPM).
// U is produced 32 rows at a time and immediately contracted into G.
// Stream per chunk c: Wu[1x4] M[GT x 2].  G rows >= (s+2)^2 are padding.
// ------------------------------------------------------------------------------------------
template <typename T, int GT, bool XLM = false, int NW = kNwUp>   // XLM: input tile in lane-major form (producer: the last k_spa2)
__global__ __launch_bounds__(64 * NW) void k_up(const T* __restrict__ X, const T* __restrict__ ws, float* __restrict__ G,
                                            long long ntok, int nchunk, int gp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const long long tok_raw = ((long long)blockIdx.x * NW + (threadIdx.x >> 6)) * 32 + r;
    const bool ok = tok_raw < ntok;
    const long long tok = ok ? tok_raw : ntok - 1;
    const long long t0 = tok_raw - r;
    const int nvalid = (int)max(0LL, min(32LL, ntok - t0));
    char* scr = smem + WRing<T, kUpChunk, NW>::LDS_BYTES + (threadIdx.x >> 6) * TileIO<2, T>::BYTES;
    WRing<T, kUpChunk, NW> ring;
    ring.init(ws, smem, nchunk * (4 + 2 * GT));         // first: the weight DMA is in flight while the tile is fetched
    f32x16 x[2];
    if constexpr (XLM) load_tile_lm<2, T>(X + t0 * 64, lane, x);
    else load_tile<2, T>(X + min(t0, ntok - 1) * 64, nvalid, lane, x, scr);
    Frag<T> xf[4];
    acc_frags<2, T>(x, xf);
    f32x16 g[GT];
    zero_acc<GT>(g);
#pragma unroll 2
    for (int c = 0; c < nchunk; ++c) {
        f32x16 u[1];
        zero_acc<1>(u);
        linear_ring<1, 4, T>(ring, xf, u);
#pragma unroll
        for (int i = 0; i < 16; ++i) u[0][i] = lrelu02_fast(u[0][i]);
        Frag<T> uf[2];
        acc_frags<1, T>(u, uf);
        linear_ring<GT, 2, T>(ring, uf, g);
    }
    if (!ok) return;
#pragma unroll
    for (int nt = 0; nt < GT; ++nt)
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int base = 32 * nt + 8 * grp + 4 * hh;
            if (base < gp)
                store4(G + tok * gp + base, f32x4{g[nt][4 * grp], g[nt][4 * grp + 1], g[nt][4 * grp + 2], g[nt][4 * grp + 3]});
        }
}

// ------------------------------------------------------------------------------------------
// Assemble (reference LFT.py:43 border handling, :81 skip add, :255-266 bicubic):
//   out[b, Y, X] = bicubic(lr)[Y, X] + sum over the <= 4 LR mosaic pixels whose (s+2)^2 footprint covers (Y, X)
// The 3x3 conv runs over the whole mosaic: footprints cross view borders, only the outer mosaic border pads.
// Bicubic is per view: source = (dst + 0.5)/s - 0.5, taps floor-1..floor+2 clamped to the view, Keys A = -0.75
// (torch upsample_bicubic2d, align_corners=False).
// ------------------------------------------------------------------------------------------
LFT_DEV void cubic_coef(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
LFT_DEV float bicubic_at(const float* __restrict__ view, int stride, int h, int w, int Y, int X, int s) {
    const float sy = ((float)Y + 0.5f) / (float)s - 0.5f, sx = ((float)X + 0.5f) / (float)s - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float cy[4], cx[4];
    cubic_coef(sy - fy, cy);
    cubic_coef(sx - fx, cx);
    const int iy = (int)fy, ix = (int)fx;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), h - 1);
        float row = 0.0f;
#pragma unroll
        for (int b = 0; b < 4; ++b) row += cx[b] * view[(size_t)yy * stride + min(max(ix - 1 + b, 0), w - 1)];
        acc += cy[a] * row;
    }
    return acc;
}
// The per-view bicubic skip on its own (lft_bicubic_fwd: unit tests of reference LFT.py:255-266).
__global__ __launch_bounds__(256) void k_bicubic(const float* __restrict__ lr, float* __restrict__ out, int B, int A, int h, int w, int s) {
    // grid: x = 256-pixel column groups, y = HR mosaic row, z = batch
    const int HR_H = A * h * s, HR_W = A * w * s;
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y, b = blockIdx.z;
    if (X >= HR_W) return;
    const int a1 = Y / (h * s), a2 = X / (w * s);
    const float* view = lr + (size_t)b * (A * h) * (A * w) + (size_t)(a1 * h) * (A * w) + a2 * w;
    out[((size_t)b * HR_H + Y) * HR_W + X] = bicubic_at(view, A * w, h, w, Y - a1 * h * s, X - a2 * w * s, s);
}

// The gather is tiled: a workgroup owns an 8 x 8 block of LR mosaic pixels =
// an 8S x 8S block of output pixels, and stages the (8+2)^2 footprints it can touch in LDS with coalesced 16-byte
// loads (a footprint is (S+2)^2 contiguous floats) -- every footprint is fetched once per block instead of once per
// output row that needs it, and the 4-float groups read by neighbouring lanes fall on distinct banks (row stride
// (S+2)^2 = 36 or 16 floats).  The bicubic taps come from the LR mosaic, which is cache-resident (0.1 MB per patch).
template <int S>
__global__ __launch_bounds__(256) void k_assemble_t(const float* __restrict__ lr, const float* __restrict__ G, float* __restrict__ out,
                                                    int B, int A, int h, int w, int gld, unsigned* __restrict__ status) {      // gld: floats between the footprints of consecutive tokens (>= GP)
    constexpr int TL = 8, HL = TL + 2, GP = (S + 2) * (S + 2), P4 = GP / 4, TS = TL * S;
    __shared__ __attribute__((aligned(16))) float gs[HL * HL * GP];
    const int MH = A * h, MW = A * w, HR_W = MW * S, HR_H = MH * S;
    // 1-D grid, XCD-aware: neighbouring blocks stage 36 of each other's 100 footprints -- with the round-robin placement every one of
    // those came from beyond an L2 once per block that touched it (34.8 MB per launch for 21.7 MB of contract, PMC)
    const int tiles_x = (MW + TL - 1) / TL, tiles_y = (MH + TL - 1) / TL;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int bx0 = (bid % tiles_x) * TL, by0 = ((bid / tiles_x) % tiles_y) * TL, b = bid / (tiles_x * tiles_y);
    const int hw = h * w, V = A * A;
    const float* Gb = G + (size_t)b * V * hw * gld;
    {   // every piece of the staging is requested before the first one is written to LDS (branch-free, from clamped addresses): one
        // memory round trip per workgroup -- as a load / store loop with a conditional load it was four dependent ones
        constexpr int NPC = HL * HL * P4, NIT = (NPC + 255) / 256;
        f32x4 vv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pidx = min((int)threadIdx.x + 256 * i, NPC - 1);
            const int slot = pidx / P4, part = pidx - slot * P4;
            const int by = min(max(by0 - 1 + slot / HL, 0), MH - 1), bx = min(max(bx0 - 1 + slot % HL, 0), MW - 1);
            const int vy = by / h, py = by - vy * h, vx = bx / w, px = bx - vx * w;
            vv[i] = load4(Gb + ((size_t)(vy * A + vx) * hw + py * w + px) * gld + part * 4);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pidx = (int)threadIdx.x + 256 * i;
            const int slot = pidx / P4, part = pidx - slot * P4;
            const int by = by0 - 1 + slot / HL, bx = bx0 - 1 + slot % HL;
            const bool in = by >= 0 && by < MH && bx >= 0 && bx < MW;  // outside the mosaic: zero padding of the final conv (LFT.py:43)
            if (pidx < NPC) *reinterpret_cast<f32x4*>(gs + slot * GP + part * 4) = in ? vv[i] : f32x4{0, 0, 0, 0};
        }
    }
    // The bicubic taps of the block: when the views are whole numbers of blocks (h, w multiples of 8 -- every BASELINE shape) the block
    // lies inside ONE view and its taps are the 12 x 12 LR pixels around it, clamped to that view: staged once per workgroup (they were
    // 16 cache-line-scattered global loads per output pixel), already clamped, so a tap is one LDS read.  A thread's pixels (rows 256 / TS
    // apart, same column) share their sub-pixel phase (Y mod S, X mod S): the cubic weights are computed once per thread.  Same values,
    // same order of additions as bicubic_at: bit-identical.  Other shapes take bicubic_at per pixel.
    constexpr int LT = TL + 4;
    __shared__ float lt[LT * LT];
    const bool one_view = (h % TL == 0) && (w % TL == 0);
    if (one_view && threadIdx.x < LT * LT) {
        const int r = threadIdx.x / LT, c = threadIdx.x - r * LT;
        const int vy0 = (by0 / h) * h, vx0 = (bx0 / w) * w;
        const int yy = min(max(by0 - 2 + r, vy0), vy0 + h - 1), xx = min(max(bx0 - 2 + c, vx0), vx0 + w - 1);
        lt[threadIdx.x] = lr[(size_t)b * MH * MW + (size_t)yy * MW + xx];
    }
    __syncthreads();
    unsigned bad = 0;                                                  // the network's output is the last place an overflow can show
    float cy[4] = {0, 0, 0, 0}, cx[4] = {0, 0, 0, 0};
    int ry = 0, rx = 0;                                                // tile row / column of the first tap, relative to the pixel's LR cell
    if (one_view) {
        const int Yl0 = threadIdx.x / TS, Xl0 = threadIdx.x - Yl0 * TS;               // (TS * TS is a multiple of 256 or equal to it: every thread has a first pixel)
        const float sy = ((float)Yl0 + 0.5f) / (float)S - 0.5f, sx = ((float)Xl0 + 0.5f) / (float)S - 0.5f;   // S is a power of two: exact, and so is the
        const float fy = floorf(sy), fx = floorf(sx);                                                        // fraction, whichever multiple of S is added to Y
        cubic_coef(sy - fy, cy);
        cubic_coef(sx - fx, cx);
        ry = (int)fy - Yl0 / S;                                        // -1 or 0: floor(src) relative to the LR cell of the pixel
        rx = (int)fx - Xl0 / S;
    }
    for (int idx = threadIdx.x; idx < TS * TS; idx += 256) {
        const int Yl = idx / TS, Xl = idx - Yl * TS;
        const int Y = by0 * S + Yl, X = bx0 * S + Xl;
        if (Y >= HR_H || X >= HR_W) continue;
        float v;
        if (one_view) {
            // LR cell (Yl / S, Xl / S) of the block = tile (2 + Yl / S, 2 + Xl / S); taps floor - 1 .. floor + 2
            const float* t0 = lt + (2 + Yl / S + ry - 1) * LT + (2 + Xl / S + rx - 1);
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float row = 0.0f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) row += cx[bb] * t0[a * LT + bb];
                acc += cy[a] * row;
            }
            v = acc;
        } else {
            const int a1 = Y / (h * S), a2 = X / (w * S);
            const float* view = lr + (size_t)b * MH * MW + (size_t)(a1 * h) * MW + a2 * w;
            v = bicubic_at(view, MW, h, w, Y - a1 * h * S, X - a2 * w * S, S);
        }
        const int ql = Yl / S + 1, qc = Xl / S + 1, i = Yl % S, j = Xl % S;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int I = i - S * dy;
            if (I < -1 || I > S) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int J = j - S * dx;
                if (J < -1 || J > S) continue;
                v += gs[((ql + dy) * HL + qc + dx) * GP + (I + 1) * (S + 2) + (J + 1)];
            }
        }
        bad |= not_finite(v);
        out[((size_t)b * HR_H + Y) * HR_W + X] = v;
    }
    publish_status(status, bad);
}

// ------------------------------------------------------------------------------------------
// Scene tiling around the hot path (reference utils/utils.py:91-157, driven by test.py:83-101): a whole scene is
// cut into overlapping A*patch x A*patch mosaics on the GPU, super-resolved as ONE batch, and re-assembled from
// the central stride*s region of every SR patch -- instead of the reference's batch-1 Python double loop.
// Both kernels are pure gathers (one thread per output element), HBM-bound.
// ------------------------------------------------------------------------------------------
LFT_DEV int reflect_idx(int i, int n) {             // symmetric extension, edge repeated (ImageExtend, utils.py:126-138)
    i = i < 0 ? -i - 1 : i;
    return i >= n ? 2 * n - 1 - i : i;
}
__global__ __launch_bounds__(256) void k_scene_divide(const float* __restrict__ scene, float* __restrict__ patches,
                                                      int A, int h0, int w0, int patch, int stride, int nv) {
    // grid: x over the A*patch columns, y = row of the patch mosaic, z = patch index ku*nv + kv
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y, n = blockIdx.z;
    const int P = A * patch;
    if (X >= P) return;
    const int bdr = (patch - stride) / 2, ku = n / nv, kv = n - ku * nv;
    const int u = Y / patch, y = Y - u * patch, v = X / patch, x = X - v * patch;
    const int ey = ku * stride + y, ex = kv * stride + x;
    float val = 0.0f;                                   // beyond the extended view: zero fill (utils.py:109-113)
    if (ey < h0 + 2 * bdr && ex < w0 + 2 * bdr)
        val = scene[(size_t)(u * h0 + reflect_idx(ey - bdr, h0)) * (A * w0) + v * w0 + reflect_idx(ex - bdr, w0)];
    patches[((size_t)n * P + Y) * P + X] = val;
}
__global__ __launch_bounds__(256) void k_scene_integrate(const float* __restrict__ sub, float* __restrict__ out,
                                                         int A, int pz, int stride, int h0, int w0, int nv) {
    // grid: x over the A*w0 columns of the SR scene mosaic, y = its rows; pz, stride, h0, w0 are in SR pixels
    const int Xm = blockIdx.x * 256 + threadIdx.x, Ym = blockIdx.y;
    if (Xm >= A * w0) return;
    const int bdr = (pz - stride) / 2;
    const int u = Ym / h0, Y = Ym - u * h0, v = Xm / w0, X = Xm - v * w0;
    const int ku = Y / stride, i = Y - ku * stride, kv = X / stride, j = X - kv * stride;
    const int P = A * pz;
    out[(size_t)Ym * (A * w0) + Xm] = sub[((size_t)(ku * nv + kv) * P + u * pz + bdr + i) * P + v * pz + bdr + j];
}

// ------------------------------------------------------------------------------------------
// MFMA layout self-test: C[32x32] = A[32x16] * B[16x32] through the same fragment helpers the
// kernels use (natural k order), and the acc-order re-use path D = W2 * C.  Checked from Python
// with asymmetric integer data (guide rule: never validate a layout with symmetric operands).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_selftest(const float* __restrict__ Am, const float* __restrict__ Bm,
                                                 const float* __restrict__ W2, float* __restrict__ Cout, float* __restrict__ Dout) {
    const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
    Frag<T> a, b;
    float ta[8], tb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ta[j] = Am[r * 16 + 8 * hh + j]; tb[j] = Bm[(8 * hh + j) * 32 + r]; }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a.lo[j] = ta[j]; a.hi[j] = ta[4 + j]; b.lo[j] = tb[j]; b.hi[j] = tb[4 + j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a.v[j] = (T)ta[j]; b.v[j] = (T)tb[j]; }
    }
    f32x16 c[1];
    zero_acc<1>(c);
    mma(a, b, c[0]);
#pragma unroll
    for (int i = 0; i < 16; ++i) Cout[acc_row(i, hh) * 32 + r] = c[0][i];
    // D[32x32] = W2[32x32] * C, with W2 fragments built in acc order on the fly
    Frag<T> cf[2];
    acc_frags<1, T>(c, cf);
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Frag<T> wf;
        float tw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tw[j] = W2[r * 32 + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { wf.lo[j] = tw[j]; wf.hi[j] = tw[4 + j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) wf.v[j] = (T)tw[j];
        }
        mma(wf, cf[s], d);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Dout[acc_row(i, hh) * 32 + r] = d[i];
}
