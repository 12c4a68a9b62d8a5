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
// Stream: conv[36 x 4] Wv[4x8] Wq[4x8] Wk[4x8]  (240 fragments).
// PE_ONLY: embed the position image itself and write the tokens (pack-time precompute).
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
template <typename T, bool PE_ONLY, int CH = kSpaChunk, bool TOKLM = false>   // TOKLM: the token tile goes to k_spa2 in lane-major tile format
__global__ __launch_bounds__(256, LFT_SPA_OCC) void k_spa1(const T* __restrict__ X, const T* __restrict__ ws,
                                              const float* __restrict__ ln, const T* __restrict__ petok,
                                              T* __restrict__ TOK, T* __restrict__ Q, T* __restrict__ K, T* __restrict__ Vv,
                                              T* __restrict__ pe_out, int nimg, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5, wave = threadIdx.x >> 6;
    const int hw = h * w, tpi = (hw + 127) >> 7;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);                  // neighbouring tiles (shared halo rows) on one XCD
    const int im = bid / tpi, p0 = (bid % tpi) * 128;
    const int tl = wave * 32 + r, p = p0 + tl;
    const bool ok = p < hw;
    const int pc = min(p, hw - 1);
    LFT_STAMP(0);
    typename RawPiece<T>::type pe_raw[16];                                    // position tokens of this lane's token, kept packed
    if (!PE_ONLY) load_lane_major_raw<4, T>(petok + (size_t)((p0 >> 5) + wave) * 4096, lane, pe_raw);   // early, 8 coalesced loads
    char* lds_in = smem + WRing<T, CH>::LDS_BYTES;
    float* lds_ln = reinterpret_cast<float*>(lds_in + ConvIn<T>::bytes(w));
    raw16 lnv = raw16{0u, 0u, 0u, 0u};
    if (!PE_ONLY) lnv = params_load(ln, 256);                         // norm.{weight,bias}; ln is null in the pack-time PE_ONLY launch
    WRing<T, CH> ring;
    ring.init(ws, smem, PE_ONLY ? 144 : 240, p0 + 128 <= hw);
    stage_conv_input<T>(X + (size_t)im * hw * 64, p0, hw, w, lds_in);
    LFT_STAMP(12);
    wait_staged();
    LFT_STAMP(13);
    if (!PE_ONLY) params_store(lds_ln, 256, lnv);
    __syncthreads();                                                  // input tile, first weight chunks and LN parameters published
    LFT_STAMP(1);
    f32x16 t[4];
    zero_acc<4>(t);
    conv3x3_tile<4, T>(lds_in, tl, p / w, p % w, ok, h, w, hh, ring, t);
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
    if constexpr (TOKLM) ring.note_vm(store_tile_lm<4, T>(TOK + ((size_t)im * hw + t0) * 128, lane, t));   // hw % 128 == 0: every tile is full
    else ring.note_vm(store_tile<4, T>(TOK + tile_off, nvalid, lane, t, scr));
    LFT_STAMP(4);
    Frag<T> nf[8];
    // Each projection is produced and stored in two 64-channel halves: 32 accumulator registers instead of 64
    // keep the kernel inside 256 VGPRs (2 waves/SIMD) without scratch spills -- a spill reload forces
    // s_waitcnt vmcnt(0), which drains the weight DMA and every output store in flight.
    acc_frags<4, T>(t, nf);                    // V = tok Wv^T first (raw tokens, reference LFT.py:185); tok is then normalised in place
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 a[2];
        zero_acc<2>(a);
        linear_ring<2, 8, T>(ring, nf, a);
        ring.note_vm(store_tile<2, T, 128>(Vv + tile_off + 64 * half, nvalid, lane, a, scr));
    }
    LFT_STAMP(6);
    add_acc_raw<4, T>(t, pe_raw, ok);
    layernorm_acc<4>(t, lds_ln, lds_ln + 128, hh);
    acc_frags<4, T>(t, nf);
    LFT_STAMP(7);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 a[2];
        zero_acc<2>(a);
        linear_ring<2, 8, T>(ring, nf, a);
        ring.note_vm(store_tile<2, T, 128>(Q + tile_off + 64 * half, nvalid, lane, a, scr));
    }
    LFT_STAMP(9);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 a[2];
        zero_acc<2>(a);
        linear_ring<2, 8, T>(ring, nf, a);
        ring.note_vm(store_tile<2, T, 128>(K + tile_off + 64 * half, nvalid, lane, a, scr));
    }
    LFT_STAMP(11);
}

// ------------------------------------------------------------------------------------------
// SpaTrans windowed attention core (reference LFT.py:147-162 mask + :183-187 attention): keys = clamped 5x5 window around
// the query.  The reference bounds the window columns by min(h, x+3) (LFT.py:155, "h" where "w" is meant) and slicing
// clips at w; reproduced as-is: for h < w some queries see no key at all and get a zero attention output (what the
// reference gives under torch >= 2.5).  Q is pre-scaled by scale*log2(e); softmax uses exp2.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
LFT_DEV void load_pairs16(const bf16_t* p, bf16x2 (&o)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p), b = *reinterpret_cast<const bf16x8*>(p + 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = bf16x2{a[2 * i], a[2 * i + 1]}; o[4 + i] = bf16x2{b[2 * i], b[2 * i + 1]}; }
}
// LDS-tiled bf16 windowed attention on the VALU (v_dot2c): superseded by the MFMA kernel below, kept for A/B builds
// (-DLFT_ATT_VALU).  The fp32 path uses the LDS-tiled kernel of lft_train.cuh (k_win_attn_lds<0, true>).
// A 512-thread workgroup owns a 4 x 32 tile of queries of one view image and 4 of the 8 heads (64 channels).
// The (4+4) x (32+4) halo tile of K -- then of V, re-using the same 41 KiB -- is staged once into LDS (every key is
// used by up to 25 queries x 4 heads), token rows padded to 144 B so the 16-byte reads of a half-wave (32 different
// query columns) fall on distinct banks.  wave = (head, 2 query rows), lane = (row parity, column).
#ifndef LFT_ATT_TY
#define LFT_ATT_TY 4
#endif
constexpr int kAttTY = LFT_ATT_TY, kAttTX = 32, kAttHR = kAttTY + 4, kAttHC = kAttTX + 4, kAttRow = 144;
constexpr int kAttThreads = kAttTY * kAttTX * 4;      // one thread per (query, head) for 4 heads
constexpr int kAttLds = kAttHR * kAttHC * kAttRow;
LFT_DEV void att_stage(const bf16_t* __restrict__ src, char* lds, long long img_tok0, int ty, int tx, int hg, int h, int w) {
    constexpr int N = kAttHR * kAttHC * 8, ITER = (N + kAttThreads - 1) / kAttThreads;   // 16-byte pieces of the halo tile
    raw16 v[ITER];
#pragma unroll
    for (int u = 0; u < ITER; ++u) {                                       // all loads first (branch-free), then all stores
        const int idx = min((int)threadIdx.x + kAttThreads * u, N - 1);
        const int slot = idx >> 3, piece = idx & 7;
        const int gy = ty * kAttTY - 2 + slot / kAttHC, gx = tx * kAttTX - 2 + slot % kAttHC;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        const long long t = in ? img_tok0 + gy * w + gx : img_tok0;
        const raw16 r = load_raw16(reinterpret_cast<const char*>(src + t * 128 + hg * 64) + piece * 16);
        v[u] = in ? r : raw16{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
        const int idx = (int)threadIdx.x + kAttThreads * u;
        if (idx < N) store_raw16(lds + (idx >> 3) * kAttRow + (idx & 7) * 16, v[u]);
    }
}
LFT_DEV void lds_pairs16(const char* p, bf16x2 (&o)[8]) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, load_raw16(p)), b = __builtin_bit_cast(bf16x8, load_raw16(p + 16));
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = bf16x2{a[2 * i], a[2 * i + 1]}; o[4 + i] = bf16x2{b[2 * i], b[2 * i + 1]}; }
}
__global__ __launch_bounds__(kAttThreads) void k_spa_attn_lds(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                      const bf16_t* __restrict__ Vv, bf16_t* __restrict__ O, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_x = (w + kAttTX - 1) / kAttTX, tiles_y = (h + kAttTY - 1) / kAttTY;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, im = blockIdx.x / (tiles_x * tiles_y);
    const int hg = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = wave & 3, qrow = (wave >> 2) * 2 + (lane >> 5), qcol = lane & 31;
    const int y = ty * kAttTY + qrow, x = tx * kAttTX + qcol;
    const bool valid = y < h && x < w;
    const long long img0 = (long long)im * h * w;
    const long long tok = img0 + min(y, h - 1) * w + min(x, w - 1);
    const int y0 = max(0, y - 2), y1 = min(h, y + 3), x0 = max(0, x - 2), x1 = min(min(h, x + 3), w);   // reference LFT.py:155 (sic)
#ifdef LFT_ATT_FMA
    // Experiment (tools/ab_build.py fma:-DLFT_ATT_FMA): unpack bf16 by shift / mask and use plain fp32 FMAs.
    // Measured 102 us vs 64 us for the v_dot2c_f32_bf16 form below: the kernel is VALU-issue-bound, fewer instructions win.
    float q[16];
    {
        bf16x2 qp[8];
        load_pairs16(Q + tok * 128 + hg * 64 + hl * 16, qp);
#pragma unroll
        for (int c = 0; c < 8; ++c) { q[2 * c] = (float)qp[c][0]; q[2 * c + 1] = (float)qp[c][1]; }
    }
    att_stage(K, smem, img0, ty, tx, hg, h, w);
    __syncthreads();
    float s[25];
    float m = -INFINITY;
    const char* base = smem + (qrow * kAttHC + qcol) * kAttRow + hl * 32;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const int ky = y - 2 + t / 5, kx = x - 2 + t % 5;
        const raw16 k0 = load_raw16(base + ((t / 5) * kAttHC + t % 5) * kAttRow), k1 = load_raw16(base + ((t / 5) * kAttHC + t % 5) * kAttRow + 16);
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d += q[2 * c] * __builtin_bit_cast(float, k0[c] << 16) + q[2 * c + 1] * __builtin_bit_cast(float, k0[c] & 0xffff0000u);
            d += q[8 + 2 * c] * __builtin_bit_cast(float, k1[c] << 16) + q[8 + 2 * c + 1] * __builtin_bit_cast(float, k1[c] & 0xffff0000u);
        }
        s[t] = (ky >= y0 && ky < y1 && kx >= x0 && kx < x1) ? d : -INFINITY;
        m = fmaxf(m, s[t]);
    }
    __syncthreads();                       // everyone is done with K
    att_stage(Vv, smem, img0, ty, tx, hg, h, w);
    __syncthreads();
    float sum = 0.0f, o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = 0.0f;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        float pr = (s[t] != -INFINITY) ? fast_exp2(s[t] - m) : 0.0f;
        sum += pr;
        pr = (float)(bf16_t)pr;            // P rounded to bf16, as an MFMA operand would be
        const raw16 v0 = load_raw16(base + ((t / 5) * kAttHC + t % 5) * kAttRow), v1 = load_raw16(base + ((t / 5) * kAttHC + t % 5) * kAttRow + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o[2 * c] += pr * __builtin_bit_cast(float, v0[c] << 16);
            o[2 * c + 1] += pr * __builtin_bit_cast(float, v0[c] & 0xffff0000u);
            o[8 + 2 * c] += pr * __builtin_bit_cast(float, v1[c] << 16);
            o[8 + 2 * c + 1] += pr * __builtin_bit_cast(float, v1[c] & 0xffff0000u);
        }
    }
#else
    bf16x2 q[8], kv[8];
    load_pairs16(Q + tok * 128 + hg * 64 + hl * 16, q);
    att_stage(K, smem, img0, ty, tx, hg, h, w);
    __syncthreads();
    float s[25];
    float m = -INFINITY;
    const char* base = smem + (qrow * kAttHC + qcol) * kAttRow + hl * 32;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const int ky = y - 2 + t / 5, kx = x - 2 + t % 5;
        lds_pairs16(base + ((t / 5) * kAttHC + t % 5) * kAttRow, kv);
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) d = __builtin_amdgcn_fdot2_f32_bf16(q[c], kv[c], d, false);
        s[t] = (ky >= y0 && ky < y1 && kx >= x0 && kx < x1) ? d : -INFINITY;
        m = fmaxf(m, s[t]);
    }
    __syncthreads();                       // everyone is done with K
    att_stage(Vv, smem, img0, ty, tx, hg, h, w);
    __syncthreads();
    float sum = 0.0f, o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = 0.0f;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const float pr = (s[t] != -INFINITY) ? fast_exp2(s[t] - m) : 0.0f;
        sum += pr;
        const bf16_t pb = (bf16_t)pr;
        const bf16x2 p0 = bf16x2{pb, (bf16_t)0.0f}, p1 = bf16x2{(bf16_t)0.0f, pb};
        lds_pairs16(base + ((t / 5) * kAttHC + t % 5) * kAttRow, kv);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            o[2 * c] = __builtin_amdgcn_fdot2_f32_bf16(p0, kv[c], o[2 * c], false);
            o[2 * c + 1] = __builtin_amdgcn_fdot2_f32_bf16(p1, kv[c], o[2 * c + 1], false);
        }
    }
#endif
    if (!valid) return;
    const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;      // empty window (h < w quirk) -> 0, see the generic kernel
#pragma unroll
    for (int g = 0; g < 4; ++g)
        store4(O + tok * 128 + hg * 64 + hl * 16 + 4 * g, f32x4{o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv});
}

// ------------------------------------------------------------------------------------------
// MFMA windowed attention (bf16 production path; same maths as the two kernels above).
// A workgroup = a 4 x 32 tile of queries of one view image; wave c owns the 8 x 4 block of columns 8c..8c+7
// (32 queries on its 32 MFMA columns).  The block's clamped 5x5 windows live inside a 12 x 8 neighbourhood =
// 96 keys = three 32-key tiles.  Per pair of heads (32 channels) the K and V halo tiles (8 x 36 tokens) are
// staged in LDS once for the four waves, then per head
//   S^T[key, q]  = K_tile . Q^T          3 MFMAs (head_dim 16 = exactly one k-step); the accumulator is
//                                        initialised with a 0 / -inf bias that encodes window, image border
//                                        and the reference's min(h, x+3) column clamp (LFT.py:155)
//   softmax over the 96 rows             in registers + one lane^32 exchange (raw v_exp_f32)
//   O^T[d, q]   += V^T_tile . P^T        6 MFMAs; V^T operand fragments come straight out of the row-major LDS
//                                        tile through ds_read_b64_tr_b16 (hardware transposing read)
// i.e. the dot products run on the matrix pipe and the VALU only does the softmax: about half the VALU
// instructions per query of the dot2c kernel above, which was VALU-issue-bound.
// ------------------------------------------------------------------------------------------
constexpr int kAmRow = 80;                                   // 32 channels (2 heads) x bf16 + 16 B pad per staged token
constexpr int kAmSlots = kAttHR * kAttHC;                    // 8 x 36 halo tokens (kAttTY = 4, kAttTX = 32)
constexpr int kAmLds = 2 * kAmSlots * kAmRow;                // K tile + V tile
static_assert(LFT_ATT_TY == 4, "k_spa_attn_mfma is written for 4-row tiles");
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Per-thread staging plan of the 8 x 36 x (2 heads) halo tile: which global token / LDS slot each of the thread's
// 16-byte pieces touches.  Independent of the head pair, so it is computed once per workgroup tile.
struct AmPlan {
    static constexpr int N = kAmSlots * 4, ITER = (N + 255) / 256;      // 16-byte pieces: 4 per token
    long long gofs[ITER];                                               // element offset of the piece in Q/K/V ([tok][128]), head pair 0
    int lofs[ITER];                                                     // byte offset in the LDS tile (or -1: no store)
    unsigned inmask;                                                    // bit u: the token exists in the image
};
LFT_DEV void am_plan(AmPlan& p, long long img0, int y0, int x0, int h, int w) {
    p.inmask = 0;
#pragma unroll
    for (int u = 0; u < AmPlan::ITER; ++u) {
        const int raw = (int)threadIdx.x + 256 * u, idx = min(raw, AmPlan::N - 1);
        const int slot = idx >> 2, piece = idx & 3;
        const int gy = y0 - 2 + slot / kAttHC, gx = x0 - 2 + slot % kAttHC;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        p.gofs[u] = (in ? img0 + gy * w + gx : img0) * 128 + piece * 8;
        p.lofs[u] = raw < AmPlan::N ? slot * kAmRow + piece * 16 : -1;
        p.inmask |= (in ? 1u : 0u) << u;
    }
}
LFT_DEV void am_load(const AmPlan& p, const bf16_t* __restrict__ src, int hg, raw16 (&v)[AmPlan::ITER]) {
#pragma unroll
    for (int u = 0; u < AmPlan::ITER; ++u) {
        const raw16 r = load_raw16(reinterpret_cast<const char*>(src + p.gofs[u] + hg * 32));
        v[u] = ((p.inmask >> u) & 1u) ? r : raw16{0u, 0u, 0u, 0u};
    }
}
LFT_DEV void am_store(const AmPlan& p, char* lds, const raw16 (&v)[AmPlan::ITER]) {
#pragma unroll
    for (int u = 0; u < AmPlan::ITER; ++u)
        if (p.lofs[u] >= 0) store_raw16(lds + p.lofs[u], v[u]);
}

__global__ __launch_bounds__(256, 2) void k_spa_attn_mfma(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ Vv, bf16_t* __restrict__ O, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;
    char* ldsV = smem + kAmSlots * kAmRow;
    const int tiles_x = (w + kAttTX - 1) / kAttTX, tiles_y = (h + kAttTY - 1) / kAttTY;
    const int bid = xcd_tile(blockIdx.x, gridDim.x);                  // vertically adjacent tiles share 4 of their 8 halo rows
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, im = bid / (tiles_x * tiles_y);
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y0 = ty * kAttTY, x0 = tx * kAttTX, bxl = 8 * wave;          // block origin: (y0, x0 + bxl)
    const long long img0 = (long long)im * h * w;
    AmPlan plan;
    am_plan(plan, img0, y0, x0, h, w);
    // this lane's query
    const int qy = y0 + (r >> 3), qx = x0 + bxl + (r & 7);
    const bool qok = qy < h && qx < w;
    const long long qtok = img0 + min(qy, h - 1) * w + min(qx, w - 1);
    // 0 / -inf bias of the three score tiles: key kk = 32 j + row -> (jy, jx) = (kk / 12, kk % 12) of the 12 x 8 neighbourhood
    f32x16 bias[3];
    {
        const int wy0 = max(0, qy - 2), wy1 = min(h, qy + 3), wx0 = max(0, qx - 2), wx1 = min(min(h, qx + 3), w);   // reference LFT.py:155 (sic)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k0 = 32 * j + acc_row(i, 0), k1 = k0 + 4;          // key index for lane half 0 / 1 (compile-time)
                const int ky = y0 - 2 + (hh ? k1 / 12 : k0 / 12), kx = x0 + bxl - 2 + (hh ? k1 % 12 : k0 % 12);
                bias[j][i] = (ky >= wy0 && ky < wy1 && kx >= wx0 && kx < wx1) ? 0.0f : -INFINITY;
            }
    }
    // LDS byte offsets of this lane's operand rows: K fragment rows (key = 32 j + r), V^T transposing reads
    int kofs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int kk = 32 * j + r;
        kofs[j] = ((kk / 12) * kAttHC + bxl + kk % 12) * kAmRow + hh * 16;
    }
    const int li = lane & 15, tq = li >> 2, tp = li & 3, g2 = (lane >> 4) & 1;
    int vofs[3][2][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int wh = 0; wh < 2; ++wh) {
                const int kk = 32 * j + 16 * s2 + 8 * wh + 4 * hh + tq;                 // acc-order key of element group `wh`, row tq of the 4-row block
                vofs[j][s2][wh] = ((kk / 12) * kAttHC + bxl + kk % 12) * kAmRow + (16 * g2 + 4 * tp) * 2;
            }

#pragma unroll 1
    for (int hg = 0; hg < 4; ++hg) {
        Frag<bf16_t> qf[2];
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
            qf[hl].v = *reinterpret_cast<const bf16x8*>(Q + qtok * 128 + (2 * hg + hl) * 16 + 8 * hh);
        {
            raw16 kv[AmPlan::ITER], vv[AmPlan::ITER];
            am_load(plan, K, hg, kv);
            am_load(plan, Vv, hg, vv);
            if (hg) __syncthreads();                                       // previous head pair fully consumed
            am_store(plan, ldsK, kv);
            am_store(plan, ldsV, vv);
        }
        __syncthreads();
        // one accumulator per head: the V^T operand holds BOTH heads' 32 channels on its 32 rows, so each product
        // also fills the other head's 16 rows with garbage -- simply never read (cheaper than masking the operand)
        f32x16 o[2];
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            f32x16 S[3];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Frag<bf16_t> kf;
                kf.v = __builtin_bit_cast(bf16x8, load_raw16(ldsK + kofs[j] + hl * 32));
                S[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v, qf[hl].v, bias[j], 0, 0, 0);   // S^T[key, q] + mask bias
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, S[j][i]);
            }
            m = fmaxf(xhalf_max(m), -1.0e30f);                               // empty window: keep exp2(-inf - m) = 0, not NaN
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) { S[j][i] = fast_exp2(S[j][i] - m); sum += S[j][i]; }
            sum = xhalf_sum(sum);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;                // empty window (h < w quirk): 0, as the pinned reference
#pragma unroll
            for (int i = 0; i < 16; ++i) o[hl][i] = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(ldsV + vofs[j][s2][0]));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(ldsV + vofs[j][s2][1]));
                    Frag<bf16_t> vf;
                    vf.v = __builtin_bit_cast(bf16x8, (short __attribute__((ext_vector_type(8)))){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
                    mma(vf, acc_to_frag(S[j], s2, bf16_t()), o[hl]);          // O^T[d, q] += V^T P^T
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) o[hl][8 * hl + i] *= inv;           // rows 16 hl .. 16 hl + 15 are this head's
        }
        if (qok) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& oo = o[g >> 1];                                // pieces 0,1 (rows 0-15): head 0; pieces 2,3: head 1
                store4(O + qtok * 128 + hg * 32 + 8 * g + 4 * hh, f32x4{oo[4 * g], oo[4 * g + 1], oo[4 * g + 2], oo[4 * g + 3]});
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// SpaTrans part 2 (reference LFT.py:187-189, 171-174): per 32-token tile
//   t  = tok + O Wo^T ;  t2 = t + W2 relu(W1 LN'(t)) ;  y = Wl t2  (Conv3d 1x1x1 128->64)  [+ global skip, LFT.py:76]
// FFN hidden width 256 is processed in four 64-wide chunks so the hidden activations never leave registers.
// Stream: Wo[4x8, natural k] {W1c[2x8] W2c[4x4]} x4  Wl[2x8]  (176 fragments).
// ------------------------------------------------------------------------------------------
template <typename T, bool SKIP, bool TOKLM = false, bool YLM = false>   // YLM: output tile in lane-major form (consumer: k_up)
__global__ __launch_bounds__(256, LFT_SPA_OCC) void k_spa2(const T* __restrict__ TOK, const T* __restrict__ O, const T* __restrict__ ws,
                                              const float* __restrict__ ln, const T* __restrict__ skip, T* __restrict__ Y,
                                              long long ntok) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const long long t0 = ((long long)blockIdx.x * 4 + wave) * 32;                     // this wave's 32 consecutive tokens
    const int nvalid = (int)max(0LL, min(32LL, ntok - t0));
    const long long tb = min(t0, ntok - 1);
    char* scr = smem + WRing<T, kSpaChunk>::LDS_BYTES + 1024 + wave * TileIO<4, T>::BYTES;   // wave-private tile I/O scratch
    WRing<T, kSpaChunk> ring;
    ring.init(ws, smem, 176);                 // first: the weight DMA is in flight while the activation tiles are fetched
    f32x16 t[4], n[4];
    if constexpr (TOKLM) load_tile_lm<4, T>(TOK + tb * 128, lane, t);       // written by k_spa1 in the same 32-token tiling
    else load_tile<4, T>(TOK + tb * 128, nvalid, lane, t, scr);
    Frag<T> f[8];
    load_tile_frags<8, T>(O + tb * 128, nvalid, lane, f, scr);
    f32x16 sk[2];
    if (SKIP) load_tile<2, T>(skip + tb * 64, nvalid, lane, sk, scr);
    float* lds_ln = reinterpret_cast<float*>(smem + WRing<T, kSpaChunk>::LDS_BYTES);
    params_store(lds_ln, 256, params_load(ln + 256, 256));            // feed_forward.0.{weight,bias}; the tile loads above were waited for anyway; published by the first ring barrier
    linear_ring<4, 8, T>(ring, f, t);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) n[nt] = t[nt];
    layernorm_acc<4>(n, lds_ln, lds_ln + 128, hh);
    acc_frags<4, T>(n, f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f32x16 hid[2];
        zero_acc<2>(hid);
        linear_ring<2, 8, T>(ring, f, hid);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) hid[nt][i] = fmaxf(hid[nt][i], 0.0f);
        Frag<T> hf[4];
        acc_frags<2, T>(hid, hf);
        linear_ring<4, 4, T>(ring, hf, t);
    }
    acc_frags<4, T>(t, f);
    f32x16 y[2];
    zero_acc<2>(y);
    linear_ring<2, 8, T>(ring, f, y);
    if (SKIP) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) y[nt] += sk[nt];
    }
    if constexpr (YLM) store_tile_lm<2, T>(Y + tb * 64, lane, y);
    else store_tile<2, T>(Y + tb * 64, nvalid, lane, y, scr);
}

// ------------------------------------------------------------------------------------------
// Up-sampler, fused per LR token (reference LFT.py:39-44): never materialises the [B,64,A*h*s,A*w*s] map.
//   U = Wu x (64 s^2 values) -> LeakyReLU(0.2) -> these are the 64 x s x s HR features of this LR pixel
//   G = M lrelu(U): the token's contribution to the final 3x3 convolution on the (s+2)x(s+2) HR
//       neighbourhood of its s x s block ("overlap-add"; M built from upsampling.3.weight by k_pack/UPM).
// U is produced 32 rows at a time and immediately contracted into G.
// Stream per chunk c: Wu[1x4] M[GT x 2].  G rows >= (s+2)^2 are padding.
// ------------------------------------------------------------------------------------------
template <typename T, int GT, bool XLM = false>   // XLM: input tile in lane-major form (producer: the last k_spa2)
__global__ __launch_bounds__(256) void k_up(const T* __restrict__ X, const T* __restrict__ ws, float* __restrict__ G,
                                            long long ntok, int nchunk, int gp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const long long tok_raw = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + r;
    const bool ok = tok_raw < ntok;
    const long long tok = ok ? tok_raw : ntok - 1;
    const long long t0 = tok_raw - r;
    const int nvalid = (int)max(0LL, min(32LL, ntok - t0));
    char* scr = smem + WRing<T, kUpChunk>::LDS_BYTES + (threadIdx.x >> 6) * TileIO<2, T>::BYTES;
    WRing<T, kUpChunk> ring;
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
        for (int i = 0; i < 16; ++i) u[0][i] = u[0][i] > 0.0f ? u[0][i] : 0.2f * u[0][i];
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
template <int S>
__global__ __launch_bounds__(256) void k_assemble(const float* __restrict__ lr, const float* __restrict__ G, float* __restrict__ out,
                                                  int B, int A, int h, int w, int with_body) {
    // grid: x = 256-pixel column groups, y = HR mosaic row, z = batch.  Row quantities are block-uniform (scalar),
    // S is a compile-time power of two, so the per-thread index math has a single division (view column).
    const int HR_H = A * h * S, HR_W = A * w * S;
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y, b = blockIdx.z;
    if (X >= HR_W) return;
    const int a1 = Y / (h * S), a2 = X / (w * S);
    const float* view = lr + (size_t)b * (A * h) * (A * w) + (size_t)(a1 * h) * (A * w) + a2 * w;
    float v = bicubic_at(view, A * w, h, w, Y - a1 * h * S, X - a2 * w * S, S);
    if (with_body) {
        constexpr int GP = (S + 2) * (S + 2);
        const int hw = h * w, V = A * A;
        const int qy = Y / S, qx = X / S, i = Y % S, j = X % S;
        const float* Gb = G + (size_t)b * V * hw * GP;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int I = i - S * dy, by = qy + dy;
            if (I < -1 || I > S || by < 0 || by >= A * h) continue;
            const int vy = by / h, py = by - vy * h;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int J = j - S * dx, bx = qx + dx;
                if (J < -1 || J > S || bx < 0 || bx >= A * w) continue;
                const int vx = bx / w, px = bx - vx * w;
                v += Gb[((size_t)(vy * A + vx) * hw + py * w + px) * GP + (I + 1) * (S + 2) + (J + 1)];
            }
        }
    }
    out[((size_t)b * HR_H + Y) * HR_W + X] = v;
}

// Tiled form of the same gather (the one the forward uses): a workgroup owns an 8 x 8 block of LR mosaic pixels =
// an 8S x 8S block of output pixels, and stages the (8+2)^2 footprints it can touch in LDS with coalesced 16-byte
// loads (a footprint is (S+2)^2 contiguous floats) -- every footprint is fetched once per block instead of once per
// output row that needs it, and the 4-float groups read by neighbouring lanes fall on distinct banks (row stride
// (S+2)^2 = 36 or 16 floats).  The bicubic taps come from the LR mosaic, which is cache-resident (0.1 MB per patch).
template <int S>
__global__ __launch_bounds__(256) void k_assemble_t(const float* __restrict__ lr, const float* __restrict__ G, float* __restrict__ out,
                                                    int B, int A, int h, int w, int gld) {      // gld: floats between the footprints of consecutive tokens (>= GP)
    constexpr int TL = 8, HL = TL + 2, GP = (S + 2) * (S + 2), P4 = GP / 4, TS = TL * S;
    __shared__ __attribute__((aligned(16))) float gs[HL * HL * GP];
    const int MH = A * h, MW = A * w, HR_W = MW * S, HR_H = MH * S;
    const int by0 = blockIdx.y * TL, bx0 = blockIdx.x * TL, b = blockIdx.z;
    const int hw = h * w, V = A * A;
    const float* Gb = G + (size_t)b * V * hw * gld;
    for (int pidx = threadIdx.x; pidx < HL * HL * P4; pidx += 256) {
        const int slot = pidx / P4, part = pidx - slot * P4;
        const int by = by0 - 1 + slot / HL, bx = bx0 - 1 + slot % HL;
        f32x4 v = f32x4{0, 0, 0, 0};                                   // outside the mosaic: zero padding of the final conv (LFT.py:43)
        if (by >= 0 && by < MH && bx >= 0 && bx < MW) {
            const int vy = by / h, py = by - vy * h, vx = bx / w, px = bx - vx * w;
            v = load4(Gb + ((size_t)(vy * A + vx) * hw + py * w + px) * gld + part * 4);
        }
        *reinterpret_cast<f32x4*>(gs + slot * GP + part * 4) = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TS * TS; idx += 256) {
        const int Yl = idx / TS, Xl = idx - Yl * TS;
        const int Y = by0 * S + Yl, X = bx0 * S + Xl;
        if (Y >= HR_H || X >= HR_W) continue;
        const int a1 = Y / (h * S), a2 = X / (w * S);
        const float* view = lr + (size_t)b * MH * MW + (size_t)(a1 * h) * MW + a2 * w;
        float v = bicubic_at(view, MW, h, w, Y - a1 * h * S, X - a2 * w * S, S);
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
        out[((size_t)b * HR_H + Y) * HR_W + X] = v;
    }
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
