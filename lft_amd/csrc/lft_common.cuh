// lft_common.cuh -- gfx950 (CDNA4) building blocks shared by every LFT kernel.
//
// Design: "token on lane".  A wave owns a tile of 32 tokens; token t sits on MFMA column t
// (lanes t and t+32).  Every per-token linear layer is computed transposed,
//     Y^T[n, tok] = sum_k W[n, k] * X^T[k, tok],
// with the (pre-packed) weight as the MFMA A operand and the activations as the B operand.  The
// 32x32 accumulator of one product (channel on the row = register index, token on the lane) is,
// after a register-local down-conversion, directly the B operand of the next product -- so whole
// chains  LN -> Linear -> ReLU -> Linear -> residual  run in registers with no LDS round trip for
// activations.  LayerNorm / softmax reductions run over a lane's own registers plus one exchange
// with lane^32.
//
// Two operand precisions share all code:
//   T = float : v_mfma_f32_32x32x2_f32 (exact fp32, the parity path)
//   T = bf16  : v_mfma_f32_32x32x16_bf16 (fp32 accumulate, the throughput path; BASELINE configs[1] dtype)
//   T = f16   : v_mfma_f32_32x32x16_f16, the same kernels and layouts with 11 significant bits instead of 8 -- the
//               16-bit path that meets the 1e-3 parity tolerance (2e-4 observed); range 65504
// One "k-step" always covers 16 k values; a fragment holds 8 of them per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

typedef __bf16 bf16_t;
// Register pairs for the vector-instruction-bound phases.  LFT_PK = 1: a real 2-vector (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: one instruction per pair); 0: two scalars with the same interface.
#ifndef LFT_PK
#define LFT_PK 1
#endif
#if LFT_PK
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 p2_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f32x2 {
    float v[2];
    __device__ __forceinline__ float& operator[](int i) { return v[i]; }
    __device__ __forceinline__ const float& operator[](int i) const { return v[i]; }
};
__device__ __forceinline__ f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.v[0] - b.v[0], a.v[1] - b.v[1]}; }
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.v[0] + b.v[0], a.v[1] + b.v[1]}; }
__device__ __forceinline__ f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.v[0] * b.v[0], a.v[1] * b.v[1]}; }
__device__ __forceinline__ f32x2& operator+=(f32x2& a, f32x2 b) { a.v[0] += b.v[0]; a.v[1] += b.v[1]; return a; }
__device__ __forceinline__ f32x2 p2_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{__builtin_fmaf(a.v[0], b.v[0], c.v[0]), __builtin_fmaf(a.v[1], b.v[1], c.v[1])}; }
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// The two 16-bit operand types share every layout; H16<T> names their vector types.
template <typename T> struct H16;
template <> struct H16<bf16_t> { typedef bf16x4 v4; typedef bf16x8 v8; };
template <> struct H16<f16_t> { typedef f16x4 v4; typedef f16x8 v8; };
// Raw 16-byte moves (LDS staging, fragment reads) go through a may_alias type, so that type-based alias
// analysis can never treat an LDS store and a differently-typed LDS load of the same bytes as independent.
typedef unsigned int raw16 __attribute__((ext_vector_type(4), may_alias));

#define LFT_DEV static __device__ __forceinline__
#define LFT_MEM __device__ __forceinline__

// In-kernel phase stamps (LFT_STAMP) and other timing experiments live in lft_experiment.cuh, which only a diagnostic
// build includes (-DLFT_EXPERIMENT, tools/stamp_report.py); in the product build LFT_STAMP() is empty and nothing else of
// that header exists.
#ifdef LFT_EXPERIMENT
#include "lft_experiment.cuh"
#else
#define LFT_STAMP(slot) ((void)0)
#define LFT_STAMP_IT(slot, it_ofs) ((void)0)
#endif

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one private 4 MiB L2): consecutive block
// ids -- i.e. neighbouring tiles, which share halo rows -- land on eight different L2s and each fetches the shared rows
// from HBM itself.  This bijective remap gives every XCD a CONTIGUOUS range of tiles instead, so a halo row is fetched
// into one L2 once.  Pure performance: placement is not part of any correctness argument.
LFT_DEV int xcd_tile(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

constexpr int LFT_C = 64;            // feature channels (reference option.py --channels, LFT.py:11)
constexpr int LFT_E = 128;           // spatial token width 2C (reference LFT.py:124)
constexpr float LFT_LN_EPS = 1e-5f;  // nn.LayerNorm default
constexpr float LFT_LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------
// Fragments: 8 k-values per lane of one 16-deep k-step.
//   lane = 32*h + r.   A operand: A[row r][k(h,j)]   B operand: B[k(h,j)][col r]   j = 0..7
// For bf16 the hardware fixes k(h,j) = 8h + j inside the instruction; for fp32 we issue 8
// v_mfma_f32_32x32x2_f32, the j-th consuming element j of both operands (k pair {h=0,h=1}).
// Any labelling of the 16 k values works as long as A and B agree, which is what lets an
// accumulator tile be re-used as an operand ("acc order", see acc_to_frag).
// ------------------------------------------------------------------------------------------
template <typename T> struct Frag { typename H16<T>::v8 v; };          // bf16 / f16
template <> struct Frag<float> { f32x4 lo, hi; };

LFT_DEV void mma(const Frag<float>& a, const Frag<float>& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], c, 0, 0, 0);
}
LFT_DEV f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
LFT_DEV f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename T> LFT_DEV void mma(const Frag<T>& a, const Frag<T>& b, f32x16& c) { c = mfma16(a.v, b.v, c); }

LFT_DEV Frag<float> frag_zero(float) { Frag<float> f; f.lo = f32x4{0, 0, 0, 0}; f.hi = f.lo; return f; }
template <typename T> LFT_DEV Frag<T> frag_zero(T) {
    Frag<T> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (T)0.0f;
    return f;
}
// keep elements j<4 (which==0) or j>=4 (which==1); the rest become zero
LFT_DEV Frag<float> frag_half(const Frag<float>& f, int which) {
    Frag<float> g = f;
    if (which == 0) g.hi = f32x4{0, 0, 0, 0}; else g.lo = f32x4{0, 0, 0, 0};
    return g;
}
template <typename T> LFT_DEV Frag<T> frag_half(const Frag<T>& f, int which) {
    Frag<T> g = f;
#pragma unroll
    for (int j = 0; j < 4; ++j) g.v[which == 0 ? j + 4 : j] = (T)0.0f;
    return g;
}
template <typename T> LFT_DEV Frag<T> frag_select(bool keep, const Frag<T>& f) { return keep ? f : frag_zero(T()); }

// ------------------------------------------------------------------------------------------
// Accumulator layout (hardware C/D map of every 32x32 MFMA on gfx950):
//   element [row][col]:  col = lane & 31,  row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5),  i = 0..15
// Rows 16s .. 16s+15 (s = 0,1) of a tile are registers 8s .. 8s+7; taken as a fragment they label
//   k(h, j) = 16 s + 8 (j >> 2) + 4 h + (j & 3)                       ("acc order")
// and weights consumed against such a fragment are packed with the same labelling.
// ------------------------------------------------------------------------------------------
LFT_DEV int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

LFT_DEV Frag<float> acc_to_frag(const f32x16& a, int s, float) {
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.lo[j] = a[8 * s + j]; f.hi[j] = a[8 * s + 4 + j]; }
    return f;
}
template <typename T> LFT_DEV Frag<T> acc_to_frag(const f32x16& a, int s, T) {
    Frag<T> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (T)a[8 * s + j];
    return f;
}

// ------------------------------------------------------------------------------------------
// Packed weight streams.  A stream is a sequence of 1 KiB "pieces", each lane-linear (lane l owns bytes
// 16 l .. 16 l + 15) -- exactly what one wave-wide global_load_lds_dwordx4 moves.  A bf16 fragment is one
// piece (8 bf16 per lane); an fp32 fragment is two pieces (elements 0-3, then elements 4-7).
// ------------------------------------------------------------------------------------------
template <typename T> struct FragInfo { static constexpr int PIECES = 1; };      // bf16 / f16
template <> struct FragInfo<float> { static constexpr int PIECES = 2; };

LFT_DEV raw16 load_raw16(const char* p) { return *reinterpret_cast<const raw16*>(p); }
LFT_DEV void store_raw16(char* p, raw16 v) { *reinterpret_cast<raw16*>(p) = v; }

LFT_DEV Frag<float> frag_from_pieces(const char* base, int lane, float) {
    Frag<float> r;
    r.lo = __builtin_bit_cast(f32x4, load_raw16(base + lane * 16));
    r.hi = __builtin_bit_cast(f32x4, load_raw16(base + 1024 + lane * 16));
    return r;
}
template <typename T> LFT_DEV Frag<T> frag_from_pieces(const char* base, int lane, T) {
    Frag<T> r;
    r.v = __builtin_bit_cast(typename H16<T>::v8, load_raw16(base + lane * 16));
    return r;
}
template <typename T> LFT_DEV Frag<T> load_wfrag(const T* __restrict__ stream, int f, int lane) {
    return frag_from_pieces(reinterpret_cast<const char*>(stream) + (size_t)f * 1024 * FragInfo<T>::PIECES, lane, T());
}

// Asynchronous global -> LDS copy of one 1 KiB piece by one wave (LDS-DMA, no VGPR staging).
// The LDS destination is wave-uniform (M0) + 16 * lane; the global source is per lane.
LFT_DEV void glds_piece(const char* __restrict__ gsrc, char* lds_dst, int lane) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + lane * 16),
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// The same LDS-DMA issued from inline asm, for pipelines whose in-flight pieces must NOT be known to hipcc: after the
// builtin form the compiler guards every later LDS read it cannot disambiguate with a vmcnt wait that drains the
// prefetch (k_spa_b: an s_waitcnt vmcnt(4) in front of the first V read of a head pair, i.e. "next tile landed").  The
// caller orders data by its own counted s_waitcnt + barrier.  M0 (LDS base of the DMA) is compiler-reserved: saved,
// set and restored inside the one statement.  lds_dst must be wave-uniform.
LFT_DEV void glds16_asm(const char* gsrc_lane, char* lds_dst) {
    unsigned keep;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(dst) : "memory");
}

// s_waitcnt vmcnt(n) for a value that is a compile-time constant after inlining (the switch folds away).
#define LFT_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
LFT_DEV void wait_vmcnt(int n) {
    switch (n < 0 ? 0 : (n > 63 ? 63 : n)) {
        LFT_VMCNT_CASE(0) LFT_VMCNT_CASE(1) LFT_VMCNT_CASE(2) LFT_VMCNT_CASE(3) LFT_VMCNT_CASE(4) LFT_VMCNT_CASE(5) LFT_VMCNT_CASE(6) LFT_VMCNT_CASE(7)
        LFT_VMCNT_CASE(8) LFT_VMCNT_CASE(9) LFT_VMCNT_CASE(10) LFT_VMCNT_CASE(11) LFT_VMCNT_CASE(12) LFT_VMCNT_CASE(13) LFT_VMCNT_CASE(14) LFT_VMCNT_CASE(15)
        LFT_VMCNT_CASE(16) LFT_VMCNT_CASE(17) LFT_VMCNT_CASE(18) LFT_VMCNT_CASE(19) LFT_VMCNT_CASE(20) LFT_VMCNT_CASE(21) LFT_VMCNT_CASE(22) LFT_VMCNT_CASE(23)
        LFT_VMCNT_CASE(24) LFT_VMCNT_CASE(25) LFT_VMCNT_CASE(26) LFT_VMCNT_CASE(27) LFT_VMCNT_CASE(28) LFT_VMCNT_CASE(29) LFT_VMCNT_CASE(30) LFT_VMCNT_CASE(31)
        LFT_VMCNT_CASE(32) LFT_VMCNT_CASE(33) LFT_VMCNT_CASE(34) LFT_VMCNT_CASE(35) LFT_VMCNT_CASE(36) LFT_VMCNT_CASE(37) LFT_VMCNT_CASE(38) LFT_VMCNT_CASE(39)
        LFT_VMCNT_CASE(40) LFT_VMCNT_CASE(41) LFT_VMCNT_CASE(42) LFT_VMCNT_CASE(43) LFT_VMCNT_CASE(44) LFT_VMCNT_CASE(45) LFT_VMCNT_CASE(46) LFT_VMCNT_CASE(47)
        LFT_VMCNT_CASE(48) LFT_VMCNT_CASE(49) LFT_VMCNT_CASE(50) LFT_VMCNT_CASE(51) LFT_VMCNT_CASE(52) LFT_VMCNT_CASE(53) LFT_VMCNT_CASE(54) LFT_VMCNT_CASE(55)
        LFT_VMCNT_CASE(56) LFT_VMCNT_CASE(57) LFT_VMCNT_CASE(58) LFT_VMCNT_CASE(59) LFT_VMCNT_CASE(60) LFT_VMCNT_CASE(61) LFT_VMCNT_CASE(62) LFT_VMCNT_CASE(63)
    }
}

// Workgroup barrier that does NOT drain vector memory: own LDS operations complete (lgkmcnt(0)), then s_barrier.
// __syncthreads() makes hipcc wait vmcnt(0) whenever LDS-DMA is outstanding, which serialises the ring's
// 2-chunk lookahead and forces every in-flight output store to be acknowledged at each chunk boundary.  The
// "memory" clobber keeps the compiler from moving LDS/global accesses across it; LDS-DMA completion is handled
// explicitly by the counted wait in WRing.
LFT_DEV void wg_barrier_keep_vm() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The same with the LDS wait as a builtin (0xC07F = lgkmcnt(0), vmcnt / expcnt untouched): hipcc's wait-count bookkeeping then
// KNOWS that every earlier ds_read has returned.  After the all-asm form it re-waits for register sets fetched a boundary ago
// -- and, with younger reads in flight, does so with lgkmcnt(0): the pipelined ring's fetch latency exposed again.
LFT_DEV void wg_barrier_keep_vm_lds_visible() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("s_barrier" ::: "memory");
}

// Protocol notes for tools/lds_dma_hazards.py (the static check of every LDS-DMA pipeline): comment-only asm statements -- no
// instruction is emitted; they only name, in the assembly listing, the LDS slot the following DMA pieces fill (DMA), the point from
// which a slot is read (USE) and the point at which its reads have been issued for the last time (DONE).  A note is written only
// where the slot is a compile-time constant after inlining and unrolling (the straight-line kernels); in run-time loops (k_up,
// k_ang, k_linr) the slot is a register value and the tool reports the ring as not modelled there.
#define LFT_NOTE_ASM_(kind, ring, slot) asm volatile("; LFT_NOTE " kind " ring=%0 slot=%1" :: "i"(ring), "i"(slot))
#define LFT_DMA_NOTE(kind, ring, slot) do { if (__builtin_constant_p(slot)) LFT_NOTE_ASM_(kind, ring, slot); } while (0)
constexpr int kNoteWRing = 0, kNoteWRingPipe = 1, kNoteKV = 2, kNoteConvIn = 3, kNoteAngW = 4;

// Weight ring: the 4 waves of a workgroup consume the same fragment stream in lock-step.  The stream is cut
// into chunks of CH fragments held in a 3-slot LDS ring: while chunk c feeds the MFMAs, chunks c+1 and c+2 are
// in flight / landed (LDS-DMA issued two chunks ahead: one chunk of MFMAs, ~0.25 us, is shorter than the
// L2 -> LDS latency, so a 2-slot ring stalled at every chunk boundary).  next() must be called by all 256
// threads at the same program points: at the first fragment of a chunk it
//   1. waits until this wave's own DMA pieces of chunk c have landed -- a COUNTED vmcnt that leaves the
//      pieces of chunk c+1 (the youngest VM operations) in flight.  vmcnt retires in issue order, so any
//      younger ordinary load/store only makes the wait stricter, never unsafe.  hipcc does NOT insert this
//      wait reliably for LDS-DMA (k_up's loop had a bare "s_waitcnt lgkmcnt(0); s_barrier"), so it is explicit;
//   2. executes the workgroup barrier: chunk c is published, chunk c-1 is retired by every wave;
//   3. issues the DMA of chunk c+2 into the slot chunk c-1 just vacated.
template <typename T, int CH, int NW = 4>       // NW: waves of the workgroup that share the ring
struct WRing {
    static constexpr int NBUF = 3;
    static constexpr int FRAG_BYTES = 1024 * FragInfo<T>::PIECES;
    static constexpr int CHUNK_BYTES = CH * FRAG_BYTES;
    static constexpr int LDS_BYTES = NBUF * CHUNK_BYTES;
    static constexpr int PIECES_PER_WAVE = CH * FragInfo<T>::PIECES / NW;
    static_assert((CH * FragInfo<T>::PIECES) % NW == 0, "chunk must split over the workgroup's waves");
    static_assert(PIECES_PER_WAVE <= 12, "counted vmcnt immediates below cover up to 12 pieces per wave");
    const char* g;
    char* lds;
    int lane, wave, pos, nfrag;
    LFT_MEM void init(const T* stream, char* lds_base, int total_frags) {
        g = reinterpret_cast<const char*>(stream); lds = lds_base; nfrag = total_frags; pos = 0;
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        issue(0);
        issue(1);
    }
    // Every wave issues exactly PIECES_PER_WAVE DMA instructions per chunk (clamped to the last piece of the
    // stream when the chunk is short) so that the counted wait below is exact.
    LFT_MEM void issue(int c) {
        if (c * CH >= nfrag) return;                                   // uniform: chunk does not exist
        const char* src = g + (size_t)c * CHUNK_BYTES;
        char* dst = lds + (c % NBUF) * CHUNK_BYTES;
        const int last = (nfrag - c * CH) * FragInfo<T>::PIECES - 1;    // last piece that exists in this chunk
        LFT_DMA_NOTE("DMA", kNoteWRing, c % NBUF);
#pragma unroll
        for (int i = 0; i < PIECES_PER_WAVE; ++i) {
            const int piece = min(wave * PIECES_PER_WAVE + i, last);
            glds_piece(src + piece * 1024, dst + piece * 1024, lane);
        }
    }
    // Chunk c's DMA was issued right after barrier c-2; the only operations this wait may leave in flight are the
    // PIECES_PER_WAVE pieces of chunk c+1, the youngest DMA.  Any other VM operation the kernel has issued since (tile
    // stores, loads) is younger still and only makes the wait stricter -- never unsafe.  (Round 1 also subtracted the
    // kernel's own store counts to let stores stay in flight; that relied on hipcc emitting exactly the counted number of
    // store instructions, failed once the stores changed form, and bought nothing measurable: removed.)
    LFT_MEM void wait_landed(bool next_in_flight) { wait_vmcnt(next_in_flight ? PIECES_PER_WAVE : 0); }
    LFT_MEM Frag<T> next() {
        const int c = pos / CH, i = pos % CH;
        if (i == 0) {
            if (c > 0) LFT_DMA_NOTE("DONE", kNoteWRing, (c + NBUF - 1) % NBUF);   // chunk c-1 has been read for the last time
            wait_landed((c + 1) * CH < nfrag);
            wg_barrier_keep_vm();
            issue(c + 2);
            LFT_DMA_NOTE("USE", kNoteWRing, c % NBUF);
        }
        ++pos;
        return frag_from_pieces(lds + (c % NBUF) * CHUNK_BYTES + i * FRAG_BYTES, lane, T());
    }
};


// Register-pipelined weight ring.  WRing reads a chunk's fragments from LDS right behind the barrier that publishes
// it, so every chunk boundary costs a barrier, an LDS round trip and only then the chunk's MFMAs (~700 cycles per 8-fragment
// chunk of which 256 are matrix work: in-kernel stamps).  Here the fragments of chunk c+1 are fetched into a SECOND register set
// at the boundary in front of chunk c's MFMAs, which then run from registers loaded one boundary earlier: the LDS latency and
// the barrier skew hide under matrix work.  Costs CH fragments of registers.  A chunk's slot is free as soon as every wave
// holds it in registers, i.e. at the next boundary (the barrier helper waits lgkmcnt(0) first): the DMA runs NBUF chunks ahead.
// Every position is a TEMPLATE argument (get<POS>()): the two register sets are only ever indexed by constants, whatever the
// optimiser does (with a run-time position member the sets ended up in scratch).  Protocol: setup(); issue(0 .. NBUF-1) by the
// caller as the slots become available; start(); then get<0>(), get<1>(), ... in order, at the same program points in every
// wave (linear_ring_at).  NFRAG = fragments in the stream.
template <typename T, int CH, int NW, int NBUF, int NFRAG>
struct WRingPipe {
    static constexpr int FRAG_BYTES = 1024 * FragInfo<T>::PIECES;
    static constexpr int CHUNK_BYTES = CH * FRAG_BYTES;
    static constexpr int LDS_BYTES = NBUF * CHUNK_BYTES;
    static constexpr int PIECES_PER_WAVE = CH * FragInfo<T>::PIECES / NW;
    static constexpr int NCHUNK = (NFRAG + CH - 1) / CH;
    static_assert((CH * FragInfo<T>::PIECES) % NW == 0, "chunk must split over the workgroup's waves");
    static_assert(PIECES_PER_WAVE * (NBUF - 1) <= 63, "counted vmcnt immediate");
    static constexpr int cmin(int a, int b) { return a < b ? a : b; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    const char* g;
    char* lds;
    int lane, wave, split_slot, split_gap;
    Frag<T> fr0[CH], fr1[CH];
    LFT_MEM void setup(const T* stream, char* lds_base, int split_slot_ = NBUF, int split_gap_ = 0) {
        g = reinterpret_cast<const char*>(stream); lds = lds_base;
        split_slot = split_slot_; split_gap = split_gap_;
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    }
    LFT_MEM char* slot(int c) const {
        const int s = c % NBUF;
        return lds + s * CHUNK_BYTES + (s >= split_slot ? split_gap : 0);
    }
    LFT_MEM void issue(int c) {                                         // every wave issues exactly PIECES_PER_WAVE pieces per existing chunk
        if (c * CH >= NFRAG) return;
        const char* src = g + (size_t)c * CHUNK_BYTES;
        char* dst = slot(c);
        const int last = (NFRAG - c * CH) * FragInfo<T>::PIECES - 1;
        LFT_DMA_NOTE("DMA", kNoteWRingPipe, c % NBUF);
#pragma unroll
        for (int i = 0; i < PIECES_PER_WAVE; ++i) {
            const int piece = min(wave * PIECES_PER_WAVE + i, last);
            glds_piece(src + piece * 1024, dst + piece * 1024, lane);
        }
    }
    template <int C> LFT_MEM void fetch() {                             // LDS -> register set C & 1; the reads are in flight on return
        const char* base = slot(C);
        LFT_NOTE_ASM_("USE", kNoteWRingPipe, C % NBUF);
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (C * CH + i < NFRAG) {
                if constexpr (C & 1) fr1[i] = frag_from_pieces(base + i * FRAG_BYTES, lane, T());
                else fr0[i] = frag_from_pieces(base + i * FRAG_BYTES, lane, T());
            }
        LFT_NOTE_ASM_("DONE", kNoteWRingPipe, C % NBUF);                  // the chunk now lives in registers: its slot is free after the next barrier
    }
    LFT_MEM void init(const T* stream, char* lds_base) {                // all slots free from the start
        setup(stream, lds_base);
#pragma unroll
        for (int c = 0; c < NBUF; ++c) issue(c);
    }
    LFT_MEM void start() {                                              // chunks 0 .. NBUF-1 have been issued
        wait_vmcnt(cmax(0, cmin(NBUF - 1, NCHUNK - 1)) * PIECES_PER_WAVE);
        wg_barrier_keep_vm_lds_visible();
        fetch<0>();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int POS> LFT_MEM Frag<T> get() {
        constexpr int C = POS / CH, I = POS % CH;
        static_assert(POS >= 0 && POS < NFRAG, "position outside the stream");
        if constexpr (I == 0 && C + 1 < NCHUNK) {
            __builtin_amdgcn_sched_barrier(0);                          // chunk C-1's MFMAs are issued BEFORE the barrier: they execute while the wave waits in it
            wait_vmcnt(cmax(0, cmin(NBUF - 2, NCHUNK - 2 - C)) * PIECES_PER_WAVE);   // chunks C+2 .. C+NBUF-1 stay in flight behind chunk C+1
            wg_barrier_keep_vm_lds_visible();                           // chunk C+1 published; chunk C is in every wave's registers: its slot is free
            issue(C + NBUF);
            fetch<C + 1>();
            // pin the fetch here: left alone, the scheduler sinks half of these reads to just in front of the NEXT boundary (fewer
            // live registers), where the barrier's lgkmcnt(0) then waits for them -- the latency this ring exists to hide
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (C & 1) return fr1[I];
        else return fr0[I];
    }
};
// Y^T[nt] += sum_ks W(nt, ks) x[ks] with the fragments at stream positions POS0 .. POS0 + NT_OUT * KS - 1 (nt-major)
template <int POS0, int NT_OUT, int KS, typename Ring, typename T, int... I>
LFT_DEV void linear_ring_seq(Ring& ring, const Frag<T> (&x)[KS], f32x16 (&y)[NT_OUT], std::integer_sequence<int, I...>) {
    (mma(ring.template get<POS0 + I>(), x[I % KS], y[I / KS]), ...);
}
template <int POS0, int NT_OUT, int KS, typename Ring, typename T>
LFT_DEV void linear_ring_at(Ring& ring, const Frag<T> (&x)[KS], f32x16 (&y)[NT_OUT]) {
    linear_ring_seq<POS0, NT_OUT, KS>(ring, x, y, std::make_integer_sequence<int, NT_OUT * KS>());
}
// 8 consecutive channels of a token row in memory -> fragment in NATURAL k order (k = 8h + j);
// used where an operand comes straight from HBM (attention output).  Branch-free: the caller passes an
// address that is always readable (clamped for out-of-range lanes) and the result is zeroed by `ok`, so the
// compiler can issue all loads of a tile back to back instead of one exec-masked load + wait at a time.
LFT_DEV Frag<float> load_row8(const float* __restrict__ p, bool ok, float) {
    Frag<float> r;
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    r.lo = q[0]; r.hi = q[1];
    if (!ok) r = frag_zero(0.0f);
    return r;
}
template <typename T> LFT_DEV Frag<T> load_row8(const T* __restrict__ p, bool ok, T) {
    Frag<T> r;
    r.v = *reinterpret_cast<const typename H16<T>::v8*>(p);
    if (!ok) r = frag_zero(T());
    return r;
}

// same from LDS (always in bounds; caller zeroes by predicate)
LFT_DEV Frag<float> lds_row8(const char* p, bool ok, float) {
    Frag<float> r;
    r.lo = __builtin_bit_cast(f32x4, load_raw16(p));
    r.hi = __builtin_bit_cast(f32x4, load_raw16(p + 16));
    if (!ok) r = frag_zero(0.0f);
    return r;
}
template <typename T> LFT_DEV Frag<T> lds_row8(const char* p, bool ok, T) {
    Frag<T> r;
    r.v = __builtin_bit_cast(typename H16<T>::v8, load_raw16(p));
    if (!ok) r = frag_zero(T());
    return r;
}

// ------------------------------------------------------------------------------------------
// Token rows <-> accumulator layout.  Token row = NT*32 channels contiguous in memory.
// Lane (h, r) owns channels 32 nt + 8 g + 4 h + {0..3} (g = 0..3) of token r: 4-channel pieces.
// ------------------------------------------------------------------------------------------
LFT_DEV f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <typename T> LFT_DEV f32x4 load4(const T* p) {
    typename H16<T>::v4 v = *reinterpret_cast<const typename H16<T>::v4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
LFT_DEV void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <typename T> LFT_DEV void store4(T* p, f32x4 v) {
    typename H16<T>::v4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
    *reinterpret_cast<typename H16<T>::v4*>(p) = o;
}

// `row` must be readable for every lane (clamp the token index); lanes with !ok get zeros.
template <int NT, typename T>
LFT_DEV void load_acc(const T* __restrict__ row, bool ok, int h, f32x16 (&a)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = load4(row + 32 * nt + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[nt][4 * g + j] = ok ? v[j] : 0.0f;
        }
    }
}
// Raw (unconverted) accumulator-layout pieces of a token row, for values that are loaded early but used late:
// a bf16 row costs half the registers of its fp32 expansion while it waits.
template <typename T> struct RawPiece { typedef typename H16<T>::v4 type; };      // bf16 / f16
template <> struct RawPiece<float> { typedef f32x4 type; };
template <int NT, typename T>
LFT_DEV void load_acc_raw(const T* __restrict__ row, int h, typename RawPiece<T>::type (&p)[NT * 4]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) p[nt * 4 + g] = *reinterpret_cast<const typename RawPiece<T>::type*>(row + 32 * nt + 8 * g + 4 * h);
}
template <int NT, typename T>
LFT_DEV void add_acc_raw(f32x16 (&a)[NT], const typename RawPiece<T>::type (&p)[NT * 4], bool ok) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[nt][4 * g + j] += ok ? (float)p[nt * 4 + g][j] : 0.0f;
}

// "Lane-major" tables: input-independent per-token data that a wave consumes in accumulator layout (position
// tokens, angular PE) is stored at pack time as [tile of 32 tokens][k = nt*2 + g/2][lane][8 elements], lane (h, r)
// holding pieces g = 2(k&1)... i.e. exactly the 16 four-channel pieces of its token, so that each of the NT*2
// wave loads is one fully coalesced 64 x 16 B (bf16) or 64 x 32 B (fp32) block instead of touching 32 cache lines.
template <int NT, typename T>
LFT_DEV void store_lane_major(T* __restrict__ tile_base, int lane, const f32x16 (&a)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            store4(tile_base + ((size_t)(nt * 2 + (g >> 1)) * 64 + lane) * 8 + (g & 1) * 4,
                   f32x4{a[nt][4 * g], a[nt][4 * g + 1], a[nt][4 * g + 2], a[nt][4 * g + 3]});
}
// The same layout for an ACTIVATION tile that one kernel writes and another reads with the same 32-token tiling
// (the spatial tokens between k_spa1 and k_spa2): 16 bytes per lane and store, no LDS transposition on either side.
// Returns the number of wave-level store instructions.
template <int NT, typename T>
LFT_DEV int store_tile_lm(T* __restrict__ tile_base, int lane, const f32x16 (&a)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            T* dst = tile_base + ((size_t)(nt * 2 + k) * 64 + lane) * 8;
            if constexpr (sizeof(T) == 2) {
                typename H16<T>::v8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (T)a[nt][8 * k + j];
                store_raw16(reinterpret_cast<char*>(dst), __builtin_bit_cast(raw16, v));
            } else {
                store_raw16(reinterpret_cast<char*>(dst), __builtin_bit_cast(raw16, f32x4{a[nt][8 * k], a[nt][8 * k + 1], a[nt][8 * k + 2], a[nt][8 * k + 3]}));
                store_raw16(reinterpret_cast<char*>(dst + 4), __builtin_bit_cast(raw16, f32x4{a[nt][8 * k + 4], a[nt][8 * k + 5], a[nt][8 * k + 6], a[nt][8 * k + 7]}));
            }
        }
    return NT * 2 * (sizeof(T) == 2 ? 1 : 2);
}
template <int NT, typename T>
LFT_DEV void load_tile_lm(const T* __restrict__ tile_base, int lane, f32x16 (&a)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const T* src = tile_base + ((size_t)(nt * 2 + k) * 64 + lane) * 8;
            if constexpr (sizeof(T) == 2) {
                const auto v = __builtin_bit_cast(typename H16<T>::v8, load_raw16(reinterpret_cast<const char*>(src)));
#pragma unroll
                for (int j = 0; j < 8; ++j) a[nt][8 * k + j] = (float)v[j];
            } else {
                const f32x4 lo = __builtin_bit_cast(f32x4, load_raw16(reinterpret_cast<const char*>(src)));
                const f32x4 hi = __builtin_bit_cast(f32x4, load_raw16(reinterpret_cast<const char*>(src + 4)));
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[nt][8 * k + j] = lo[j]; a[nt][8 * k + 4 + j] = hi[j]; }
            }
        }
}

template <int NT, typename T>
LFT_DEV void load_lane_major_raw(const T* __restrict__ tile_base, int lane, typename RawPiece<T>::type (&p)[NT * 4]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            p[nt * 4 + g] = *reinterpret_cast<const typename RawPiece<T>::type*>(tile_base + ((size_t)(nt * 2 + (g >> 1)) * 64 + lane) * 8 + (g & 1) * 4);
}

template <int NT, typename T>
LFT_DEV void store_acc(T* __restrict__ row, bool ok, int h, const f32x16 (&a)[NT]) {
    if (!ok) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = f32x4{a[nt][4 * g], a[nt][4 * g + 1], a[nt][4 * g + 2], a[nt][4 * g + 3]};
            store4(row + 32 * nt + 8 * g + 4 * h, v);
        }
    }
}
// ------------------------------------------------------------------------------------------
// Coalesced tile I/O.  The accumulator layout gives a lane 4-channel pieces of ONE token row, so a direct wave
// store touches 32 different rows (32 cache lines) per instruction -- measured 10x slower than the MFMAs it
// follows.  A wave's 32 tokens are consecutive in memory (one contiguous block of 32 rows), so tiles go
// through a wave-private LDS scratch, 16 rows at a time: accumulator-layout accesses on one side, 16-byte
// lane-linear pieces (1 KiB contiguous per wave instruction) on the global-memory side.  Rows are padded
// by 16 B.  Only the owning wave touches its scratch: LDS operations of one wave execute in order, the wave
// barrier below only stops the compiler from re-ordering them.
// ------------------------------------------------------------------------------------------
template <int NT, typename T> struct TileIO {
    static constexpr int ROW_BYTES = NT * 32 * (int)sizeof(T);       // a token row in memory
    static constexpr int ROWB = ROW_BYTES + 16;                       // padded row in scratch
    static constexpr int PASS_ROWS = 16;
    static constexpr int BYTES = PASS_ROWS * ROWB;                    // scratch per wave
    static constexpr int P16 = ROW_BYTES / 16;                        // 16-byte pieces per row
};
// The LDS queue serves one wave's instructions in issue order, so a ds_read issued after a ds_write of the same
// wave sees it whichever lane wrote it: only the COMPILER must be kept from re-ordering them.  (A wavefront-
// scope C++ fence is not free here: hipcc lowered it to s_waitcnt vmcnt(0), i.e. every pass waited for the
// previous pass's global stores to be acknowledged -- 3.5k cycles per tile store.)
LFT_DEV void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
LFT_DEV void lds_store4(char* p, f32x4 v, float) { store_raw16(p, __builtin_bit_cast(raw16, v)); }
template <typename T> LFT_DEV void lds_store4(char* p, f32x4 v, T) {
    typename H16<T>::v4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
    *reinterpret_cast<unsigned long long __attribute__((may_alias))*>(p) = __builtin_bit_cast(unsigned long long, o);
}
LFT_DEV f32x4 lds_load4(const char* p, float) { return __builtin_bit_cast(f32x4, load_raw16(p)); }
template <typename T> LFT_DEV f32x4 lds_load4(const char* p, T) {
    const auto v = __builtin_bit_cast(typename H16<T>::v4, *reinterpret_cast<const unsigned long long __attribute__((may_alias))*>(p));
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// Store a wave's accumulator tile (32 tokens x NT*32 channels) to `gbase` = address of the tile's first token
// (plus a channel offset when the tile is a column slice); ROW_CH = channels of a full row in memory (row
// stride), so a 64-channel half of a 128-channel row can be written on its own.  Rows >= nvalid are not written.
// Returns the number of wave-level global store instructions issued for a full tile.
template <int NT, typename T, int ROW_CH = NT * 32>
LFT_DEV int store_tile(T* __restrict__ gbase, int nvalid, int lane, const f32x16 (&a)[NT], char* scr, size_t row_stride_bytes = 0) {
    using IO = TileIO<NT, T>;
    const size_t STRIDE = row_stride_bytes ? row_stride_bytes : (size_t)ROW_CH * sizeof(T);
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        wave_lds_fence();                                              // previous pass fully read
        if ((r >> 4) == pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    lds_store4(scr + (r & 15) * IO::ROWB + (32 * nt + 8 * g + 4 * hh) * (int)sizeof(T),
                               f32x4{a[nt][4 * g], a[nt][4 * g + 1], a[nt][4 * g + 2], a[nt][4 * g + 3]}, T());
        }
        wave_lds_fence();
        char* g0 = reinterpret_cast<char*>(gbase) + (size_t)pass * 16 * STRIDE;
        raw16 v[16 * IO::P16 / 64];
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {                     // all LDS reads first, then all stores
            const int idx = i * 64 + lane;
            v[i] = load_raw16(scr + (idx / IO::P16) * IO::ROWB + (idx % IO::P16) * 16);
        }
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane, row = idx / IO::P16, pc = idx % IO::P16;
            if (pass * 16 + row < nvalid) store_raw16(g0 + (size_t)row * STRIDE + pc * 16, v[i]);
        }
    }
    return 2 * (16 * IO::P16 / 64);
}

// Load a wave's tile into the accumulator layout; rows >= nvalid read as zero.  `gbase` must be readable for
// max(nvalid,1) rows; rows beyond are not touched.
template <int NT, typename T>
LFT_DEV void load_tile(const T* __restrict__ gbase, int nvalid, int lane, f32x16 (&a)[NT], char* scr,
                       size_t row_stride_bytes = TileIO<NT, T>::ROW_BYTES) {
    using IO = TileIO<NT, T>;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const char* g0 = reinterpret_cast<const char*>(gbase) + (size_t)pass * 16 * row_stride_bytes;
        raw16 v[16 * IO::P16 / 64];
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane, row = idx / IO::P16, pc = idx % IO::P16;
            const bool in = pass * 16 + row < nvalid;
            const raw16 t = load_raw16(in ? g0 + (size_t)row * row_stride_bytes + pc * 16 : reinterpret_cast<const char*>(gbase));
            v[i] = in ? t : raw16{0u, 0u, 0u, 0u};
        }
        wave_lds_fence();                                              // previous pass fully consumed
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane;
            store_raw16(scr + (idx / IO::P16) * IO::ROWB + (idx % IO::P16) * 16, v[i]);
        }
        wave_lds_fence();
        if ((r >> 4) == pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 t = lds_load4(scr + (r & 15) * IO::ROWB + (32 * nt + 8 * g + 4 * hh) * (int)sizeof(T), T());
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[nt][4 * g + j] = t[j];
                }
        }
    }
}

// Tile I/O for a wave whose 32 tokens are NOT consecutive in memory: an 8 x 4 block of one view image (the windowed
// attention's query block), taken as two 4 x 4 blocks side by side.  Row r of the tile = token 16 b + 4 py + px of k_spa_b:
// image row y0 + ((r >> 2) & 3), column x0 + 4 (r >> 4) + (r & 3).
struct BlkRows {
    int img_row_bytes;     // bytes from one image row to the next (w * bytes per token row)
    int tok_bytes;         // bytes per token row
    int nrow, ncol;        // rows (<= 4) and columns (<= 8) of the block that lie inside the image
    LFT_MEM static int py(int row) { return (row >> 2) & 3; }
    LFT_MEM static int px(int row) { return 4 * (row >> 4) + (row & 3); }
    LFT_MEM size_t off(int row) const { return (size_t)py(row) * img_row_bytes + (size_t)px(row) * tok_bytes; }
    LFT_MEM bool ok(int row) const { return py(row) < nrow && px(row) < ncol; }
};
// As load_tile / store_tile, with the row -> address map given by `rm` (offsets relative to gbase, which must be a
// readable address even when no row is valid).
template <int NT, typename T, typename RM>
LFT_DEV void load_tile_map(const T* __restrict__ gbase, const RM& rm, int lane, f32x16 (&a)[NT], char* scr) {
    using IO = TileIO<NT, T>;
    constexpr int NV = 16 * IO::P16 / 64;
    const int r = lane & 31, hh = lane >> 5;
    raw16 v[2][NV];                                                    // both passes' loads are issued up front: ONE memory round trip
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane, row = pass * 16 + idx / IO::P16, pc = idx % IO::P16;
            const bool in = rm.ok(row);
            const raw16 t = load_raw16(reinterpret_cast<const char*>(gbase) + (in ? rm.off(row) + pc * 16 : 0));
            v[pass][i] = in ? t : raw16{0u, 0u, 0u, 0u};
        }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        wave_lds_fence();                                              // previous pass fully consumed
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = i * 64 + lane;
            store_raw16(scr + (idx / IO::P16) * IO::ROWB + (idx % IO::P16) * 16, v[pass][i]);
        }
        wave_lds_fence();
        if ((r >> 4) == pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 t = lds_load4(scr + (r & 15) * IO::ROWB + (32 * nt + 8 * g + 4 * hh) * (int)sizeof(T), T());
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[nt][4 * g + j] = t[j];
                }
        }
    }
}
template <int NT, typename T, typename RM>
LFT_DEV void store_tile_map(T* __restrict__ gbase, const RM& rm, int lane, const f32x16 (&a)[NT], char* scr) {
    using IO = TileIO<NT, T>;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        wave_lds_fence();                                              // previous pass fully read
        if ((r >> 4) == pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    lds_store4(scr + (r & 15) * IO::ROWB + (32 * nt + 8 * g + 4 * hh) * (int)sizeof(T),
                               f32x4{a[nt][4 * g], a[nt][4 * g + 1], a[nt][4 * g + 2], a[nt][4 * g + 3]}, T());
        }
        wave_lds_fence();
        raw16 v[16 * IO::P16 / 64];
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane;
            v[i] = load_raw16(scr + (idx / IO::P16) * IO::ROWB + (idx % IO::P16) * 16);
        }
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane, row = pass * 16 + idx / IO::P16, pc = idx % IO::P16;
            if (rm.ok(row)) store_raw16(reinterpret_cast<char*>(gbase) + rm.off(row) + pc * 16, v[i]);
        }
    }
}

// Load a tile of rows as NATURAL-order B fragments (k = 16 ks + 8 h + j), e.g. the attention output.
template <int KS, typename T>
LFT_DEV void load_tile_frags(const T* __restrict__ gbase, int nvalid, int lane, Frag<T> (&f)[KS], char* scr) {
    using IO = TileIO<KS / 2, T>;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const char* g0 = reinterpret_cast<const char*>(gbase) + (size_t)pass * 16 * IO::ROW_BYTES;
        raw16 v[16 * IO::P16 / 64];
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane, row = idx / IO::P16;
            const bool in = pass * 16 + row < nvalid;
            const raw16 t = load_raw16(in ? g0 + (size_t)idx * 16 : reinterpret_cast<const char*>(gbase));
            v[i] = in ? t : raw16{0u, 0u, 0u, 0u};
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane;
            store_raw16(scr + (idx / IO::P16) * IO::ROWB + (idx % IO::P16) * 16, v[i]);
        }
        wave_lds_fence();
        if ((r >> 4) == pass) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) f[ks] = lds_row8(scr + (r & 15) * IO::ROWB + (16 * ks + 8 * hh) * (int)sizeof(T), true, T());
        }
    }
}

// Same with an explicit row stride: KS k-steps (16 * KS channels) out of wider rows.
template <int KS, typename T>
LFT_DEV void load_tile_frags_s(const T* __restrict__ gbase, size_t row_stride_bytes, int nvalid, int lane, Frag<T> (&f)[KS], char* scr) {
    using IO = TileIO<KS / 2, T>;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const char* g0 = reinterpret_cast<const char*>(gbase) + (size_t)pass * 16 * row_stride_bytes;
        raw16 v[16 * IO::P16 / 64];
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane, row = idx / IO::P16, pc = idx % IO::P16;
            const bool in = pass * 16 + row < nvalid;
            const raw16 t = load_raw16(in ? g0 + (size_t)row * row_stride_bytes + pc * 16 : reinterpret_cast<const char*>(gbase));
            v[i] = in ? t : raw16{0u, 0u, 0u, 0u};
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 16 * IO::P16 / 64; ++i) {
            const int idx = i * 64 + lane;
            store_raw16(scr + (idx / IO::P16) * IO::ROWB + (idx % IO::P16) * 16, v[i]);
        }
        wave_lds_fence();
        if ((r >> 4) == pass) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) f[ks] = lds_row8(scr + (r & 15) * IO::ROWB + (16 * ks + 8 * hh) * (int)sizeof(T), true, T());
        }
    }
}

template <int NT>
LFT_DEV void zero_acc(f32x16 (&a)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[nt][i] = 0.0f;
}

// 2^x for softmax arguments (x <= 0): the bare v_exp_f32.  exp2f() wraps it in ldexp / compare / select range
// handling (4 extra VALU instructions per call); results below 2^-126 flush to 0, which a softmax does not mind.
LFT_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// ReLU, LeakyReLU(0.2) and a plain maximum on values hipcc cannot prove canonical (MFMA results, inline-asm outputs):
// fmaxf() first quiets each such operand with a v_max_f32 x, x, x, and x > 0 ? x : 0.2 x is a compare, a multiply and a
// select -- in kernels bound by vector-instruction issue.  v_med3_f32 against +inf needs no preparation: the median of
// (x, 0, inf) is max(x, 0), of (x, 0.2 x, inf) the leaky form (any slope below 1; bit-identical to the select), of
// (a, b, inf) max(a, b).  The +inf comes from an opaque s_mov: given the constant, hipcc folds the median back into a
// canonicalising maximum.  (No inline-asm v_max on the values themselves: the hazard recogniser does not pad an asm
// statement that reads an MFMA result, and such a read returns stale registers.)
LFT_DEV float opaque_inf() { float v; asm("s_mov_b32 %0, 0x7f800000" : "=s"(v)); return v; }
LFT_DEV float relu_fast(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, opaque_inf()); }
LFT_DEV float lrelu02_fast(float x) { return __builtin_amdgcn_fmed3f(x, 0.2f * x, opaque_inf()); }
LFT_DEV float max_fast(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, opaque_inf()); }
// Exchange between the two 32-lane halves of a wave (token halves of the accumulator layout): gfx950's
// v_permlane32_swap_b32 swaps lanes 32..63 of its first operand with lanes 0..31 of its second, so from two copies of v it
// leaves (lower half's values in both halves, upper half's values in both halves) -- one VALU instruction instead of a
// ds_bpermute_b32 round trip through the LDS crossbar and the lgkmcnt wait behind it.  From inline asm: the builtin of
// this hipcc (ROCm 7.2) returns its first result twice.  s_nop 1: wait states between the VALU writes of the operands
// and the swap, as the compiler inserts for the builtin.
#ifndef LFT_XHALF_BPERMUTE
LFT_DEV void xhalf_split(float v, float& lo, float& hi) {
    lo = v; hi = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}
LFT_DEV float xhalf_sum(float v) { float lo, hi; xhalf_split(v, lo, hi); return lo + hi; }
LFT_DEV float xhalf_max(float v) { float lo, hi; xhalf_split(v, lo, hi); return max_fast(lo, hi); }
#else
LFT_DEV float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }
LFT_DEV float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
#endif

// ------------------------------------------------------------------------------------------
// 16-column MFMA forms (v_mfma_f32_16x16x16_{bf16,f16}, v_mfma_f32_16x16x32_{bf16,f16}) for the windowed attention's score
// tiles: a 4 x 4 query block sees an 8 x 8 key neighbourhood (64 keys), a 32-query block 96 -- a third fewer scores to
// exponentiate per query.  Lane maps (checked on the card by tools/micro/mfma_small.hip): lane l = 16 g + i holds
//   A[row i][k = K4 g + j],  B[k = K4 g + j][col i],  C / D[row 4 g + e][col i]        (K4 = 4 for x16, 8 for x32; e = 0..3)
// so a result tile (row on the register, column on the lane) is, converted, directly the B operand of a product that sums over
// its rows -- the same idiom as the 32-wide tiles.
// Two 16-column operands <-> one 32-column fragment: v_permlane16_swap_b32 D, S exchanges rows (16 lanes) 1, 3 of D with rows
// 0, 2 of S.  For a 32x32x16 fragment (lane 32 h + r: token r, elements j = 0..7) with X = elements 0..3 and Y = elements 4..7,
// swap(X, Y) leaves X = the 16x16x16 operand of tokens 0..15 (row g: k = 4 g .. 4 g + 3 <-> fragment labels 8 h + j in order) and
// Y = that of tokens 16..31; the same swap turns two 16-column result tiles back into one 32-column fragment.
// ------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4v __attribute__((ext_vector_type(4)));
LFT_DEV f32x4 mfma16k16(u32x2 a, u32x2 b, f32x4 c, bf16_t) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4v, a), __builtin_bit_cast(s16x4v, b), c, 0, 0, 0);
}
LFT_DEV f32x4 mfma16k16(u32x2 a, u32x2 b, f32x4 c, f16_t) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}
LFT_DEV f32x4 mfma16k32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
LFT_DEV f32x4 mfma16k32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// s_nop 1: wait states between the VALU writes of the operands and the swap (as xhalf_split).  The operands are always VALU
// results here (conversions, maxima, sums), never raw MFMA results (the hazard recogniser does not pad an asm statement).
LFT_DEV void swap16(unsigned& d, unsigned& s) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(d), "+v"(s)); }
LFT_DEV void swap16(float& d, float& s) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(d), "+v"(s)); }
LFT_DEV void swap32(float& d, float& s) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(d), "+v"(s)); }
// 32-column fragment (4 registers) -> the two 16-column operands in place: registers 0, 1 = tokens 0..15, registers 2, 3 = tokens 16..31
LFT_DEV void frag_to_blocks(raw16& f) {
    unsigned a0 = f[0], a1 = f[1], b0 = f[2], b1 = f[3];
    swap16(a0, b0);
    swap16(a1, b1);
    f = raw16{a0, a1, b0, b1};
}
// a, b: a lane's partial results for query column (lane & 15) of blocks A and B; on return EVERY lane holds, for its column, the
// combination over the four 16-lane rows -- a for block A, b for block B (3 swaps, 2 operations, 2 copies for both blocks).
template <typename OP> LFT_DEV void xrow_combine2(float& a, float& b, OP op) {
    swap16(a, b);                               // a = [A0 B0 A2 B2]   b = [A1 B1 A3 B3]   (by lane row)
    float c = op(a, b), c2 = c;                 // [A01 B01 A23 B23]
    swap32(c, c2);                              // c = [A01 B01 A01 B01]   c2 = [A23 B23 A23 B23]
    float d = op(c, c2), d2 = d;                // [A B A B]
    swap16(d, d2);                              // d = [A A A A]   d2 = [B B B B]
    a = d; b = d2;
}

// LayerNorm over the NT*32 channels of each token (biased variance, eps inside the sqrt, affine),
// as nn.LayerNorm does (reference LFT.py:127,136,199,208).  In place.  gamma/beta may point to LDS (kernels
// copy the small parameter vectors there at start: a global load in the middle of a kernel would force a
// vmcnt(0) that also drains the weight ring's in-flight LDS-DMA).
// FAST (bf16 path): v_rsq_f32 / v_rcp_f32 (1 ulp) instead of sqrt + IEEE division (a dozen instructions per token; the
// kernels that call this are bound by vector-instruction issue).  The fp32 parity path keeps the exact forms.
LFT_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// `bad` (sticky, per lane): set when the token's variance is not finite, i.e. some channel of the token is inf / NaN.  Every
// activation of the residual stream passes through one of these LayerNorms in the following block, and an fp16 range
// overflow anywhere upstream (a finite fp32 value converted to +-inf) turns into inf / NaN there: one compare per token is
// the network's overflow detector (lft_status_read; the last block's output is checked by k_assemble_t).
LFT_DEV unsigned not_finite(float v) { return !(__builtin_fabsf(v) < 3.0e38f); }
template <int NT, bool FAST = false>
LFT_DEV void layernorm_acc(f32x16 (&a)[NT], const float* gamma, const float* beta, int h, unsigned& bad) {
    if constexpr (FAST) {
        // 16-bit paths: every pass on register pairs (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32), the centred value kept
        // from the variance pass, gamma folded into the scale: 2.5 vector instructions per element instead of 6 -- these
        // kernels are bound by vector-instruction issue.  ((x - mean) * (rstd * gamma) + beta: last-bit differences from the
        // exact form below, far inside the 16-bit operand rounding that follows.)
        f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) s2 += f32x2{a[nt][i], a[nt][i + 1]};
        const float mean = xhalf_sum(s2[0] + s2[1]) * (1.0f / (NT * 32));
        const f32x2 mean2 = {mean, mean};
        f32x2 q2 = {0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 d = f32x2{a[nt][i], a[nt][i + 1]} - mean2;
                q2 = p2_fma(d, d, q2);
                a[nt][i] = d[0]; a[nt][i + 1] = d[1];
            }
        const float var = xhalf_sum(q2[0] + q2[1]) * (1.0f / (NT * 32)) + LFT_LN_EPS;
        bad |= not_finite(var);
        const float rstd = __builtin_amdgcn_rsqf(var);
        const f32x2 rstd2 = {rstd, rstd};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gm = __builtin_bit_cast(f32x4, load_raw16(reinterpret_cast<const char*>(gamma + 32 * nt + 8 * g + 4 * h)));
                const f32x4 bt = __builtin_bit_cast(f32x4, load_raw16(reinterpret_cast<const char*>(beta + 32 * nt + 8 * g + 4 * h)));
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 sc = f32x2{gm[j], gm[j + 1]} * rstd2;
                    const f32x2 o = p2_fma(f32x2{a[nt][4 * g + j], a[nt][4 * g + j + 1]}, sc, f32x2{bt[j], bt[j + 1]});
                    a[nt][4 * g + j] = o[0]; a[nt][4 * g + j + 1] = o[1];
                }
            }
        return;
    }
    float s = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[nt][i];
    const float mean = xhalf_sum(s) * (1.0f / (NT * 32));
    float q = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { float d = a[nt][i] - mean; q += d * d; }
    const float var = xhalf_sum(q) * (1.0f / (NT * 32)) + LFT_LN_EPS;
    bad |= not_finite(var);
    const float rstd = FAST ? __builtin_amdgcn_rsqf(var) : 1.0f / sqrtf(var);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gm = __builtin_bit_cast(f32x4, load_raw16(reinterpret_cast<const char*>(gamma + 32 * nt + 8 * g + 4 * h)));
            const f32x4 bt = __builtin_bit_cast(f32x4, load_raw16(reinterpret_cast<const char*>(beta + 32 * nt + 8 * g + 4 * h)));
#pragma unroll
            for (int j = 0; j < 4; ++j) a[nt][4 * g + j] = (a[nt][4 * g + j] - mean) * rstd * gm[j] + bt[j];
        }
}
// Publish a kernel's sticky "non-finite activation seen" bit (status may be null: per-stage entry points without a workspace).
// Many lanes may store the same word; the value only ever goes from 0 to non-zero until lft_status_reset.
constexpr unsigned LFT_STATUS_NONFINITE_BIT = 1u;
LFT_DEV void publish_status(unsigned* status, unsigned bad) {
    if (bad && status) *status = LFT_STATUS_NONFINITE_BIT;
}
// LayerNorm parameters global -> LDS in two halves: the load is issued early with everything else, the LDS
// store only after the kernel's one big vmcnt wait (a load -> ds_write pair in the prologue costs a full memory
// round trip there, vmcnt being in-order).  n % 4 == 0, n <= 1024.  The caller's barrier publishes the store.
LFT_DEV raw16 params_load(const float* __restrict__ src, int n) {
    const int i = min((int)threadIdx.x * 4, n - 4);     // threads beyond n/4 re-read the last piece and store nothing
    return load_raw16(reinterpret_cast<const char*>(src + i));
}
LFT_DEV void params_store(float* lds_dst, int n, raw16 v) {
    const int i = threadIdx.x * 4;
    if (i < n) store_raw16(reinterpret_cast<char*>(lds_dst + i), v);
}

// all 2*NT k-steps of an accumulator-resident activation as fragments
template <int NT, typename T>
LFT_DEV void acc_frags(const f32x16 (&a)[NT], Frag<T> (&f)[2 * NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f[2 * nt] = acc_to_frag(a[nt], 0, T());
        f[2 * nt + 1] = acc_to_frag(a[nt], 1, T());
    }
}

// ReLU'd activation as fragments.  16-bit operands: convert first, then ReLU on the PACKED halves as signed 16-bit integers
// (v_pk_max_i16 against 0: a negative bf16 / f16 has its sign bit set, i.e. is a negative integer; -0.0 becomes +0) -- one
// instruction per two values instead of one v_med3_f32 per value, in kernels bound by vector-instruction issue.  Rounding
// and ReLU commute (rounding is monotonic and keeps the sign), so the fragments are bit-identical to relu-then-convert.
template <int NT>
LFT_DEV void acc_frags_relu(const f32x16 (&a)[NT], Frag<float> (&f)[2 * NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[2 * nt + s].lo[j] = relu_fast(a[nt][8 * s + j]); f[2 * nt + s].hi[j] = relu_fast(a[nt][8 * s + 4 + j]); }
        }
}
template <int NT, typename T>
LFT_DEV void acc_frags_relu(const f32x16 (&a)[NT], Frag<T> (&f)[2 * NT]) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    acc_frags<NT, T>(a, f);
#pragma unroll
    for (int k = 0; k < 2 * NT; ++k) {
        const s16x8 v = __builtin_bit_cast(s16x8, f[k].v);
        f[k].v = __builtin_bit_cast(typename H16<T>::v8, __builtin_elementwise_max(v, (s16x8)(0)));
    }
}

// Same, weights taken in stream order from the workgroup's LDS ring.
template <int NT_OUT, int KS, typename T, int CH, int NW>
LFT_DEV void linear_ring(WRing<T, CH, NW>& ring, const Frag<T> (&x)[KS], f32x16 (&y)[NT_OUT]) {
#pragma unroll
    for (int nt = 0; nt < NT_OUT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) mma(ring.next(), x[ks], y[nt]);
}

// Y^T[nt] += sum_ks W(nt, ks) * x[ks]; stream fragments ordered nt-major, starting at f0.
template <int NT_OUT, int KS, typename T>
LFT_DEV void linear_acc(const T* __restrict__ w, int f0, int lane, const Frag<T> (&x)[KS], f32x16 (&y)[NT_OUT]) {
#pragma unroll
    for (int nt = 0; nt < NT_OUT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) mma(load_wfrag(w, f0 + nt * KS + ks, lane), x[ks], y[nt]);
}
