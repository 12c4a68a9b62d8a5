// Micro-benchmark (design tool, not product code): ceiling of a persistent "weights resident in LDS" token-GEMM chain.
// One workgroup per CU holds NFRAG 1-KiB bf16 weight fragments in LDS (one LDS-DMA prologue); each wave walks over
// 32-token tiles on its own (no workgroup barrier after the prologue): 16 x 16 B/lane of loads, a chain of MFMAs whose
// A operand comes from LDS (ds_read_b128) and whose B operand is the previous accumulator ("token on lane"), some
// VALU in between (relu + bf16 conversion), 8 x 16 B/lane of stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb_resident.hip -o ab_so/mb_resident && ab_so/mb_resident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int raw16 __attribute__((ext_vector_type(4), may_alias));

template <int NW, int NFRAG, bool VALU>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void k_chain(const char* __restrict__ wsrc, const char* __restrict__ in, char* __restrict__ out,
                                                                   int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int p = wave; p < NFRAG; p += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int tile = wave * gridDim.x + blockIdx.x; tile < ntiles; tile += NW * gridDim.x) {
        asm volatile("" ::: "memory");
        const char* src = in + (size_t)tile * 16384;
        bf16x8 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const raw16*>(src + i * 1024 + lane * 16));
        f32x16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
        int f = 0;
        for (int rep = 0; rep < reps; ++rep) {
            // 4 output row tiles x 8 k-steps = 32 MFMAs from the 8 fragments of x[], then the accumulators become the next operand
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const raw16*>(smem + ((f + n * 8 + ks) % NFRAG) * 1024 + lane * 16));
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x[ks], acc[n], 0, 0, 0);
                }
            f = (f + 32) % NFRAG;
            if (VALU) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[2 * n + s][j] = (__bf16)fmaxf(acc[n][8 * s + j], 0.0f);
            } else {
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(acc[n]));
            }
        }
        char* dst = out + (size_t)tile * 8192;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)acc[n][8 * s + j];
                *reinterpret_cast<raw16*>(dst + (2 * n + s) * 1024 + lane * 16) = __builtin_bit_cast(raw16, v);
            }
    }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <int NW, int NFRAG, bool VALU>
int run(const char* name, char* w, char* in, char* out, int ntiles, int reps) {
    const size_t lds = (size_t)NFRAG * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<NW, NFRAG, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) k_chain<NW, NFRAG, VALU><<<256, 64 * NW, lds>>>(w, in, out, ntiles, reps);
    CK(hipDeviceSynchronize());
    const int N = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) k_chain<NW, NFRAG, VALU><<<256, 64 * NW, lds>>>(w, in, out, ntiles, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / N, mfma = (double)ntiles * reps * 32, flop = mfma * 32768.0;
    printf("%-34s NW=%2d frags=%3d tiles=%d mfma/tile=%3d : %7.2f us  %6.1f TFLOP/s  (%4.1f%% of 2.5 PF)\n", name, NW, NFRAG, ntiles, reps * 32, us,
           flop / us * 1e-6, flop / us * 1e-6 / 2500 * 100);
    return 0;
}

int main() {
    const int ntiles = 3200;
    char *w, *in, *out;
    CK(hipMalloc(&w, 160 * 1024)); CK(hipMalloc(&in, (size_t)ntiles * 16384)); CK(hipMalloc(&out, (size_t)ntiles * 8192));
    std::vector<unsigned short> h(160 * 512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 22);      // bf16 values around 0.01 .. 0.03
    CK(hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<unsigned short> hin((size_t)ntiles * 8192);
    for (size_t i = 0; i < hin.size(); ++i) hin[i] = 0x3f00 + (unsigned short)((i * 40503u) & 0xff) + ((i & 1) ? 0x8000 : 0);
    CK(hipMemcpy(in, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
    // spa2-like: 144 KiB of weights, 4.5 x 32 = 144 MFMAs per tile -> reps 4 (128) and 5 (160) bracket it
    if (run<8, 144, true>("resident, relu+cvt between", w, in, out, ntiles, 4)) return 1;
    if (run<8, 144, true>("resident, relu+cvt between", w, in, out, ntiles, 5)) return 1;
    if (run<8, 144, false>("resident, MFMA only", w, in, out, ntiles, 5)) return 1;
    if (run<4, 144, true>("resident, relu+cvt between", w, in, out, ntiles, 5)) return 1;
    if (run<12, 144, true>("resident, relu+cvt between", w, in, out, ntiles, 5)) return 1;
    if (run<16, 144, true>("resident, relu+cvt between", w, in, out, ntiles, 5)) return 1;
    if (run<8, 96, true>("resident (qkv-like)", w, in, out, ntiles, 3)) return 1;
    if (run<8, 144, true>("resident, 2x tiles (B=8)", w, in, out, 2 * ntiles > 3200 ? 3200 : 3200, 5)) return 1;
    return 0;
}
