// Facts the 16-query score tiles of k_spa_b rest on, checked on the card:
//   1. operand / result lane maps of v_mfma_f32_16x16x16_bf16 (A[m = l%16][k = 4(l/16)+j], B[k][n = l%16], C[row 4(l/16)+i][col l%16])
//   2. v_permlane16_swap_b32 D, S: rows (16 lanes) 1 and 3 of D are exchanged with rows 0 and 2 of S
//   3. ds_read_b64_tr_b16 as the A operand of the 16x16x16 product (lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3)
//   4. issue cost: cycles per MFMA (one wave per SIMD, back to back) of 16x16x16, 16x16x32 and 32x32x16, alone and with v_exp fillers
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_small.hip -o ab_so/mfma_small && ab_so/mfma_small
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* A, const float* B, float* C, float* Ctr, unsigned* sw) {
    // A [16][16] row-major (m, k), B [16][16] (k, n)
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    bf16x4 a, b;
    for (int j = 0; j < 4; ++j) { a[j] = (__bf16)A[i * 16 + 4 * g + j]; b[j] = (__bf16)B[(4 * g + j) * 16 + i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];
    // permlane16 swap
    unsigned d = 1000 + l, s = 2000 + l;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(d), "+v"(s));
    sw[l] = d; sw[64 + l] = s;
    // transposed read: LDS image Vt[key][16 ch] bf16 with 64-byte rows (only the first 32 B used); operand A'[m = ch][k = key] = A^T
    __shared__ __attribute__((aligned(16))) __bf16 img[16 * 32];
    for (int e = l; e < 256; e += 64) img[(e / 16) * 32 + (e % 16)] = (__bf16)A[(e % 16) * 16 + (e / 16)];   // img[key][ch] = A[ch][key]
    __syncthreads();
    const int q = i >> 2, p = i & 3;
    const s16x4 at = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + (4 * g + q) * 32 + 4 * p));
    f32x4 c2 = {0, 0, 0, 0};
    c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at, __builtin_bit_cast(s16x4, b), c2, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Ctr[(4 * g + r) * 16 + i] = c2[r];
}

template <int MODE, int NEXP>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    f32x4 a4[4];
    f32x16 a16[4];
    for (int a = 0; a < 4; ++a) { for (int i = 0; i < 4; ++i) a4[a][i] = 0; for (int i = 0; i < 16; ++i) a16[a][i] = 0; }
    bf16x8 x8, y8;
    for (int i = 0; i < 8; ++i) { x8[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); y8[i] = (__bf16)(1.0f - threadIdx.x * 1e-3f); }
    bf16x4 x4 = {x8[0], x8[1], x8[2], x8[3]}, y4 = {y8[0], y8[1], y8[2], y8[3]};
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = -1.0f - i - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if constexpr (MODE == 0) a4[a] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, x4), __builtin_bit_cast(s16x4, y4), a4[a], 0, 0, 0);
                else if constexpr (MODE == 1) a4[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a4[a], 0, 0, 0);
                else if constexpr (MODE == 2) a16[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, y8, a16[a], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NEXP; ++f) e[(a * NEXP + f) & 7] = __builtin_amdgcn_exp2f(e[(a * NEXP + f) & 7]) - 2.0f;
            }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) { for (int i = 0; i < 4; ++i) s += a4[a][i]; for (int i = 0; i < 16; ++i) s += a16[a][i]; }
    for (int i = 0; i < 8; ++i) s += e[i];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE, int NEXP> double cycles_per_mfma(int blocks, int iters, double ghz) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<MODE, NEXP><<<blocks, 256>>>(d, 64); hipDeviceSynchronize();
    hipEventRecord(e0); k_rate<MODE, NEXP><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d);
    return ms * 1e-3 * ghz * 1e9 / ((double)iters * 16);
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double ghz = pr.clockRate / 1e6;
    float hA[256], hB[256], hC[256], hCt[256], ref[256], refT[256];
    unsigned hsw[128];
    srand(1);
    for (int i = 0; i < 256; ++i) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 5 - 2); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        float s = 0, t = 0;
        for (int k = 0; k < 16; ++k) { s += hA[m * 16 + k] * hB[k * 16 + n]; t += hA[m * 16 + k] * hB[k * 16 + n]; }
        ref[m * 16 + n] = s; refT[m * 16 + n] = t;
    }
    float *dA, *dB, *dC, *dCt; unsigned* dsw;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dCt, 1024); hipMalloc(&dsw, 512);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dC, dCt, dsw);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost); hipMemcpy(hCt, dCt, 1024, hipMemcpyDeviceToHost); hipMemcpy(hsw, dsw, 512, hipMemcpyDeviceToHost);
    int bad = 0, badt = 0;
    for (int i = 0; i < 256; ++i) { bad += hC[i] != ref[i]; badt += hCt[i] != refT[i]; }
    std::printf("16x16x16 bf16 lane maps: %s (%d mismatches); tr16 read as its A operand: %s (%d)\n", bad ? "WRONG" : "ok", bad, badt ? "WRONG" : "ok", badt);
    int swbad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4;
        const unsigned wd = (row & 1) ? 2000 + (l - 16) : 1000 + l;       // D: odd rows now hold S's even rows
        const unsigned ws = (row & 1) ? 2000 + l : 1000 + (l + 16);       // S: even rows now hold D's odd rows
        swbad += hsw[l] != wd; swbad += hsw[64 + l] != ws;
    }
    std::printf("v_permlane16_swap_b32 D.rows{1,3} <-> S.rows{0,2}: %s (%d)  D: %u %u %u %u  S: %u %u %u %u\n", swbad ? "WRONG" : "ok", swbad,
                hsw[0], hsw[16], hsw[32], hsw[48], hsw[64], hsw[80], hsw[96], hsw[112]);
    const int blocks = pr.multiProcessorCount;           // one 4-wave workgroup per CU: one wave per SIMD
    std::printf("cycles per MFMA at %.1f GHz nominal, one wave per SIMD (wall clock: relative values matter)\n", ghz);
    std::printf("  16x16x16 bf16: %.1f   +1 exp: %.1f   +2 exp: %.1f   +4 exp: %.1f\n", cycles_per_mfma<0, 0>(blocks, 200000, ghz), cycles_per_mfma<0, 1>(blocks, 200000, ghz),
                cycles_per_mfma<0, 2>(blocks, 200000, ghz), cycles_per_mfma<0, 4>(blocks, 200000, ghz));
    std::printf("  16x16x32 bf16: %.1f   +1 exp: %.1f   +2 exp: %.1f   +4 exp: %.1f\n", cycles_per_mfma<1, 0>(blocks, 200000, ghz), cycles_per_mfma<1, 1>(blocks, 200000, ghz),
                cycles_per_mfma<1, 2>(blocks, 200000, ghz), cycles_per_mfma<1, 4>(blocks, 200000, ghz));
    std::printf("  32x32x16 bf16: %.1f   +1 exp: %.1f   +2 exp: %.1f   +4 exp: %.1f\n", cycles_per_mfma<2, 0>(blocks, 200000, ghz), cycles_per_mfma<2, 1>(blocks, 200000, ghz),
                cycles_per_mfma<2, 2>(blocks, 200000, ghz), cycles_per_mfma<2, 4>(blocks, 200000, ghz));
    std::printf("  no MFMA, exp only (per 'slot'): 1: %.1f  4: %.1f\n", cycles_per_mfma<3, 1>(blocks, 200000, ghz), cycles_per_mfma<3, 4>(blocks, 200000, ghz));
    return 0;
}
