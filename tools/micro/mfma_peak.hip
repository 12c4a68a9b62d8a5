// Sustained MFMA rate of this card: every SIMD runs NW waves of back-to-back MFMAs on 4 independent accumulators, no memory.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o ab_so/mfma_peak && ab_so/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k_peak(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    const float x = 1.0f + threadIdx.x * 1e-6f, y = 1.0f - threadIdx.x * 1e-6f;
    bf16x8 xb, yb;
    for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)x; yb[i] = (__bf16)y; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if constexpr (MODE == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[a], 0, 0, 0);
            }
    }
    float s = 0.0f;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE>
double run(int blocks, int iters) {
    float* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_peak<MODE><<<blocks, 256>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_peak<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 32 * (MODE == 0 ? 32.0 * 32 * 2 * 2 : 32.0 * 32 * 16 * 2);
    hipFree(d);
    return flop / (ms * 1e-3) / 1e12;
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    std::printf("%s: %d CUs, clock %d MHz\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000);
    for (int wpc : {1, 2, 4}) {                         // workgroups (of 4 waves) per CU
        const int blocks = pr.multiProcessorCount * wpc;
        for (int rep = 0; rep < 3; ++rep)
            std::printf("wg/CU %d  fp32 32x32x2: %7.1f TFLOP/s   bf16 32x32x16: %7.1f TFLOP/s\n", wpc, run<0>(blocks, 20000), run<1>(blocks, 20000));
    }
    // a long run (about a second): the rate after the card's power management has settled
    std::printf("long  fp32: %7.1f TFLOP/s   bf16: %7.1f TFLOP/s\n", run<0>(pr.multiProcessorCount * 2, 600000), run<1>(pr.multiProcessorCount * 2, 600000));
    return 0;
}
