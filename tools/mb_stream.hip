// Micro-benchmark (design tool, not product code): what HBM-side rate do the spatial kernels' tile stores / loads reach on
// their own?  Each wave stores (or loads) NP 1-KiB pieces (16 B per lane), 4 waves per workgroup, like k_spa1's four
// lane-major output tiles of 32 tokens x 128 channels (NP = 32) or k_spa_b's input side.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb_stream.hip -o ab_so/mb_stream && ab_so/mb_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int raw16 __attribute__((ext_vector_type(4), may_alias));

template <int NP, bool STORE>
__global__ __launch_bounds__(256) void k_stream(char* __restrict__ buf, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = buf + ((size_t)blockIdx.x * 4 + wave) * NP * 1024 + lane * 16;
    raw16 acc = raw16{1u, 2u, 3u, (unsigned)lane};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (STORE) *reinterpret_cast<raw16*>(base + i * 1024) = acc;
        else { const raw16 v = *reinterpret_cast<const raw16*>(base + i * 1024); acc ^= v; }
    }
    if (!STORE && acc[0] == 0x12345u) sink[0] = acc[1];
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <int NP, bool STORE>
int run(const char* name, char* buf, unsigned* sink, int nwg, size_t stride_bufs) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t bytes = (size_t)nwg * 4 * NP * 1024;
    for (int i = 0; i < 3; ++i) k_stream<NP, STORE><<<nwg, 256>>>(buf + (i % stride_bufs) * bytes, sink);
    CK(hipDeviceSynchronize());
    const int N = 24;
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) k_stream<NP, STORE><<<nwg, 256>>>(buf + (i % stride_bufs) * bytes, sink);   // rotate over buffers: no reuse from cache
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / N;
    printf("%-40s %4d WGs x 4 waves x %2d KiB = %6.1f MB, %zu rotating buffers: %7.2f us  %6.2f TB/s\n", name, nwg, NP, bytes / 1e6, stride_bufs, us, bytes / us * 1e-6);
    return 0;
}

int main() {
    char* buf; unsigned* sink;
    const size_t cap = (size_t)12 * 110 * 1024 * 1024;
    CK(hipMalloc(&buf, cap)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, cap));
    // k_spa1's store side: 800 workgroups x 4 waves x 32 KiB = 105 MB
    if (run<32, true>("stores, k_spa1-sized", buf, sink, 800, 12)) return 1;
    if (run<32, true>("stores, same buffer every launch", buf, sink, 800, 1)) return 1;
    if (run<32, true>("stores, two alternating buffers", buf, sink, 800, 2)) return 1;
    if (run<32, false>("loads, k_spa_b-sized", buf, sink, 800, 12)) return 1;
    if (run<32, false>("loads, same buffer every launch", buf, sink, 800, 1)) return 1;
    if (run<8, true>("stores, 26 MB (one [N,128] bf16 tensor)", buf, sink, 800, 12)) return 1;
    if (run<8, true>("stores, 26 MB, same buffer", buf, sink, 800, 1)) return 1;
    if (run<32, true>("stores, 3200 WGs (420 MB)", buf, sink, 3200, 3)) return 1;
    return 0;
}
