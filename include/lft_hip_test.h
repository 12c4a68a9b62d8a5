/* lft_hip_test.h -- test-only entry points of liblft_hip.so (NOT part of the product ABI of lft_hip.h; no reference counterpart).
 * The library exports them for tests/ and tools/ only: a kernel-level conv launch for stress tools and the MFMA fragment-layout
 * self test the parity suite starts with. */
#ifndef LFT_HIP_TEST_H
#define LFT_HIP_TEST_H
#ifdef __cplusplus
extern "C" {
#endif

/* Debug aid: a single conv_init[which] launch (with_res: add `res`; extra_lds: pad the LDS request). */
int lft_debug_conv64(const void* packed, int which, int with_res, const void* in, const void* res, void* out,
                     int B, int A, int h, int w, int s, int prec, int extra_lds, void* stream);
/* MFMA fragment-layout self test: C = Am[32x16] * Bm[16x32], D = W2[32x32] * C.  All fp32 device buffers. */
int lft_mfma_selftest(const float* Am, const float* Bm, const float* W2, float* C, float* D, int prec, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LFT_HIP_TEST_H */
