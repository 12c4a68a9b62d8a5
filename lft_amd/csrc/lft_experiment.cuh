// lft_experiment.cuh -- diagnostic builds ONLY (-DLFT_EXPERIMENT; tools/stamp_report.py).  The product library never includes
// this file: lft_common.cuh pulls it in under #ifdef LFT_EXPERIMENT, and lft_amd/_lib.py never defines that macro.
//
// Per-phase s_memtime stamps of wave 0 of every workgroup go to a side buffer that no kernel reads; the report tool copies
// the buffer to the host through lft_debug_read_stamps (defined at the end of lft_api.hip under the same macro).
#pragma once

__device__ unsigned long long g_lft_stamps[4096 * 32];    // 32 slots per workgroup: k_spa1 uses 0..15, k_spa_b / k_spa2 16..31
static __device__ __forceinline__ void lft_stamp(int slot) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_lft_stamps[blockIdx.x * 32 + slot] = t;
}
#define LFT_STAMP(slot) lft_stamp(slot)
