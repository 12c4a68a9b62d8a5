// lft_experiment.cuh -- diagnostic builds ONLY (-DLFT_EXPERIMENT; tools/stamp_report.py).  The product library never includes
// this file: lft_common.cuh pulls it in under #ifdef LFT_EXPERIMENT, and lft_amd/_lib.py never defines that macro.
//
// Per-phase s_memtime stamps of wave 0 of every workgroup go to a side buffer that no kernel reads; the report tool copies
// the buffer to the host through lft_debug_read_stamps (defined at the end of lft_api.hip under the same macro).
#pragma once

#ifdef LFT_ISA_MARKS
// Static instruction budget per phase (tools/isa_mix.py --phases): every LFT_STAMP becomes a comment in the assembly listing,
// fenced by scheduling barriers so that a phase's instructions stay between its marks.  Nothing is emitted; the build is
// only ever compiled to a listing, never linked.
#define LFT_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; LFT_MARK %0" :: "n"(slot)); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LFT_STAMP_IT(slot, it_ofs) LFT_STAMP(slot)       // a loop body's marks: the same phase in every iteration
#else
__device__ unsigned long long g_lft_stamps[4096 * 32];    // 32 slots per workgroup: k_spa1 uses 0..15, k_spa_b / k_spa2 16..31
static __device__ __forceinline__ void lft_stamp(int slot) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_lft_stamps[blockIdx.x * 32 + slot] = t;
}
#define LFT_STAMP(slot) lft_stamp(slot)
#define LFT_STAMP_IT(slot, it_ofs) lft_stamp((slot) + (it_ofs))
#endif
