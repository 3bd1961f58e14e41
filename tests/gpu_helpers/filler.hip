// Test helper (never part of the product): a co-tenant kernel that HOLDS compute-unit resources for a bounded time, so
// that tests can check what libsr3hip does when the CU slots its in-place split-K convs rely on are taken by somebody
// else (tests/test_gpu_round4.py; VERDICT r3 "next" item 4). Built by __graft_entry__.build() into
// tests/gpu_helpers/libsr3_test_filler.so and loaded with ctypes in the same process as the library under test.
//
// Every block is one wavefront that owns `lds_bytes` of LDS and spins on the constant 100 MHz real-time counter until
// `ticks` have passed since the block started — an exit condition every wave reaches, whatever else runs.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void hold_kernel(long long ticks, unsigned *started) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;                                       // (the allocation is what matters)
        atomicAdd(started, 1u);
    }
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// one wave that waits `ticks` and then ORs `bits` into a device word: raises libsr3hip's device flag in the MIDDLE of a
// running call (address from sr3_test_flag_address), so that the replay logic of sr3_sample runs deterministically
__global__ __launch_bounds__(64) void poke_kernel(int *word, int bits, long long ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) atomicOr(word, bits);
}

// The helper's stream must not share a hardware queue with the stream of the library under test (ROCm maps streams onto a
// few hardware queues round-robin; two streams on one queue execute in submission order — no co-tenancy at all, and a poke
// submitted first would land before the call it is meant to interrupt). Streams of a different PRIORITY get queues of
// their own, so the helper's is created with the highest one.
static hipStream_t g_stream = nullptr;
static bool make_stream() {
    if (g_stream) return true;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);       // (numerically lowest = highest priority)
    return hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, hi) == hipSuccess;
}
static unsigned *g_started = nullptr;      // host-visible counter of blocks that have begun to run

extern "C" {

// launches `blocks` holders of `lds_bytes` LDS each for `ticks` x 10 ns on a stream of their own; returns 0 on success
int filler_launch(int blocks, int lds_bytes, long long ticks) {
    if (!make_stream()) return -1;
    if (!g_started && hipHostMalloc(reinterpret_cast<void **>(&g_started), sizeof(unsigned), hipHostMallocDefault) != hipSuccess) return -2;
    *g_started = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -3;
    hipLaunchKernelGGL(hold_kernel, dim3(blocks), dim3(64), lds_bytes, g_stream, ticks, g_started);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int filler_poke(void *word, int bits, long long ticks) {
    if (!make_stream()) return -1;
    hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(64), 0, g_stream, reinterpret_cast<int *>(word), bits, ticks);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
// blocks that have started so far (poll until it reaches the number that fits the chip)
unsigned filler_started(void) { return g_started ? *reinterpret_cast<volatile unsigned *>(g_started) : 0u; }
int filler_wait(void) { return g_stream && hipStreamSynchronize(g_stream) == hipSuccess ? 0 : -1; }

}
