// Debugging aid (round 6, the k_step4_act_enc flake of journal r05 section 7): fill the queue's PRIVATE SEGMENT (scratch) with a pattern,
// from a kernel of its own, right in front of the launch under test.  Scratch is one buffer per hardware queue that successive
// kernels carve up wave slot by wave slot and that nobody clears: a kernel that reloads a spill slot some lanes never stored (a
// spill under a partial EXEC mask, a callee-saved register saved by part of the wave) reads what the PREVIOUS kernel left there -
// its own earlier launch leaves plausible values (lane ids, addresses, constants repeat), another library's kernel does not.
// That is exactly "wrong only as the first launch behind somebody else's kernels".
//
//   hipcc --offload-arch=gfx950 -O1 -shared -fPIC scripts/micro/poison_scratch.hip -o scripts/micro/libpoison_scratch.so
//
// poison_scratch(stream, pattern): as many waves as the chip holds at once, each writes `pattern ^ lane ^ word` to WORDS dwords
// of its private segment and stays resident for a while, so that every scratch wave slot is handed out once.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int WORDS>
__global__ __launch_bounds__(64) void k_poison(uint32_t pat, uint32_t salt, uint32_t* sink) {
    volatile uint32_t buf[WORDS];
    const uint32_t v = salt ? (pat ^ (threadIdx.x * 0x9E3779B9u)) : pat;
#pragma unroll 1
    for (int i = 0; i < WORDS; i++) buf[i] = salt ? v + (uint32_t)i * salt : v;
    for (int k = 0; k < 40; k++) __builtin_amdgcn_s_sleep(127);   // ~40 x 127 x 64 clocks: long enough for the grid to fill every slot
    uint32_t acc = 0;
#pragma unroll 1
    for (int i = 0; i < WORDS; i += 61) acc += buf[i];
    if (sink && acc == 0x12345u) *sink = acc;
}

// filler(stream, blocks, ticks): `blocks` one-wave workgroups that do nothing for `ticks` of the 100 MHz clock (no scratch, no LDS, a
// handful of registers): launched on ANOTHER stream they take the first wave slot of the SIMDs they land on, so that the waves of the
// launch under test start in slot 1 - next to a foreign wave instead of a sibling of their own kernel.
__global__ __launch_bounds__(64) void k_filler(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(20);
}
extern "C" int filler(void* stream, uint32_t blocks, unsigned long long ticks) {
    hipLaunchKernelGGL(k_filler, dim3(blocks), dim3(64), 0, (hipStream_t)stream, ticks);
    return (int)hipGetLastError();
}

extern "C" int poison_scratch(void* stream, uint32_t bytes_per_lane, uint32_t pattern, uint32_t salt, uint32_t blocks) {
    hipStream_t s = (hipStream_t)stream;
    if (blocks == 0) blocks = 256u * 32u * 2u;
    if (bytes_per_lane <= 1024) hipLaunchKernelGGL(k_poison<256>, dim3(blocks), dim3(64), 0, s, pattern, salt, (uint32_t*)nullptr);
    else if (bytes_per_lane <= 2048) hipLaunchKernelGGL(k_poison<512>, dim3(blocks), dim3(64), 0, s, pattern, salt, (uint32_t*)nullptr);
    else hipLaunchKernelGGL(k_poison<1024>, dim3(blocks), dim3(64), 0, s, pattern, salt, (uint32_t*)nullptr);
    return (int)hipGetLastError();
}
