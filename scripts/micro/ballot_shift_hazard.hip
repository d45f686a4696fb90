// Round 6 micro-probe: rballot() of the four-games-per-wave step is compiled to
//     v_cmp_ne_u32_e32 vcc, 0, vX ; s_nop N ; v_lshrrev_b64 v[2:3], v87, vcc ; v_bfe_u32 v0, v2, v48, 16
// (the ballot as a 64-bit VALU operand, shifted by the lane's row half, v87 = lane & 32).  The library built with
// -mllvm -disable-machine-licm and a 96-register budget for k_step4_act_enc saw rows 2 / 3 (the upper half of the ballot) pick up
// other rows' bits - only in waves that were not the first of their SIMD and only when the wave was allocated exactly the 88 registers
// it names.  This probe runs that sequence in a grid of one-wave blocks and counts wrong slices per wave slot, varying: the wait
// states between compare and shift, where the ballot lives (vcc / an SGPR pair written by the VALU / an SGPR pair copied by the SALU),
// the register that holds the shift amount (the TOP allocated one or a middle one), 64-bit shift vs 32-bit select, and the
// allocation (88 / 96 registers).  The expected slice is computed per lane from the generator's state, without any cross-lane or
// SGPR traffic, so the reference cannot share the hazard.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/ballot_shift_hazard.hip -o scripts/micro/ballot_shift_hazard && scripts/micro/ballot_shift_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define NVAR 12
#define STR2(x) #x
#define STR(x) STR2(x)
#define PRE(AMT) "v_and_b32 v" AMT ", 32, %[lane]\n\tv_and_b32 v48, 16, %[lane]\n\t"
#define CMP_VCC "v_cmp_ne_u32_e32 vcc, 0, %[p]\n\t"
#define CMP_S "v_cmp_ne_u32_e64 s[20:21], 0, %[p]\n\t"
#define SH(AMT, SRC) "v_lshrrev_b64 v[2:3], v" AMT ", " SRC "\n\tv_bfe_u32 %[res], v2, v48, 16"
#define ASM(BODY) asm volatile(BODY : [res] "=&v"(r) : [lane] "v"(lane), [p] "v"(p) : "v2", "v3", "v40", "v48", "v" STR(TOP), "v" STR(TOP1), "v" STR(TOP2), "v" STR(TOP8), "vcc", "s20", "s21", "s22", "s23")
#define DEFK(NAME)                                                                                                       \
    __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t* hw, uint32_t iters) {                            \
        uint32_t bad[NVAR] = {0};                                                                                        \
        const uint32_t lane = threadIdx.x & 63u;                                                                         \
        uint32_t xs[16];                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 16; j++) xs[j] = blockIdx.x * 64u + (lane & 48u) + j;                       \
        for (uint32_t it = 0; it < iters; it++) {                                                                        \
            uint32_t want = 0, p = 0;                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 16; j++) {                                                             \
                xs[j] = xs[j] * 1664525u + 1013904223u;                                                                  \
                const uint32_t pj = (xs[j] >> 13) & 1u;                                                                  \
                want |= pj << j;                                                                                         \
                p = (int)(lane & 15u) == j ? pj : p;                                                                     \
            }                                                                                                            \
            uint32_t r;                                                                                                  \
            _Pragma("unroll") for (int v = 0; v < NVAR; v++) {                                                           \
                switch (v) {                                                                                             \
                    case 0: ASM(PRE(STR(TOP)) CMP_VCC "s_nop 4\n\t" SH(STR(TOP), "vcc")); break;                           \
                    case 1: ASM(PRE(STR(TOP1)) CMP_VCC "s_nop 4\n\t" SH(STR(TOP1), "vcc")); break;                         \
                    case 2: ASM(PRE(STR(TOP2)) CMP_VCC "s_nop 4\n\t" SH(STR(TOP2), "vcc")); break;                         \
                    case 3: ASM(PRE(STR(TOP8)) CMP_VCC "s_nop 4\n\t" SH(STR(TOP8), "vcc")); break;                         \
                    case 4: ASM(PRE("40") CMP_VCC "s_nop 4\n\t" SH("40", "vcc")); break;                                  \
                    case 5: ASM(PRE(STR(TOP)) CMP_VCC "s_mov_b64 s[22:23], vcc\n\ts_nop 4\n\t" SH(STR(TOP), "s[22:23]")); break; \
                    case 6: ASM(PRE(STR(TOP)) CMP_VCC "s_mov_b64 s[22:23], vcc\n\ts_nop 4\n\tv_mov_b32 v40, v" STR(TOP) "\n\ts_nop 4\n\t" SH("40", "s[22:23]")); break; \
                    case 7: ASM(PRE(STR(TOP)) CMP_VCC "s_mov_b64 s[22:23], vcc\n\ts_nop 4\n\tv_cmp_ne_u32_e64 s[20:21], 0, v" STR(TOP) "\n\tv_mov_b32 v2, s22\n\tv_mov_b32 v3, s23\n\ts_nop 4\n\tv_cndmask_b32_e64 v2, v2, v3, s[20:21]\n\tv_bfe_u32 %[res], v2, v48, 16"); break; \
                    case 8: ASM(PRE(STR(TOP)) CMP_VCC "s_mov_b64 s[22:23], vcc\n\ts_nop 4\n\tv_lshrrev_b32 v2, v" STR(TOP) ", s22\n\tv_bfe_u32 %[res], v2, v48, 16\n\tv_cmp_ne_u32_e64 s[20:21], 0, v" STR(TOP) "\n\tv_mov_b32 v3, s23\n\tv_bfe_u32 v3, v3, v48, 16\n\ts_nop 4\n\tv_cndmask_b32_e64 %[res], %[res], v3, s[20:21]"); break; \
                    case 9: ASM("v_mov_b32 v40, 0x0C0C0100\n\tv_lshrrev_b32 v2, 4, %[lane]\n\tv_and_b32 v2, 3, v2\n\tv_mov_b32 v3, 0x0202\n\tv_mad_u32_u24 v40, v2, v3, v40\n\t" CMP_VCC "s_nop 4\n\tv_mov_b32 v2, vcc_hi\n\tv_perm_b32 v2, v2, vcc_lo, v40\n\tv_mov_b32 v3, 0\n\tv_and_b32 v" STR(TOP) ", 32, %[lane]\n\tv_or_b32 v" STR(TOP) ", 4, v" STR(TOP) "\n\tv_lshlrev_b64 v[2:3], v" STR(TOP) ", v[2:3]\n\tv_mov_b32 %[res], v2"); break; \
                    case 10: ASM("v_mov_b32 v" STR(TOP) ", 0x0C0C0100\n\tv_lshrrev_b32 v2, 4, %[lane]\n\tv_and_b32 v2, 3, v2\n\tv_mov_b32 v3, 0x0202\n\tv_mad_u32_u24 v" STR(TOP) ", v2, v3, v" STR(TOP) "\n\t" CMP_VCC "s_nop 1\n\tv_mov_b32 v2, vcc_hi\n\tv_perm_b32 %[res], v2, vcc_lo, v" STR(TOP)); break; \
                    default: ASM(PRE(STR(TOP)) CMP_VCC "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" SH(STR(TOP), "vcc")); break; \
                }                                                                                                        \
                bad[v] += __popcll(__ballot(r != (v == 9 ? ((lane & 32u) ? 0u : want << 4) : want))) ? 1u : 0u;                                                       \
            }                                                                                                            \
        }                                                                                                                \
        if (lane == 0) {                                                                                                 \
            for (int v = 0; v < NVAR; v++) out[blockIdx.x * NVAR + v] = bad[v];                                          \
            hw[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                                  \
        }                                                                                                                \
    }
#define TOP 87
#define TOP1 86
#define TOP2 85
#define TOP8 79
DEFK(k_top87)
#undef TOP
#undef TOP1
#undef TOP2
#undef TOP8
#define TOP 95
#define TOP1 94
#define TOP2 93
#define TOP8 87
DEFK(k_top95)
#undef TOP

template <typename Kern> void run(Kern kern, const char* what, uint32_t blocks, uint32_t iters) {
    const char* names[NVAR] = {"shift amount in the LAST allocated register (v87 / v95)", "amount in last - 1", "amount in last - 2", "amount in last - 8 (v79 / v87)", "amount in v40",
                               "last register, ballot copied by the SALU first", "amount copied from the last register to v40 first", "no 64-bit shift: select of two 32-bit halves on (last != 0)",
                               "32-bit shift by the last register + select", "v_lshlrev_b64 of the slice by (lane & 32 | 4) held in the last register", "the fix: v_perm_b32 with the selector in the last register (row_ballot16)", "last register, 32 wait states behind the compare"};
    uint32_t *out, *hw;
    hipMalloc(&out, blocks * NVAR * 4); hipMalloc(&hw, blocks * 4);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, hw, iters);
    hipDeviceSynchronize();
    std::vector<uint32_t> o(blocks * NVAR), h(blocks);
    hipMemcpy(o.data(), out, blocks * NVAR * 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), hw, blocks * 4, hipMemcpyDeviceToHost);
    printf("== %s, %u one-wave blocks x %u trials\n", what, blocks, iters);
    for (int v = 0; v < NVAR; v++) {
        unsigned long long tot[4] = {0}, waves[4] = {0};
        for (uint32_t b = 0; b < blocks; b++) { const int s = (h[b] & 15) > 2 ? 3 : (h[b] & 15); tot[s] += o[b * NVAR + v]; waves[s]++; }
        printf("   %-66s wrong slices, wave slot 0 / 1 / 2 / 3+: %llu / %llu / %llu / %llu (waves %llu / %llu / %llu / %llu)\n", names[v], tot[0], tot[1], tot[2], tot[3], waves[0], waves[1], waves[2], waves[3]);
    }
    hipFree(out); hipFree(hw);
}
int main() {
    for (int rep = 0; rep < 2; rep++) {
        run(k_top87, "88 registers allocated", 5120, 12000);
        run(k_top95, "96 registers allocated", 5120, 12000);
    }
    return 0;
}
