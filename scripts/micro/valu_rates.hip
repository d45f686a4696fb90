// Issue cost of single VALU instructions on gfx950, relative to v_add_u32 (round 6): every wave of a full chip runs N independent copies of one
// instruction in a loop; time per instruction and wave.  build: hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rates.hip -o /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITER 2000
#define K(name, ASM)                                                                                         \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned a0) {                                 \
        unsigned x0 = threadIdx.x + a0, x1 = x0 * 3u + 1u, x2 = x0 ^ 0x55u, x3 = x0 + 77u;                   \
        unsigned long long y0 = x0, y1 = x1, y2 = x2, y3 = x3;                                                \
        unsigned s = (threadIdx.x & 7u) + 1u;                                                                 \
        for (int it = 0; it < ITER; it++) {                                                                   \
            _Pragma("unroll") for (int r = 0; r < REP / 4; r++) { ASM }                                      \
        }                                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)(y0 + y1 + y2 + y3);              \
    }
K(k_add, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_shl64, asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(s));)
K(k_shr64, asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(s));)
K(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(s) : "vcc");)
K(k_lshl_add64, asm volatile("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));)
K(k_perm, asm volatile("v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %4, %4\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_bfe, asm volatile("v_bfe_u32 %0, %0, %4, 5\n v_bfe_u32 %1, %1, %4, 5\n v_bfe_u32 %2, %2, %4, 5\n v_bfe_u32 %3, %3, %4, 5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_dpp, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
K(k_bcnt, asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "vcc");)
K(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "vcc");)
K(k_ffbl, asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
K(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) :: "s20", "s21", "s22", "s23");)
K(k_and, asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_xor, asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_shl32, asm volatile("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_min, asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_add3, asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_add2, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_salu_mix, asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %4\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, %4\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %4\n s_add_u32 s23, s23, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "s20", "s21", "s22", "s23", "scc");)
K(k_perm_salu, asm volatile("v_perm_b32 %0, %0, %4, %4\n s_add_u32 s20, s20, 1\n v_perm_b32 %1, %1, %4, %4\n s_add_u32 s21, s21, 1\n v_perm_b32 %2, %2, %4, %4\n s_add_u32 s22, s22, 1\n v_perm_b32 %3, %3, %4, %4\n s_add_u32 s23, s23, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "s20", "s21", "s22", "s23", "scc");)
K(k_salu, asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");)
K(k_perm_2salu, asm volatile("v_perm_b32 %0, %0, %4, %4\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_perm_b32 %1, %1, %4, %4\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n v_perm_b32 %2, %2, %4, %4\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_perm_b32 %3, %3, %4, %4\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "s20", "s21", "s22", "s23", "scc");)
K(k_perm_3salu, asm volatile("v_perm_b32 %0, %0, %4, %4\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s24, s24, 1\n v_perm_b32 %1, %1, %4, %4\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s25, s25, 1\n v_perm_b32 %2, %2, %4, %4\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s24, s24, 1\n v_perm_b32 %3, %3, %4, %4\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s25, s25, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");)
K(k_2perm_salu, asm volatile("v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %4, %4\n s_add_u32 s20, s20, 1\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %4, %4\n s_add_u32 s21, s21, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s) : "s20", "s21", "scc");)
K(k_perm_add, asm volatile("v_perm_b32 %0, %0, %4, %4\n v_add_u32 %1, %1, %4\n v_perm_b32 %2, %2, %4, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(s));)
K(k_nop, asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");)
K(k_lds, asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n s_waitcnt lgkmcnt(0)" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"((threadIdx.x & 63u) * 4u));)
int main() {
    unsigned* out; hipMalloc(&out, 4096 * 256 * 4);
    struct { const char* n; void (*k)(unsigned*, unsigned); } ks[] = {{"v_add_u32", k_add}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24},
        {"v_mad_u32_u24", k_mad_u24}, {"v_lshlrev_b64", k_shl64}, {"v_lshrrev_b64", k_shr64}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshl_add64}, {"v_perm_b32", k_perm},
        {"v_bfe_u32", k_bfe}, {"v_mov_b32_dpp", k_dpp}, {"v_bcnt_u32_b32", k_bcnt}, {"v_cmp + v_cndmask", k_cmp_cnd}, {"v_add_co + v_addc_co", k_add_co}, {"v_ffbl_b32", k_ffbl}, {"v_readlane_b32", k_readlane}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_shl32}, {"v_mov_b32", k_mov}, {"v_min_u32", k_min}, {"v_add3_u32", k_add3}, {"v_add_u32 (again)", k_add2}, {"v_add_u32 + s_add_u32 (per pair)", k_salu_mix}, {"v_perm_b32 + s_add_u32 (per pair)", k_perm_salu}, {"s_add_u32", k_salu}, {"v_perm + 2 s_add (per triple)", k_perm_2salu}, {"v_perm + 3 s_add (per quad)", k_perm_3salu}, {"2 v_perm + s_add (4 v + 2 s per asm: per 1.5)", k_2perm_salu}, {"v_perm + v_add (avg)", k_perm_add}, {"s_nop 0", k_nop}, {"ds_read_b32 (+ waitcnt per 4)", k_lds}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    for (auto& k : ks) {
        const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipEventRecord(e0); hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e-3 / ((double)ITER * REP * 8);   // seconds per instruction and SIMD (8 waves per SIMD in turn)
        if (base == 0) base = per;
        printf("%-22s %8.3f ms  %6.2f ns per instruction and SIMD  = %5.2f x v_add_u32\n", k.n, ms, per * 1e9, per / base);
    }
    return 0;
}
