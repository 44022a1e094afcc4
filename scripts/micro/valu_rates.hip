// Measured on MI355X (2026-09): full rate (~2.5 cyc @2.4GHz nominal): fma add mul sub fmac fmaak and xor add_u32 mov;
// half rate (~4.3): min max med3 cmp cndmask(e64) ldexp rndne cvt shifts bfe lshl_add dpp mul_lo mad_u24 and ANY op with an SGPR
// source; quarter (~8.1): rcp sqrt exp permlane32_swap; v_pk_fma_f32 = same flops as v_fma_f32.  Always run under `timeout`.
// Microbenchmark (dev tool): issue rate of individual gfx950 VALU opcodes, 8 independent
// chains per lane, 8 waves per SIMD.  Prints wave-instructions per SIMD-cycle-equivalent.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define DEF_KERNEL(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b, int iters) { \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int it = 0; it < iters; ++it) {                                               \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)       \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                             : "v"(a), "v"(b), "s"(a) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");                                    \
            }                                                                              \
        }                                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;      \
    }
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define A_CMPS(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n"
#define A_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %8\n"
#define A_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define A_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_CVTF(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_SUB(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define A_FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %8, 0x3c088908\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define A_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_PERM32(i) "v_permlane32_swap_b32 %" #i ", %8\n"
#define A_FMAS(i) "v_fma_f32 %" #i ", %" #i ", %10, %9\n"
#define A_MULS(i) "v_mul_f32 %" #i ", %10, %" #i "\n"
#define A_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define A_CNDK(i) "v_cndmask_b32 %" #i ", 0, %" #i ", vcc\n"
#define A_CNDD(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define A_FMAK1(i) "v_fma_f32 %" #i ", %" #i ", %8, 1.0\n"
#define A_MULK(i) "v_mul_f32 %" #i ", 0.5, %" #i "\n"
#define A_MULLIT(i) "v_mul_f32 %" #i ", 0x3fb8aa3b, %" #i "\n"
#define A_SUBREV(i) "v_sub_f32 %" #i ", 1.0, %" #i "\n"
#define A_FMANEG(i) "v_fma_f32 %" #i ", -%" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_CMPU(i) "v_cmp_le_u32 vcc, %" #i ", %8\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n s_waitcnt lgkmcnt(0)\n"
#define LIST(X) X(fma, A_FMA) X(add, A_ADD) X(mul, A_MUL) X(sub, A_SUB) X(fmac, A_FMAC) X(fmaak, A_FMAAK) X(fma_sgpr, A_FMAS) X(mul_sgpr, A_MULS) \
    X(min, A_MIN) X(max, A_MAX) X(cndmask, A_CNDMASK) X(cmp_vcc, A_CMP) X(cmp_sgpr, A_CMPS) X(ldexp, A_LDEXP) X(rndne, A_RNDNE) \
    X(cvt_i32_f32, A_CVTI) X(cvt_f32_u32, A_CVTF) X(and_b32, A_AND) X(add_u32, A_ADDU) X(lshl, A_LSHL) X(mov, A_MOV) \
    X(rcp, A_RCP) X(sqrt, A_SQRT) X(exp, A_EXP) X(add_dpp, A_DPP) X(mul_lo_u32, A_MULLO) X(mad_u24, A_MAD24) X(cnd_e64, A_CND64) X(cnd_k, A_CNDK) X(cnd_nodep, A_CNDD) X(med3, A_MED3) X(max3, A_MAX3) X(bfe, A_BFE) X(fma_k1, A_FMAK1) X(mul_k, A_MULK) X(mul_lit, A_MULLIT) X(sub_k, A_SUBREV) X(fma_neg, A_FMANEG) X(lshl_add, A_LSHLADD) X(xor, A_XOR) X(cmp_u32, A_CMPU) X(perm32swap, A_PERM32)
#define X(n, a) DEF_KERNEL(k_##n, a)
LIST(X)
#undef X
typedef void (*kfn)(float*, float, float, int);
static void run(const char* name, kfn f, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 1000;
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * 64;  // wave-instructions
    const double per_simd_per_s = winst / (ms * 1e-3) / 1024.0;
    printf("%-14s %8.3f ms   %.3f G wave-inst/s/SIMD  -> %.2f cycles/inst @2.4GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define X(n, a) run(#n, k_##n, d);
    return 0;
}
