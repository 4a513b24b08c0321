// Microbenchmark: issue cost of the instructions k_fine is made of, per wave64 instruction and SIMD, on MI355X.
// Settles what "VALU issue bound" means for an integer / DPP / SDWA / packed-f32 mix (the guide quotes 2
// cycles per wave64 v_fma_f32 on CDNA4's SIMD-32, DESIGN.md assumed 4).  Not part of the product.
//
// Every test is a loop of 64 instructions of one kind over 8 independent registers (no dependent chain shorter than 8
// instructions), run by W waves per SIMD on every SIMD of the chip: 256-thread workgroups (a workgroup's four waves go to
// the CU's four SIMDs), W workgroups per CU, W = 1, 2, 4, 8; the kernels hold < 64 VGPRs, so all of them are resident at
// once.  Reported: shader cycles (s_memtime) per instruction PER SIMD = a wave's elapsed cycles / (instructions issued by
// the W waves of its SIMD), median over waves; and the same from the launch's wall time at a nominal 2.4 GHz (the ratio of
// the two is the clock the run sustained).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

struct Regs {
    uint32_t r[8];
};

#define DECL_TEST(NAME, ASM_LINE)                                                                                    \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t iters, uint32_t *sink, unsigned long long *cyc) {          \
        uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,      \
                 r7 = r0 + 7;                                                                                        \
        uint32_t s0 = blockIdx.x, s1 = s0 + 1;                                                                        \
        unsigned long long p0 = r0, p1 = r1, p2 = r2, p3 = r3;                                                       \
        __shared__ __attribute__((aligned(16))) uint32_t lds[1024];                                                  \
        lds[threadIdx.x] = 0;                                                                                        \
        uint32_t la = threadIdx.x * 16u;                                                                             \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (uint32_t i = 0; i < iters; i++) {                                                                       \
            asm volatile(ASM_LINE(0) ASM_LINE(1) ASM_LINE(2) ASM_LINE(3) ASM_LINE(4) ASM_LINE(5) ASM_LINE(6)         \
                             ASM_LINE(7) ASM_LINE(0) ASM_LINE(1) ASM_LINE(2) ASM_LINE(3) ASM_LINE(4) ASM_LINE(5)     \
                                 ASM_LINE(6) ASM_LINE(7) ASM_LINE(0) ASM_LINE(1) ASM_LINE(2) ASM_LINE(3)             \
                                     ASM_LINE(4) ASM_LINE(5) ASM_LINE(6) ASM_LINE(7) ASM_LINE(0) ASM_LINE(1)         \
                                         ASM_LINE(2) ASM_LINE(3) ASM_LINE(4) ASM_LINE(5) ASM_LINE(6) ASM_LINE(7)     \
                                             ASM_LINE(0) ASM_LINE(1) ASM_LINE(2) ASM_LINE(3) ASM_LINE(4)             \
                                                 ASM_LINE(5) ASM_LINE(6) ASM_LINE(7) ASM_LINE(0) ASM_LINE(1)         \
                                                     ASM_LINE(2) ASM_LINE(3) ASM_LINE(4) ASM_LINE(5) ASM_LINE(6)     \
                                                         ASM_LINE(7) ASM_LINE(0) ASM_LINE(1) ASM_LINE(2)             \
                                                             ASM_LINE(3) ASM_LINE(4) ASM_LINE(5) ASM_LINE(6)         \
                                                                 ASM_LINE(7) ASM_LINE(0) ASM_LINE(1) ASM_LINE(2)     \
                                                                     ASM_LINE(3) ASM_LINE(4) ASM_LINE(5)             \
                                                                         ASM_LINE(6) ASM_LINE(7)                     \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(s0),  \
                           "+s"(s1)                                                                                  \
                         : "v"(la)                                                                                   \
                         : "vcc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",            \
                           "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s40", "s41", "s42", "s43", "s44",        \
                           "s45", "s46", "s47");                                                                     \
        }                                                                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        if ((threadIdx.x & 63u) == 0) cyc[blockIdx.x * 4u + (threadIdx.x >> 6)] = t1 - t0;                           \
        if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ s0 ^ s1 ^ (uint32_t)(p0 ^ p1 ^ p2 ^ p3)) == 0x12345u)            \
            *sink = r0 + lds[r1 & 255u];                                                                             \
    }

// operand numbering of the asm block: %0..%7 = r0..r7 (VGPRs), %8, %9 = s0, s1 (SGPRs), %10 = la (LDS byte address)
#define L_ADD(k) "v_add_u32 %" #k ", %" #k ", %" #k "\n"
#define L_AND(k) "v_and_b32 %" #k ", 0x7fffffff, %" #k "\n"
#define L_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define L_XOR(k) "v_xor_b32 %" #k ", %" #k ", %10\n"
#define L_BITOP3(k) "v_bitop3_b32 %" #k ", %" #k ", %10, %" #k " bitop3:0x78\n"
#define L_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %10, %" #k "\n"
#define L_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %10\n"
#define L_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 2, %10\n"
#define L_BFI(k) "v_bfi_b32 %" #k ", %10, %" #k ", %" #k "\n"
#define L_PERM(k) "v_perm_b32 %" #k ", %" #k ", %" #k ", %10\n"
#define L_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 9\n"
#define L_BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %" #k "\n"
#define L_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %10, vcc\n"
#define L_CNDMASK_S(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %10, s[44:45]\n"
#define L_CNDMASK_C(k) "v_cndmask_b32_e64 %" #k ", 1.0, 0, vcc\n"
#define L_CMP_CND(k) "v_cmp_eq_u32 vcc, %" #k ", %10\nv_cndmask_b32 %" #k ", %" #k ", %10, vcc\n"
#define L_CMP_CND_S(k) "v_cmp_eq_u32 s[44:45], %" #k ", %10\nv_cndmask_b32_e64 %" #k ", %" #k ", %10, s[44:45]\n"
#define L_CND_SPACED(k) "v_cndmask_b32 %" #k ", %" #k ", %10, vcc\nv_add_u32 %" #k ", %" #k ", %10\nv_xor_b32 %" #k ", %" #k ", %10\nv_sub_u32 %" #k ", %" #k ", %10\n"
#define L_CND_E64_VCC(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %10, vcc\n"
#define L_CND_SWAP(k) "v_cndmask_b32 %" #k ", %10, %" #k ", vcc\n"
#define L_CND_SMOV(k) "s_mov_b64 vcc, s[44:45]\nv_cndmask_b32 %" #k ", %" #k ", %10, vcc\n"
#define L_CND_CMP4(k) "v_cmp_eq_u32 vcc, %" #k ", %10\nv_cndmask_b32 %" #k ", %" #k ", %10, vcc\nv_cndmask_b32 %" #k ", %10, %" #k ", vcc\nv_cndmask_b32 %" #k ", %" #k ", %10, vcc\nv_cndmask_b32 %" #k ", %10, %" #k ", vcc\n"
#define L_LSHL3(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define L_LSHR1(k) "v_lshrrev_b32 %" #k ", 1, %" #k "\n"
#define L_LSHLV(k) "v_lshlrev_b32 %" #k ", %10, %" #k "\n"
#define L_LSHRV(k) "v_lshrrev_b32 %" #k ", %10, %" #k "\n"
#define L_ASHR(k) "v_ashrrev_i32 %" #k ", 3, %" #k "\n"
#define L_ADD_LIT(k) "v_add_u32 %" #k ", 0x7f7f7f80, %" #k "\n"
#define L_ADD_SGPR(k) "v_add_u32 %" #k ", %8, %" #k "\n"
#define L_MAX_F32(k) "v_max_f32 %" #k ", %" #k ", %" #k "\n"
#define L_AND_OR(k) "v_and_or_b32 %" #k ", %" #k ", %10, %" #k "\n"
#define L_OR3(k) "v_or3_b32 %" #k ", %" #k ", %10, %" #k "\n"
#define L_XAD(k) "v_xad_u32 %" #k ", %" #k ", %10, %" #k "\n"
#define L_BFE_I(k) "v_bfe_i32 %" #k ", %" #k ", 3, 1\n"
#define L_SUB(k) "v_sub_u32 %" #k ", %" #k ", %10\n"
#define L_OR(k) "v_or_b32 %" #k ", %" #k ", %10\n"
#define L_LSHR(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define L_MIN(k) "v_min_u32 %" #k ", %" #k ", %10\n"
#define L_ADDF(k) "v_add_f32 %" #k ", %" #k ", %" #k "\n"
#define L_SUBF_C(k) "v_sub_f32 %" #k ", 1.0, %" #k "\n"
#define L_MIX_VV(k) "v_add_u32 %" #k ", %" #k ", %" #k "\nv_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define L_CMP(k) "v_cmp_eq_u32 vcc, %" #k ", %10\n"
#define L_CMP_SGPR(k) "v_cmp_eq_u32 s[40:41], %" #k ", %10\n"
#define L_DPP(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define L_ADD_DPP(k) "v_add_u32_dpp %" #k ", %" #k ", %" #k " quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n"
#define L_SDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define L_CMP_SDWA(k) "v_cmp_eq_u32_sdwa vcc, %" #k ", %10 src0_sel:BYTE_2 src1_sel:DWORD\n"
#define L_READLANE(k) "v_readlane_b32 s42, %" #k ", 5\n"
#define L_READLANE_S(k) "v_readlane_b32 s42, %" #k ", %8\n"
#define L_READFIRST(k) "v_readfirstlane_b32 s42, %" #k "\n"
#define L_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %" #k "\n"
#define L_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %" #k "\n"
#define L_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %" #k ", %10\n"
#define L_FMA(k) "v_fma_f32 %" #k ", %" #k ", %" #k ", %" #k "\n"
#define L_MULF(k) "v_mul_f32 %" #k ", %" #k ", %" #k "\n"
#define L_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define L_CVTU(k) "v_cvt_u32_f32 %" #k ", %" #k "\n"
#define L_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n"
#define L_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define L_MOV(k) "v_mov_b32 %" #k ", %10\n"
// 64-bit / packed forms use the clobbered pairs v[100:115]
#define L_PKMUL(k) "v_pk_mul_f32 v[" PAIR(k) "], v[" PAIR(k) "], v[" PAIR(k) "]\n"
#define L_PKADD(k) "v_pk_add_f32 v[" PAIR(k) "], v[" PAIR(k) "], v[" PAIR(k) "]\n"
#define L_PKFMA(k) "v_pk_fma_f32 v[" PAIR(k) "], v[" PAIR(k) "], v[" PAIR(k) "], v[" PAIR(k) "]\n"
#define L_MOV64(k) "v_mov_b64 v[" PAIR(k) "], v[" PAIR2(k) "]\n"
#define L_LSHLADD64(k) "v_lshl_add_u64 v[" PAIR(k) "], v[" PAIR(k) "], 2, v[" PAIR(k) "]\n"
#define PAIR(k) PAIR_##k
#define PAIR2(k) PAIR2_##k
#define PAIR_0 "40:41"
#define PAIR_1 "42:43"
#define PAIR_2 "44:45"
#define PAIR_3 "46:47"
#define PAIR_4 "48:49"
#define PAIR_5 "50:51"
#define PAIR_6 "52:53"
#define PAIR_7 "54:55"
#define PAIR2_0 "42:43"
#define PAIR2_1 "44:45"
#define PAIR2_2 "46:47"
#define PAIR2_3 "48:49"
#define PAIR2_4 "50:51"
#define PAIR2_5 "52:53"
#define PAIR2_6 "54:55"
#define PAIR2_7 "40:41"
// LDS
#define L_DSREAD(k) "ds_read_b32 %" #k ", %10\n"
#define L_DSWRITE(k) "ds_write_b32 %10, %" #k "\n"
#define L_DSADD(k) "ds_add_u32 %10, %" #k "\n"
#define L_DSBPERM(k) "ds_bpermute_b32 %" #k ", %10, %" #k "\n"
#define L_DSREAD128(k) "ds_read_b128 v[" QUAD(k) "], %10\n"
#define L_DSWRITE128(k) "ds_write_b128 %10, v[" QUAD(k) "]\n"
#define QUAD(k) QUAD_##k
#define QUAD_0 "40:43"
#define QUAD_1 "44:47"
#define QUAD_2 "48:51"
#define QUAD_3 "52:55"
#define QUAD_4 "40:43"
#define QUAD_5 "44:47"
#define QUAD_6 "48:51"
#define QUAD_7 "52:55"
// scalar
#define L_SADD(k) "s_add_u32 s43, s43, %8\n"
#define L_SAND64(k) "s_and_b64 s[44:45], s[44:45], s[46:47]\n"
#define L_SBFE(k) "s_bfe_u32 s43, %9, 0xa000a\n"
// mixes: half VALU, half something else -- does the other pipe issue beside VALU?
#define L_MIX_VS(k) "v_add_u32 %" #k ", %" #k ", %" #k "\ns_add_u32 s43, s43, %8\n"
#define L_MIX_VL(k) "v_add_u32 %" #k ", %" #k ", %" #k "\nds_read_b32 v4" #k ", %10\n"

#define ALL_TESTS(T)                                                                                                 \
    T(add, L_ADD, 1) T(and_lit, L_AND, 1) T(lshl, L_LSHL, 1) T(xor, L_XOR, 1) T(bitop3, L_BITOP3, 1)                 \
    T(add3, L_ADD3, 1) T(lshl_add, L_LSHLADD, 1) T(lshl_or, L_LSHLOR, 1) T(bfi, L_BFI, 1) T(perm, L_PERM, 1)         \
    T(bfe, L_BFE, 1) T(bcnt, L_BCNT, 1) T(cndmask, L_CNDMASK, 1) T(cndmask_sgpr, L_CNDMASK_S, 1)                    \
    T(cndmask_consts, L_CNDMASK_C, 1) T(cndmask_spaced, L_CND_SPACED, 4) T(cndmask_e64_vcc, L_CND_E64_VCC, 1)    \
    T(cndmask_swapped, L_CND_SWAP, 1) T(smov_vcc_cndmask, L_CND_SMOV, 2) T(cmp_4cndmask, L_CND_CMP4, 5)              \
    T(lshl3, L_LSHL3, 1) T(lshr1, L_LSHR1, 1) T(lshl_v, L_LSHLV, 1) T(lshr_v, L_LSHRV, 1) T(ashr, L_ASHR, 1)         \
    T(add_lit, L_ADD_LIT, 1) T(add_sgpr, L_ADD_SGPR, 1) T(max_f32, L_MAX_F32, 1) T(and_or, L_AND_OR, 1)             \
    T(or3, L_OR3, 1) T(xad, L_XAD, 1) T(cmp_cndmask, L_CMP_CND, 2) T(cmp_cndmask_sgpr, L_CMP_CND_S, 2)               \
    T(bfe_i32, L_BFE_I, 1) T(sub, L_SUB, 1) T(or, L_OR, 1) T(lshr, L_LSHR, 1) T(min_u32, L_MIN, 1) T(add_f32, L_ADDF, 1) \
    T(sub_f32_const, L_SUBF_C, 1) T(mix_add_lshl, L_MIX_VV, 2) T(cmp_vcc, L_CMP, 1) T(cmp_sgpr, L_CMP_SGPR, 1)            \
    T(mov_dpp, L_DPP, 1) T(add_dpp, L_ADD_DPP, 1) T(add_sdwa, L_SDWA, 1) T(cmp_sdwa, L_CMP_SDWA, 1)                  \
    T(readlane, L_READLANE, 1) T(readlane_s, L_READLANE_S, 1) T(readfirstlane, L_READFIRST, 1)                       \
    T(mul_lo, L_MULLO, 1) T(mul_u24, L_MUL24, 1) T(mad_u24, L_MAD24, 1) T(fma_f32, L_FMA, 1) T(mul_f32, L_MULF, 1)   \
    T(cvt_ubyte, L_CVTUB, 1) T(cvt_u32_f32, L_CVTU, 1) T(floor, L_FLOOR, 1) T(rcp, L_RCP, 1) T(mov, L_MOV, 1)        \
    T(pk_mul_f32, L_PKMUL, 1) T(pk_add_f32, L_PKADD, 1) T(pk_fma_f32, L_PKFMA, 1) T(mov_b64, L_MOV64, 1)             \
    T(lshl_add_u64, L_LSHLADD64, 1) T(ds_read_b32, L_DSREAD, 1) T(ds_write_b32, L_DSWRITE, 1)                        \
    T(ds_add_u32, L_DSADD, 1) T(ds_bpermute, L_DSBPERM, 1) T(ds_read_b128, L_DSREAD128, 1)                           \
    T(ds_write_b128, L_DSWRITE128, 1) T(s_add, L_SADD, 1) T(s_and_b64, L_SAND64, 1) T(s_bfe, L_SBFE, 1)              \
    T(mix_valu_salu, L_MIX_VS, 2) T(mix_valu_lds, L_MIX_VL, 2)

#define MAKE(NAME, LINE, PER) DECL_TEST(NAME, LINE)
ALL_TESTS(MAKE)

typedef void (*kern_t)(uint32_t, uint32_t *, unsigned long long *);
struct Test {
    const char *name;
    kern_t k;
    int per_line;
};
#define ENTRY(NAME, LINE, PER) {#NAME, k_##NAME, PER},
static const Test tests[] = {ALL_TESTS(ENTRY)};

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    uint32_t *sink;
    unsigned long long *cyc;
    hipMalloc((void **)&sink, 4);
    const int max_waves = n_cu * 4 * 8;
    hipMalloc((void **)&cyc, max_waves * sizeof(unsigned long long));
    const uint32_t iters = 1024;
    printf("%-16s %s\n", "instruction", "cycles per wave64 instruction per SIMD at 1 | 2 | 4 | 8 waves per SIMD  (s_memtime median ; from wall time at 2.4 GHz)");
    for (const Test &t : tests) {
        printf("%-16s", t.name);
        for (int w : {1, 2, 4, 8}) {
            const int blocks = n_cu * w, waves = blocks * 4;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            float best = 1e9f;
            std::vector<unsigned long long> h(waves);
            double med = 0.0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, iters, sink, cyc);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) {
                    best = ms;
                    hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                    std::sort(h.begin(), h.end());
                    med = (double)h[waves / 2];
                }
            }
            const double n_instr = (double)iters * 64.0 * t.per_line;  // per wave
            // a wave's elapsed cycles cover the instructions of all w waves of its SIMD (if the dispatcher spread them evenly)
            printf("  %6.2f ; %6.2f", med / (n_instr * w), best * 1e-3 * 2.4e9 / (n_instr * w));
        }
        printf("\n");
    }
    return 0;
}
