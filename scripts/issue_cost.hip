// issue_cost.hip -- VALU issue cost per wave64 instruction on the gfx950 SIMD, by opcode (diagnostic, not product).
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/issue_cost scripts/issue_cost.hip && /tmp/issue_cost > profiles/r03/issue_cost.txt
//
// Every kernel runs `iters` passes of 64 independent instructions of ONE opcode (8 register chains x 8) in every wave, W waves per SIMD
// (256 * W blocks of 256 threads on 256 CUs).  Printed per opcode for W = 4 and 8:
//   Ginst/s        wave-instructions per second of the whole chip from the HIP-event wall time of the launch
//   GHz            shader clock held during the launch: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), median over waves
//   cyc/inst/SIMD  = 1024 SIMDs x clock / (inst/s): the issue cost of the opcode with the SIMD kept busy by W waves
// bench.py / scripts/isa_mix.py price a kernel's VALU stream with this table (profiles/r03/issue_cost.txt).
// Operands: %0-%7 eight 64-bit VGPR pairs, %8-%15 eight 32-bit VGPRs, %16 / %17 64-bit constants in VGPRs, %18 / %19 32-bit constants.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(S) S S S S S S S S

#define KERNEL(NAME, PRE, ASM8)                                                                                         \
  __global__ void __launch_bounds__(256) NAME(unsigned long long* out, int iters, double c1, double c2) {               \
    __shared__ double lds[256 * 17];                                                                                    \
    for (int i = threadIdx.x; i < 256 * 17; i += 256) lds[i] = i;                                                       \
    __syncthreads();                                                                                                    \
    double d0 = threadIdx.x * 1e-3 + 1., d1 = d0 + 1., d2 = d0 + 2., d3 = d0 + 3., d4 = d0 + 4., d5 = d0 + 5., d6 = d0 + 6., d7 = d0 + 7.; \
    int r0 = threadIdx.x + 1, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    int k1 = (int)(size_t)(lds + threadIdx.x), k2 = (int)c2 + 3;                                                        \
    unsigned long long t0 = __builtin_readcyclecounter(), q0 = __builtin_amdgcn_s_memrealtime();                        \
    for (int i = 0; i < iters; i++) {                                                                                   \
      asm volatile(PRE REP8(ASM8)                                                                                       \
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7),                    \
                     "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                     \
                   : "v"(c1), "v"(c2), "v"(k1), "v"(k2) : "vcc", "s20", "s21", "s22", "s23", "memory");                 \
    }                                                                                                                   \
    unsigned long long t1 = __builtin_readcyclecounter(), q1 = __builtin_amdgcn_s_memrealtime();                        \
    double s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);                         \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
      size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                                          \
      out[2 * w] = (t1 - t0) + (s == 12345.678 ? 1 : 0); out[2 * w + 1] = q1 - q0;                                      \
    }                                                                                                                   \
  }

// one instruction template applied to the eight chains: D(i) 64-bit chain i, R(i) 32-bit chain i
#define E8(T) T(0, 8) T(1, 9) T(2, 10) T(3, 11) T(4, 12) T(5, 13) T(6, 14) T(7, 15)
#define S_(x) #x

#define T_FMA64(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, %17\n"
#define T_FMAC64(d, r) "v_fmac_f64 %" S_(d) ", %16, %17\n"
#define T_FMA64NEG(d, r) "v_fma_f64 %" S_(d) ", -%" S_(d) ", %16, %17\n"
#define T_FMA64LIT(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, -0.5\n"
#define T_ADD64(d, r) "v_add_f64 %" S_(d) ", %" S_(d) ", %16\n"
#define T_ADD64S(d, r) "v_add_f64 %" S_(d) ", %" S_(d) ", s[22:23]\n"
#define T_MUL64(d, r) "v_mul_f64 %" S_(d) ", %" S_(d) ", %16\n"
#define T_MAX64(d, r) "v_max_f64 %" S_(d) ", %" S_(d) ", %16\n"
#define T_FLOOR64(d, r) "v_floor_f64 %" S_(d) ", %" S_(d) "\n"
#define T_CEIL64(d, r) "v_ceil_f64 %" S_(d) ", %" S_(d) "\n"
#define T_RNDNE64(d, r) "v_rndne_f64 %" S_(d) ", %" S_(d) "\n"
#define T_FRACT64(d, r) "v_fract_f64 %" S_(d) ", %" S_(d) "\n"
#define T_RCP64(d, r) "v_rcp_f64 %" S_(d) ", %" S_(d) "\n"
#define T_SQRT64(d, r) "v_sqrt_f64 %" S_(d) ", %" S_(d) "\n"
#define T_LDEXP64(d, r) "v_ldexp_f64 %" S_(d) ", %" S_(d) ", 1\n"
#define T_FRMANT64(d, r) "v_frexp_mant_f64 %" S_(d) ", %" S_(d) "\n"
#define T_DIVFIX64(d, r) "v_div_fixup_f64 %" S_(d) ", %" S_(d) ", %16, %17\n"
#define T_DIVSCALE64(d, r) "v_div_scale_f64 %" S_(d) ", vcc, %" S_(d) ", %16, %17\n"
#define T_DIVFMAS64(d, r) "v_div_fmas_f64 %" S_(d) ", %" S_(d) ", %16, %17\n"
#define T_MOV64(d, r) "v_mov_b64 %" S_(d) ", %16\n"
#define T_MOV64S(d, r) "v_mov_b64 %" S_(d) ", s[22:23]\n"
#define T_LSHL64(d, r) "v_lshlrev_b64 %" S_(d) ", 1, %" S_(d) "\n"
#define T_LSHLADD64(d, r) "v_lshl_add_u64 %" S_(d) ", %" S_(d) ", 1, %16\n"
#define T_MADU64(d, r) "v_mad_u64_u32 %" S_(d) ", vcc, %" S_(r) ", %18, %" S_(d) "\n"
#define T_CVTI32F64(d, r) "v_cvt_i32_f64 %" S_(r) ", %" S_(d) "\n"
#define T_CVTF64I32(d, r) "v_cvt_f64_i32 %" S_(d) ", %" S_(r) "\n"
#define T_CMPF64VCC(d, r) "v_cmp_lt_f64 vcc, %" S_(d) ", %16\n"
#define T_CMPF64SG(d, r) "v_cmp_lt_f64 s[20:21], %" S_(d) ", %16\n"
#define T_CMPF64CLASS(d, r) "v_cmp_class_f64 vcc, %" S_(d) ", %18\n"
#define T_FREXPEXP(d, r) "v_frexp_exp_i32_f64 %" S_(r) ", %" S_(d) "\n"

#define T_FMA32(d, r) "v_fma_f32 %" S_(r) ", %" S_(r) ", %18, %19\n"
#define T_MUL32(d, r) "v_mul_f32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_ADDU32(d, r) "v_add_u32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_SUBU32(d, r) "v_sub_u32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_ADDCO(d, r) "v_add_co_u32 %" S_(r) ", vcc, %" S_(r) ", %18\n"
#define T_ADD3(d, r) "v_add3_u32 %" S_(r) ", %" S_(r) ", %18, %19\n"
#define T_AND(d, r) "v_and_b32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_OR3(d, r) "v_or3_b32 %" S_(r) ", %" S_(r) ", %18, %19\n"
#define T_LSHLREV32(d, r) "v_lshlrev_b32 %" S_(r) ", 3, %" S_(r) "\n"
#define T_ASHR32(d, r) "v_ashrrev_i32 %" S_(r) ", 31, %" S_(r) "\n"
#define T_MAXI32(d, r) "v_max_i32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_MINI32(d, r) "v_min_i32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_MED3(d, r) "v_med3_i32 %" S_(r) ", %" S_(r) ", %18, %19\n"
#define T_MULLO(d, r) "v_mul_lo_u32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_MULHI(d, r) "v_mul_hi_u32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_LSHLADD32(d, r) "v_lshl_add_u32 %" S_(r) ", %" S_(r) ", 3, %18\n"
#define T_BFE(d, r) "v_bfe_u32 %" S_(r) ", %" S_(r) ", 3, 7\n"
#define T_MOV32(d, r) "v_mov_b32 %" S_(r) ", %18\n"
#define T_MOV32S(d, r) "v_mov_b32 %" S_(r) ", s22\n"
#define T_MOV32LIT(d, r) "v_mov_b32 %" S_(r) ", 0x7ff80000\n"
#define T_MOVDPP(d, r) "v_mov_b32_dpp %" S_(r) ", %" S_(r) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_MOVDPPBC(d, r) "v_mov_b32_dpp %" S_(r) ", %" S_(r) " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define T_ADDDPP(d, r) "v_add_u32_dpp %" S_(r) ", %" S_(r) ", %" S_(r) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_CNDVCC(d, r) "v_cndmask_b32 %" S_(r) ", %" S_(r) ", %18, vcc\n"
#define T_CNDSG(d, r) "v_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %18, s[20:21]\n"
#define T_CNDLIT(d, r) "v_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", 0, s[20:21]\n"
#define T_CMPI32VCC(d, r) "v_cmp_lt_i32 vcc, %" S_(r) ", %18\n"
#define T_CMPI32SG(d, r) "v_cmp_lt_i32 s[20:21], %" S_(r) ", %18\n"
#define T_READLANE(d, r) "v_readlane_b32 s20, %" S_(r) ", 3\n"
#define T_READFIRST(d, r) "v_readfirstlane_b32 s20, %" S_(r) "\n"
#define T_WRITELANE(d, r) "v_writelane_b32 %" S_(r) ", s22, 3\n"
#define T_EXP32(d, r) "v_exp_f32 %" S_(r) ", %" S_(r) "\n"
#define T_CVTF32I32(d, r) "v_cvt_f32_i32 %" S_(r) ", %" S_(r) "\n"
#define T_CVTF64F32(d, r) "v_cvt_f64_f32 %" S_(d) ", %" S_(r) "\n"
#define T_CVTF32F64(d, r) "v_cvt_f32_f64 %" S_(r) ", %" S_(d) "\n"
#define T_NOP(d, r) "s_nop 0\n"
#define T_SMOV(d, r) "s_mov_b32 s20, s22\n"
#define T_DSREAD64(d, r) "ds_read_b64 %" S_(d) ", %18 offset:" S_(d) "*2048\n"
#define T_DSREAD2(d, r) "ds_read2_b64 %" S_(d) ", %18 offset0:" S_(d) "*32 offset1:" S_(d) "*32+1\n"
#define T_MIXFMAADD(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, %17\nv_add_u32 %" S_(r) ", %" S_(r) ", %18\n"
#define T_MIXFMAMOVDPP(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, %17\nv_mov_b32_dpp %" S_(r) ", %" S_(r) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_MIXFMACND(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, %17\nv_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %18, s[20:21]\n"
#define T_MIXFMASALU(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, %17\ns_and_b32 s20, s22, s23\n"

#define T_CNDE64VCC(d, r) "v_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %18, vcc\n"
#define T_ADDC(d, r) "v_addc_co_u32 %" S_(r) ", vcc, %" S_(r) ", %18, vcc\n"
#define T_MIXFMACND32(d, r) "v_fma_f64 %" S_(d) ", %" S_(d) ", %16, %17\nv_cndmask_b32 %" S_(r) ", %" S_(r) ", %18, vcc\n"
#define T_SELVCC(d, r) "v_cmp_lt_f64 vcc, %" S_(d) ", %16\ns_nop 3\nv_cndmask_b32 %" S_(r) ", %" S_(r) ", %18, vcc\nv_cndmask_b32 %" S_(r) ", %" S_(r) ", %19, vcc\n"
#define T_SELSG(d, r) "v_cmp_lt_f64 s[20:21], %" S_(d) ", %16\ns_nop 3\nv_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %18, s[20:21]\nv_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %19, s[20:21]\n"
#define T_SELVCC64(d, r) "v_cmp_lt_f64 vcc, %" S_(d) ", %16\ns_nop 3\nv_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %18, vcc\nv_cndmask_b32_e64 %" S_(r) ", %" S_(r) ", %19, vcc\n"
#define T_CNDDIFF(d, r) "v_cndmask_b32 %" S_(r) ", %18, %19, vcc\n"
#define INITS "s_mov_b64 s[20:21], 0x5555\ns_mov_b64 s[22:23], 0x3ff\ns_mov_b64 vcc, 0x3333\n"
#define K(NAME, T) KERNEL(NAME, INITS, E8(T))

K(k01, T_FMA64) K(k02, T_FMAC64) K(k03, T_FMA64NEG) K(k04, T_FMA64LIT) K(k05, T_ADD64) K(k06, T_ADD64S) K(k07, T_MUL64) K(k08, T_MAX64)
K(k09, T_FLOOR64) K(k10, T_CEIL64) K(k11, T_RNDNE64) K(k12, T_FRACT64) K(k13, T_RCP64) K(k14, T_SQRT64) K(k15, T_LDEXP64) K(k16, T_FRMANT64)
K(k17, T_DIVFIX64) K(k18, T_DIVSCALE64) K(k19, T_DIVFMAS64) K(k20, T_MOV64) K(k21, T_MOV64S) K(k22, T_LSHL64) K(k23, T_LSHLADD64) K(k24, T_MADU64)
K(k25, T_CVTI32F64) K(k26, T_CVTF64I32) K(k27, T_CMPF64VCC) K(k28, T_CMPF64SG) K(k29, T_CMPF64CLASS) K(k30, T_FREXPEXP)
K(k31, T_FMA32) K(k32, T_MUL32) K(k33, T_ADDU32) K(k34, T_SUBU32) K(k35, T_ADDCO) K(k36, T_ADD3) K(k37, T_AND) K(k38, T_OR3) K(k39, T_LSHLREV32)
K(k40, T_ASHR32) K(k41, T_MAXI32) K(k42, T_MINI32) K(k43, T_MED3) K(k44, T_MULLO) K(k45, T_MULHI) K(k46, T_LSHLADD32) K(k47, T_BFE) K(k48, T_MOV32)
K(k49, T_MOV32S) K(k50, T_MOV32LIT) K(k51, T_MOVDPP) K(k52, T_MOVDPPBC) K(k53, T_ADDDPP) K(k54, T_CNDVCC) K(k55, T_CNDSG) K(k56, T_CNDLIT)
K(k57, T_CMPI32VCC) K(k58, T_CMPI32SG) K(k59, T_READLANE) K(k60, T_READFIRST) K(k61, T_WRITELANE) K(k62, T_EXP32) K(k63, T_CVTF32I32)
K(k64, T_CVTF64F32) K(k65, T_CVTF32F64) K(k66, T_NOP) K(k67, T_SMOV)
KERNEL(k68, INITS, E8(T_DSREAD64) "s_waitcnt lgkmcnt(0)\n")
K(k70, T_MIXFMAADD) K(k71, T_MIXFMAMOVDPP) K(k72, T_MIXFMACND) K(k73, T_MIXFMASALU)

K(k74, T_CNDE64VCC) K(k75, T_ADDC) K(k76, T_MIXFMACND32) K(k77, T_SELVCC) K(k78, T_SELSG) K(k79, T_SELVCC64) K(k80, T_CNDDIFF)
typedef void (*kern_t)(unsigned long long*, int, double, double);
struct Entry { const char* name; kern_t k; int per_pass; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  std::vector<Entry> es = {
    {"v_fma_f64", k01, 64}, {"v_fmac_f64 (VOP2)", k02, 64}, {"v_fma_f64 neg modifier", k03, 64}, {"v_fma_f64 inline constant", k04, 64}, {"v_add_f64", k05, 64},
    {"v_add_f64 SGPR operand", k06, 64}, {"v_mul_f64", k07, 64}, {"v_max_f64", k08, 64}, {"v_floor_f64", k09, 64}, {"v_ceil_f64", k10, 64}, {"v_rndne_f64", k11, 64},
    {"v_fract_f64", k12, 64}, {"v_rcp_f64", k13, 64}, {"v_sqrt_f64", k14, 64}, {"v_ldexp_f64", k15, 64}, {"v_frexp_mant_f64", k16, 64}, {"v_div_fixup_f64", k17, 64},
    {"v_div_scale_f64", k18, 64}, {"v_div_fmas_f64", k19, 64}, {"v_mov_b64", k20, 64}, {"v_mov_b64 SGPR source", k21, 64}, {"v_lshlrev_b64", k22, 64},
    {"v_lshl_add_u64", k23, 64}, {"v_mad_u64_u32", k24, 64}, {"v_cvt_i32_f64", k25, 64}, {"v_cvt_f64_i32", k26, 64}, {"v_cmp_lt_f64 -> vcc", k27, 64},
    {"v_cmp_lt_f64 -> SGPR pair", k28, 64}, {"v_cmp_class_f64 -> vcc", k29, 64}, {"v_frexp_exp_i32_f64", k30, 64},
    {"v_fma_f32", k31, 64}, {"v_mul_f32", k32, 64}, {"v_add_u32", k33, 64}, {"v_sub_u32", k34, 64}, {"v_add_co_u32", k35, 64}, {"v_add3_u32", k36, 64}, {"v_and_b32", k37, 64},
    {"v_or3_b32", k38, 64}, {"v_lshlrev_b32", k39, 64}, {"v_ashrrev_i32", k40, 64}, {"v_max_i32", k41, 64}, {"v_min_i32", k42, 64}, {"v_med3_i32", k43, 64},
    {"v_mul_lo_u32", k44, 64}, {"v_mul_hi_u32", k45, 64}, {"v_lshl_add_u32", k46, 64}, {"v_bfe_u32", k47, 64}, {"v_mov_b32", k48, 64}, {"v_mov_b32 SGPR source", k49, 64},
    {"v_mov_b32 literal", k50, 64}, {"v_mov_b32_dpp row_shr:1", k51, 64}, {"v_mov_b32_dpp row_bcast:15", k52, 64}, {"v_add_u32_dpp row_shr:1", k53, 64},
    {"v_cndmask_b32 vcc", k54, 64}, {"v_cndmask_b32_e64 SGPR pair", k55, 64}, {"v_cndmask_b32_e64 SGPR pair, const 0", k56, 64}, {"v_cmp_lt_i32 -> vcc", k57, 64},
    {"v_cmp_lt_i32 -> SGPR pair", k58, 64}, {"v_readlane_b32", k59, 64}, {"v_readfirstlane_b32", k60, 64}, {"v_writelane_b32", k61, 64}, {"v_exp_f32", k62, 64},
    {"v_cvt_f32_i32", k63, 64}, {"v_cvt_f64_f32", k64, 64}, {"v_cvt_f32_f64", k65, 64}, {"s_nop 0", k66, 64}, {"s_mov_b32", k67, 64},
    {"ds_read_b64 (8 + wait)", k68, 64},
    {"v_fma_f64 + v_add_u32 (pairs)", k70, 64}, {"v_fma_f64 + v_mov_b32_dpp (pairs)", k71, 64}, {"v_fma_f64 + v_cndmask_b32_e64 (pairs)", k72, 64},
    {"v_fma_f64 + s_and_b32 (pairs)", k73, 64},
    {"v_cndmask_b32_e64 with vcc as the mask", k74, 64}, {"v_addc_co_u32 (reads vcc)", k75, 64}, {"v_fma_f64 + v_cndmask_b32 vcc (pairs)", k76, 64},
    {"cmp_f64->vcc, s_nop 3, 2 cndmask vcc (groups)", k77, 64}, {"cmp_f64->SGPR, s_nop 3, 2 cndmask_e64 (groups)", k78, 64},
    {"cmp_f64->vcc, s_nop 3, 2 cndmask_e64 vcc (groups)", k79, 64}, {"v_cndmask_b32 vcc, sources not the destination", k80, 64},
  };
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs, clockRate %d kHz; %d passes of 64 instructions (or pairs) per wave; W waves per SIMD\n", prop.name, prop.multiProcessorCount, prop.clockRate, iters);
  printf("%-42s %2s %10s %7s %14s\n", "opcode", "W", "Ginst/s", "GHz", "cyc/inst/SIMD");
  unsigned long long* d;
  const int maxw = 256 * 8 * 4;
  CK(hipMalloc(&d, 2 * maxw * sizeof(unsigned long long)));
  std::vector<unsigned long long> h(2 * maxw);
  std::vector<double> clk(maxw);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* only = argc > 2 ? argv[2] : nullptr;
  for (auto& en : es) {
    if (only && !strstr(en.name, only)) continue;
    for (int W : {4, 8}) {
      const int blocks = prop.multiProcessorCount * W, nw = blocks * 4;
      hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, d, iters / 10 + 1, 1.0000001, 1e-9);      // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(en.k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), d, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      for (int i = 0; i < nw; i++) clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.;      // GHz: shader ticks per 10 ns
      std::sort(clk.begin(), clk.begin() + nw);
      const double ghz = clk[nw / 2], n = (double)iters * en.per_pass, ginst = nw * n / (ms * 1e-3) / 1e9;
      printf("%-42s %2d %10.1f %7.3f %14.3f\n", en.name, W, ginst, ghz, 1024. * ghz / ginst);
    }
  }
  return 0;
}
