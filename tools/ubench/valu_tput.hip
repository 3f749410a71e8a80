// What one SIMD of a CU can issue per cycle, by instruction class (development tool; round 4).
//
// tools/ubench/pk_rate.hip timed wave 0 only.  Issue arbitration between the waves of a SIMD goes by age, so the
// oldest wave runs at its single-wave rate however many others are resident, and "0.97 wave-instructions per cycle
// per SIMD at four waves" was wave 0's rate times four.  Here EVERY wave records its first and last s_memtime and the
// rate is instructions of all waves of a SIMD over (latest end - earliest start).
//
// One workgroup of 4*W waves on one CU (W waves per SIMD), each wave a stream of 16 independent instructions of one
// class, 256 times.  Instructions are emitted 16 to an asm statement (the compiler pads each asm STATEMENT with one
// s_nop, not each instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_tput.hip -o tools/ubench/valu_tput && tools/ubench/valu_tput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define REP 64
#define STMTS 8
#define PER 16

#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

// 32-bit destination classes: a[i] = op(a[i], b, c)
#define DEF32(NAME, TEXT)                                                                                       \
    struct NAME {                                                                                               \
        static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e) \
        {                                                                                                       \
            asm volatile(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7) TEXT(8) TEXT(9) TEXT(10) TEXT(11) TEXT(12) TEXT(13) TEXT(14) TEXT(15) \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),          \
                           "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])     \
                         : "v"(b), "v"(c)                                                                       \
                         : "vcc", "s20", "s21");                                                                              \
        }                                                                                                       \
    };
#define DEF64(NAME, TEXT)                                                                                       \
    struct NAME {                                                                                               \
        static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e) \
        {                                                                                                       \
            asm volatile(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7) TEXT(8) TEXT(9) TEXT(10) TEXT(11) TEXT(12) TEXT(13) TEXT(14) TEXT(15) \
                         : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),          \
                           "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15])     \
                         : "v"(e), "v"(b)                                                                       \
                         : "vcc", "s20", "s21");                                                                              \
        }                                                                                                       \
    };
// f64 <- f32 conversions and the like: d[i] = op(a[i])
#define DEFMIX(NAME, TEXT)                                                                                      \
    struct NAME {                                                                                               \
        static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e) \
        {                                                                                                       \
            asm volatile(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7)                        \
                         : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),          \
                           "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])           \
                         : "v"(e), "v"(b)                                                                       \
                         : "vcc", "s20", "s21");                                                                              \
            asm volatile(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7)                        \
                         : "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]),    \
                           "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])     \
                         : "v"(e), "v"(b)                                                                       \
                         : "vcc", "s20", "s21");                                                                              \
        }                                                                                                       \
    };

#define T_MUL(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define T_ADD(i) "v_add_f32 %" #i ", %" #i ", %16\n"
#define T_SUB(i) "v_sub_f32 %" #i ", %16, %" #i "\n"
#define T_FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define T_MAX(i) "v_max_f32 %" #i ", %" #i ", %16\n"
#define T_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %16, %17\n"
#define T_MOV(i) "v_mov_b32 %" #i ", %16\n"
#define T_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n"
#define T_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define T_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %16\n"
#define T_CMPX(i) "v_cmp_lt_f32 s[20:21], %" #i ", %16\n"
#define T_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define T_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define T_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define T_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define T_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %16\n"
#define T_ADDU(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define T_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %16, %" #i "\n"
#define T_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %16, %17\n"
#define T_DIVFIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %16, %17\n"
#define T_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n"
#define T_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %16\n"
#define T_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %16\n"
#define T_ADD64(i) "v_add_f64 %" #i ", %" #i ", %16\n"
#define T_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %16, %16\n"
#define T_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define T_CVT64_32(i) "v_cvt_f64_f32 %" #i ", %1" #i "\n"            /* d[i] = a[i]; operand numbers 8 + i below */
#define T_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n"
#define T_MED3(i) "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define T_OR(i) "v_or_b32 %" #i ", %" #i ", %16\n"
#define T_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define T_CNDMASK_S(i) "v_cndmask_b32 %" #i ", %" #i ", %16, s[20:21]\n"
#define T_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_MAXDPP(i) "v_max_f32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", exec_lo, %" #i "\n"
#define T_READLANE(i) "v_readlane_b32 s20, %" #i ", 3\n"
#define T_MIN(i) "v_min_f32 %" #i ", %" #i ", %16\n"
#define T_MINU(i) "v_min_u32 %" #i ", %" #i ", %16\n"
#define T_MAXI(i) "v_max_i32 %" #i ", %" #i ", %16\n"
#define T_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
#define T_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %16\n"
#define T_FMAC(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define T_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", 48, %16\n"
#define T_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 6, %16\n"
#define T_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 20, 10\n"
#define T_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n"
#define T_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define T_CVTUF(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define T_LSHLV(i) "v_lshlrev_b32 %" #i ", %16, %" #i "\n"
#define T_LSHL6(i) "v_lshlrev_b32 %" #i ", 6, %" #i "\n"
#define T_MULNEG(i) "v_mul_f32_e64 %" #i ", -%" #i ", %16\n"
#define T_MULE64(i) "v_mul_f32_e64 %" #i ", %" #i ", %16\n"
#define T_MULSGPR(i) "v_mul_f32 %" #i ", s20, %" #i "\n"
#define T_MULCONST(i) "v_mul_f32 %" #i ", 0x3f800347, %" #i "\n"
#define T_ADDC(i) "v_addc_co_u32 %" #i ", vcc, 0, %" #i ", vcc\n"
#define T_CNDMASK_E64VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, vcc\n"
#define T_CMPCND_VCC(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define T_CMPCND_S(i) "v_cmp_lt_f32 s[20:21], %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\n"
#define T_CMPCND_VCC2(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_mul_f32 %" #i ", %" #i ", %16\nv_add_f32 %" #i ", %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define T_CMPCND_S2(i) "v_cmp_lt_f32 s[20:21], %" #i ", %16\nv_mul_f32 %" #i ", %" #i ", %16\nv_add_f32 %" #i ", %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\n"
#define T_MULADD(i) "v_mul_f32 %" #i ", %" #i ", %16\nv_add_f32 %" #i ", %" #i ", %16\n"
#define T_MULMAX(i) "v_mul_f32 %" #i ", %" #i ", %16\nv_max_f32 %" #i ", %" #i ", %16\n"
#define T_CC2(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\nv_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define T_CC2S(i) "v_cmp_lt_f32 s[20:21], %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\nv_cndmask_b32 %" #i ", %" #i ", %16, s[20:21]\n"
#define T_CC4(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\nv_cndmask_b32 %" #i ", %" #i ", %16, vcc\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\nv_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define T_CC4S(i) "v_cmp_lt_f32 s[20:21], %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\nv_cndmask_b32 %" #i ", %" #i ", %16, s[20:21]\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\nv_cndmask_b32 %" #i ", %" #i ", %16, s[20:21]\n"
#define T_CNDMUL(i) "v_cndmask_b32 %" #i ", %" #i ", %17, vcc\nv_mul_f32 %" #i ", %" #i ", %16\n"
#define T_CNDMULS(i) "v_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\nv_mul_f32 %" #i ", %" #i ", %16\n"
#define T_MAXMIN(i) "v_max_f32 %" #i ", %" #i ", %16\nv_min_f32 %" #i ", %" #i ", %17\n"
#define T_SUBMULMIN(i) "v_sub_f32 %" #i ", %" #i ", %16\nv_mul_f32 %" #i ", %" #i ", %17\nv_min_f32 %" #i ", %" #i ", %17\n"
#define T_BPERM(i) "ds_bpermute_b32 %" #i ", %16, %" #i "\n"
#define T_SNOP(i) "s_nop 0\n"
#define T_SAND(i) "s_and_b64 s[20:21], s[20:21], exec\n"

DEF32(K_MUL, T_MUL) DEF32(K_ADD, T_ADD) DEF32(K_SUB, T_SUB) DEF32(K_FMA, T_FMA) DEF32(K_MAX, T_MAX) DEF32(K_MIN3, T_MIN3)
DEF32(K_MOV, T_MOV) DEF32(K_AND, T_AND) DEF32(K_LSHL, T_LSHL) DEF32(K_CNDMASK, T_CNDMASK) DEF32(K_CMP, T_CMP)
DEF32(K_RCP, T_RCP) DEF32(K_SQRT, T_SQRT) DEF32(K_RSQ, T_RSQ) DEF32(K_MULLO, T_MULLO) DEF32(K_MULHI, T_MULHI) DEF32(K_ADDU, T_ADDU)
DEF32(K_DIVSCALE, T_DIVSCALE) DEF32(K_DIVFMAS, T_DIVFMAS) DEF32(K_DIVFIXUP, T_DIVFIXUP)
DEF32(K_SNOP, T_SNOP) DEF32(K_MAX3, T_MAX3) DEF32(K_MED3, T_MED3) DEF32(K_OR, T_OR) DEF32(K_LSHR, T_LSHR)
DEF32(K_CNDMASK_S, T_CNDMASK_S) DEF32(K_CMPX, T_CMPX) DEF32(K_MOVDPP, T_MOVDPP) DEF32(K_MAXDPP, T_MAXDPP) DEF32(K_MBCNT, T_MBCNT)
DEF32(K_MIN, T_MIN) DEF32(K_MINU, T_MINU) DEF32(K_MAXI, T_MAXI) DEF32(K_XOR, T_XOR) DEF32(K_SUBU, T_SUBU) DEF32(K_FMAC, T_FMAC)
DEF32(K_MADU24, T_MADU24) DEF32(K_LSHLADD, T_LSHLADD) DEF32(K_BFE, T_BFE) DEF32(K_ADD3, T_ADD3) DEF32(K_CVTFU, T_CVTFU) DEF32(K_CVTUF, T_CVTUF)
DEF32(K_LSHLV, T_LSHLV) DEF32(K_LSHL6, T_LSHL6) DEF32(K_MULNEG, T_MULNEG) DEF32(K_MULE64, T_MULE64) DEF32(K_MULSGPR, T_MULSGPR) DEF32(K_MULCONST, T_MULCONST)
DEF32(K_ADDC, T_ADDC) DEF32(K_CNDMASK_E64VCC, T_CNDMASK_E64VCC) DEF32(K_CMPCND_VCC, T_CMPCND_VCC) DEF32(K_CMPCND_S, T_CMPCND_S)
DEF32(K_CMPCND_VCC2, T_CMPCND_VCC2) DEF32(K_CMPCND_S2, T_CMPCND_S2) DEF32(K_MULADD, T_MULADD) DEF32(K_MULMAX, T_MULMAX)
DEF32(K_CC2, T_CC2) DEF32(K_CC2S, T_CC2S) DEF32(K_CC4, T_CC4) DEF32(K_CC4S, T_CC4S) DEF32(K_CNDMUL, T_CNDMUL) DEF32(K_CNDMULS, T_CNDMULS)
DEF32(K_MAXMIN, T_MAXMIN) DEF32(K_SUBMULMIN, T_SUBMULMIN)
DEF32(K_READLANE, T_READLANE) DEF32(K_BPERM, T_BPERM)
DEF64(K_PKMUL, T_PKMUL) DEF64(K_PKADD, T_PKADD) DEF64(K_MUL64, T_MUL64) DEF64(K_ADD64, T_ADD64) DEF64(K_FMA64, T_FMA64) DEF64(K_RCP64, T_RCP64)

// d[i] = (double)a[i]  /  a[i] = (float)d[i]  (8 per statement: operands 0-7 are d, 8-15 are a)
#define T_CVT_D_F(i) "v_cvt_f64_f32 %" #i ", %" #i "+8\n"
struct K_CVT64_32 {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                     "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                     "v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                     "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    }
};
struct K_CVT32_64 {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                     "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n"
                     "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                     "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
    }
};
// scalar stream: what the SALU issues beside nothing
struct K_SALU {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        asm volatile(R16(T_SAND) : : : "s20", "s21", "scc");
    }
};
// half vector, half scalar, interleaved: do the scalar instructions cost vector issue slots?
struct K_MUL_SALU {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        asm volatile("v_mul_f32 %0, %0, %8\ns_and_b64 s[20:21], s[20:21], exec\nv_mul_f32 %1, %1, %8\ns_and_b64 s[20:21], s[20:21], exec\n"
                     "v_mul_f32 %2, %2, %8\ns_and_b64 s[20:21], s[20:21], exec\nv_mul_f32 %3, %3, %8\ns_and_b64 s[20:21], s[20:21], exec\n"
                     "v_mul_f32 %4, %4, %8\ns_and_b64 s[20:21], s[20:21], exec\nv_mul_f32 %5, %5, %8\ns_and_b64 s[20:21], s[20:21], exec\n"
                     "v_mul_f32 %6, %6, %8\ns_and_b64 s[20:21], s[20:21], exec\nv_mul_f32 %7, %7, %8\ns_and_b64 s[20:21], s[20:21], exec\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b)
                     : "s20", "s21", "scc");
    }
};
// LDS: 16 ds_read_b128 of per-lane addresses that change every call (conflict-free: consecutive 16-byte slots)
struct K_LDS128 {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        extern __shared__ v4f lds[];
        unsigned base = __float_as_uint(a[15]);
        v4f s = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const v4f q = lds[(threadIdx.x + base + 67 * i) & 1023];
            s += q;
        }
        a[0] += s.x + s.y + s.z + s.w;
        a[15] = __uint_as_float(base + 1);
    }
};
// LDS gather as the node fetch does it: 4 x (4 x ds_read_b128 of one 64-byte record chosen per lane)
struct K_LDSNODE {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        extern __shared__ v4f lds[];
        v4f s = {0, 0, 0, 0};
        unsigned h = __float_as_uint(a[15]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const v4f *n = lds + 4 * ((h >> 8) % 255u);
            s += n[0] + n[1] + n[2] + n[3];
            h = h * 1664525u + 1013904223u;
        }
        a[0] += s.x + s.y + s.z + s.w;
        a[15] = __uint_as_float(h);
    }
};
// ... and three 16-byte reads of a 48-byte record
struct K_LDSNODE48 {
    static __device__ __forceinline__ void run(float (&a)[16], float b, float c, double (&d)[16], double e)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        extern __shared__ v4f lds[];
        v4f s = {0, 0, 0, 0};
        unsigned h = __float_as_uint(a[15]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const v4f *n = lds + 3 * ((h >> 8) % 340u);
            s += n[0] + n[1] + n[2];
            h = h * 1664525u + 1013904223u;
        }
        a[0] += s.x + s.y + s.z + s.w;
        a[15] = __uint_as_float(h);
    }
};

template <class K, int LANES>
__global__ __launch_bounds__(1024) void bench(float *out, unsigned long long *stamps, float x)
{
    extern __shared__ float lds_f[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds_f[i] = x + i;
    float a[16];
    double d[16];
    for (int i = 0; i < 16; i++) { a[i] = x + threadIdx.x + i; d[i] = a[i] * 1.5; }
    const float b = 1.0001f * x, c = 0.9999f * x;
    const double e = 1.0001 * x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    const int ln = threadIdx.x & 63;
    const bool on = LANES == 64 ? true : LANES == 32 ? ln < 32 : LANES == 16 ? ln < 16 : LANES == 1 ? ln == 0 : LANES == 33 ? (ln & 1) == 0 : (ln & 3) == 0;
    if (on)
#pragma unroll 1
    for (int it = 0; it < REP; it++) {
#pragma unroll
        for (int u = 0; u < STMTS; u++) K::run(a, b, c, d, e);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + (float)d[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        stamps[4 * w] = t0; stamps[4 * w + 1] = t1; stamps[4 * w + 2] = w0; stamps[4 * w + 3] = w1;
    }
}

template <class K, int LANES = 64>
static void run(const char *name, int per_iter, float *out, unsigned long long *stamps)
{
    printf("%-28s", name);
    for (int W = 1; W <= 4; W *= 2) {
        const int waves = 4 * W;
        std::vector<unsigned long long> h(4 * 16);
        for (int r = 0; r < 2; r++) {
            hipLaunchKernelGGL((bench<K, LANES>), dim3(1), dim3(64 * waves), 16384, 0, out, stamps, 1.0f);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * waves, hipMemcpyDeviceToHost);
        }
        unsigned long long first = ~0ull, last = 0, wf = ~0ull, wl = 0, own = 0;
        for (int w = 0; w < waves; w++) {
            first = std::min(first, h[4 * w]); last = std::max(last, h[4 * w + 1]);
            wf = std::min(wf, h[4 * w + 2]); wl = std::max(wl, h[4 * w + 3]);
            own = std::max(own, h[4 * w + 1] - h[4 * w]);
        }
        const double span = (double)(last - first), n = (double)REP * STMTS * per_iter;
        // cycles of the SIMD per wave-instruction (all W waves of a SIMD together), and the slowest wave's own cycles per instruction
        printf("  W=%d: %6.2f cyc/inst/SIMD (slowest wave %6.2f)", W, span / (n * W), (double)own / n);
    }
    printf("\n");
}

int main()
{
    float *out; unsigned long long *stamps;
    (void)hipMalloc(&out, 8192); (void)hipMalloc(&stamps, 8 * 4 * 16);
    printf("# cycles the SIMD spends per wave-instruction (s_memtime span of all its waves / instructions issued), W waves per SIMD;\n"
           "# GHz = s_memtime ticks per wall_clock64 tick (100 MHz)\n");
#define RUN(K, n) run<K>(#K, n, out, stamps)
    RUN(K_MUL, 16); RUN(K_ADD, 16); RUN(K_SUB, 16); RUN(K_FMA, 16); RUN(K_MAX, 16); RUN(K_MIN3, 16);
    RUN(K_MOV, 16); RUN(K_AND, 16); RUN(K_LSHL, 16); RUN(K_ADDU, 16); RUN(K_CNDMASK, 16); RUN(K_CMP, 16);
    RUN(K_RCP, 16); RUN(K_SQRT, 16); RUN(K_RSQ, 16); RUN(K_MULLO, 16); RUN(K_MULHI, 16);
    RUN(K_DIVSCALE, 16); RUN(K_DIVFMAS, 16); RUN(K_DIVFIXUP, 16);
    RUN(K_PKMUL, 16); RUN(K_PKADD, 16); RUN(K_MUL64, 16); RUN(K_ADD64, 16); RUN(K_FMA64, 16); RUN(K_RCP64, 16);
    RUN(K_CVT64_32, 16); RUN(K_CVT32_64, 16);
    RUN(K_SNOP, 16); RUN(K_SALU, 16); RUN(K_MUL_SALU, 16);
    RUN(K_MAX3, 16); RUN(K_MED3, 16); RUN(K_OR, 16); RUN(K_LSHR, 16); RUN(K_CNDMASK_S, 16); RUN(K_CMPX, 16);
    RUN(K_MOVDPP, 16); RUN(K_MAXDPP, 16); RUN(K_MBCNT, 16); RUN(K_READLANE, 16); RUN(K_BPERM, 16);
    RUN(K_MIN, 16); RUN(K_MINU, 16); RUN(K_MAXI, 16); RUN(K_XOR, 16); RUN(K_SUBU, 16); RUN(K_FMAC, 16); RUN(K_MADU24, 16); RUN(K_LSHLADD, 16);
    RUN(K_BFE, 16); RUN(K_ADD3, 16); RUN(K_CVTFU, 16); RUN(K_CVTUF, 16); RUN(K_LSHLV, 16); RUN(K_LSHL6, 16); RUN(K_MULNEG, 16); RUN(K_MULE64, 16);
    RUN(K_MULSGPR, 16); RUN(K_MULCONST, 16); RUN(K_ADDC, 16); RUN(K_CNDMASK_E64VCC, 16);
    printf("# pairs and quads (cycles per INSTRUCTION of the group)\n");
    RUN(K_CMPCND_VCC, 32); RUN(K_CMPCND_S, 32); RUN(K_CMPCND_VCC2, 64); RUN(K_CMPCND_S2, 64); RUN(K_MULADD, 32); RUN(K_MULMAX, 32);
    RUN(K_CC2, 48); RUN(K_CC2S, 48); RUN(K_CC4, 80); RUN(K_CC4S, 80); RUN(K_CNDMUL, 32); RUN(K_CNDMULS, 32); RUN(K_MAXMIN, 32); RUN(K_SUBMULMIN, 48);
    RUN(K_LDS128, 16); RUN(K_LDSNODE, 16); RUN(K_LDSNODE48, 12);
    printf("# EXEC-mask dependence: lanes 0-31 / 0-15 / lane 0 only / even lanes / every fourth lane\n");
#define RUNL(K, L) run<K, L>(#K " lanes=" #L, 16, out, stamps)
    RUNL(K_MUL, 32); RUNL(K_MUL, 16); RUNL(K_MUL, 1); RUNL(K_MUL, 33); RUNL(K_MUL, 17);
    RUNL(K_FMA, 32); RUNL(K_FMA, 16); RUNL(K_FMA, 1); RUNL(K_FMA, 33); RUNL(K_FMA, 17);
    RUNL(K_MAX, 32); RUNL(K_MAX, 16); RUNL(K_RCP, 32); RUNL(K_RCP, 16); RUNL(K_MUL64, 32); RUNL(K_MUL64, 16);
    return 0;
}
