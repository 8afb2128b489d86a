// Issue rate of single vector instructions on one SIMD, opcode by opcode.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_rate.hip -o build/issue_rate && build/issue_rate
// Every wave runs a stream of the ONE instruction named, written in assembly so that the compiler neither packs two
// of them into one (v_pk_*) nor folds them: eight independent destination registers, so no instruction waits for
// the one before it.  A grid of CUs x 4 SIMDs x W waves (one-wave workgroups) is timed for W = 1 and 4.
// Prints cycles per wave-instruction per SIMD (clock taken from the device's reported rate).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(OP) \
    OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")

#define KERNEL(NAME, TEXT)                                                                                         \
    __global__ __launch_bounds__(64) void NAME(float* out, int iters, float seed)                                 \
    {                                                                                                              \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, \
              a7 = seed + 7;                                                                                       \
        float b = seed * 0.5f, c = 3.0f;                                                                           \
        for (int i = 0; i < iters; ++i) {                                                                          \
            asm volatile(REP8(TEXT) REP8(TEXT) REP8(TEXT) REP8(TEXT)                                               \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                         : "v"(b), "v"(c)                                                                          \
                         : "vcc", "s20", "s21");                                                                   \
        }                                                                                                          \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 1.0f;                                    \
    }

#define T_ADD_U32(D) "v_add_u32 " D ", " D ", %8\n"
#define T_FMA_F32(D) "v_fma_f32 " D ", " D ", %8, %9\n"
#define T_MUL_F32(D) "v_mul_f32 " D ", " D ", %8\n"
#define T_ADD_F32(D) "v_add_f32 " D ", " D ", %8\n"
#define T_CNDMASK(D) "v_cndmask_b32 " D ", " D ", %8, vcc\n"
#define T_CVT_F32_I32(D) "v_cvt_f32_i32 " D ", " D "\n"
#define T_CVT_I32_F32(D) "v_cvt_i32_f32 " D ", " D "\n"
#define T_MUL_I24(D) "v_mul_i32_i24 " D ", " D ", %8\n"
#define T_MAD_I24(D) "v_mad_i32_i24 " D ", " D ", %8, %9\n"
#define T_LSHL_ADD(D) "v_lshl_add_u32 " D ", " D ", 3, %8\n"
#define T_ADD3(D) "v_add3_u32 " D ", " D ", %8, %9\n"
#define T_OR3(D) "v_or3_b32 " D ", " D ", %8, %9\n"
#define T_AND(D) "v_and_b32 " D ", " D ", %8\n"
#define T_MAX_I32(D) "v_max_i32 " D ", " D ", %8\n"
#define T_MAX3_I32(D) "v_max3_i32 " D ", " D ", %8, %9\n"
#define T_ASHR(D) "v_ashrrev_i32 " D ", 8, " D "\n"
#define T_BFE(D) "v_bfe_u32 " D ", " D ", 5, 5\n"
#define T_MUL_LO(D) "v_mul_lo_u32 " D ", " D ", %8\n"
#define T_RCP(D) "v_rcp_f32 " D ", " D "\n"
#define T_FLOOR(D) "v_floor_f32 " D ", " D "\n"
#define T_RNDNE(D) "v_rndne_f32 " D ", " D "\n"
#define T_LDEXP(D) "v_ldexp_f32 " D ", " D ", %8\n"
#define T_MOV(D) "v_mov_b32 " D ", %8\n"
#define T_MOV_DPP(D) "v_mov_b32_dpp " D ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_CMP(D) "v_cmp_lt_i32 vcc, " D ", %8\n"
#define T_MIN_F32(D) "v_min_f32 " D ", " D ", %8\n"
#define T_MED3_F32(D) "v_med3_f32 " D ", " D ", %8, %9\n"
#define T_SUB_U32(D) "v_sub_u32 " D ", " D ", %8\n"
#define T_LSHL_OR(D) "v_lshl_or_b32 " D ", " D ", 8, %8\n"
#define T_XAD(D) "v_xad_u32 " D ", " D ", %8, %9\n"

#define T_OR(D) "v_or_b32 " D ", " D ", %8\n"
#define T_XOR(D) "v_xor_b32 " D ", " D ", %8\n"
#define T_LSHL(D) "v_lshlrev_b32 " D ", 3, " D "\n"
#define T_LSHR(D) "v_lshrrev_b32 " D ", 3, " D "\n"
#define T_SUB_F32(D) "v_sub_f32 " D ", " D ", %8\n"
#define T_FMAC(D) "v_fmac_f32 " D ", %8, %9\n"
#define T_MUL_U24(D) "v_mul_u32_u24 " D ", " D ", %8\n"
#define T_CVT_F32_U32(D) "v_cvt_f32_u32 " D ", " D "\n"
#define T_CVT_U32_F32(D) "v_cvt_u32_f32 " D ", " D "\n"
#define T_MIN_I32(D) "v_min_i32 " D ", " D ", %8\n"
#define T_MIN_U32(D) "v_min_u32 " D ", " D ", %8\n"
#define T_MAX_F32(D) "v_max_f32 " D ", " D ", %8\n"
#define T_MUL_HI(D) "v_mul_hi_u32 " D ", " D ", %8\n"
#define T_MAD_U24(D) "v_mad_u32_u24 " D ", " D ", %8, %9\n"
#define T_PERM(D) "v_perm_b32 " D ", " D ", %8, %9\n"
#define T_ALIGNBIT(D) "v_alignbit_b32 " D ", " D ", %8, 8\n"
#define T_AND_OR(D) "v_and_or_b32 " D ", " D ", %8, %9\n"
#define T_ADD_LSHL(D) "v_add_lshl_u32 " D ", " D ", %8, 2\n"
#define T_SUBREV(D) "v_subrev_u32 " D ", " D ", %8\n"
#define T_ADD_CO(D) "v_add_co_u32 " D ", vcc, " D ", %8\n"
#define T_CMP_CND(D) "v_cmp_lt_i32 vcc, " D ", %8\nv_cndmask_b32 " D ", " D ", %9, vcc\n"
#define T_CMP_F32(D) "v_cmp_lt_f32 vcc, " D ", %8\n"
#define T_CMP_E64(D) "v_cmp_lt_i32 s[20:21], " D ", %8\n"
#define T_CND_E64(D) "v_cndmask_b32 " D ", " D ", %8, s[20:21]\n"
#define T_READLANE(D) "v_readlane_b32 s20, " D ", 5\n"
#define T_EXP(D) "v_exp_f32 " D ", " D "\n"
#define T_SQRT(D) "v_sqrt_f32 " D ", " D "\n"
#define T_TRUNC(D) "v_trunc_f32 " D ", " D "\n"
#define T_FRACT(D) "v_fract_f32 " D ", " D "\n"

#define T_MIX_FMA_ADD(D) "v_fma_f32 " D ", " D ", %8, %9\nv_add_u32 " D ", " D ", %8\n"
#define T_MIX_FMA3_ADD(D) "v_fma_f32 " D ", " D ", %8, %9\nv_fma_f32 " D ", " D ", %8, %9\nv_fma_f32 " D ", " D ", %8, %9\nv_add_u32 " D ", " D ", %8\n"
#define T_MIX_FMA_ADD3(D) "v_fma_f32 " D ", " D ", %8, %9\nv_add_u32 " D ", " D ", %8\nv_and_b32 " D ", " D ", %8\nv_sub_u32 " D ", " D ", %8\n"
#define T_MIX_CVT_AND(D) "v_cvt_f32_i32 " D ", " D "\nv_and_b32 " D ", " D ", %8\n"
#define T_MIX_ADD_MUL(D) "v_add_u32 " D ", " D ", %8\nv_mul_f32 " D ", " D ", %8\n"
#define T_MIX_CND_ADD(D) "v_cndmask_b32 " D ", " D ", %8, s[20:21]\nv_add_u32 " D ", " D ", %8\n"

KERNEL(k_add_u32, T_ADD_U32)
KERNEL(k_mix_fma_add, T_MIX_FMA_ADD)
KERNEL(k_mix_fma3_add, T_MIX_FMA3_ADD)
KERNEL(k_mix_fma_add3, T_MIX_FMA_ADD3)
KERNEL(k_mix_cvt_and, T_MIX_CVT_AND)
KERNEL(k_mix_add_mul, T_MIX_ADD_MUL)
KERNEL(k_mix_cnd_add, T_MIX_CND_ADD)

KERNEL(k_or, T_OR)
KERNEL(k_xor, T_XOR)
KERNEL(k_lshl, T_LSHL)
KERNEL(k_lshr, T_LSHR)
KERNEL(k_sub_f32, T_SUB_F32)
KERNEL(k_fmac, T_FMAC)
KERNEL(k_mul_u24, T_MUL_U24)
KERNEL(k_cvt_f32_u32, T_CVT_F32_U32)
KERNEL(k_cvt_u32_f32, T_CVT_U32_F32)
KERNEL(k_min_i32, T_MIN_I32)
KERNEL(k_min_u32, T_MIN_U32)
KERNEL(k_max_f32, T_MAX_F32)
KERNEL(k_mul_hi, T_MUL_HI)
KERNEL(k_mad_u24, T_MAD_U24)
KERNEL(k_perm, T_PERM)
KERNEL(k_alignbit, T_ALIGNBIT)
KERNEL(k_and_or, T_AND_OR)
KERNEL(k_add_lshl, T_ADD_LSHL)
KERNEL(k_subrev, T_SUBREV)
KERNEL(k_add_co, T_ADD_CO)
KERNEL(k_cmp_cnd, T_CMP_CND)
KERNEL(k_cmp_f32, T_CMP_F32)
KERNEL(k_cmp_e64, T_CMP_E64)
KERNEL(k_cnd_e64, T_CND_E64)
KERNEL(k_readlane, T_READLANE)
KERNEL(k_exp, T_EXP)
KERNEL(k_sqrt, T_SQRT)
KERNEL(k_trunc, T_TRUNC)
KERNEL(k_fract, T_FRACT)

KERNEL(k_sub_u32, T_SUB_U32)
KERNEL(k_fma_f32, T_FMA_F32)
KERNEL(k_mul_f32, T_MUL_F32)
KERNEL(k_add_f32, T_ADD_F32)
KERNEL(k_cndmask, T_CNDMASK)
KERNEL(k_cvt_f32_i32, T_CVT_F32_I32)
KERNEL(k_cvt_i32_f32, T_CVT_I32_F32)
KERNEL(k_mul_i24, T_MUL_I24)
KERNEL(k_mad_i24, T_MAD_I24)
KERNEL(k_lshl_add, T_LSHL_ADD)
KERNEL(k_add3, T_ADD3)
KERNEL(k_or3, T_OR3)
KERNEL(k_and, T_AND)
KERNEL(k_max_i32, T_MAX_I32)
KERNEL(k_max3_i32, T_MAX3_I32)
KERNEL(k_ashr, T_ASHR)
KERNEL(k_bfe, T_BFE)
KERNEL(k_mul_lo, T_MUL_LO)
KERNEL(k_rcp, T_RCP)
KERNEL(k_floor, T_FLOOR)
KERNEL(k_rndne, T_RNDNE)
KERNEL(k_ldexp, T_LDEXP)
KERNEL(k_mov, T_MOV)
KERNEL(k_mov_dpp, T_MOV_DPP)
KERNEL(k_cmp, T_CMP)
KERNEL(k_min_f32, T_MIN_F32)
KERNEL(k_med3_f32, T_MED3_F32)
KERNEL(k_lshl_or, T_LSHL_OR)
KERNEL(k_xad, T_XAD)

// two fp32 operations per lane and instruction: register pairs
__global__ __launch_bounds__(64) void k_pk_fma_f32(float* out, int iters, float seed)
{
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7};
    float2v a4 = a0 + 8.0f, a5 = a1 + 8.0f, a6 = a2 + 8.0f, a7 = a3 + 8.0f;
    float2v b = {seed * 0.5f, seed * 0.25f}, c = {3.0f, 4.0f};
#define T_PK(D) "v_pk_fma_f32 " D ", " D ", %8, %9\n"
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8(T_PK) REP8(T_PK) REP8(T_PK) REP8(T_PK)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    }
    const float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 12345.678f) out[0] = 1.0f;
}

typedef void (*Kernel)(float*, int, float);

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const double clock_hz = prop.clockRate * 1e3;
    const int cus = prop.multiProcessorCount;
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, clock_hz / 1e6);
    struct { const char* name; Kernel k; int per; } table[] = {
        {"v_add_u32", k_add_u32}, {"v_sub_u32", k_sub_u32}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32}, {"v_min_f32", k_min_f32}, {"v_med3_f32", k_med3_f32},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_i32", k_cmp}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
        {"v_mul_i32_i24", k_mul_i24}, {"v_mad_i32_i24", k_mad_i24}, {"v_lshl_add_u32", k_lshl_add}, {"v_lshl_or_b32", k_lshl_or},
        {"v_add3_u32", k_add3}, {"v_xad_u32", k_xad}, {"v_or3_b32", k_or3}, {"v_and_b32", k_and}, {"v_max_i32", k_max_i32},
        {"v_max3_i32", k_max3_i32}, {"v_ashrrev_i32", k_ashr}, {"v_bfe_u32", k_bfe}, {"v_mul_lo_u32", k_mul_lo},
        {"v_rcp_f32", k_rcp}, {"v_floor_f32", k_floor}, {"v_rndne_f32", k_rndne}, {"v_ldexp_f32", k_ldexp},
        {"v_mov_b32", k_mov}, {"v_mov_b32 dpp", k_mov_dpp},
        {"v_or_b32", k_or},
        {"v_xor_b32", k_xor},
        {"v_lshlrev_b32", k_lshl},
        {"v_lshrrev_b32", k_lshr},
        {"v_sub_f32", k_sub_f32},
        {"v_fmac_f32", k_fmac},
        {"v_mul_u32_u24", k_mul_u24},
        {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_cvt_u32_f32", k_cvt_u32_f32},
        {"v_min_i32", k_min_i32},
        {"v_min_u32", k_min_u32},
        {"v_max_f32", k_max_f32},
        {"v_mul_hi_u32", k_mul_hi},
        {"v_mad_u32_u24", k_mad_u24},
        {"v_perm_b32", k_perm},
        {"v_alignbit_b32", k_alignbit},
        {"v_and_or_b32", k_and_or},
        {"v_add_lshl_u32", k_add_lshl},
        {"v_subrev_u32", k_subrev},
        {"v_add_co_u32", k_add_co},
        {"mix: fma + add_u32", k_mix_fma_add, 2}, {"mix: 3 fma + add_u32", k_mix_fma3_add, 4}, {"mix: fma + add + and + sub", k_mix_fma_add3, 4},
        {"mix: cvt_f32_i32 + and", k_mix_cvt_and, 2}, {"mix: add_u32 + mul_f32", k_mix_add_mul, 2}, {"mix: cndmask(sgpr) + add_u32", k_mix_cnd_add, 2},
        {"v_cmp + v_cndmask (vcc)", k_cmp_cnd, 2},
        {"v_cmp_lt_f32", k_cmp_f32},
        {"v_cmp_lt_i32 e64 sgpr", k_cmp_e64},
        {"v_cndmask_b32 e64 sgpr", k_cnd_e64},
        {"v_readlane_b32", k_readlane},
        {"v_exp_f32", k_exp},
        {"v_sqrt_f32", k_sqrt},
        {"v_trunc_f32", k_trunc},
        {"v_fract_f32", k_fract},
    };
    for (auto& t : table) {
        printf("%-30s", t.name);
        for (int w : {1, 4}) {
            const int blocks = cus * 4 * w;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(t.k, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
                (void)hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
            }
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = (double)iters * 32 * w * (t.per ? t.per : 1);
            printf("  %d wave%s/SIMD: %5.2f cycles each", w, w == 1 ? " " : "s", ms * 1e-3 * clock_hz / per_simd);
        }
        printf("\n");
    }
    return 0;
}
