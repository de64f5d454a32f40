// valu_rate.hip -- what one gfx950 SIMD really issues: wave64 instructions per cycle by instruction kind,
// operand form and occupancy. Each kind is a block of 32 instructions written as inline assembly on explicit
// registers (values are irrelevant for timing; explicit registers fix the VGPR bank of every operand: bank =
// register number mod 4), so the compiler cannot pack, fuse, reorder or drop anything. Waves stamp themselves
// with s_memtime and their hardware id, so rates are computed per SIMD from the waves that SIMD really hosted,
// whatever the dispatcher's placement was.
//   hipcc --offload-arch=gfx950 -O2 -w tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate [filter]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess) {                                                                                         \
            printf("hip error %s at line %d\n", hipGetErrorString(e), __LINE__);                                       \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

// destination / chain registers: eight in banks 2 and 3 ("D"), eight in bank 0 ("Z")
#define D8(M) M(22) M(23) M(26) M(27) M(30) M(31) M(34) M(35)
#define Z8(M) M(24) M(28) M(32) M(36) M(40) M(44) M(48) M(52)
#define X4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define CLOBBER                                                                                                        \
    "v16", "v17", "v20", "v22", "v23", "v26", "v27", "v30", "v31", "v34", "v35", "v24", "v28", "v32", "v36", "v40",    \
        "v44", "v48", "v52", "v56", "v57", "v58", "v59", "v60", "v61", "v64", "v65", "v68", "v69", "v72", "v73", "s20", "s21", "s22", "s23", "vcc"
// v16: bank 0, v17: bank 1, v20: bank 0

#define I_FMA_FREE(d) "v_fma_f32 v" #d ", v" #d ", v16, v17\n"
#define I_FMA_AB(d) "v_fma_f32 v" #d ", v" #d ", v16, v20\n"
#define I_FMA_SGPR(d) "v_fma_f32 v" #d ", v" #d ", s20, v17\n"
#define I_FMA_SGPR2(d) "v_fma_f32 v" #d ", s20, v16, v17\n"
#define I_FMA_CONST(d) "v_fma_f32 v" #d ", v" #d ", 2.0, v17\n"
#define I_FMAC(d) "v_fmac_f32_e32 v" #d ", v16, v17\n"
#define I_FMAC_SGPR(d) "v_fmac_f32_e32 v" #d ", s20, v17\n"
#define I_FMAAK(d) "v_fmaak_f32 v" #d ", v16, v17, 0x3a83126f\n"
#define I_MUL(d) "v_mul_f32_e32 v" #d ", v16, v" #d "\n"
#define I_MUL_SGPR(d) "v_mul_f32_e32 v" #d ", s20, v" #d "\n"
#define I_MUL_LIT(d) "v_mul_f32_e32 v" #d ", 0x3a83126f, v" #d "\n"
#define I_MUL_E64(d) "v_mul_f32_e64 v" #d ", v16, v" #d "\n"
#define I_MUL_NEG(d) "v_mul_f32_e64 v" #d ", v16, -v" #d "\n"
#define I_ADD(d) "v_add_f32_e32 v" #d ", v16, v17\n"
#define I_MAX(d) "v_max_f32_e32 v" #d ", v16, v17\n"
#define I_MOV(d) "v_mov_b32_e32 v" #d ", v16\n"
#define I_MOV_SGPR(d) "v_mov_b32_e32 v" #d ", s20\n"
#define I_MOV_CONST(d) "v_mov_b32_e32 v" #d ", 1.0\n"
#define I_CND(d) "v_cndmask_b32_e32 v" #d ", v16, v17, vcc\n"
#define I_CND_E64(d) "v_cndmask_b32_e64 v" #d ", v16, v17, s[22:23]\n"
#define I_CMP(d) "v_cmp_gt_f32_e32 vcc, v16, v" #d "\n"
#define I_CMP_E64(d) "v_cmp_gt_f32_e64 s[22:23], v16, v" #d "\n"
#define I_CMP_SGPR(d) "v_cmp_gt_f32_e32 vcc, s20, v" #d "\n"
#define I_CMP_ABS(d) "v_cmp_nlt_f32_e64 s[22:23], |v" #d "|, s20\n"
#define I_ADDU(d) "v_add_u32_e32 v" #d ", v16, v" #d "\n"
#define I_XOR(d) "v_xor_b32_e32 v" #d ", v16, v" #d "\n"
#define I_ALIGN(d) "v_alignbit_b32 v" #d ", v" #d ", v" #d ", 7\n"
#define I_MULLO(d) "v_mul_lo_u32 v" #d ", v" #d ", v16\n"
#define I_LSHLADD(d) "v_lshl_add_u32 v" #d ", v" #d ", 3, v16\n"
#define I_CVT(d) "v_cvt_f32_u32_e32 v" #d ", v16\n"
#define I_RCP(d) "v_rcp_f32_e32 v" #d ", v16\n"
#define I_RSQ(d) "v_rsq_f32_e32 v" #d ", v16\n"
#define I_SQRT(d) "v_sqrt_f32_e32 v" #d ", v16\n"
#define I_SIN(d) "v_sin_f32_e32 v" #d ", v16\n"
#define I_EXP(d) "v_exp_f32_e32 v" #d ", v16\n"
#define I_LOG(d) "v_log_f32_e32 v" #d ", v16\n"
#define I_PKFMA(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[16:17], v[56:57]\n"
#define P8(M) M(22) M(26) M(30) M(34) M(60) M(64) M(68) M(72)
#define I_READLANE(d) "v_readlane_b32 s21, v" #d ", 3\n"
#define I_FMA_SALU(d) "v_fma_f32 v" #d ", v" #d ", v16, v17\n s_add_u32 s21, s21, 1\n"
#define I_FMA_SALU2(d) "v_fma_f32 v" #d ", v" #d ", v16, v17\n s_add_u32 s21, s21, 1\n s_and_b64 s[22:23], s[22:23], vcc\n"

#define I_SUB(d) "v_sub_f32_e32 v" #d ", v16, v17\n"
#define I_AND(d) "v_and_b32_e32 v" #d ", v16, v17\n"
#define I_OR(d) "v_or_b32_e32 v" #d ", v16, v17\n"
#define I_LSHL(d) "v_lshlrev_b32_e32 v" #d ", 3, v17\n"
#define I_LSHR(d) "v_lshrrev_b32_e32 v" #d ", 3, v17\n"
#define I_BFE(d) "v_bfe_u32 v" #d ", v16, 3, 8\n"
#define I_MAD64(d) "v_mad_u64_u32 v[" #d ":" #d "+1], s[22:23], v16, v17, v[56:57]\n"
#define I_ADDCO(d) "v_add_co_u32_e32 v" #d ", vcc, v16, v17\n"
#define I_ADDC(d) "v_addc_co_u32_e32 v" #d ", vcc, v16, v17, vcc\n"
#define I_MINF(d) "v_min_f32_e32 v" #d ", v16, v17\n"
#define I_MED3(d) "v_med3_f32 v" #d ", v16, v17, v" #d "\n"
#define I_MIN3(d) "v_min3_f32 v" #d ", v16, v17, v" #d "\n"
#define I_MINU(d) "v_min_u32_e32 v" #d ", v16, v17\n"
#define I_MAXI(d) "v_max_i32_e32 v" #d ", v16, v17\n"
#define I_CMPU(d) "v_cmp_le_u32_e32 vcc, v16, v" #d "\n"
#define I_CMPCND(d) "v_cmp_gt_f32_e32 vcc, v16, v" #d "\n v_cndmask_b32_e32 v" #d ", v16, v17, vcc\n"
#define I_CND_VCC64(d) "v_cndmask_b32_e64 v" #d ", v16, v17, vcc\n"
#define I_CND_CONST(d) "v_cndmask_b32_e64 v" #d ", 0, 1.0, s[22:23]\n"
#define I_FMA_CLAMP(d) "v_fma_f32 v" #d ", v" #d ", v16, v17 clamp\n"
#define I_MUL_CLAMP(d) "v_mul_f32_e64 v" #d ", v16, v" #d " clamp\n"
#define I_MUL_ABS(d) "v_mul_f32_e64 v" #d ", |v16|, v" #d "\n"
#define I_CVTU(d) "v_cvt_u32_f32_e32 v" #d ", v16\n"
#define I_CVTI(d) "v_cvt_f32_i32_e32 v" #d ", v16\n"
#define I_MAD24(d) "v_mad_u32_u24 v" #d ", v16, v17, v" #d "\n"
#define I_MUL24(d) "v_mul_u32_u24_e32 v" #d ", v16, v17\n"
#define I_MULHI(d) "v_mul_hi_u32 v" #d ", v16, v17\n"
#define I_ADD3(d) "v_add3_u32 v" #d ", v16, v17, v" #d "\n"
#define I_ANDOR(d) "v_and_or_b32 v" #d ", v16, v17, v" #d "\n"
#define I_LSHLOR(d) "v_lshl_or_b32 v" #d ", v16, 3, v" #d "\n"
#define I_SUBU(d) "v_sub_u32_e32 v" #d ", v16, v17\n"
#define I_FLOOR(d) "v_floor_f32_e32 v" #d ", v16\n"
#define I_FRACT(d) "v_fract_f32_e32 v" #d ", v16\n"
#define I_RCPIFLAG(d) "v_rcp_iflag_f32_e32 v" #d ", v16\n"
#define I_WRITELANE(d) "v_writelane_b32 v" #d ", s20, 3\n"
#define I_DSREAD(d) "ds_read_b128 v[56:59], v20\n"
#define I_SAND(d) "s_and_b64 s[22:23], s[22:23], exec\n"
#define I_SADD(d) "s_add_u32 s21, s21, 1\n"
#define I_SMOV(d) "s_mov_b32 s21, 1\n"

#define I_MUL_SAME(d) "v_mul_f32_e32 v" #d ", v16, v20\n"
#define I_ADD_SAME(d) "v_add_f32_e32 v" #d ", v16, v20\n"
#define I_FMAC_S01(d) "v_fmac_f32_e32 v" #d ", v16, v20\n"
#define I_FMAC_DS0(d) "v_fmac_f32_e32 v" #d ", v16, v17\n"
#define I_FMAC_DS1(d) "v_fmac_f32_e32 v" #d ", v17, v16\n"
#define I_FMA_S02(d) "v_fma_f32 v" #d ", v16, v17, v20\n"
#define I_FMA_S01(d) "v_fma_f32 v" #d ", v16, v20, v17\n"
#define I_FMA_S12(d) "v_fma_f32 v" #d ", v17, v16, v20\n"
#define I_CND_SAME(d) "v_cmp_gt_f32_e32 vcc, v16, v" #d "\n v_cndmask_b32_e32 v" #d ", v16, v20, vcc\n"
// one transcendental among seven FMAs / one per three
#define I_MIX_RCP8 "v_rcp_f32_e32 v22, v16\n" I_FMA_FREE(23) I_FMA_FREE(26) I_FMA_FREE(27) I_FMA_FREE(30) I_FMA_FREE(31) I_FMA_FREE(34) I_FMA_FREE(35)
#define I_MIX_RCP4 "v_rcp_f32_e32 v22, v16\n" I_FMA_FREE(23) I_FMA_FREE(26) I_FMA_FREE(27) "v_rcp_f32_e32 v30, v16\n" I_FMA_FREE(31) I_FMA_FREE(34) I_FMA_FREE(35)
// LDS broadcast read next to VALU work: 1 ds_read_b128 per 8 FMAs (v56..v59 receive)
#define I_MIX_DS8 "ds_read_b128 v[56:59], v20\n" I_FMA_FREE(22) I_FMA_FREE(23) I_FMA_FREE(26) I_FMA_FREE(27) I_FMA_FREE(30) I_FMA_FREE(31) I_FMA_FREE(34) I_FMA_FREE(35)

#define KINDS(K)                                                                                                       \
    K(fma_free, "v_fma d(2,3) a(0) b(1)", X4(D8(I_FMA_FREE)))                                                          \
    K(fma_ab, "v_fma a,b same bank", X4(D8(I_FMA_AB)))                                                                 \
    K(fma_da, "v_fma d,a same bank", X4(Z8(I_FMA_FREE)))                                                               \
    K(fma_all, "v_fma d,a,b same bank", X4(Z8(I_FMA_AB)))                                                              \
    K(fma_sgpr, "v_fma v,s,v", X4(D8(I_FMA_SGPR)))                                                                     \
    K(fma_sgpr2, "v_fma s,v,v", X4(D8(I_FMA_SGPR2)))                                                                   \
    K(fma_const, "v_fma v,2.0,v", X4(D8(I_FMA_CONST)))                                                                 \
    K(fmac, "v_fmac_e32 v,v", X4(D8(I_FMAC)))                                                                          \
    K(fmac_sgpr, "v_fmac_e32 s,v", X4(D8(I_FMAC_SGPR)))                                                                \
    K(fmaak, "v_fmaak literal", X4(D8(I_FMAAK)))                                                                       \
    K(mul, "v_mul_e32 v,v", X4(D8(I_MUL)))                                                                             \
    K(mul_sgpr, "v_mul_e32 s,v", X4(D8(I_MUL_SGPR)))                                                                   \
    K(mul_lit, "v_mul_e32 literal,v", X4(D8(I_MUL_LIT)))                                                               \
    K(mul_e64, "v_mul_e64 v,v", X4(D8(I_MUL_E64)))                                                                     \
    K(mul_neg, "v_mul_e64 v,-v", X4(D8(I_MUL_NEG)))                                                                    \
    K(add, "v_add_e32 (no chain)", X4(D8(I_ADD)))                                                                      \
    K(max, "v_max_e32", X4(D8(I_MAX)))                                                                                 \
    K(mov, "v_mov v", X4(D8(I_MOV)))                                                                                   \
    K(mov_sgpr, "v_mov s", X4(D8(I_MOV_SGPR)))                                                                         \
    K(mov_const, "v_mov 1.0", X4(D8(I_MOV_CONST)))                                                                     \
    K(cnd, "v_cndmask_e32 vcc", X4(D8(I_CND)))                                                                         \
    K(cnd_e64, "v_cndmask_e64 s[]", X4(D8(I_CND_E64)))                                                                 \
    K(cmp, "v_cmp_e32 -> vcc", X4(D8(I_CMP)))                                                                          \
    K(cmp_e64, "v_cmp_e64 -> s[]", X4(D8(I_CMP_E64)))                                                                  \
    K(cmp_sgpr, "v_cmp_e32 s,v", X4(D8(I_CMP_SGPR)))                                                                   \
    K(cmp_abs, "v_cmp_e64 |v|,s -> s[]", X4(D8(I_CMP_ABS)))                                                            \
    K(addu, "v_add_u32", X4(D8(I_ADDU)))                                                                               \
    K(xor_, "v_xor_b32", X4(D8(I_XOR)))                                                                                \
    K(align, "v_alignbit_b32", X4(D8(I_ALIGN)))                                                                        \
    K(mullo, "v_mul_lo_u32", X4(D8(I_MULLO)))                                                                          \
    K(lshladd, "v_lshl_add_u32", X4(D8(I_LSHLADD)))                                                                    \
    K(cvt, "v_cvt_f32_u32", X4(D8(I_CVT)))                                                                             \
    K(rcp, "v_rcp_f32", X4(D8(I_RCP)))                                                                                 \
    K(rsq, "v_rsq_f32", X4(D8(I_RSQ)))                                                                                 \
    K(sqrt_, "v_sqrt_f32", X4(D8(I_SQRT)))                                                                             \
    K(sin_, "v_sin_f32", X4(D8(I_SIN)))                                                                                \
    K(exp_, "v_exp_f32", X4(D8(I_EXP)))                                                                                \
    K(log_, "v_log_f32", X4(D8(I_LOG)))                                                                                \
    K(pkfma, "v_pk_fma_f32 (2 FMA)", X4(P8(I_PKFMA)))                                                                  \
    K(readlane, "v_readlane_b32", X4(D8(I_READLANE)))                                                                  \
    K(sub, "v_sub_f32", X4(D8(I_SUB)))                                                                                 \
    K(and_, "v_and_b32", X4(D8(I_AND)))                                                                                \
    K(or_, "v_or_b32", X4(D8(I_OR)))                                                                                   \
    K(lshl, "v_lshlrev_b32", X4(D8(I_LSHL)))                                                                           \
    K(lshr, "v_lshrrev_b32", X4(D8(I_LSHR)))                                                                           \
    K(bfe, "v_bfe_u32", X4(D8(I_BFE)))                                                                                 \
    K(mad64, "v_mad_u64_u32", X4(P8(I_MAD64)))                                                                         \
    K(addco, "v_add_co_u32 -> vcc", X4(D8(I_ADDCO)))                                                                   \
    K(addc, "v_addc_co_u32", X4(D8(I_ADDC)))                                                                           \
    K(minf, "v_min_f32", X4(D8(I_MINF)))                                                                               \
    K(med3, "v_med3_f32", X4(D8(I_MED3)))                                                                              \
    K(min3, "v_min3_f32", X4(D8(I_MIN3)))                                                                              \
    K(minu, "v_min_u32", X4(D8(I_MINU)))                                                                               \
    K(maxi, "v_max_i32", X4(D8(I_MAXI)))                                                                               \
    K(cmpu, "v_cmp_le_u32 -> vcc", X4(D8(I_CMPU)))                                                                     \
    K(cmpcnd, "v_cmp + v_cndmask (16 pairs)", X4(I_CMPCND(22) I_CMPCND(23) I_CMPCND(26) I_CMPCND(27)))                 \
    K(cnd_vcc64, "v_cndmask_e64 vcc", X4(D8(I_CND_VCC64)))                                                             \
    K(cnd_const, "v_cndmask_e64 0,1.0,s[]", X4(D8(I_CND_CONST)))                                                       \
    K(fma_clamp, "v_fma clamp", X4(D8(I_FMA_CLAMP)))                                                                   \
    K(mul_clamp, "v_mul_e64 clamp", X4(D8(I_MUL_CLAMP)))                                                               \
    K(mul_abs, "v_mul_e64 |v|", X4(D8(I_MUL_ABS)))                                                                     \
    K(cvtu, "v_cvt_u32_f32", X4(D8(I_CVTU)))                                                                           \
    K(cvti, "v_cvt_f32_i32", X4(D8(I_CVTI)))                                                                           \
    K(mad24, "v_mad_u32_u24", X4(D8(I_MAD24)))                                                                         \
    K(mul24, "v_mul_u32_u24", X4(D8(I_MUL24)))                                                                         \
    K(mulhi, "v_mul_hi_u32", X4(D8(I_MULHI)))                                                                          \
    K(add3, "v_add3_u32", X4(D8(I_ADD3)))                                                                              \
    K(andor, "v_and_or_b32", X4(D8(I_ANDOR)))                                                                          \
    K(lshlor, "v_lshl_or_b32", X4(D8(I_LSHLOR)))                                                                       \
    K(subu, "v_sub_u32", X4(D8(I_SUBU)))                                                                               \
    K(floor_, "v_floor_f32", X4(D8(I_FLOOR)))                                                                          \
    K(fract, "v_fract_f32", X4(D8(I_FRACT)))                                                                           \
    K(rcpiflag, "v_rcp_iflag_f32", X4(D8(I_RCPIFLAG)))                                                                 \
    K(writelane, "v_writelane_b32", X4(D8(I_WRITELANE)))                                                               \
    K(dsread, "ds_read_b128 (broadcast)", X4(D8(I_DSREAD)))                                                            \
    K(sand, "s_and_b64 (chain)", X4(D8(I_SAND)))                                                                       \
    K(sadd, "s_add_u32 (chain)", X4(D8(I_SADD)))                                                                       \
    K(smov, "s_mov_b32", X4(D8(I_SMOV)))                                                                               \
    K(bk_mul01, "v_mul src0,src1 same bank", X4(D8(I_MUL_SAME)))                                                       \
    K(bk_add01, "v_add src0,src1 same bank", X4(D8(I_ADD_SAME)))                                                       \
    K(bk_fmac01, "v_fmac src0,src1 same bank", X4(D8(I_FMAC_S01)))                                                     \
    K(bk_fmacd0, "v_fmac dst,src0 same bank", X4(Z8(I_FMAC_DS0)))                                                      \
    K(bk_fmacd1, "v_fmac dst,src1 same bank", X4(Z8(I_FMAC_DS1)))                                                      \
    K(bk_fma02, "v_fma src0,src2 same bank", X4(D8(I_FMA_S02)))                                                        \
    K(bk_fma01, "v_fma src0,src1 same bank", X4(D8(I_FMA_S01)))                                                        \
    K(bk_fma12, "v_fma src1,src2 same bank", X4(D8(I_FMA_S12)))                                                        \
    K(fma_salu, "v_fma + 1 SALU (32 VALU)", X4(D8(I_FMA_SALU)))                                                        \
    K(fma_salu2, "v_fma + 2 SALU (32 VALU)", X4(D8(I_FMA_SALU2)))                                                      \
    K(mix_rcp8, "1 rcp + 7 fma", X4(I_MIX_RCP8))                                                                       \
    K(mix_rcp4, "2 rcp + 6 fma", X4(I_MIX_RCP4))                                                                       \
    K(mix_ds8, "8 fma + ds_read_b128 (32 VALU)", X4(I_MIX_DS8))

#define DEFINE_KERNEL(id, label, body)                                                                                 \
    __global__ void __launch_bounds__(256) k_##id(unsigned long long* rec, int iters)                                  \
    {                                                                                                                  \
        __shared__ float lds[64];                                                                                      \
        lds[threadIdx.x & 63] = 1.0f;                                                                                  \
        asm volatile("v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0.5\n v_mov_b32 v20, 0\n s_mov_b32 s20, 1.0\n s_mov_b64 vcc, exec\n s_mov_b64 s[22:23], exec\n" ::: CLOBBER);  \
        __syncthreads();                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < iters; i++)                                                                                \
            asm volatile(body "s_waitcnt lgkmcnt(0)\n" ::: CLOBBER, "memory");                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if ((threadIdx.x & 63) == 0) {                                                                                 \
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  /* HW_REG_HW_ID */       \
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); /* HW_REG_XCC_ID */      \
            unsigned long long* r = rec + 3 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);                          \
            r[0] = t0;                                                                                                 \
            r[1] = t1;                                                                                                 \
            r[2] = ((unsigned long long)xcc << 32) | (hw & 0xfff0u); /* simd, pipe, cu, sh, se */                      \
        }                                                                                                              \
    }
KINDS(DEFINE_KERNEL)

typedef void (*KernelFn)(unsigned long long*, int);
struct Kind
{
    const char* id;
    const char* label;
    KernelFn fn;
};
#define TABLE_ENTRY(id, label, body) {#id, label, k_##id},
static const Kind kinds[] = {KINDS(TABLE_ENTRY)};

static unsigned long long* gRec;
static hipEvent_t gE0, gE1;

static int run(const Kind& k, int wavesPerSimd, int iters)
{
    const int blocks = 256 * wavesPerSimd; // a 256-thread block puts one wave on each SIMD of a CU
    for (int rep = 0; rep < 2; rep++) {    // the first is a warm-up (clocks up)
        if (rep == 1)
            CHECK(hipEventRecord(gE0));
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, gRec, iters);
    }
    CHECK(hipEventRecord(gE1));
    CHECK(hipEventSynchronize(gE1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, gE0, gE1));
    std::vector<unsigned long long> h(blocks * 4 * 3);
    CHECK(hipMemcpy(h.data(), gRec, h.size() * 8, hipMemcpyDeviceToHost));
    struct Simd { unsigned long long lo = ~0ull, hi = 0, busy = 0; int waves = 0; };
    std::map<unsigned long long, Simd> simds;
    for (int w = 0; w < blocks * 4; w++) {
        Simd& s = simds[h[3 * w + 2]];
        s.lo = std::min(s.lo, h[3 * w]);
        s.hi = std::max(s.hi, h[3 * w + 1]);
        s.busy += h[3 * w + 1] - h[3 * w];
        s.waves++;
    }
    // SIMDs that hosted exactly the intended number of waves: ticks per instruction = span / instructions issued,
    // overlap = how much of that span all of its waves were resident together
    const double instrPerWave = (double)iters * 32;
    double rate = 0, overlap = 0;
    int n = 0;
    for (auto& kv : simds) {
        const Simd& s = kv.second;
        if (s.waves != wavesPerSimd)
            continue;
        const double span = (double)(s.hi - s.lo);
        rate += span / (instrPerWave * s.waves);
        overlap += (double)s.busy / (span * s.waves);
        n++;
    }
    if (getenv("VALU_RATE_DUMP")) { // residency of the waves of the first few SIMDs
        int shown = 0;
        for (auto& kv : simds) {
            if (shown++ >= 3)
                break;
            printf("   simd %llx:", kv.first);
            for (int w = 0; w < blocks * 4; w++)
                if (h[3 * w + 2] == kv.first)
                    printf(" [%.2f..%.2f ms]", (h[3 * w] - kv.second.lo) / 2.4e6, (h[3 * w + 1] - kv.second.lo) / 2.4e6);
            printf("\n");
        }
    }
    const double wall = ms * 1e-3 * 2.4e9 / (instrPerWave * wavesPerSimd);
    printf("%-10s %-32s %dw: %5.2f ticks/instr (%4d SIMDs with %d waves, overlap %.2f)   wall %6.2f ms = %5.2f cyc/instr at 2.4 GHz\n",
           k.id, k.label, wavesPerSimd, n ? rate / n : 0.0, n, wavesPerSimd, n ? overlap / n : 0.0, ms, wall);
    return 0;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    CHECK(hipMalloc(&gRec, 4096 * 4 * 24));
    CHECK(hipEventCreate(&gE0));
    CHECK(hipEventCreate(&gE1));
    const int it = 60000; // x 32 instructions per wave at 4 waves per SIMD
    for (const Kind& k : kinds) {
        if (argc > 1 && !strstr(argv[1], k.id))
            continue;
        const bool slow = strstr("rcp rsq sqrt_ sin_ exp_ log_ readlane writelane rcpiflag dsread cnd", k.id) != nullptr;
        for (int w : {4}) {
            const auto t0 = std::chrono::steady_clock::now();
            if (run(k, w, (slow ? it / 4 : it) * 4 / w))
                return 1;
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s > 1.0)
                printf("   (host: that run took %.1f s)\n", s);
        }
    }
    return 0;
}
