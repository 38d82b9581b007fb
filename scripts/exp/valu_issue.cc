// Experiment: how many cycles does ONE wave64 vector instruction occupy a gfx950 SIMD's issue port?
// The guide says 2 (SIMD-32, fp32); round 3's PMC reading (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 quad-cycle) and the
// one v_alignbit chain of overlap.cc said 4.  This program measures a table: instruction class x wavefronts per SIMD, every
// wavefront running 16 INDEPENDENT chains (register i depends on register i only), 64 instructions per loop iteration,
// cycles from s_memtime (shader clock) inside the wavefront AND from HIP events (wall), clock from s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 valu_issue.cc -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 16 chains; OP(i) is one asm statement acting on chain i
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define REP64(OP) REP16(OP) REP16(OP) REP16(OP) REP16(OP)

struct Out { unsigned long long cyc, rt; unsigned chk; unsigned pad; };

#define KERNEL_HEAD(NAME)                                                                                              \
    __global__ __launch_bounds__(1024) void NAME(int iters, Out *out, const float *seed) {                             \
        const int lane = threadIdx.x & 63;                                                                             \
        float s0 = seed[lane], s1 = seed[64 + lane];
#define TIMER_BEGIN                                                                                                    \
    __syncthreads();                                                                                                   \
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#define TIMER_END(CHKV)                                                                                                 \
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();                                   \
    if (lane == 0) {                                                                                                   \
        Out o; o.cyc = t1 - t0; o.rt = w1 - w0; o.chk = (CHKV); o.pad = 0;                                              \
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = o;                                                  \
    }                                                                                                                  \
    }

// ---- 32-bit classes: r[i] = op(r[i], a, b) -------------------------------------------------------------------------
#define DEF32(NAME, ASM)                                                                                               \
    KERNEL_HEAD(NAME)                                                                                                  \
    unsigned r[16];                                                                                                    \
    for (int i = 0; i < 16; ++i) r[i] = __float_as_uint(s0 * (float)(i + 1));                                          \
    unsigned a = __float_as_uint(s1), b = __float_as_uint(s0);                                                         \
    TIMER_BEGIN                                                                                                        \
    for (int it = 0; it < iters; ++it) {                                                                               \
        REP64(ASM)                                                                                                     \
    }                                                                                                                  \
    unsigned acc = 0;                                                                                                  \
    for (int i = 0; i < 16; ++i) acc ^= r[i];                                                                          \
    TIMER_END(acc)

#define OP_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_MUL32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[i]) : "v"(a));
#define OP_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define OP_CVT16(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));
#define OP_CMP(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define OP_CMPS(i) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m[i & 7]) : "v"(r[i]), "v"(a));
#define OP_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define OP_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(r[i]));

DEF32(k_fma_f32, OP_FMA32)
DEF32(k_mul_f32, OP_MUL32)
DEF32(k_add_u32, OP_ADDU)
DEF32(k_or3_b32, OP_OR3)
DEF32(k_alignbit, OP_ALIGN)
DEF32(k_med3_f32, OP_MED3)
DEF32(k_mov_b32, OP_MOV)
DEF32(k_cndmask, OP_CNDMASK)
DEF32(k_mul_lo_u32, OP_MULLO)
DEF32(k_rcp_f32, OP_RCP)
DEF32(k_cvt_f16_f32, OP_CVT16)
DEF32(k_lshl_add, OP_LSHLADD)
DEF32(k_cmp_vcc, OP_CMP)
DEF32(k_mov_dpp, OP_DPP)
DEF32(k_bfe_u32, OP_BFE)


#define OP_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));
#define OP_ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_ADD32E64(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_ADD32LIT(i) asm volatile("v_add_f32 %0, 0x3f800123, %0" : "+v"(r[i]));
#define OP_MAX32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_MIN32(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_SUBU(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_FMAC32(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_FMA32S(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(r[i]) : "s"(sa));
#define OP_FMA32K(i) asm volatile("v_fma_f32 %0, %0, 2.0, 1.0" : "+v"(r[i]));
#define OP_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(r[i]));
#define OP_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_CVT32_16(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[i]));
#define OP_ADD32DPP(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
#define OP_CNDE64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(mk));
#define OP_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc");
#define OP_PKADD16(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_FMA16(i) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
DEF32(k_or_b32, OP_OR)
DEF32(k_and_b32, OP_AND)
DEF32(k_xor_b32, OP_XOR)
DEF32(k_lshlrev_b32, OP_LSHL)
DEF32(k_add_f32, OP_ADD32)
DEF32(k_add_f32_e64, OP_ADD32E64)
DEF32(k_add_f32_lit, OP_ADD32LIT)
DEF32(k_max_f32, OP_MAX32)
DEF32(k_min_f32, OP_MIN32)
DEF32(k_sub_u32, OP_SUBU)
DEF32(k_fmac_f32, OP_FMAC32)
DEF32(k_fma_f32_const, OP_FMA32K)
DEF32(k_min3_f32, OP_MIN3)
DEF32(k_and_or_b32, OP_ANDOR)
DEF32(k_lshl_or_b32, OP_LSHLOR)
DEF32(k_add3_u32, OP_ADD3)
DEF32(k_perm_b32, OP_PERM)
DEF32(k_ffbh_u32, OP_FFBH)
DEF32(k_bcnt, OP_BCNT)
DEF32(k_mbcnt, OP_MBCNT)
DEF32(k_cvt_f32_f16, OP_CVT32_16)
DEF32(k_add_f32_dpp, OP_ADD32DPP)
DEF32(k_add_co_u32, OP_ADDCO)
DEF32(k_pk_add_f16, OP_PKADD16)
DEF32(k_fma_f16, OP_FMA16)
KERNEL_HEAD(k_fma_f32_sgpr)
    unsigned r[16];
    for (int i = 0; i < 16; ++i) r[i] = __float_as_uint(s0 * (float)(i + 1));
    float sa = __builtin_amdgcn_readfirstlane(__float_as_uint(s1)) * 1.0f;
    TIMER_BEGIN
    for (int it = 0; it < iters; ++it) {
        REP64(OP_FMA32S)
    }
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    TIMER_END(acc)
KERNEL_HEAD(k_cndmask_e64)
    unsigned r[16];
    for (int i = 0; i < 16; ++i) r[i] = __float_as_uint(s0 * (float)(i + 1));
    unsigned a = __float_as_uint(s1);
    unsigned long long mk = __builtin_amdgcn_ballot_w64(s0 > 1.03f);
    TIMER_BEGIN
    for (int it = 0; it < iters; ++it) {
        REP64(OP_CNDE64)
    }
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    TIMER_END(acc)

// v_cmp into SGPR pairs (the scorer's form: comparison result = wave mask in SGPRs)
KERNEL_HEAD(k_cmp_sgpr)
    unsigned r[16];
    for (int i = 0; i < 16; ++i) r[i] = __float_as_uint(s0 * (float)(i + 1));
    unsigned a = __float_as_uint(s1);
    unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    TIMER_BEGIN
    for (int it = 0; it < iters; ++it) {
        REP64(OP_CMPS)
    }
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= (unsigned)m[i];
    TIMER_END(acc)

// ---- 64-bit classes ---------------------------------------------------------------------------------------------------
#define DEF64(NAME, ASM)                                                                                               \
    KERNEL_HEAD(NAME)                                                                                                  \
    double r[16];                                                                                                      \
    for (int i = 0; i < 16; ++i) r[i] = (double)s0 * (i + 1);                                                          \
    double a = (double)s1, b = (double)s0;                                                                             \
    TIMER_BEGIN                                                                                                        \
    for (int it = 0; it < iters; ++it) {                                                                               \
        REP64(ASM)                                                                                                     \
    }                                                                                                                  \
    double acc = 0;                                                                                                    \
    for (int i = 0; i < 16; ++i) acc += r[i];                                                                          \
    TIMER_END((unsigned)__double2loint(acc))

#define OP_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[i]));
#define OP_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(r[i]));
#define OP_CMP64(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define OP_DIVFIX(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));

DEF64(k_fma_f64, OP_FMA64)
DEF64(k_mul_f64, OP_MUL64)
DEF64(k_add_f64, OP_ADD64)
DEF64(k_pk_fma_f32, OP_PKFMA)
DEF64(k_pk_mul_f32, OP_PKMUL)
DEF64(k_pk_add_f32, OP_PKADD)
DEF64(k_rcp_f64, OP_RCP64)
DEF64(k_lshl_b64, OP_LSHL64)
DEF64(k_cmp_f64, OP_CMP64)
DEF64(k_div_fixup_f64, OP_DIVFIX)

// the scorer's tile-loop mix: per pair v_or3 / v_or / v_alignbit (int32), independent chains
#define OP_MIX(i)                                                                                                      \
    asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));                                          \
    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[(i + 8) & 15]) : "v"(a));
KERNEL_HEAD(k_mix_or3_alignbit)
    unsigned r[16];
    for (int i = 0; i < 16; ++i) r[i] = __float_as_uint(s0 * (float)(i + 1));
    unsigned a = __float_as_uint(s1), b = __float_as_uint(s0);
    TIMER_BEGIN
    for (int it = 0; it < iters; ++it) {
        REP16(OP_MIX) REP16(OP_MIX)
    }
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    TIMER_END(acc)

typedef void (*kern_t)(int, Out *, const float *);
struct Entry { const char *name; kern_t k; int instr_per_iter; int iters; };

int main(int argc, char **argv) {
    Out *out; CK(hipMalloc(&out, sizeof(Out) * 16 * 1024)); float *seed; CK(hipMalloc(&seed, 4 * 128));
    std::vector<float> hs(128); for (int i = 0; i < 128; ++i) hs[i] = 1.0f + 1e-3f * i;
    CK(hipMemcpy(seed, hs.data(), 4 * 128, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n\n", prop.name, cus, prop.clockRate);
    const Entry tab[] = {
        {"v_fma_f32", k_fma_f32, 64, 4000},       {"v_mul_f32", k_mul_f32, 64, 4000},
        {"v_add_u32", k_add_u32, 64, 4000},       {"v_or3_b32", k_or3_b32, 64, 4000},
        {"v_alignbit_b32", k_alignbit, 64, 4000}, {"v_lshl_add_u32", k_lshl_add, 64, 4000},
        {"v_bfe_u32", k_bfe_u32, 64, 4000},       {"v_med3_f32", k_med3_f32, 64, 4000},
        {"v_mov_b32", k_mov_b32, 64, 4000},       {"v_mov_b32 dpp row_shr", k_mov_dpp, 64, 4000},
        {"v_cndmask_b32", k_cndmask, 64, 4000},   {"v_cmp_gt_f32 vcc", k_cmp_vcc, 64, 4000},
        {"v_cmp_gt_f32 sgpr", k_cmp_sgpr, 64, 4000}, {"v_cvt_f16_f32", k_cvt_f16_f32, 64, 4000},
        {"v_mul_lo_u32", k_mul_lo_u32, 64, 2000}, {"v_rcp_f32", k_rcp_f32, 64, 2000},
        {"v_pk_fma_f32", k_pk_fma_f32, 64, 4000}, {"v_pk_mul_f32", k_pk_mul_f32, 64, 4000},
        {"v_pk_add_f32", k_pk_add_f32, 64, 4000}, {"v_fma_f64", k_fma_f64, 64, 2000},
        {"v_mul_f64", k_mul_f64, 64, 2000},       {"v_add_f64", k_add_f64, 64, 2000},
        {"v_cmp_gt_f64 vcc", k_cmp_f64, 64, 2000}, {"v_div_fixup_f64", k_div_fixup_f64, 64, 2000},
        {"v_lshlrev_b64", k_lshl_b64, 64, 2000},  {"v_rcp_f64", k_rcp_f64, 64, 1000},
        {"v_or_b32 (VOP2)", k_or_b32, 64, 4000}, {"v_and_b32 (VOP2)", k_and_b32, 64, 4000}, {"v_xor_b32 (VOP2)", k_xor_b32, 64, 4000},
        {"v_lshlrev_b32 (VOP2)", k_lshlrev_b32, 64, 4000}, {"v_sub_u32 (VOP2)", k_sub_u32, 64, 4000},
        {"v_add_f32 (VOP2)", k_add_f32, 64, 4000}, {"v_add_f32_e64 (VOP3 encoding)", k_add_f32_e64, 64, 4000},
        {"v_add_f32 + 32-bit literal", k_add_f32_lit, 64, 4000}, {"v_add_f32 dpp", k_add_f32_dpp, 64, 4000},
        {"v_max_f32 (VOP2)", k_max_f32, 64, 4000}, {"v_min_f32 (VOP2)", k_min_f32, 64, 4000},
        {"v_fmac_f32 (VOP2)", k_fmac_f32, 64, 4000}, {"v_fma_f32 v,s,const", k_fma_f32_sgpr, 64, 4000},
        {"v_fma_f32 v,const,const", k_fma_f32_const, 64, 4000}, {"v_min3_f32", k_min3_f32, 64, 4000},
        {"v_and_or_b32", k_and_or_b32, 64, 4000}, {"v_lshl_or_b32", k_lshl_or_b32, 64, 4000}, {"v_add3_u32", k_add3_u32, 64, 4000},
        {"v_perm_b32", k_perm_b32, 64, 4000}, {"v_ffbh_u32 (VOP1)", k_ffbh_u32, 64, 4000}, {"v_bcnt_u32_b32", k_bcnt, 64, 4000},
        {"v_mbcnt_lo_u32_b32", k_mbcnt, 64, 4000}, {"v_cvt_f32_f16 (VOP1)", k_cvt_f32_f16, 64, 4000},
        {"v_cndmask_b32_e64 (sgpr mask)", k_cndmask_e64, 64, 4000}, {"v_add_co_u32 vcc", k_add_co_u32, 64, 4000},
        {"v_pk_add_f16", k_pk_add_f16, 64, 4000}, {"v_fma_f16", k_fma_f16, 64, 4000},
        {"mix v_or3 + v_alignbit", k_mix_or3_alignbit, 64, 4000},
    };
    const char *filter = argc > 1 ? argv[1] : nullptr;  // substring of the instruction name
    std::vector<int> waves_per_simd = {1, 2, 4, 6, 8};
    if (argc > 2) waves_per_simd = {atoi(argv[2])};  // PMC runs: one kernel, one occupancy
    printf("cycles per wave64 instruction per SIMD: A = from s_memtime inside the wavefronts (shader clock; mean over wavefronts of\n"
           "(cycles of the loop / instructions) / wavefronts on the SIMD), B = from the HIP-event wall time at the measured clock\n\n");
    printf("| instruction | 1 wave/SIMD A / B | 2 A / B | 4 A / B | 6 A / B | 8 A / B | MHz |\n|---|---|---|---|---|---|---|\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Entry &e : tab) {
        if (filter && !strstr(e.name, filter)) continue;
        printf("| %s |", e.name);
        double mhz_last = 0;
        for (int w : waves_per_simd) {
            // w waves per SIMD = 4w waves per CU: one block of 256 w threads per CU for w <= 4, two blocks of 128 w for 6 / 8
            const int per_cu = w <= 4 ? 1 : 2, threads = 256 * w / per_cu, blocks = cus * per_cu, waves = blocks * threads / 64;
            e.k<<<blocks, threads>>>(e.iters / 8, out, seed); // warm-up
            CK(hipEventRecord(e0, 0));
            e.k<<<blocks, threads>>>(e.iters, out, seed);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Out> h(waves); CK(hipMemcpy(h.data(), out, sizeof(Out) * waves, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; for (const Out &o : h) { cyc += (double)o.cyc; rt += (double)o.rt; }
            cyc /= waves; rt /= waves;
            const double mhz = cyc / (rt / 100.0);  // s_memrealtime ticks at 100 MHz
            const double n = (double)e.iters * e.instr_per_iter;
            const double A = cyc / n / w, B = ms * 1e-3 * mhz * 1e6 / n / w;
            printf(" %.2f / %.2f |", A, B);
            mhz_last = mhz;
        }
        printf(" %.0f |\n", mhz_last);
        fflush(stdout);
    }
    return 0;
}
