// Developer tool: issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the
// sweep kernel is made of, measured on the whole chip at 4 waves/SIMD with 8 independent chains
// per wave.  hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL_F64_3(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(double* out, double a, double b)             \
    {                                                                                        \
        double r[8];                                                                         \
        for (int i = 0; i < 8; ++i) r[i] = a + i + threadIdx.x;                              \
        for (int it = 0; it < ITER; ++it) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                    \
                asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b));                             \
        }                                                                                    \
        double s = 0;                                                                        \
        for (int i = 0; i < 8; ++i) s += r[i];                                               \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                             \
    }

KERNEL_F64_3(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_F64_3(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_F64_3(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_F64_3(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_F64_3(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_F64_3(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_F64_3(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL_F64_3(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
KERNEL_F64_3(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
KERNEL_F64_3(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
KERNEL_F64_3(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL_F64_3(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_F64_3(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")
KERNEL_F64_3(k_cmp_f64, "v_cmp_gt_f64 vcc, %0, %1")
KERNEL_F64_3(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_F64_3(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_F64_3(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
__global__ __launch_bounds__(256) void k_mad_u64_u32(double* out, double a, double b)
{
    unsigned long long r[8];
    unsigned ua = (unsigned)a + threadIdx.x, ub = 0xD2511F53u;
    for (int i = 0; i < 8; ++i) r[i] = i + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(ua), "v"(ub) : "vcc");
    }
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = (double)s;
}

#define KERNEL_U32_3(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(double* out, double ad, double bd)           \
    {                                                                                        \
        unsigned a = (unsigned)ad, b = (unsigned)bd;                                         \
        unsigned r[8];                                                                       \
        for (int i = 0; i < 8; ++i) r[i] = a + i + threadIdx.x;                              \
        for (int it = 0; it < ITER; ++it) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                    \
                asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b));                             \
        }                                                                                    \
        unsigned s = 0;                                                                      \
        for (int i = 0; i < 8; ++i) s += r[i];                                               \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                             \
    }

KERNEL_U32_3(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL_U32_3(k_bitop3_b32, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL_U32_3(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_U32_3(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_U32_3(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL_U32_3(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL_U32_3(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_U32_3(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_U32_3(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_U32_3(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL_U32_3(k_log_f32, "v_log_f32 %0, %0")
KERNEL_U32_3(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_U32_3(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_U32_3(k_sin_f32, "v_sin_f32 %0, %0")
KERNEL_U32_3(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL_U32_3(k_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL_U32_3(k_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 11")
KERNEL_U32_3(k_mov_b32, "v_mov_b32 %0, %1")

__global__ __launch_bounds__(256) void k_cvt_f64_u32(double* out, double a, double b)
{
    double r[8];
    unsigned u = (unsigned)a + threadIdx.x;
    for (int i = 0; i < 8; ++i) r[i] = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r[i]) : "v"(u));
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// round 3: the select and compare forms the kernels really use, and the 32-bit / conversion opcodes of their loops that
// the first table priced by class
KERNEL_U32_3(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL_U32_3(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KERNEL_U32_3(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
KERNEL_U32_3(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL_U32_3(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 12")
KERNEL_U32_3(k_bfi_b32, "v_bfi_b32 %0, %1, %2, %0")
KERNEL_U32_3(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL_U32_3(k_or3_b32, "v_or3_b32 %0, %0, %1, %2")
KERNEL_U32_3(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_U32_3(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL_F64_3(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
KERNEL_F64_3(k_min_f64, "v_min_f64 %0, %0, %1")

// v_cndmask_b32 with the lane mask in an SGPR pair (the VOP3 form the compiler emits for selects on a ballot) ...
__global__ __launch_bounds__(256) void k_cndmask_sgpr(double* out, double ad, double bd)
{
    unsigned a = (unsigned)ad, b = (unsigned)bd;
    unsigned r[8];
    unsigned long long m = __builtin_amdgcn_ballot_w64((threadIdx.x & 3) != 0);
    for (int i = 0; i < 8; ++i) r[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "s"(m));
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// ... with vcc written once before the loop by a VALU compare (the VOP2 form) ...
__global__ __launch_bounds__(256) void k_cndmask_vcc_set(double* out, double ad, double bd)
{
    unsigned a = (unsigned)ad, b = (unsigned)bd;
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + i + threadIdx.x;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\ts_nop 4" ::"v"(r[1]), "v"(r[2]) : "vcc");
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b));
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// ... and the pair the kernels' selects consist of: a compare into an SGPR pair followed by the select on it
__global__ __launch_bounds__(256) void k_cmp_cndmask_pair(double* out, double ad, double bd)
{
    unsigned a = (unsigned)ad, b = (unsigned)bd;
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITER / 2; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned long long m;
            asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(b));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(m));
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// compares into SGPR pairs
__global__ __launch_bounds__(256) void k_cmp_f32_sgpr(double* out, double ad, double bd)
{
    float a = (float)ad + threadIdx.x, b = (float)bd;
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned long long m;
            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
            acc ^= m;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (double)acc;
}
__global__ __launch_bounds__(256) void k_cmp_f64_sgpr(double* out, double a, double b)
{
    a += threadIdx.x;
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned long long m;
            asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
            acc ^= m;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (double)acc;
}
__global__ __launch_bounds__(256) void k_cvt_f32_f64(double* out, double a, double b)
{
    float r[8];
    a += threadIdx.x;
    for (int i = 0; i < 8; ++i) r[i] = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(a));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_readlane(double* out, double ad, double bd)
{
    unsigned a = (unsigned)ad + threadIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned sv;
            asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sv) : "v"(a));
            acc ^= sv;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kfn)(double*, double, double);

int main()
{
    int blocks_per_cu = 4;   // 4 waves per SIMD
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    int grid = cus * blocks_per_cu;
    double* out;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
    struct { const char* name; kfn fn; } tests[] = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_rsq_f64", k_rsq_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_div_fixup_f64", k_div_fixup_f64},
        {"v_div_fmas_f64", k_div_fmas_f64}, {"v_div_scale_f64", k_div_scale_f64}, {"v_ldexp_f64", k_ldexp_f64},
        {"v_max_f64", k_max_f64}, {"v_mov_b64", k_mov_b64}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"v_cmp_gt_f64", k_cmp_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
        {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24},
        {"v_xor_b32", k_xor_b32}, {"v_bitop3_b32", k_bitop3_b32}, {"v_add_u32", k_add_u32},
        {"v_cndmask_b32", k_cndmask_b32}, {"v_lshl_or_b32", k_lshl_or_b32}, {"v_alignbit_b32", k_alignbit_b32},
        {"v_mov_b32", k_mov_b32}, {"v_fma_f32", k_fma_f32}, {"v_exp_f32", k_exp_f32}, {"v_log_f32", k_log_f32},
        {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_sin_f32", k_sin_f32}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_cndmask_b32_sgpr", k_cndmask_sgpr}, {"v_cndmask_b32_vcc_set", k_cndmask_vcc_set}, {"v_cmp+v_cndmask_pair", k_cmp_cndmask_pair},
        {"v_cmp_gt_f32_sgpr", k_cmp_f32_sgpr}, {"v_cmp_lt_f64_sgpr", k_cmp_f64_sgpr}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_readlane_b32", k_readlane}, {"v_and_b32", k_and_b32}, {"v_lshrrev_b32", k_lshrrev_b32}, {"v_lshlrev_b32", k_lshlrev_b32},
        {"v_add3_u32", k_add3_u32}, {"v_bfe_u32", k_bfe_u32}, {"v_bfi_b32", k_bfi_b32}, {"v_perm_b32", k_perm_b32},
        {"v_or3_b32", k_or3_b32}, {"v_mul_f32", k_mul_f32}, {"v_max_f32", k_max_f32}, {"v_fmac_f64", k_fmac_f64},
        {"v_min_f64", k_min_f64},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%d CUs, grid %d blocks x 256 (= %d waves/SIMD), %d x 8 instr per wave\n", cus, grid, blocks_per_cu, ITER);
    for (auto& t : tests) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, out, 1.5, 0.75);
        hipEventRecord(e0);
        const int reps = 5;
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(t.fn, dim3(grid), dim3(256), 0, 0, out, 1.5, 0.75);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double sec = ms * 1e-3 / reps;
        double wave_instr_per_simd = (double)blocks_per_cu * ITER * 8;   // each SIMD hosts blocks_per_cu waves
        double ns_per = sec * 1e9 / wave_instr_per_simd;
        printf("%-18s %7.3f ns/wave-instr/SIMD  = %5.2f cycles @2.4GHz\n", t.name, ns_per, ns_per * 2.4);
    }
    return 0;
}
