// Per-opcode issue cost of the fp64-class VALU instructions on gfx950 (round 5, second session): companion of valu_ops.hip for the
// instruction mix of the precision = f64 step kernels (2 waves per SIMD).  One opcode per kernel from inline asm, 8 independent
// chains per wave; ns per wave-instruction per SIMD.  The question it answers: do v_mul_f64 / v_add_f64 / v_max_f64 / compares /
// SGPR-pair operands cost the same issue time as v_fma_f64 with VGPR operands (2.1-2.3 ns), or more?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_ops64.hip -o tools/valu_ops64 && tools/valu_ops64
#include <hip/hip_runtime.h>
#include <stdio.h>

#define KERNEL64(name, BODY)                                                                                          \
    __global__ __launch_bounds__(64) void name(float* out, int iters, double a, double b) {                           \
        double d0 = threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7; \
        double da = a + d0 * 1e-9, db = b + d0 * 1e-9;                                                                \
        for (int it = 0; it < iters; it++) {                                                                          \
            _Pragma("unroll") for (int r = 0; r < 8; r++) { BODY(0) BODY(1) BODY(2) BODY(3) BODY(4) BODY(5) BODY(6) BODY(7) } \
        }                                                                                                             \
        out[blockIdx.x * 64 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + da + db);               \
    }

#define B_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(da), "v"(db));
#define B_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(da));
#define B_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(db));
#define B_MAX(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d##i) : "v"(da));
#define B_FMAS(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "s"(a), "v"(db));
#define B_MULS(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "s"(a));
#define B_FMAI(i) asm volatile("v_fma_f64 %0, %0, 0.5, %1" : "+v"(d##i) : "v"(db));
#define B_CMP(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(d##i), "v"(da) : "vcc");
#define B_CMPCND(i)                                                                                                     \
    asm volatile("v_cmp_gt_f64 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc"        \
                 : "+v"(lo##i), "+v"(hi##i) : "v"(d##i), "v"(da), "v"(fa) : "vcc");
#define B_FMAC(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d##i) : "v"(da), "v"(db));
#define B_RDL(i) asm volatile("v_readlane_b32 s20, %0, 3" ::"v"((float)i + fa_rdl) : "s20");
#define B_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(d##i) : "v"(da));

KERNEL64(k_fma, B_FMA)
KERNEL64(k_mul, B_MUL)
KERNEL64(k_add, B_ADD)
KERNEL64(k_max, B_MAX)
KERNEL64(k_fma_s, B_FMAS)
KERNEL64(k_mul_s, B_MULS)
KERNEL64(k_fma_inl, B_FMAI)
KERNEL64(k_cmp, B_CMP)
KERNEL64(k_fmac, B_FMAC)
KERNEL64(k_mov64, B_MOV64)

// what the fp64 kernel's constant multiplies look like: two s_mov_b32 materialising a 64-bit constant in front of every use
#define B_SMOV_FMA(i) asm volatile("s_mov_b32 s20, 0x9999999a\n\ts_mov_b32 s21, 0x3fb99999\n\tv_fma_f64 %0, %0, s[20:21], %1" : "+v"(d##i) : "v"(db) : "s20", "s21");
KERNEL64(k_smov_fma, B_SMOV_FMA)

// a 64-bit select as the compiler emits it: one compare, two 32-bit selects (here on separate 32-bit chains)
__global__ __launch_bounds__(64) void k_cmpcnd(float* out, int iters, double a, double b) {
    double d0 = threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    double da = a + d0 * 1e-9;
    float fa = (float)b + threadIdx.x;
    float lo0 = 0, lo1 = 1, lo2 = 2, lo3 = 3, lo4 = 4, lo5 = 5, lo6 = 6, lo7 = 7, hi0 = 0, hi1 = 1, hi2 = 2, hi3 = 3, hi4 = 4, hi5 = 5, hi6 = 6, hi7 = 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) { B_CMPCND(0) B_CMPCND(1) B_CMPCND(2) B_CMPCND(3) B_CMPCND(4) B_CMPCND(5) B_CMPCND(6) B_CMPCND(7) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + lo0 + lo1 + lo2 + lo3 + lo4 + lo5 + lo6 + lo7 + hi0 + hi1 + hi2 + hi3 + hi4 + hi5 + hi6 + hi7;
}

__global__ __launch_bounds__(64) void k_rdl(float* out, int iters, double a, double b) {
    const float fa_rdl = (float)a + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) { B_RDL(0) B_RDL(1) B_RDL(2) B_RDL(3) B_RDL(4) B_RDL(5) B_RDL(6) B_RDL(7) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = fa_rdl;
}

typedef void (*kern_t)(float*, int, double, double);
static void run(const char* what, kern_t k, int valu_per_iter, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 2000;
    float* d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters / 4, 0.999, 0.001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999, 0.001);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * iters * valu_per_iter;
    printf("%-58s %d waves/SIMD: %8.3f ms  %.3f ns per VALU wave-instruction per SIMD\n", what, waves_per_simd, ms, ms * 1e6 * 1024 / winst);
    (void)hipFree(d);
}

int main() {
    for (int pass = 0; pass < 2; pass++) {
        printf("--- pass %d\n", pass);
        for (int w = 2; w <= 4; w += 2) {
            run("v_fma_f64 (3 VGPR operands)", k_fma, 64, w);
            run("v_fmac_f64 (VOP2 form)", k_fmac, 64, w);
            run("v_mul_f64", k_mul, 64, w);
            run("v_add_f64", k_add, 64, w);
            run("v_max_f64", k_max, 64, w);
            run("v_fma_f64 with an SGPR-pair operand", k_fma_s, 64, w);
            run("v_mul_f64 with an SGPR-pair operand", k_mul_s, 64, w);
            run("v_fma_f64 with an inline constant (0.5)", k_fma_inl, 64, w);
            run("2 x s_mov_b32 + v_fma_f64 s[pair] (per VALU instr)", k_smov_fma, 64, w);
            run("v_cmp_gt_f64 -> vcc", k_cmp, 64, w);
            run("v_cmp_f64 + 2 v_cndmask_b32 (per VALU instr, 3 each)", k_cmpcnd, 192, w);
            run("v_readlane_b32", k_rdl, 64, w);
            run("v_mov_b64", k_mov64, 64, w);
        }
    }
    return 0;
}
