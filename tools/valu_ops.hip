// Per-opcode issue cost of the VALU on gfx950 (round 5): the step kernels are bound by instruction issue under the board's power cap,
// and ~40 % of their instructions are not FMAs.  Each kernel below issues ONE opcode (inline asm, so that the compiler neither fuses
// nor packs it) in 8 independent chains per wave, 4 waves per SIMD, and reports ns per wave-instruction per SIMD.  What costs more
// than an fma is worth replacing; what costs the same is not.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_ops.hip -o tools/valu_ops && tools/valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(name, ASM)                                                                                        \
    __global__ __launch_bounds__(64) void name(float* out, int iters, float a, float b) {                        \
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;  \
        float va = a + x0 * 1e-9f, vb = b + x0 * 1e-9f;                                                          \
        for (int it = 0; it < iters; it++) {                                                                     \
            _Pragma("unroll") for (int r = 0; r < 8; r++) { ASM }                                                \
        }                                                                                                        \
        out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                              \
    }
#define A1(op, i) asm volatile(op " %0, %0, %1" : "+v"(x##i) : "v"(va));
#define A2(op, i) asm volatile(op " %0, %0, %1, %2" : "+v"(x##i) : "v"(va), "v"(vb));
#define ALL(M, op) M(op, 0) M(op, 1) M(op, 2) M(op, 3) M(op, 4) M(op, 5) M(op, 6) M(op, 7)

KERNEL(k_fma, ALL(A2, "v_fma_f32"))
KERNEL(k_mul, ALL(A1, "v_mul_f32"))
KERNEL(k_add, ALL(A1, "v_add_f32"))
KERNEL(k_max, ALL(A1, "v_max_f32"))
KERNEL(k_med3, ALL(A2, "v_med3_f32"))
KERNEL(k_max3, ALL(A2, "v_max3_f32"))
KERNEL(k_xor, ALL(A1, "v_xor_b32"))
KERNEL(k_pkfma, ALL(A2, "v_fma_f32") )   /* placeholder slot: replaced below by the explicit packed kernel */
#define ACND(op, i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##i) : "v"(va) : );
KERNEL(k_cnd, asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(va), "v"(vb) : "vcc"); ALL(ACND, ""))
#define ACMP(op, i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x##i), "v"(va) : "vcc");
KERNEL(k_cmp, ALL(ACMP, ""))
#define ACMPS(op, i) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" :: "v"(x##i), "v"(va) : "s20", "s21");
KERNEL(k_cmps, ALL(ACMPS, ""))
#define ACC(op, i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x##i) : "v"(va), "v"(vb) : "vcc");
KERNEL(k_cmpcnd, ALL(ACC, ""))
#define ALIT(op, i) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x##i) : "v"(vb));
KERNEL(k_fmamk, ALL(ALIT, ""))
#define ASG(op, i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##i) : "s"(a), "v"(vb));
KERNEL(k_fma_sgpr, ALL(ASG, ""))
#define ARCP(op, i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x##i));
KERNEL(k_rcp, ALL(ARCP, ""))
#define ACVT(op, i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x##i));
KERNEL(k_cvt, ALL(ACVT, ""))
#define AMOV(op, i) asm volatile("v_mov_b32 %0, %1" : "+v"(x##i) : "v"(va));
KERNEL(k_mov, ALL(AMOV, ""))
#define AABS(op, i) asm volatile("v_fma_f32 %0, |%0|, %1, -%2" : "+v"(x##i) : "v"(va), "v"(vb));
KERNEL(k_fma_mod, ALL(AABS, ""))
#define AF64(op, i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(da), "v"(db));

__global__ __launch_bounds__(64) void k_fma64(float* out, int iters, float a, float b) {
    double d0 = threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    double da = a + d0 * 1e-9, db = b + d0 * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) { AF64("", 0) AF64("", 1) AF64("", 2) AF64("", 3) AF64("", 4) AF64("", 5) AF64("", 6) AF64("", 7) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
__global__ __launch_bounds__(64) void k_pk(float* out, int iters, float a, float b) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {threadIdx.x * 1e-3f, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(va), "v"(vb));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(va), "v"(vb));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(va), "v"(vb));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(va), "v"(vb));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = p0.x + p1.y + p2.x + p3.y;
}
__global__ __launch_bounds__(64) void k_lds(float* out, int iters, float a, float b) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    __shared__ v4 tile[4 * 64];
    v4 v = {a, b, a, b};
    volatile v4* tp = tile;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            tp[(r & 3) * 64 + threadIdx.x] = v;
            const v4 t = tp[((r + 1) & 3) * 64 + threadIdx.x];
            v.x += t.y;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v.x;
}

typedef void (*kern_t)(float*, int, float, float);
static void run(const char* what, kern_t k, int instr_per_iter, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 4000;
    float* d;
    hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters / 4, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * iters * instr_per_iter;
    printf("%-44s %d waves/SIMD: %8.3f ms  %.3f ns per wave-instruction per SIMD  (%.3f instr/ns/SIMD)\n", what, waves_per_simd, ms,
           ms * 1e6 * 1024 / winst, winst / 1024 / (ms * 1e6));
    hipFree(d);
}

int main() {
    for (int pass = 0; pass < 2; pass++) {
        printf("--- pass %d\n", pass);
        run("v_fma_f32 (3 VGPR operands)", k_fma, 64, 4);
        run("v_mul_f32", k_mul, 64, 4);
        run("v_add_f32", k_add, 64, 4);
        run("v_max_f32", k_max, 64, 4);
        run("v_med3_f32", k_med3, 64, 4);
        run("v_max3_f32", k_max3, 64, 4);
        run("v_xor_b32", k_xor, 64, 4);
        run("v_mov_b32", k_mov, 64, 4);
        run("v_cndmask_b32 (vcc set once)", k_cnd, 64 + 8, 4);
        run("v_cmp_gt_f32 -> vcc", k_cmp, 64, 4);
        run("v_cmp_gt_f32 -> sgpr pair", k_cmps, 64, 4);
        run("v_cmp + v_cndmask pairs", k_cmpcnd, 128, 4);
        run("v_fmamk_f32 (32-bit literal)", k_fmamk, 64, 4);
        run("v_fma_f32 with an SGPR operand", k_fma_sgpr, 64, 4);
        run("v_fma_f32 with |x| and -x modifiers", k_fma_mod, 64, 4);
        run("v_rcp_f32", k_rcp, 64, 4);
        run("v_cvt_i32_f32", k_cvt, 64, 4);
        run("v_pk_fma_f32 (2 fma per lane)", k_pk, 64, 4);
        run("v_fma_f64", k_fma64, 64, 4);
        run("v_fma_f64", k_fma64, 64, 2);
        run("v_fma_f32 (3 VGPR operands)", k_fma, 64, 2);
        run("v_fma_f32 (3 VGPR operands)", k_fma, 64, 1);
        run("ds_write_b128 + ds_read_b128 pairs", k_lds, 32, 4);
    }
    return 0;
}
