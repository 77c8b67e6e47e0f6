// What a compare-and-select costs on gfx950, pattern by pattern (round 5, second session).  tools/valu_ops.hip found `v_cndmask_b32`
// at 6.3 ns when vcc was "set once" and at ~0.9 ns right behind its compare; the step kernels hold dozens of selects per RK stage
// (dead-band, yaw wrap, cos(theta) guard; a 64-bit select in the fp64 build is one compare and TWO v_cndmask_b32), so which pattern is
// the slow one matters.  Each kernel repeats one GROUP of instructions (8 independent chains per wave); reported: ns per group per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_select.hip -o tools/valu_select && tools/valu_select
#include <hip/hip_runtime.h>
#include <stdio.h>

#define DECL32                                                                                                                   \
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    float y0 = x0 * 2, y1 = x1 * 2, y2 = x2 * 2, y3 = x3 * 2, y4 = x4 * 2, y5 = x5 * 2, y6 = x6 * 2, y7 = x7 * 2;               \
    float va = (float)a + x0 * 1e-9f, vb = (float)b + x0 * 1e-9f;
#define DECL64                                                                                                                   \
    double d0 = threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;   \
    double da = a + d0 * 1e-9, db = b + d0 * 1e-9;
#define LOOP(G) for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int r = 0; r < 8; r++) { G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7) } }
#define OUT32 out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
#define K32(name, G) __global__ __launch_bounds__(64) void name(float* out, int iters, double a, double b) { DECL32 LOOP(G) OUT32 }
#define K64(name, G)                                                                                                             \
    __global__ __launch_bounds__(64) void name(float* out, int iters, double a, double b) {                                      \
        DECL32 DECL64 LOOP(G) OUT32 out[blockIdx.x * 64 + threadIdx.x] += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + da + db); \
    }

// ---- fp32 patterns
#define G_ADJ(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x##i) : "v"(va), "v"(vb) : "vcc");
#define G_NOP(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x##i) : "v"(va), "v"(vb) : "vcc");
#define G_GAP(i)                                                                                                                 \
    asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_cndmask_b32 %0, %0, %3, vcc" \
                 : "+v"(x##i), "+v"(y##i) : "v"(va), "v"(vb) : "vcc");
#define G_SG(i) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(x##i) : "v"(va), "v"(vb) : "s20", "s21");
#define G_SGGAP(i)                                                                                                               \
    asm volatile("v_cmp_gt_f32 s[20:21], %0, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_cndmask_b32 %0, %0, %3, s[20:21]" \
                 : "+v"(x##i), "+v"(y##i) : "v"(va), "v"(vb) : "s20", "s21");
#define G_TWO(i) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(x##i), "+v"(y##i) : "v"(va), "v"(vb) : "vcc");
#define G_3FMA(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3" : "+v"(x##i), "+v"(y##i) : "v"(va), "v"(vb));
// the select as arithmetic: 0/1 mask by compare + ONE select, then a multiply (dead-band: F = f * (|f| >= fd))
#define G_MASKMUL(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %2, 0, 1.0, vcc\n\tv_mul_f32 %0, %0, %2" : "+v"(x##i) : "v"(va), "v"(y##i) : "vcc");
// dead-band without a compare: med3 picks... (not equivalent - only as a cost yardstick for one slow-class op + one fast op)
#define G_MED3MUL(i) asm volatile("v_med3_f32 %1, %0, %2, %3\n\tv_mul_f32 %0, %0, %1" : "+v"(x##i), "+v"(y##i) : "v"(va), "v"(vb));
K32(k_adj, G_ADJ)
K32(k_nop, G_NOP)
K32(k_gap, G_GAP)
K32(k_sg, G_SG)
K32(k_sggap, G_SGGAP)
K32(k_two, G_TWO)
K32(k_3fma, G_3FMA)
K32(k_maskmul, G_MASKMUL)
K32(k_med3mul, G_MED3MUL)

// ---- fp64 patterns (a 64-bit select = one compare + two 32-bit selects; the 32-bit halves are separate chains here)
#define H_TWO(i) asm volatile("v_cmp_gt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc" : "+v"(x##i), "+v"(y##i) : "v"(d##i), "v"(da), "v"(va) : "vcc");
#define H_TWONOP(i) asm volatile("v_cmp_gt_f64 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc" : "+v"(x##i), "+v"(y##i) : "v"(d##i), "v"(da), "v"(va) : "vcc");
#define H_TWOSG(i) asm volatile("v_cmp_gt_f64 s[20:21], %2, %3\n\tv_cndmask_b32 %0, %0, %4, s[20:21]\n\tv_cndmask_b32 %1, %1, %4, s[20:21]" : "+v"(x##i), "+v"(y##i) : "v"(d##i), "v"(da), "v"(va) : "s20", "s21");
#define H_ONE(i) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x##i) : "v"(d##i), "v"(da), "v"(va) : "vcc");
#define H_TWOGAP(i)                                                                                                              \
    asm volatile("v_cmp_gt_f64 vcc, %2, %3\n\tv_fma_f64 %2, %2, %3, %5\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc" \
                 : "+v"(x##i), "+v"(y##i), "+v"(d##i) : "v"(da), "v"(va), "v"(db) : "vcc");
#define H_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(da), "v"(db));
// the 64-bit select as arithmetic: compare, ONE select building the high word of 1.0 / 0.0, one multiply
#define H_MASKMUL(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, 0, %3, vcc\n\tv_mul_f64 %0, %0, %2" : "+v"(d##i) : "v"(da), "v"(x##i), "v"(va) : "vcc");
// ... with v_cmp_class-free min/max only (saturation: max + min)
#define H_MAXMIN(i) asm volatile("v_max_f64 %0, %0, %1\n\tv_min_f64 %0, %0, %2" : "+v"(d##i) : "v"(da), "v"(db));
K64(k64_two, H_TWO)
K64(k64_twonop, H_TWONOP)
K64(k64_twosg, H_TWOSG)
K64(k64_one, H_ONE)
K64(k64_twogap, H_TWOGAP)
K64(k64_fma, H_FMA)
K64(k64_maxmin, H_MAXMIN)
__global__ __launch_bounds__(64) void k64_maskmul(float* out, int iters, double a, double b) {
    DECL32 DECL64
    // (the multiply reads a 64-bit register pair whose high word the select wrote: built as a double with a zero low word)
    double m0 = 1, m1 = 1, m2 = 1, m3 = 1, m4 = 1, m5 = 1, m6 = 1, m7 = 1;
#define H_MM(i) asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %H1, 0, %3, vcc\n\tv_mul_f64 %0, %0, %1" : "+v"(d##i), "+v"(m##i) : "v"(da), "v"(va) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d0), "+v"(x0) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d0) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d1), "+v"(x1) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d2), "+v"(x2) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d2) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d3), "+v"(x3) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d3) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d4), "+v"(x4) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d4) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d5), "+v"(x5) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d5) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d6), "+v"(x6) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d6) : "v"(db));
            asm volatile("v_cmp_gt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, 0, %3, vcc" : "+v"(d7), "+v"(x7) : "v"(da), "v"(va) : "vcc"); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d7) : "v"(db));
        }
    }
    OUT32 out[blockIdx.x * 64 + threadIdx.x] += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + da + db + m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7);
}

typedef void (*kern_t)(float*, int, double, double);
static void run(const char* what, kern_t k, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 1000;
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
    const double groups = (double)blocks * iters * 64;
    printf("%-78s %d waves/SIMD: %8.3f ms  %7.3f ns per group per SIMD\n", what, waves_per_simd, ms, ms * 1e6 * 1024 / groups);
    (void)hipFree(d);
}

int main() {
    for (int pass = 0; pass < 2; pass++) {
        printf("--- pass %d\n", pass);
        run("fp32: 3 x v_fma_f32 (yardstick)", k_3fma, 4);
        run("fp32: v_cmp -> vcc ; v_cndmask (adjacent)", k_adj, 4);
        run("fp32: v_cmp -> vcc ; s_nop 1 ; v_cndmask", k_nop, 4);
        run("fp32: v_cmp -> vcc ; 2 fma ; v_cndmask   (4 VALU)", k_gap, 4);
        run("fp32: v_cmp -> s[20:21] ; v_cndmask s[20:21]", k_sg, 4);
        run("fp32: v_cmp -> s[20:21] ; 2 fma ; v_cndmask s[20:21]   (4 VALU)", k_sggap, 4);
        run("fp32: v_cmp -> vcc ; v_cndmask ; v_cndmask", k_two, 4);
        run("fp32: v_cmp ; v_cndmask 0/1.0 ; v_mul (select as arithmetic)", k_maskmul, 4);
        run("fp32: v_med3 ; v_mul (yardstick: one slow + one fast op)", k_med3mul, 4);
        for (int w = 2; w <= 4; w += 2) {
            run("fp64: v_fma_f64 (yardstick, 1 VALU)", k64_fma, w);
            run("fp64: v_max_f64 ; v_min_f64 (saturation)", k64_maxmin, w);
            run("fp64: v_cmp_f64 -> vcc ; v_cndmask (one half)", k64_one, w);
            run("fp64: v_cmp_f64 -> vcc ; v_cndmask ; v_cndmask (64-bit select)", k64_two, w);
            run("fp64: v_cmp_f64 -> vcc ; s_nop 1 ; v_cndmask ; v_cndmask", k64_twonop, w);
            run("fp64: v_cmp_f64 -> vcc ; v_fma_f64 ; v_cndmask ; v_cndmask   (4 VALU)", k64_twogap, w);
            run("fp64: v_cmp_f64 -> s[20:21] ; v_cndmask s ; v_cndmask s", k64_twosg, w);
            run("fp64: v_cmp_f64 ; v_cndmask (high word of 0/1.0) ; v_mul_f64", k64_maskmul, w);
        }
    }
    return 0;
}
