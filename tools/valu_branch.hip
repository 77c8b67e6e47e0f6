// What a TAKEN scalar branch costs a SIMD on gfx950 (round 5, second session).  The step kernels guard their rare per-lane fall-backs
// (full sincos for a lane with a large angle increment) with `s_and_saveexec ; s_cbranch_execz skip` - taken in the common case.
// Kernels: 16 independent v_fma per group, then (a) nothing, (b) a never-taken branch, (c) an always-taken forward branch over 4
// instructions, (d) v_cmp + s_and_saveexec + s_cbranch_execz (taken) + s_or exec - the fall-back guard as compiled,
// (e) the same guard with the branch NOT taken in the common case (s_cbranch_execnz to an out-of-line block).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_branch.hip -o tools/valu_branch && tools/valu_branch
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DECL                                                                                                                     \
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    float va = a + x0 * 1e-9f, vb = b + x0 * 1e-9f;
#define F(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##i) : "v"(va), "v"(vb));
#define F16 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define KERN(name, TAIL)                                                                                \
    __global__ __launch_bounds__(64) void name(float* out, int iters, float a, float b) {               \
        DECL for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int r = 0; r < 8; r++) { F16 TAIL } } \
        out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                      \
    }
#define T_NONE
#define T_NOTTAKEN asm volatile("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\tv_fma_f32 %0, %0, %1, %2\n1:" : "+v"(x0) : "v"(va), "v"(vb) : "scc");
#define T_TAKEN asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n1:" : "+v"(x0) : "v"(va), "v"(vb) : "scc");
// x0 > 1e30 is never true: exec becomes 0 inside the guard and the branch is taken
#define T_GUARD                                                                                                                      \
    asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 1f\n\tv_fma_f32 %0, %0, %1, %2\n\t" \
                 "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n1:\n\ts_or_b64 exec, exec, s[20:21]" \
                 : "+v"(x0) : "v"(va), "v"(vb), "v"(big) : "vcc", "s20", "s21", "scc");   /* s_and_saveexec / s_or write SCC: undeclared, the loop branch the compiler keeps in SCC never falls through */
// the same decision by a vote: v_cmp + s_cbranch_vccnz to the (here empty) slow path; the common case falls through
#define T_VOTE asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_cbranch_vccnz 1f\n1:" : : "v"(x0), "v"(big) : "vcc", "scc");
KERN(k_none, T_NONE)
KERN(k_nottaken, T_NOTTAKEN)
KERN(k_taken, T_TAKEN)
__global__ __launch_bounds__(64) void k_guard(float* out, int iters, float a, float b) {
    DECL const float big = 1e30f + va;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) { F16 T_GUARD }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(64) void k_vote(float* out, int iters, float a, float b) {
    DECL const float big = 1e30f + va;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) { F16 T_VOTE }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
typedef void (*kern_t)(float*, int, float, float);
static double run(const char* what, kern_t k, int waves_per_simd, double base) {
    const int blocks = 1024 * waves_per_simd, iters = 20000;
    float* d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)blocks * iters * 8;
    const double ns = ms * 1e6 * 1024 / groups;
    printf("%-78s %d waves/SIMD: %8.3f ms  %7.3f ns per group (16 fma + tail) per SIMD  (tail: %+.2f ns)\n", what, waves_per_simd, ms, ns, ns - base);
    fflush(stdout);
    (void)hipFree(d);
    return ns;
}
int main() {
    for (int pass = 0; pass < 2; pass++) {
        printf("--- pass %d\n", pass);
        for (int w = 2; w <= 4; w += 2) {
            const double b0 = run("16 v_fma", k_none, w, 0);
            run("16 v_fma ; s_cmp ; s_cbranch (never taken) ; 1 v_fma", k_nottaken, w, b0);
            run("16 v_fma ; s_cmp ; s_cbranch (always taken, skips 4 v_fma)", k_taken, w, b0);
            run("16 v_fma ; v_cmp ; s_and_saveexec ; s_cbranch_execz (taken) ; s_or exec", k_guard, w, b0);
            run("16 v_fma ; v_cmp ; s_cbranch_vccnz (not taken)", k_vote, w, b0);
        }
    }
    return 0;
}
