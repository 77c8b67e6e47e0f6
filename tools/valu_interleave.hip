// Do "slow-class" VALU instructions (v_med3 / v_max / v_cmp: 1.9 ns back to back) overlap with fma-class ones (1.24 ns) when they
// alternate?  tools/valu_select.hip measured v_med3 + v_mul pairs at 2.1 ns - less than either sum.  Here: the SAME 8 slow + 8 fast
// instructions per group, once as two runs (8 slow, then 8 fast) and once alternating, at 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_interleave.hip -o tools/valu_interleave && tools/valu_interleave
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DECL                                                                                                                     \
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    float y0 = x0 * 2, y1 = x1 * 2, y2 = x2 * 2, y3 = x3 * 2, y4 = x4 * 2, y5 = x5 * 2, y6 = x6 * 2, y7 = x7 * 2;               \
    float va = a + x0 * 1e-9f, vb = b + x0 * 1e-9f;
#define S(i) asm volatile(SLOW " %0, %0, %1, %2" : "+v"(x##i) : "v"(va), "v"(vb));
#define F(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y##i) : "v"(va), "v"(vb));
#define GROUPED S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define ALTERN S(0) F(0) S(1) F(1) S(2) F(2) S(3) F(3) S(4) F(4) S(5) F(5) S(6) F(6) S(7) F(7)
#define PAIRS S(0) S(1) F(0) F(1) S(2) S(3) F(2) F(3) S(4) S(5) F(4) F(5) S(6) S(7) F(6) F(7)
#define ONLYS S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ONLYF F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
// three fast per slow: the step kernel's ratio is about 7 : 1
#define MIX31 S(0) F(0) F(1) F(2) S(1) F(3) F(4) F(5) S(2) F(6) F(7) F(0) S(3) F(1) F(2) F(3)
#define RUN31 S(0) S(1) S(2) S(3) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3)
#define KERN(name, BODY)                                                                              \
    __global__ __launch_bounds__(64) void name(float* out, int iters, float a, float b) {             \
        DECL for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int r = 0; r < 4; r++) { BODY } } \
        out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7; \
    }
#define SLOW "v_med3_f32"
KERN(k_grouped, GROUPED)
KERN(k_altern, ALTERN)
KERN(k_pairs, PAIRS)
KERN(k_onlys, ONLYS)
KERN(k_onlyf, ONLYF)
KERN(k_mix31, MIX31)
KERN(k_run31, RUN31)
#undef S
// compares as the slow op (result to an SGPR pair nobody reads)
#define S(i) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" ::"v"(x##i), "v"(va) : "s20", "s21");
KERN(k_cgrouped, GROUPED)
KERN(k_caltern, ALTERN)

typedef void (*kern_t)(float*, int, float, float);
static void run(const char* what, kern_t k, int instr_per_iter, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 40000;   // ~10-25 ms per launch: the clock ramp of the first ~0.5 ms is noise
    float* d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);   // warm-up: same length, so the timed launch starts at the settled clock
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * iters * instr_per_iter;
    printf("%-64s %d waves/SIMD: %8.3f ms  %.3f ns per wave-instruction per SIMD\n", what, waves_per_simd, ms, ms * 1e6 * 1024 / winst);
    (void)hipFree(d);
}
int main() {
    for (int pass = 0; pass < 2; pass++) {
        printf("--- pass %d\n", pass);
        for (int w = 1; w <= 4; w *= 2) {
            run("8 v_med3 only", k_onlys, 32, w);
            run("8 v_fma only", k_onlyf, 32, w);
            run("8 v_med3 then 8 v_fma (two runs)", k_grouped, 64, w);
            run("v_med3, v_fma alternating", k_altern, 64, w);
            run("2 v_med3, 2 v_fma alternating", k_pairs, 64, w);
            run("1 v_med3 : 3 v_fma, spread", k_mix31, 64, w);
            run("4 v_med3 then 12 v_fma (runs)", k_run31, 64, w);
            run("8 v_cmp then 8 v_fma (two runs)", k_cgrouped, 64, w);
            run("v_cmp, v_fma alternating", k_caltern, 64, w);
        }
    }
    return 0;
}
