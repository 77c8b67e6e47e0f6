// Calibration micro-benchmark (round 5): sustained fp64 VALU issue rate of the chip next to the fp32 one, for the kinds of
// instruction the fp64 step kernel is made of: v_fma_f64 (dependent / independent chains, 1..4 waves per SIMD), v_max/min_f64
// clamps, 64-bit selects (two v_cndmask_b32), and an fma with an SGPR-pair constant.  Answers: what is the fp64 issue peak under
// the board's power cap, how many waves per SIMD does it take, and what does a 32-bit op cost next to an fp64 one.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_f64.hip -o tools/valu_f64 && tools/valu_f64
#include <hip/hip_runtime.h>
#include <stdio.h>

template <class T, int ILP, int KIND>
__global__ __launch_bounds__(64) void chain_kernel(T* out, int iters, T a, T b, T lo, T hi) {
    T x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = (T)threadIdx.x * (T)1e-3 + (T)k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (KIND == 0) x[k] = x[k] * a + b;                           // fma, VGPR/SGPR operands
                if (KIND == 1) x[k] = fmin(hi, fmax(lo, x[k] * a + b));       // fma + clamp (max + min): 3 instructions
                if (KIND == 2) { T y = x[k] * a + b; x[k] = (y > hi) ? lo : y; }   // fma + compare + 64-bit select: 2 + 2 x cndmask
            }
    }
    T s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class T, int ILP, int KIND>
void run(const char* what, int waves_per_simd, int iters, int instr_per_op) {
    const int blocks = 1024 * waves_per_simd;   // one-wave blocks, 1024 SIMDs
    T* d;
    hipMalloc(&d, (size_t)blocks * 64 * sizeof(T));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chain_kernel<T, ILP, KIND><<<blocks, 64>>>(d, iters / 4, (T)0.999, (T)0.001, (T)-1e30, (T)1e30);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain_kernel<T, ILP, KIND><<<blocks, 64>>>(d, iters, (T)0.999, (T)0.001, (T)-1e30, (T)1e30);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * iters * 16 * ILP;   // wave-level operations
    printf("%-34s %s ILP=%d %d waves/SIMD: %8.3f ms  %.3f wave-ops/ns/SIMD (~%d instr each -> %.3f wave-instr/ns/SIMD), %.1f TFLOP/s of fma\n", what,
           sizeof(T) == 8 ? "fp64" : "fp32", ILP, waves_per_simd, ms, ops / 1024 / (ms * 1e6), instr_per_op, instr_per_op * ops / 1024 / (ms * 1e6),
           ops * 64 * 2 / ms / 1e9);
    hipFree(d);
}

int main() {
    const int N = 6000;
    for (int rep = 0; rep < 2; rep++) {   // second pass: clocks settled under load
        printf("--- pass %d\n", rep);
        run<double, 1, 0>("fma, dependent chain", 1, N, 1);
        run<double, 1, 0>("fma, dependent chain", 2, N, 1);
        run<double, 1, 0>("fma, dependent chain", 4, N, 1);
        run<double, 4, 0>("fma, 4 independent chains", 1, N, 1);
        run<double, 4, 0>("fma, 4 independent chains", 2, N, 1);
        run<double, 4, 0>("fma, 4 independent chains", 4, N / 2, 1);
        run<double, 8, 0>("fma, 8 independent chains", 2, N / 2, 1);
        run<double, 4, 1>("fma + clamp (max, min)", 2, N / 2, 3);
        run<double, 4, 2>("fma + cmp + 64-bit select", 2, N / 2, 4);
        run<float, 4, 0>("fma, 4 independent chains", 4, N, 1);
        run<float, 8, 0>("fma, 8 independent chains", 4, N, 1);
        run<float, 4, 1>("fma + clamp (med3)", 4, N, 2);
        run<float, 4, 2>("fma + cmp + select", 4, N, 3);
    }
    return 0;
}
