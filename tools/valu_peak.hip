// Calibration micro-benchmark: sustained fp32 VALU issue rate of the chip (independent and dependent FMA chains),
// to express the step kernel's instruction throughput against what the hardware actually sustains.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int k = 0; k < ILP; k++) x[k] = fmaf(x[k], a, b);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ILP>
void run(int blocks, int iters) {
    float* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fma_kernel<ILP><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_kernel<ILP><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * 16 * ILP;  // wave-instructions
    printf("ILP=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.2f Twave-instr/s/1024SIMD -> %.3f wave-instr/ns/SIMD = %.2f GHz-equivalent at 2 cyc/instr, %.1f TFLOP/s\n",
           ILP, blocks, blocks * 4 / 1024.0, ms, winst / ms / 1e9, winst / 1024 / (ms * 1e6), winst / 1024 / (ms * 1e6) * 2, winst * 64 * 2 / ms / 1e9);
    hipFree(d);
}
int main() {
    run<1>(256 * 2, 4000);   // 2 waves/SIMD, fully dependent chain
    run<1>(256 * 3, 4000);
    run<1>(256 * 4, 4000);
    run<4>(256 * 2, 2000);
    run<4>(256 * 3, 2000);
    run<4>(256 * 4, 2000);
    run<8>(256 * 8, 1000);
    return 0;
}
