// Calibration micro-benchmark no. 3 (round 2): what a wave64 fp32 VALU instruction costs on gfx950 as a function of
//   (a) the distance to its producer inside the wave's own instruction stream (ILP interleaved chains = distance ILP),
//   (b) how many waves share the SIMD (1 .. 8),
//   (c) the instruction's size (4-byte VOP2 vs 8-byte literal / VOP3 forms) in a loop body too large for the instruction
//       buffer (16 .. 32 KB of straight-line code: instruction-cache / fetch bandwidth),
// one wave per workgroup like the step kernels.  hipcc -O3 --offload-arch=gfx950 tools/valu_dep.hip -o valu_dep
#include <hip/hip_runtime.h>
#include <stdio.h>

// FORM 0: v_fmac_f32 x, y, z (4 B)   1: v_fmac_f32 x, LITERAL, z (8 B)   2: v_fma_f32 x, x, y, z (VOP3, 8 B)
template <int ILP, int FORM, int UNROLL>
__global__ __launch_bounds__(64) void dep_kernel(float* out, int iters) {
    float x[ILP], y = 0.999f + threadIdx.x * 1e-9f, z = 1e-3f + threadIdx.x * 1e-9f;
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < UNROLL; r++)
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (FORM == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
                if (FORM == 1) asm volatile("v_fmac_f32 %0, 0x3a83126f, %1" : "+v"(x[k]) : "v"(z));
                if (FORM == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
            }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, int FORM, int UNROLL>
void run(const char* tag, int waves_per_simd, int iters) {
    const int blocks = 1024 * waves_per_simd;
    float* d;
    hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dep_kernel<ILP, FORM, UNROLL><<<blocks, 64>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    dep_kernel<ILP, FORM, UNROLL><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * iters * UNROLL * ILP;
    const double rate = winst / 1024 / (ms * 1e6);
    printf("%-10s dist %d  form %d  body %5d instr  %d waves/SIMD: %8.3f ms  %.3f wave-instr/ns/SIMD  (%.2f cyc/instr/SIMD @2.4GHz, %.2f cyc per wave)\n",
           tag, ILP, FORM, UNROLL * ILP, waves_per_simd, ms, rate, 2.4 / rate, 2.4 / rate * waves_per_simd);
    hipFree(d);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int ILP>
void sweep_waves() {
    const int total = 1 << 21;  // instructions per wave
    for (int w : {1, 2, 3, 4, 8}) run<ILP, 0, 16>("dep", w, total / (16 * ILP));
}

int main() {
    sweep_waves<1>(); sweep_waves<2>(); sweep_waves<3>(); sweep_waves<4>(); sweep_waves<6>(); sweep_waves<8>();
    // instruction footprint: 2048 / 4096 instructions of straight-line code per loop iteration, 3 waves per SIMD
    for (int w : {2, 3}) {
        run<8, 0, 32>("small4B", w, 2048);
        run<8, 1, 32>("small8B", w, 2048);
        run<8, 0, 256>("big4B", w, 256);     // 2048 instr x 4 B =  8 KB
        run<8, 1, 256>("big8B", w, 256);     // 2048 instr x 8 B = 16 KB
        run<8, 2, 256>("bigVOP3", w, 256);
        run<8, 0, 512>("huge4B", w, 128);    // 4096 instr x 4 B = 16 KB
        run<8, 1, 512>("huge8B", w, 128);    // 4096 instr x 8 B = 32 KB
    }
    return 0;
}
