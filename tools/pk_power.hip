// Is packed fp32 math (v_pk_fma_f32: two FMAs per lane and instruction) cheaper per flop when the chip sits at its power
// limit?  Kernels with known flop counts - 8 independent chains per lane of v_fma_f32 vs 4 chains of v_pk_fma_f32 on
// register pairs - each launched back to back for ~1.5 s at 4 waves per SIMD; sustained TFLOP/s is the answer (the clock
// the power controller settles at differs).  hipcc -O3 --offload-arch=gfx950 tools/pk_power.hip -o tools/pk_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    float x[8], y = 0.999f + threadIdx.x * 1e-9f, z = 1e-3f + threadIdx.x * 1e-9f;
    float2v p[4], py = {y, y * 1.0001f}, pz = {z, z * 1.0001f};
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < 4; i++) { p[i].x = x[2 * i]; p[i].y = x[2 * i + 1]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(py), "v"(pz));
            } else if (MODE == 2) {   // VOP2 with a literal: the cheapest encoding
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fmac_f32 %0, 0x3a83126f, %1" : "+v"(x[i]) : "v"(z));
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(py), "v"(pz));
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(py));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double flops_per_inner) {
    const int blocks = 1024 * 4, iters = 20000;
    float* d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(d, 100);
    (void)hipDeviceSynchronize();
    int launches = 0;
    (void)hipEventRecord(e0);
    float ms = 0;
    do {   // ~1.5 s of back-to-back launches
        for (int q = 0; q < 10; q++) { k<MODE><<<blocks, 64>>>(d, iters); launches++; }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 1500.f);
    (void)hipEventRecord(e0);
    for (int q = 0; q < 10; q++) k<MODE><<<blocks, 64>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 10.0 * blocks * 64.0 * iters * 32.0 * flops_per_inner;
    printf("%-34s %8.2f ms per 10 launches  %7.1f TFLOP/s sustained (after %d warm launches)\n", name, ms, flops / ms / 1e9, launches);
    (void)hipFree(d);
}

int main() {
    run<0>("8 x v_fma_f32 (VOP3, 3 VGPR)", 8 * 2.0);
    run<1>("4 x v_pk_fma_f32", 4 * 4.0);
    run<2>("8 x v_fmac_f32 literal (VOP2)", 8 * 2.0);
    run<3>("4 x (v_pk_fma_f32 + v_pk_mul_f32)", 4 * (4.0 + 2.0));
    run<0>("8 x v_fma_f32 again", 8 * 2.0);
    return 0;
}
