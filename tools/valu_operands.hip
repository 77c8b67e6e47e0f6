// Calibration micro-benchmark no. 2: sustained fp32 VALU issue rate as a function of how many VGPR source operands an
// instruction reads (constants / SGPR operands are free).  8 independent chains per lane, 2-8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_operands.hip -o valu_operands && ./valu_operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ILP 8
// MODE 0: x = fma(x, s, s)      1 VGPR source      (v_fmaak / v_fma with 2 SGPRs)
// MODE 1: x = fma(x, y, s)      2 VGPR sources
// MODE 2: x = fma(x, y, z)      3 VGPR sources     (v_fma_f32 v,v,v,v)
// MODE 3: x = fma(y, z, x)      v_fmac_f32 (3 VGPR reads, dst = accumulator)
// MODE 4: x = x * y             v_mul_f32 2 VGPR
// MODE 5: x = (x > y) ? z : x   v_cmp + v_cndmask pair
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[ILP], y[ILP], z[ILP];
    unsigned long long m = __ballot(threadIdx.x & 1);
#pragma unroll
    for (int i = 0; i < ILP; i++) { x[i] = threadIdx.x * 1e-3f + i; y[i] = 0.999f + i * 1e-6f + threadIdx.x * 1e-9f; z[i] = 1e-3f * (i + 1) + threadIdx.x * 1e-9f; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x[i]) : "s"(a));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(b));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
                if (MODE == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (MODE == 6) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x[i]));
                if (MODE == 7) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
                if (MODE == 8) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(x[i]) : "v"(y[i]));
                if (MODE == 9) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x[i]) : "v"(z[i]));
                if (MODE == 10) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(x[i]));
                if (MODE == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "s"(b), "v"(z[i]));
                if (MODE == 12) asm volatile("v_fmac_f32 %0, 0x3a83126f, %1" : "+v"(x[i]) : "v"(z[i]));
                if (MODE == 13) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(z[i]));
                if (MODE == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(m));
                if (MODE == 15) asm volatile("v_cmp_ge_f32_e64 s[20:21], %0, %1" : : "v"(x[i]), "v"(y[i]) : "s20", "s21");
                if (MODE == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
                if (MODE == 17) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");
                if (MODE == 18) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (MODE == 19) asm volatile("v_sub_f32 %0, 0x3a83126f, %0" : "+v"(x[i]));
                if (MODE == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]) : "vcc");
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i] + y[i] + z[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int blocks, int iters) {
    float* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * 16 * ILP * (MODE == 5 ? 2 : 1);
    printf("%-28s %4.1f waves/SIMD: %.3f ms -> %.3f wave-instr/ns/SIMD\n", name, blocks * 4 / 1024.0, ms, winst / 1024 / (ms * 1e6));
    hipFree(d);
}
int main() {
    for (int w = 3; w <= 8; w += 5) {
        int blocks = 256 * w, iters = 4000 / w;
        run<6>("mul x,LITERAL,x (VOP2)", blocks, iters);
        run<7>("mul x,SGPR,x (VOP2)", blocks, iters);
        run<10>("mul x,1.0,x (inline const)", blocks, iters);
        run<8>("fmaak x,x,y,LITERAL", blocks, iters);
        run<9>("fmamk x,x,LITERAL,z", blocks, iters);
        run<11>("fmac x+=SGPR*z (VOP2)", blocks, iters);
        run<12>("fmac x+=LITERAL*z (VOP2)", blocks, iters);
        run<19>("sub x,LITERAL,x (VOP2)", blocks, iters);
        run<18>("add x,x,y (VOP2 2 VGPR)", blocks, iters);
        run<13>("med3 x,x,SGPR,z (VOP3)", blocks, iters);
        run<16>("med3 x,x,y,z (VOP3)", blocks, iters);
        run<14>("cndmask_e64 x,x,y,SGPRmask", blocks, iters);
        run<17>("cndmask_e32 x,x,y,vcc", blocks, iters);
        run<15>("cmp_e64 -> SGPR pair", blocks, iters);
        run<0>("fma x,s,s (1 VGPR src)", blocks, iters);
        run<1>("fma x,y,s (2 VGPR src)", blocks, iters);
        run<2>("fma x,y,z (3 VGPR src)", blocks, iters);
        run<3>("fmac x+=y*z (3 VGPR reads)", blocks, iters);
        run<4>("mul x,y (2 VGPR src)", blocks, iters);
        run<5>("cmp+cndmask pair", blocks, iters);
    }
    return 0;
}
