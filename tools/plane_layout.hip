// Calibration micro-benchmark no. 5 (round 2): HBM bandwidth of the environment kernels' access pattern as a function of
// the state layout.  Every lane (env) reads W_IN words and writes W_OUT words, one wave per workgroup like the step kernels:
//   layout 0  "planes"  word k of env i at base[k * n + i]                      (one 256-B access per wave and word, the
//                                                                                words of a wave 4 n bytes apart)
//   layout 1  "tiles"   word k of env i at base[(i / 64) * W * 64 + k * 64 + i % 64]   (the W words of a wave contiguous:
//                                                                                W x 256 B in one run)
// Reported: GB/s of (W_IN + W_OUT) x 4 B x n per launch.  hipcc -O3 --offload-arch=gfx950 tools/plane_layout.hip -o tools/plane_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int W, int W_IN, int W_OUT, int TILED, int WORK>
__global__ __launch_bounds__(64) void touch(float* __restrict__ st, unsigned n) {
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float v[W_IN];
    float* base = TILED ? st + (size_t)blockIdx.x * (W * 64) + threadIdx.x : st + i;
    const size_t stride = TILED ? 64 : n;
#pragma unroll
    for (int k = 0; k < W_IN; k++) v[k] = base[k * stride];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < W_IN; k++) s += v[k];
    // a little dependent arithmetic, like a step kernel between its loads and its stores
#pragma unroll 1
    for (int r = 0; r < WORK; r++) s = fmaf(s, 0.999f, 1e-3f);
#pragma unroll
    for (int k = 0; k < W_OUT; k++) base[k * stride] = v[k] + s;
}

// AuvEnv-like with its other traffic: 8 scattered 8-byte gathers per lane from a 40 MB table (two time slices, a 2 x 2
// stencil each), GATHER = 0 none / 1 lanes spread over the table like envs with random time offsets / 2 all lanes of a wave
// in one time slice; ROWS = row-major [n][11] observation store (44 B per lane, strided) or not
template <int GATHER, int ROWS, int WORK>
__global__ __launch_bounds__(64) void auv_like(float* __restrict__ st, const float2* __restrict__ table, float* __restrict__ obs, unsigned n, unsigned salt) {
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    constexpr int W = 56, W_IN = 53, W_OUT = 14;
    float v[W_IN];
    float* base = st + i;
#pragma unroll
    for (int k = 0; k < W_IN; k++) v[k] = base[(size_t)k * n];
    float s = 0.f;
    if (GATHER) {
        // nT = 2000 slices of 41 x 61 float2; time index from the lane (GATHER 1: hashed over 500 slices; 2: per wave)
        unsigned h = (GATHER == 1 ? i : blockIdx.x) * 2654435761u + salt;
        const unsigned kk = (h >> 8) % 500u, jj = (i * 7u + salt) % 40u, ii = (i * 13u + salt) % 60u;
        const float2* p0 = table + ((size_t)kk * 41 + jj) * 61 + ii;
        const float2* p1 = p0 + 41 * 61;
        if (GATHER <= 2) {
            float2 g[8] = {p0[0], p0[1], p0[61], p0[62], p1[0], p1[1], p1[61], p1[62]};
#pragma unroll
            for (int k = 0; k < 8; k++) s += g[k].x * 0.3f + g[k].y;
        } else if (GATHER == 3) {   // the x-neighbours of a corner pair are adjacent: 4 gathers of 16 B (8-byte aligned)
            struct __attribute__((packed, aligned(4))) Pair { float a, b, c, d; };
            const Pair q[4] = {*reinterpret_cast<const Pair*>(p0), *reinterpret_cast<const Pair*>(p0 + 61), *reinterpret_cast<const Pair*>(p1),
                               *reinterpret_cast<const Pair*>(p1 + 61)};
#pragma unroll
            for (int k = 0; k < 4; k++) s += q[k].a * 0.3f + q[k].b + q[k].c * 0.3f + q[k].d;
        } else if (GATHER == 5) {   // time-paired 16-B cells (the layout the library uses): x neighbours adjacent, 2 grid rows
            const float4* q = reinterpret_cast<const float4*>(table) + ((size_t)kk * 41 + jj) * 61 + ii;
            const float4 a = q[0], b2 = q[1], c = q[61], d2 = q[62];
            s += a.x + a.y + a.z + a.w + b2.x + b2.y + b2.z + b2.w + c.x + c.y + c.z + c.w + d2.x + d2.y + d2.z + d2.w;
        } else if (GATHER == 6) {   // one 64-B cell per (t, y, x) holds the whole 2 x 2 x 2 stencil: one cache line per lookup
            const float4* q = reinterpret_cast<const float4*>(table) + (((size_t)kk * 41 + jj) * 61 + ii) * 4;
            const float4 a = q[0], b2 = q[1], c = q[2], d2 = q[3];
            s += a.x + a.y + a.z + a.w + b2.x + b2.y + b2.z + b2.w + c.x + c.y + c.z + c.w + d2.x + d2.y + d2.z + d2.w;
        } else {                    // stencil-packed table: one 32-B cell per (t, y, x) holds the 2 x 2 stencil: 2 gathers of 32 B
            struct __attribute__((aligned(16))) Cell { float4 lo, hi; };
            const Cell* c0 = reinterpret_cast<const Cell*>(table) + ((size_t)kk * 41 + jj) * 61 + ii;
            const Cell* c1 = c0 + 41 * 61;
            const Cell a = *c0, bq = *c1;
            s += a.lo.x + a.lo.y + a.lo.z + a.lo.w + a.hi.x + a.hi.y + a.hi.z + a.hi.w + bq.lo.x + bq.lo.y + bq.lo.z + bq.lo.w + bq.hi.x + bq.hi.y + bq.hi.z + bq.hi.w;
        }
    }
#pragma unroll
    for (int k = 0; k < W_IN; k++) s += v[k];
#pragma unroll 1
    for (int r = 0; r < WORK; r++) s = fmaf(s, 0.999f, 1e-3f);
#pragma unroll
    for (int k = 0; k < W_OUT; k++) base[(size_t)k * n] = v[k] + s;
    if (ROWS) {
#pragma unroll
        for (int k = 0; k < 11; k++) obs[(size_t)i * 11 + k] = s + k;
    }
}

static int g_lds_bytes = 0;   // dynamic LDS per one-wave block: 160 KB / g_lds_bytes blocks fit a CU (occupancy limiter)
template <int GATHER, int ROWS, int WORK>
void run_auv(const char* tag, unsigned n, int iters) {
    float *d, *obs;
    float2* table;
    hipMalloc(&d, (size_t)n * 56 * 4);
    hipMemset(d, 0, (size_t)n * 56 * 4);
    hipMalloc(&obs, (size_t)n * 11 * 4);
    const size_t tbytes = (size_t)2001 * 41 * 61 * (GATHER == 4 ? 32 : (GATHER == 5 ? 16 : (GATHER == 6 ? 64 : 8)));
    hipMalloc(&table, tbytes);
    hipMemset(table, 0, tbytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; w++) auv_like<GATHER, ROWS, WORK><<<(n + 63) / 64, 64, g_lds_bytes>>>(d, table, obs, n, w);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < iters; it++) auv_like<GATHER, ROWS, WORK><<<(n + 63) / 64, 64, g_lds_bytes>>>(d, table, obs, n, it);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, gb = (double)((53 + 14 + (ROWS ? 11 : 0)) * 4 + (GATHER ? 64 : 0)) * n / (us * 1e-6) / 1e9;
    printf("%-44s n %8u  %7.1f us/launch  %7.0f GB/s\n", tag, n, us, gb);
    hipFree(d); hipFree(obs); hipFree(table);
}

template <int W, int W_IN, int W_OUT, int TILED, int WORK>
void run(const char* tag, unsigned n, int iters) {
    float* d;
    const size_t bytes = (size_t)((n + 63) / 64) * 64 * W * sizeof(float);
    hipMalloc(&d, bytes);
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; w++) touch<W, W_IN, W_OUT, TILED, WORK><<<(n + 63) / 64, 64>>>(d, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < iters; it++) touch<W, W_IN, W_OUT, TILED, WORK><<<(n + 63) / 64, 64>>>(d, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, gb = (double)(W_IN + W_OUT) * 4.0 * n / (us * 1e-6) / 1e9;
    printf("%-44s n %8u  %7.1f us/launch  %7.0f GB/s\n", tag, n, us, gb);
    hipFree(d);
}

int main() {
    const unsigned n = 1048576;
    // AuvEnv-like: 56 words, 53 read, 14 written
    run<56, 53, 14, 0, 0>("auv-like planes, no work", n, 300);
    run<56, 53, 14, 1, 0>("auv-like tiles,  no work", n, 300);
    run<56, 53, 14, 0, 400>("auv-like planes, 400 dependent fma", n, 300);
    run<56, 53, 14, 1, 400>("auv-like tiles,  400 dependent fma", n, 300);
    // 6-DoF-like: 41 words, 33 read, 27 written
    run<41, 33, 27, 0, 0>("rov6-like planes, no work", n, 300);
    run<41, 33, 27, 1, 0>("rov6-like tiles,  no work", n, 300);
    // copy-like reference: 32 read, 32 written
    run<32, 32, 32, 0, 0>("copy-like planes", n, 300);
    run<32, 32, 32, 1, 0>("copy-like tiles", n, 300);
    // what the AuvEnv step adds to the plane traffic
    run_auv<0, 0, 0>("auv planes only", n, 300);
    run_auv<1, 0, 0>("auv + 8 gathers, lanes spread over 500 slices", n, 300);
    run_auv<2, 0, 0>("auv + 8 gathers, a wave in one slice", n, 300);
    run_auv<3, 0, 0>("auv + 4 gathers of 16 B (x pairs)", n, 300);
    run_auv<4, 0, 0>("auv + 2 gathers of 32 B (stencil cells, 160 MB)", n, 300);
    run_auv<5, 0, 0>("auv + 4 gathers of 16 B (time pairs, 2 rows, 80 MB)", n, 300);
    run_auv<6, 0, 0>("auv + 1 cell of 64 B (full stencil, 320 MB)", n, 300);
    run_auv<0, 1, 0>("auv + row-major obs store (44 B / lane)", n, 300);
    run_auv<1, 1, 0>("auv + gathers + obs rows", n, 300);
    run_auv<1, 1, 200>("auv + gathers + obs rows + 200 dependent fma", n, 300);
    run_auv<1, 1, 800>("auv + gathers + obs rows + 800 dependent fma", n, 300);
    // occupancy: the real AuvEnv step kernel holds 123 VGPRs = 4 waves per SIMD; these kernels need ~70 (7 waves)
    for (int waves_per_simd : {8, 6, 5, 4, 3, 2}) {
        g_lds_bytes = 160 * 1024 / (4 * waves_per_simd);
        char tag[96];
        snprintf(tag, sizeof(tag), "auv + 64-B cell + obs rows, 200 fma, %d waves/SIMD", waves_per_simd);
        run_auv<6, 1, 200>(tag, n, 300);
    }
    g_lds_bytes = 0;
    return 0;
}
