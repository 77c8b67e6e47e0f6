// mvrl_flow.hip - stand-alone turbulence-field operators + RNG fill.
//   flow_interp_kernel       ReconstructedFlow.interp  (tag_00.../flowGenerator.py:97-136), any n_comp <= 4
//   flow_reconstruct_kernel  ReconstructedFlow.__init__ + scale (flowGenerator.py:19-23, :76-92) - one-time setup
//   fill_uniform_kernel      uniform(lo, hi) actions for synthetic roll-outs (benchmark helper)
#include "mvrl_kernels.hpp"

namespace mvrl {

__global__ __launch_bounds__(MVRL_BLOCK) void flow_interp_kernel(const float* __restrict__ table, int n_t, int n_y, int n_x,
                                                                 int n_comp, float inv_dt, float inv_dx, float inv_dy,
                                                                 const float* __restrict__ t, const float* __restrict__ x,
                                                                 const float* __restrict__ y, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    float tt = t[i] * inv_dt, xx = x[i] * inv_dx, yy = y[i] * inv_dy;
    int kk = min(n_t - 2, max(0, (int)floorf(tt)));
    int ii = min(n_x - 2, max(0, (int)floorf(xx)));
    int jj = min(n_y - 2, max(0, (int)floorf(yy)));
    float ft = tt - (float)kk, fx = xx - (float)ii, fy = yy - (float)jj;
    float wt0 = 1.f - ft, wx0 = 1.f - fx, wy0 = 1.f - fy;
    const size_t row = (size_t)n_x * n_comp, plane = (size_t)n_y * row;
    const float* p0 = table + (size_t)kk * plane + (size_t)jj * row + (size_t)ii * n_comp;
    const float* p1 = p0 + plane;
    for (int c = 0; c < n_comp; c++) {
        float a = wy0 * (p0[c] * wx0 + p0[n_comp + c] * fx) + fy * (p0[row + c] * wx0 + p0[row + n_comp + c] * fx);
        float b = wy0 * (p1[c] * wx0 + p1[n_comp + c] * fx) + fy * (p1[row + c] * wx0 + p1[row + n_comp + c] * fx);
        out[i * n_comp + c] = a * wt0 + b * ft;
    }
}

// out[t][s] = mul[s % 3] * ( sum_k (Re m[s][k] * Re c[k][t] - Im m[s][k] * Im c[k][t]) + ltm[s] ) + add[s % 3]
// Block = 256 consecutive s for one t; the K coefficients of that t are staged in LDS (wave-uniform reads ->
// broadcast), each lane streams its own contiguous row of modes.
__global__ __launch_bounds__(MVRL_BLOCK) void flow_reconstruct_kernel(const float* __restrict__ mre, const float* __restrict__ mim,
                                                                      const float* __restrict__ cre, const float* __restrict__ cim,
                                                                      const float* __restrict__ ltm, int n_space3, int n_modes,
                                                                      int n_t, float mul0, float mul1, float mul2, float add0,
                                                                      float add1, float add2, float* __restrict__ out) {
    extern __shared__ float lds[];  // [2][n_modes]
    const int t = blockIdx.y;
    for (int k = threadIdx.x; k < n_modes; k += MVRL_BLOCK) {
        lds[k] = cre[(size_t)k * n_t + t];
        lds[n_modes + k] = cim[(size_t)k * n_t + t];
    }
    __syncthreads();
    const int s = blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (s >= n_space3) return;
    const float* r = mre + (size_t)s * n_modes;
    const float* q = mim + (size_t)s * n_modes;
    float acc = 0.f;
    for (int k = 0; k < n_modes; k++) acc = fmaf(r[k], lds[k], fmaf(-q[k], lds[n_modes + k], acc));
    acc += ltm[s];
    const int c = s % 3;
    const float mul = (c == 0) ? mul0 : ((c == 1) ? mul1 : mul2);
    const float add = (c == 0) ? add0 : ((c == 1) ? add1 : add2);
    out[(size_t)t * n_space3 + s] = fmaf(mul, acc, add);
}

__global__ __launch_bounds__(MVRL_BLOCK) void fill_uniform_kernel(float* __restrict__ dst, int64_t n, uint64_t seed,
                                                                  uint64_t counter, float lo, float hi) {
    const int64_t q = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;  // one Philox block = 4 outputs
    if (q * 4 >= n) return;
    Philox4 r = philox4x32_10((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)counter, (uint32_t)(counter >> 32),
                              (uint32_t)seed, (uint32_t)(seed >> 32));
    const float w = hi - lo;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int64_t j = q * 4 + k;
        if (j < n) dst[j] = fmaf(u01(r.v[k]), w, lo);
    }
}

hipError_t launch_flow_interp(const float* table, int n_t, int n_y, int n_x, int n_comp, float inv_dt, float inv_dx,
                              float inv_dy, const float* t, const float* x, const float* y, int64_t n, float* out,
                              hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(flow_interp_kernel, grid, block, 0, stream, table, n_t, n_y, n_x, n_comp, inv_dt, inv_dx, inv_dy, t, x,
                       y, n, out);
    return hipGetLastError();
}

hipError_t launch_flow_reconstruct(const float* modes_re, const float* modes_im, const float* coeffs_re,
                                   const float* coeffs_im, const float* ltm, int n_space3, int n_modes, int n_t,
                                   const float* scale_mul, const float* scale_add, float* out, hipStream_t stream) {
    dim3 grid((unsigned)((n_space3 + MVRL_BLOCK - 1) / MVRL_BLOCK), (unsigned)n_t), block(MVRL_BLOCK);
    size_t lds = (size_t)2 * n_modes * sizeof(float);
    hipLaunchKernelGGL(flow_reconstruct_kernel, grid, block, lds, stream, modes_re, modes_im, coeffs_re, coeffs_im, ltm,
                       n_space3, n_modes, n_t, scale_mul[0], scale_mul[1], scale_mul[2], scale_add[0], scale_add[1],
                       scale_add[2], out);
    return hipGetLastError();
}

// One idle wave that holds its stream for `ticks` ticks of the 100 MHz real-time counter (mvrl_delay_dev).  Exit is
// guaranteed: the loop ends after a bounded number of polls whatever the counter does.
__global__ __launch_bounds__(64) void delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int guard = 0; guard < (1 << 22); guard++) {
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// The step kernels' turbulence table (FlowDev::table): cell (t, y, x) = the whole 2 x 2 x 2 stencil, [dt][dy][dx] x (u, v),
// built from the caller's plain [n_t][n_y][n_x][2] table (neighbours beyond the last row / column / slice repeat the last
// one; lookups never read those: indices are clamped to n - 2).  T = float or double.
template <class T>
__global__ __launch_bounds__(MVRL_BLOCK) void flow_cell_kernel(const T* __restrict__ src, T* __restrict__ dst, int n_t, int n_y, int n_x) {
    const int64_t cells = (int64_t)n_y * n_x;
    const int64_t c = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (c >= cells) return;
    const int y = (int)(c / n_x), x = (int)(c % n_x);
    const int ys[2] = {y, y + 1 < n_y ? y + 1 : y}, xs[2] = {x, x + 1 < n_x ? x + 1 : x};
    for (int t = blockIdx.y; t < n_t; t += gridDim.y) {   // grid.y is capped at 65 535
        const int ts[2] = {t, t + 1 < n_t ? t + 1 : t};
        T* d = dst + ((int64_t)t * cells + c) * 16;
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) {
                    const T* a = src + (((int64_t)ts[it] * n_y + ys[iy]) * n_x + xs[ix]) * 2;
                    d[((it * 2 + iy) * 2 + ix) * 2 + 0] = a[0];
                    d[((it * 2 + iy) * 2 + ix) * 2 + 1] = a[1];
                }
    }
}

hipError_t launch_flow_cells(const void* src, void* dst, int n_t, int n_y, int n_x, bool f64, hipStream_t stream) {
    const int64_t cells = (int64_t)n_y * n_x;
    dim3 grid((unsigned)((cells + MVRL_BLOCK - 1) / MVRL_BLOCK), (unsigned)(n_t < 65535 ? n_t : 65535)), block(MVRL_BLOCK);
    if (f64) hipLaunchKernelGGL(flow_cell_kernel<double>, grid, block, 0, stream, (const double*)src, (double*)dst, n_t, n_y, n_x);
    else hipLaunchKernelGGL(flow_cell_kernel<float>, grid, block, 0, stream, (const float*)src, (float*)dst, n_t, n_y, n_x);
    return hipGetLastError();
}

hipError_t launch_delay(int microseconds, hipStream_t stream) {
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long)microseconds * 100ull);
    return hipGetLastError();
}

hipError_t launch_fill_uniform(float* dst, int64_t n, uint64_t seed, uint64_t counter, float lo, float hi,
                               hipStream_t stream) {
    int64_t nq = (n + 3) / 4;
    dim3 grid((unsigned)((nq + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(fill_uniform_kernel, grid, block, 0, stream, dst, n, seed, counter, lo, hi);
    return hipGetLastError();
}

}  // namespace mvrl
