// mvrl_policy.hip - the reference's two hand-written baseline policies as batched device kernels, so that closed-loop
// roll-outs (evaluation against the PD baseline, imitation-data generation, LOS demos) never leave the GPU:
//   pd_policy_kernel    PDController.predict    tag_00_Dec2023_simpleControlTurbulence/verySimpleAuv.py:22-50
//   los_policy_kernel   LOSNavigation.predict + lineOfSight   dynamicsModel_BlueROV2_Heavy_3DoF.py:517-607
// One lane per environment; observations/actions row-major [n, dim] like everywhere at the ABI.
#include "mvrl_kernels.hpp"

namespace mvrl {

// actions = clip(x*P + (x - oldObs)/dt*D, -1, 1) with x = obs[:3]; oldObs = x afterwards (first call: oldObs = x).
// With noise_sigma > 0 a normal deviate is added before the second clip (verySimpleAuv.py:44-45); the deviates come
// from Philox + Box-Muller, not from numpy's global generator.
__global__ __launch_bounds__(MVRL_BLOCK) void pd_policy_kernel(const float* __restrict__ obs, int obs_dim, float* __restrict__ old_obs,
                                                               uint32_t* __restrict__ calls, float* __restrict__ actions,
                                                               int64_t n, float inv_dt, float p0, float p1, float p2, float d0,
                                                               float d1, float d2, float noise_sigma, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float P[3] = {p0, p1, p2}, D[3] = {d0, d1, d2};
    // calls[i] = predict() calls since the last reset: "oldObs is None" on the first, and the counter of this env's noise
    // stream (device-resident, so the launch carries nothing that changes from call to call - graph-replayable)
    const uint32_t epoch = calls[i];
    const bool ho = epoch != 0;
    float nz[3] = {0.f, 0.f, 0.f};
    if (noise_sigma > 0.f) {
        Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), epoch, 0x50444eu, (uint32_t)seed, (uint32_t)(seed >> 32));
        const float u1 = fmaxf(u01(r.v[0]), 1e-7f), u2 = u01(r.v[1]), u3 = fmaxf(u01(r.v[2]), 1e-7f), u4 = u01(r.v[3]);
        const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u3));
        nz[0] = noise_sigma * r1 * cosf(MVRL_TWO_PI_HI * u2);
        nz[1] = noise_sigma * r1 * sinf(MVRL_TWO_PI_HI * u2);
        nz[2] = noise_sigma * r2 * cosf(MVRL_TWO_PI_HI * u4);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float x = obs[i * obs_dim + k];
        const float xo = ho ? old_obs[i * 3 + k] : x;
        float a = clampf(x * P[k] + (x - xo) * inv_dt * D[k], -1.f, 1.f);
        a = clampf(a + nz[k], -1.f, 1.f);
        actions[i * 3 + k] = a;
        old_obs[i * 3 + k] = x;
    }
    calls[i] = epoch + 1u;
}

// lineOfSight(p0, p1, Rnav) (3DoF.py:517-581), branch for branch
__device__ __forceinline__ void line_of_sight(float p0x, float p0y, float p1x, float p1y, float Rnav, float& tx, float& ty) {
    const float dToWp = sqrtf(p1x * p1x + p1y * p1y);
    if (dToWp < Rnav) { tx = p1x; ty = p1y; return; }
    const float vx = p1x - p0x, vy = p1y - p0y;
    const float dSegment = sqrtf(vx * vx + vy * vy);
    const float hx = vx / dSegment, hy = vy / dSegment;
    const float det = p0x * p1y - p1x * p0y;
    const float delta = Rnav * Rnav * dSegment * dSegment - det * det;
    if (delta < 0.f) {
        const float dAlong = -p0x * hx - p0y * hy;
        if (dAlong > dSegment) { tx = p1x; ty = p1y; }
        else if (dAlong < 0.f) { tx = p0x; ty = p0y; }
        else { tx = p0x + dAlong * hx; ty = p0y + dAlong * hy; }
        return;
    }
    float sy = fsign(vy);
    if (fabsf(sy) < 1e-12f) sy = 1.f;
    const float sq = sqrtf(delta);
    const float dd = fmaxf(1e-6f, dSegment);
    const float den = dd * dd;
    const float a0x = (det * vy + sy * vx * sq) / den, a0y = (-det * vx + fabsf(vy) * sq) / den;
    const float a1x = (det * vy - sy * vx * sq) / den, a1y = (-det * vx - fabsf(vy) * sq) / den;
    const float s0 = (hx * (a0x - p0x) + hy * (a0y - p0y)) / dd;
    const float s1 = (hx * (a1x - p0x) + hy * (a1y - p0y)) / dd;
    if (s0 >= 0.f && s0 <= 1.f && s0 > s1) { tx = a0x; ty = a0y; }
    else if (s1 >= 0.f && s1 <= 1.f) { tx = a1x; ty = a1y; }
    else if (sqrtf(p1x * p1x + p1y * p1y) < sqrtf(p0x * p0x + p0y * p0y)) { tx = p1x; ty = p1y; }
    else { tx = p0x; ty = p0y; }
}

// LOSNavigation.predict (3DoF.py:584-607): obs = [p0(2), p1(2), psi_e] -> action = [target(2), psi_e]
__global__ __launch_bounds__(MVRL_BLOCK) void los_policy_kernel(const float* __restrict__ obs, int obs_dim, float* __restrict__ actions,
                                                                int64_t n, float Rnav) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* o = obs + i * obs_dim;
    float tx, ty;
    line_of_sight(o[0], o[1], o[2], o[3], Rnav, tx, ty);
    actions[i * 3 + 0] = tx;
    actions[i * 3 + 1] = ty;
    actions[i * 3 + 2] = o[4];
}

hipError_t launch_pd_policy(const float* obs, int obs_dim, float* old_obs, uint32_t* calls, float* actions, int64_t n, float dt,
                            const float* P, const float* D, float noise_sigma, uint64_t seed, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(pd_policy_kernel, grid, block, 0, stream, obs, obs_dim, old_obs, calls, actions, n, 1.0f / dt, P[0], P[1], P[2],
                       D[0], D[1], D[2], noise_sigma, seed);
    return hipGetLastError();
}

hipError_t launch_los_policy(const float* obs, int obs_dim, float* actions, int64_t n, float Rnav, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(los_policy_kernel, grid, block, 0, stream, obs, obs_dim, actions, n, Rnav);
    return hipGetLastError();
}

}  // namespace mvrl
