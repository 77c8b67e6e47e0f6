// mvrl_kernels.hpp - host-callable launchers of the HIP kernels (internal to libmvrl.so).
#pragma once
#include "mvrl_device.hpp"

namespace mvrl {

hipError_t launch_rov6_step(const Rov6Dev* p_dev, const StepIO& io, const FlowDev& fl, bool baked, bool ctrl, bool sym, bool zoh,
                            bool flow, bool rk45, hipStream_t stream);
hipError_t launch_rov6_derivs(const Rov6Dev* p, bool baked, bool sym, int64_t n, const float* t, const float* y, const float* sp,
                              const float* cur, float* eold, float* eint, float* told, const uint8_t* has_old, float* dy, float* aux,
                              hipStream_t stream);
hipError_t launch_rov6_components(const Rov6Dev* p, int64_t n, const float* angles, const float* vel, const float* rpm_in, float* comp,
                                  hipStream_t stream);
hipError_t launch_rov6_mass_solve(const Rov6Dev* p, bool baked, bool sym, int64_t n, const float* rhs, float* acc, hipStream_t stream);
hipError_t launch_rov6_unit(const Rov6Dev* p, bool baked, bool sym, int64_t n, const float* angles, const float* gcf,
                            const float* rpm_in, const float* vel, float* axes, float* rpm_out, float* rhs, float* h_out,
                            hipStream_t stream);
hipError_t launch_rov6_observe(const Rov6Dev* p, const float* state, int64_t n, float* obs, hipStream_t stream);
hipError_t launch_rov3_observe(const Rov3Dev* p, const float* state, int64_t n, float* obs, hipStream_t stream);
hipError_t launch_auv_observe(const AuvDev& p, const float* state, int64_t n, float* obs, hipStream_t stream);
hipError_t launch_rov3_derivs(const Rov3Dev* p, bool baked, int64_t n, const float* t, const float* y, const float* sp, const float* cur,
                              float* eold, float* eint, float* told, const uint8_t* has_old, float* dy, float* aux, hipStream_t stream);
hipError_t launch_rov6_reset(const Rov6Dev* p_dev, float* state, int64_t n, const uint8_t* mask, const float* init, float* obs,
                             uint64_t seed, int64_t env_offset, float t_quarter, hipStream_t stream);

hipError_t launch_rov3_step(const Rov3Dev* p_dev, const StepIO& io, const FlowDev& fl, bool baked, bool zoh, bool flow,
                            bool rk45, hipStream_t stream);
hipError_t launch_rov3_reset(const Rov3Dev* p_dev, float* state, int64_t n, const uint8_t* mask, const float* init, float* obs,
                             uint64_t seed, int64_t env_offset, float t_quarter, hipStream_t stream);

hipError_t launch_auv_step(const AuvDev& p, const StepIO& io, const FlowDev& fl, bool flow, hipStream_t stream);
hipError_t launch_auv_pd_episodes(const AuvDev& p, const FlowDev& fl, bool flow, float* state, int64_t n, double dt, int max_steps,
                                  int n_steps, float policy_dt, const float* P, const float* D, float* returns, int32_t* lengths,
                                  hipStream_t stream);
hipError_t launch_auv_reset(const AuvDev& p, float* state, int64_t n, const uint8_t* mask, const float* init, float* obs,
                            uint64_t seed, int64_t env_offset, float t_quarter, hipStream_t stream);

hipError_t launch_flow_interp(const float* table, int n_t, int n_y, int n_x, int n_comp, float inv_dt, float inv_dx,
                              float inv_dy, const float* t, const float* x, const float* y, int64_t n, float* out,
                              hipStream_t stream);
hipError_t launch_flow_reconstruct(const float* modes_re, const float* modes_im, const float* coeffs_re,
                                   const float* coeffs_im, const float* ltm, int n_space3, int n_modes, int n_t,
                                   const float* scale_mul, const float* scale_add, float* out, hipStream_t stream);
hipError_t launch_pd_policy(const float* obs, int obs_dim, float* old_obs, uint32_t* calls, float* actions, int64_t n, float dt,
                            const float* P, const float* D, float noise_sigma, uint64_t seed, hipStream_t stream);
hipError_t launch_los_policy(const float* obs, int obs_dim, float* actions, int64_t n, float Rnav, hipStream_t stream);
hipError_t launch_replay_add_sym(const float* obs, const float* next_obs, const float* act, const float* rew, const uint8_t* done,
                                 const uint8_t* timeout, int64_t n_envs, float* b_obs, float* b_next, float* b_act, float* b_rew,
                                 uint8_t* b_done, uint8_t* b_timeout, int64_t buffer_size, int64_t pos, int n_tr, hipStream_t stream);
hipError_t launch_flow_cells(const void* src, void* dst, int n_t, int n_y, int n_x, bool f64, hipStream_t stream);
hipError_t launch_delay(int microseconds, hipStream_t stream);
hipError_t launch_fill_uniform(float* dst, int64_t n, uint64_t seed, uint64_t counter, float lo, float hi,
                               hipStream_t stream);

enum { R6_WORDS_ = 41, R3_WORDS_ = 24, AUV_WORDS_ = 56 };

}  // namespace mvrl
