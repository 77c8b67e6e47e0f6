// mvrl_rk45.hpp - scipy.integrate.solve_ivp(method="RK45") restated per lane (fp64 build only).
//
// The reference advances every env step with
//     solve_ivp(vehicle.derivs, (t-dt, t), y, 'RK45', t_eval=[t], max_step=dt, rtol=1e-3, atol=1e-3)
// (dynamicsModel_BlueROV2_Heavy_6DoF.py:555-557, _3DoF.py:475-477).  This is the same control loop as
// scipy 1.15.3 runs it (scipy/integrate/_ivp/rk.py: rk_step :14-75, RungeKutta.__init__ :85-106, _step_impl :111-179,
// RK45 tableau :377-405, RkDenseOutput :552-574; common.py: norm :63-65, select_initial_step :68-134), one
// independent adaptive integration per lane: a fresh solver per env step, whose f0 evaluation and initial-step probe
// both go through the stateful PID, exactly like the reference.  Lanes of a wave take different numbers of steps;
// the wave iterates until its slowest lane has reached t_bound.  This mode exists for exactness (it reproduces the
// reference's own env.step trajectories, goldens g10), not for throughput; the stage slopes K live in scratch.
#pragma once
#include <hip/hip_runtime.h>

namespace mvrl64 {

__device__ __forceinline__ double rk45_rms(const double* x, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s) / sqrt((double)n);
}

// RHS: void operator()(double t, const double* y, double* dy)  (may mutate controller state)
template <int N, class RHS>
__device__ inline int rk45_solve(RHS& f, double t0, double t_bound, double max_step, double rtol, double atol, double* y,
                                 int* nfev) {
    const double C[6] = {0, 1. / 5, 3. / 10, 4. / 5, 8. / 9, 1};
    const double A[6][5] = {{0, 0, 0, 0, 0},
                            {1. / 5, 0, 0, 0, 0},
                            {3. / 40, 9. / 40, 0, 0, 0},
                            {44. / 45, -56. / 15, 32. / 9, 0, 0},
                            {19372. / 6561, -25360. / 2187, 64448. / 6561, -212. / 729, 0},
                            {9017. / 3168, -355. / 33, 46732. / 5247, 49. / 176, -5103. / 18656}};
    const double B[6] = {35. / 384, 0, 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84};
    const double E[7] = {-71. / 57600, 0, 71. / 16695, -71. / 1920, 17253. / 339200, -22. / 525, 1. / 40};
    const double P[7][4] = {
        {1, -8048581381. / 2820520608, 8663915743. / 2820520608, -12715105075. / 11282082432},
        {0, 0, 0, 0},
        {0, 131558114200. / 32700410799, -68118460800. / 10900136933, 87487479700. / 32700410799},
        {0, -1754552775. / 470086768, 14199869525. / 1410260304, -10690763975. / 1880347072},
        {0, 127303824393. / 49829197408, -318862633887. / 49829197408, 701980252875. / 199316789632},
        {0, -282668133. / 205662961, 2019193451. / 616988883, -1453857185. / 822651844},
        {0, 40617522. / 29380423, -110615467. / 29380423, 69997945. / 29380423}};
    double K[7][N], fc[N], y_new[N], y_old[N], tmp[N], scale[N];
    int calls = 0;
    double t = t0;
    f(t, y, fc);  // RungeKutta.__init__: self.f = fun(t0, y0)
    calls++;
    double h_abs;
    {   // select_initial_step(fun, t0, y0, t_bound, max_step, f0, +1, order=4, rtol, atol)
        const double interval = fabs(t_bound - t0);
        if (interval == 0.0) {
            h_abs = 0.0;
        } else {
            for (int i = 0; i < N; i++) scale[i] = atol + fabs(y[i]) * rtol;
            for (int i = 0; i < N; i++) tmp[i] = y[i] / scale[i];
            const double d0 = rk45_rms(tmp, N);
            for (int i = 0; i < N; i++) tmp[i] = fc[i] / scale[i];
            const double d1 = rk45_rms(tmp, N);
            double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            h0 = fmin(h0, interval);
            for (int i = 0; i < N; i++) y_new[i] = y[i] + h0 * fc[i];
            f(t0 + h0, y_new, K[0]);
            calls++;
            for (int i = 0; i < N; i++) tmp[i] = (K[0][i] - fc[i]) / scale[i];
            const double d2 = rk45_rms(tmp, N) / h0;
            const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
            h_abs = fmin(fmin(100 * h0, h1), fmin(interval, max_step));
        }
    }
    bool have_step = false;
    double h_last = 0.0;
    int status = 0;
    while (t != t_bound) {  // OdeSolver.step until finished
        // (scipy: nextafter(t, direction * inf); the largest finite double gives the same neighbour and stays defined when the build
        // tells the compiler that infinities do not occur)
        const double min_step = 10 * fabs(nextafter(t, 1.7976931348623157e308) - t);
        double ha = (h_abs > max_step) ? max_step : ((h_abs < min_step) ? min_step : h_abs);
        bool accepted = false, rejected = false;
        double t_new = t, h = 0.0;
        while (!accepted) {
            if (ha < min_step) { status = -1; break; }  // TOO_SMALL_STEP
            // Not in scipy: an exit every lane reaches whatever its numbers are.  A non-finite state makes `err` a NaN; scipy's loop then
            // shrinks the step (fmax(0.2, NaN) = 0.2) until TOO_SMALL_STEP ends it, but this file is compiled with -fno-honor-nans, under
            // which the compiler owes a NaN nothing - and a lane that never leaves this loop takes the GPU with it.  The reference's own
            // steps take 10^2 ... 10^3 right-hand-side calls.
            if (calls > 2000000) { status = -2; break; }
            h = ha;
            t_new = t + h;
            if (t_new - t_bound > 0) t_new = t_bound;
            h = t_new - t;
            ha = fabs(h);
            for (int i = 0; i < N; i++) K[0][i] = fc[i];
            for (int s = 1; s < 6; s++) {  // rk_step
                for (int i = 0; i < N; i++) {
                    double a = 0.0;
                    for (int j = 0; j < s; j++) a += K[j][i] * A[s][j];
                    tmp[i] = y[i] + a * h;
                }
                f(t + C[s] * h, tmp, K[s]);
                calls++;
            }
            for (int i = 0; i < N; i++) {
                double a = 0.0;
                for (int j = 0; j < 6; j++) a += K[j][i] * B[j];
                y_new[i] = y[i] + h * a;
            }
            f(t + h, y_new, K[6]);
            calls++;
            for (int i = 0; i < N; i++) {
                const double sc = atol + fmax(fabs(y[i]), fabs(y_new[i])) * rtol;
                double e = 0.0;
                for (int j = 0; j < 7; j++) e += K[j][i] * E[j];
                tmp[i] = e * h / sc;
            }
            const double err = rk45_rms(tmp, N);
            if (err < 1) {
                double factor = (err == 0) ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                ha *= factor;
                accepted = true;
            } else {
                ha *= fmax(0.2, 0.9 * pow(err, -0.2));
                rejected = true;
            }
        }
        if (status != 0) break;
        for (int i = 0; i < N; i++) { y_old[i] = y[i]; y[i] = y_new[i]; fc[i] = K[6][i]; }
        t = t_new;
        h_abs = ha;
        h_last = h;
        have_step = true;
    }
    if (have_step && status == 0) {  // t_eval = [t_bound]: dense output of the final step at x = 1
        for (int i = 0; i < N; i++) {
            double acc = 0.0;
            for (int j = 0; j < 7; j++) acc += K[j][i] * (((P[j][0] + P[j][1]) + P[j][2]) + P[j][3]);
            y[i] = y_old[i] + h_last * acc;
        }
    }
    if (nfev) *nfev = calls;
    return status;
}

}  // namespace mvrl64
