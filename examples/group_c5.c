/* BASELINE configs[4] from plain C, ONE host process: "6-DoF + turbulence, 8 388 608 envs sharded 8 x MI355X with RCCL obs gather" -
 * what the reference does with SB3's SubprocVecEnv (one Python process per env, tag/main_00_sbl.py:145-146), here one
 * mvrl_group over the node's GPUs: contiguous shards, one launch per device and step, one grouped RCCL send / recv of
 * (observation, done) rows to the root device per step, overlapped with the next step (two message buffers).
 *
 *   gcc -O2 -I include examples/group_c5.c -L marinevehiclereinforcementlearning_amd -lmvrl -lm -Wl,-rpath,$PWD/marinevehiclereinforcementlearning_amd -o /tmp/group_c5
 *   /tmp/group_c5 [envs_per_device = 1048576] [steps = 200] [n_devices = all visible] [repeat_device0 = 0]
 *
 * repeat_device0 = 1 lists device 0 n_devices times (rehearsal on a 1-GPU box: the messages then move by device-to-device copies,
 * RCCL refuses duplicate devices).  Prints env-steps/s with the outputs left on the shards and with the per-step gather to the
 * root; exits 2 without a HIP device (no CPU fallback).  UNMEASURED on 8 GPUs by the build (its box has one): DESIGN.md 6. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mvrl.h"

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != MVRL_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mvrl_group_last_error(g)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const long per_dev = argc > 1 ? atol(argv[1]) : 1048576;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    int n_dev = argc > 3 ? atoi(argv[3]) : mvrl_device_count();
    const int repeat0 = argc > 4 ? atoi(argv[4]) : 0;
    if (mvrl_device_count() < 1) { fprintf(stderr, "no HIP device visible (the library has no CPU fallback)\n"); return 2; }
    if (n_dev < 1 || n_dev > 64) n_dev = 1;
    int32_t devices[64];
    for (int i = 0; i < n_dev; i++) devices[i] = repeat0 ? 0 : i;

    mvrl_config cfg;
    if (mvrl_default_config(MVRL_MODEL_ROV6, per_dev * n_dev, &cfg) != MVRL_OK) { fprintf(stderr, "default_config: %s\n", mvrl_last_error(NULL)); return 1; }
    cfg.use_flow = 1;                 /* 6-DoF + turbulence */
    cfg.seed = 12345;
    mvrl_group* g = NULL;
    int rc = mvrl_group_create(&cfg, devices, n_dev, 0, &g);
    if (rc != MVRL_OK) { fprintf(stderr, "mvrl_group_create failed (%d): %s\n", rc, mvrl_group_last_error(NULL)); return rc == MVRL_ENODEV ? 2 : 1; }
    mvrl_group_layout lay;
    mvrl_group_info(g, &lay);

    /* a small synthetic turbulence table of the shipped grid size (the reference's coeffs / modes blobs are not distributed):
     * 64 snapshots of 41 x 61 (u, v), mean current 1 m/s, AuvEnv's scaling (dx = dy = 0.055 m, dt = 0.022 s) */
    const int nt = 64, ny = 41, nx = 61;
    float* table = (float*)malloc(sizeof(float) * nt * ny * nx * 2);
    for (int t = 0; t < nt; t++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                float* c = table + (((size_t)t * ny + j) * nx + i) * 2;
                c[0] = 1.0f + 0.08f * sinf(0.21f * i + 0.13f * t) * cosf(0.17f * j);
                c[1] = 0.06f * cosf(0.19f * i - 0.11f * t) * sinf(0.23f * j + 0.2f);
            }
    mvrl_flow_desc fd;
    memset(&fd, 0, sizeof(fd));
    fd.n_t = nt; fd.n_y = ny; fd.n_x = nx; fd.dt = 0.022; fd.dx = 0.055; fd.dy = 0.055;
    CHECK(mvrl_group_set_flow(g, table, &fd));
    free(table);

    CHECK(mvrl_group_reset(g));
    CHECK(mvrl_group_gather_dev(g));
    CHECK(mvrl_group_wait(g));
    printf("group of %d device(s), %ld envs each, %s transport, message %ld B per shard (%.1f B per env)\n", n_dev, per_dev,
           lay.transport ? "RCCL" : "device-to-device copy", (long)lay.msg_bytes, (double)lay.msg_bytes / (double)lay.cmax);

    for (int pass = 0; pass < 2; pass++) {          /* pass 0: outputs stay on the shards; pass 1: gathered to the root every step */
        for (int k = 0; k < 20; k++) {               /* warm-up */
            CHECK(mvrl_group_fill_actions(g, 12345, (uint64_t)k, -1.0f, 1.0f));
            CHECK(mvrl_group_step_dev(g, NULL));
            if (pass) CHECK(mvrl_group_gather_dev(g));
        }
        CHECK(mvrl_group_synchronize(g));
        const double t0 = now();
        for (int k = 0; k < steps; k++) {
            CHECK(mvrl_group_step_dev(g, NULL));     /* the warm-up's last actions again: random-action roll-out, no host data */
            if (pass) CHECK(mvrl_group_gather_dev(g));
        }
        CHECK(mvrl_group_synchronize(g));
        const double el = now() - t0;
        printf("%s: %d steps of %ld envs in %.3f s -> %.3e env-steps/s (%.1f us per step)\n",
               pass ? "with the per-step gather to the root" : "outputs left on the shards", steps, per_dev * n_dev, el,
               (double)per_dev * n_dev * steps / el, 1e6 * el / steps);
    }
    const float *obs, *rew;
    const uint8_t* done;
    int64_t first, count;
    CHECK(mvrl_group_root_views(g, n_dev - 1, &obs, &rew, &done, &first, &count));
    printf("last shard: global envs [%ld, %ld), rows at %p on the root device (reward plane: %s)\n", (long)first, (long)(first + count),
           (const void*)obs, rew ? "yes" : "none - identically 0");
    mvrl_group_destroy(g);
    return 0;
}
