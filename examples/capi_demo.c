/* A plain-C consumer of the C ABI (include/atacom_hip.h): no Python, no torch -- only the HIP runtime for the device
 * buffers the caller owns.  It is what a maintainer binding the library from another language would write first.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/capi_demo.c -Iinclude -I/opt/rocm/include -Lrl_on_manifold_amd \
 *       -latacom_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/rl_on_manifold_amd -Wl,-rpath,/opt/rocm/lib -o capi_demo
 *   ./capi_demo [batch] [steps]
 *
 * Steps `batch` IiwaAirHockey-7H environments (the reference's AirHockeyIiwaAtacom, iiwa_hit_atacom.py:10-40) with a
 * fixed action, then prints get_constraints_logs (atacom.py:207-216).  Exit code 0 iff every call succeeded and the
 * statistics are finite and small. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "atacom_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s failed: %s\n", #call, atacom_last_error());      \
            return 1;                                                            \
        }                                                                        \
    } while (0)
#define HIP(call)                                                                \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));           \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 1024;
    const int steps = argc > 2 ? atoi(argv[2]) : 120;
    atacom_config cfg;
    atacom_dims dims;
    CHECK(atacom_default_config(ATACOM_ENV_IIWA, &cfg));
    CHECK(atacom_get_dims(ATACOM_ENV_IIWA, &dims));
    cfg.batch = batch;
    cfg.auto_reset = 1;
    atacom_handle* h = NULL;
    CHECK(atacom_create(&cfg, 0, &h));

    float *d_act, *d_obs, *d_rew;
    uint8_t *d_abs, *d_last;
    HIP(hipMalloc((void**)&d_act, sizeof(float) * batch * dims.n_null));
    HIP(hipMalloc((void**)&d_obs, sizeof(float) * batch * dims.obs_dim));
    HIP(hipMalloc((void**)&d_rew, sizeof(float) * batch));
    HIP(hipMalloc((void**)&d_abs, batch));
    HIP(hipMalloc((void**)&d_last, batch));
    float* act = (float*)malloc(sizeof(float) * batch * dims.n_null);
    for (int i = 0; i < batch * dims.n_null; ++i) act[i] = 0.3f * (float)((i % 7) - 3) / 3.0f;
    HIP(hipMemcpy(d_act, act, sizeof(float) * batch * dims.n_null, hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    CHECK(atacom_reset(h, NULL, NULL, d_obs, stream));
    for (int t = 0; t < steps; ++t) CHECK(atacom_step(h, d_act, d_obs, d_rew, d_abs, d_last, stream));
    double stats[3];
    CHECK(atacom_get_stats(h, stats, 1, stream)); /* synchronises the stream */
    float obs0[32];
    HIP(hipMemcpy(obs0, d_obs, sizeof(float) * dims.obs_dim, hipMemcpyDeviceToHost));
    printf("%s: %d envs x %d steps  c_avg %.3e  c_max %.3e  c_dq_max %.3e  obs[0][6..8] = %.4f %.4f %.4f\n",
           atacom_version(), batch, steps, stats[0], stats[1], stats[2], obs0[6], obs0[7], obs0[8]);
    CHECK(atacom_destroy(h));
    hipFree(d_act); hipFree(d_obs); hipFree(d_rew); hipFree(d_abs); hipFree(d_last);
    free(act);
    return (isfinite(stats[0]) && isfinite(stats[1]) && stats[1] < 0.05 && stats[2] <= 1e-4) ? 0 : 2;
}
