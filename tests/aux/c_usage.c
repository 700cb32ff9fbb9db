/* Plain-C client of libamc.so: no Python, no C++ -- what a `ccall` / cgo / JNI binding sees. */
#include <math.h>
#include <stdio.h>
#include "amc.h"

int main(void)
{
    double sigma[2] = {0.1, 1.0}, weight[2] = {0.5, 0.5}, red[AMC_RED_HEADER + 2];
    amc_config cfg = {sizeof(amc_config), 0, 200000, 0, 200000, AMC_POTENTIAL_DOUBLE_WELL, 2, 2.0,
                      sigma, weight, 1, 1, 1, NULL};
    amc_handle *h = NULL;
    if (amc_create(&cfg, &h)) { fprintf(stderr, "amc_create: %s\n", amc_last_error()); return 1; }
    if (amc_init_uniform(h, -2.0, 2.0) || amc_sweep(h, 500) || amc_reduce(h, red)) {
        fprintf(stderr, "%s\n", amc_last_error());
        return 1;
    }
    printf("double_well mean_U %.6f mean_x2 %.6f acc0 %.5f acc1 %.5f\n", red[AMC_RED_SUM_E] / red[AMC_RED_COUNT],
           red[AMC_RED_SUM_XX] / red[AMC_RED_COUNT], red[4] / red[AMC_RED_COUNT], red[5] / red[AMC_RED_COUNT]);
    amc_destroy(h);

    /* the script-defined potential of the reference, as a C expression compiled for the GPU at run time */
    cfg.potential = AMC_POTENTIAL_CUSTOM;
    cfg.n_moves = 1;
    weight[0] = 1.0;
    sigma[0] = 0.5;
    cfg.per_chain_counters = 0;
    if (amc_create_custom(&cfg, "0.5*x*x + 0.25*x*x*x*x", &h)) { fprintf(stderr, "custom: %s\n", amc_last_error()); return 1; }
    if (amc_init_uniform(h, -2.0, 2.0) || amc_sweep(h, 500) || amc_reduce(h, red)) {
        fprintf(stderr, "%s\n", amc_last_error());
        return 1;
    }
    printf("custom mean_U %.6f mean_x %.6f\n", red[AMC_RED_SUM_E] / red[AMC_RED_COUNT], red[AMC_RED_SUM_X] / red[AMC_RED_COUNT]);
    amc_destroy(h);

    /* Particle{Float32}: the same calls, Float32 state on the device, double host buffers */
    cfg.potential = AMC_POTENTIAL_HARMONIC;
    cfg.state_dtype = AMC_DTYPE_F32;
    sigma[0] = 0.1;
    if (amc_create(&cfg, &h)) { fprintf(stderr, "f32: %s\n", amc_last_error()); return 1; }
    if (amc_init_uniform(h, -2.0, 2.0) || amc_sweep(h, 2000) || amc_reduce(h, red)) {
        fprintf(stderr, "%s\n", amc_last_error());
        return 1;
    }
    printf("f32 harmonic mean_U %.6f acc %.5f\n", red[AMC_RED_SUM_E] / red[AMC_RED_COUNT], red[4] / red[AMC_RED_COUNT]);
    amc_destroy(h);
    cfg.state_dtype = AMC_DTYPE_F64;

    /* errors come back as codes + message, never as exceptions */
    cfg.n_moves = 0;
    if (amc_create(&cfg, &h) != AMC_ERR_BAD_ARG) return 2;
    printf("error path: %s\n", amc_last_error());
    return 0;
}
