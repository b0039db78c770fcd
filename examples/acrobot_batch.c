/* Plain-C caller of the C-ABI (include/ilqr_hip.h): what examples/acrobot.jl of the reference does
 * (T = 101, x1 = 0, ū ~ N(0,1), terminal goal [π,0,0,0]) for a batch of instances on one MI355X.
 *
 *   gcc -O2 -Iinclude examples/acrobot_batch.c -o acrobot_batch \
 *       -Literativelqr.jl_amd/lib -lilqr_hip -Wl,-rpath,$PWD/iterativelqr.jl_amd/lib -lm
 *   ./acrobot_batch 1024
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ilqr_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != ILQR_OK) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ilqr_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static double gauss(uint64_t* s) {   /* splitmix64 + Box-Muller */
    double u[2];
    for (int i = 0; i < 2; ++i) {
        uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        u[i] = ((double)(z >> 11) + 0.5) / 9007199254740992.0;
    }
    return sqrt(-2.0 * log(u[0])) * cos(6.283185307179586 * u[1]);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = 101, nx = 4, nu = 1;
    ilqr_problem_desc desc = {"acrobot", NULL, T, B, 0, 1};
    ilqr_handle* h = NULL;
    CHECK(ilqr_create(&desc, &h));
    ilqr_options opt;
    CHECK(ilqr_default_options(&opt));
    opt.verbose = 0;
    CHECK(ilqr_set_options(h, &opt));

    double* x1 = calloc((size_t)B * nx, sizeof(double));
    double* ub = malloc((size_t)B * (T - 1) * nu * sizeof(double));
    uint64_t seed = 20240607;
    for (size_t i = 0; i < (size_t)B * (T - 1) * nu; ++i) ub[i] = gauss(&seed);

    CHECK(ilqr_initialize_rollout(h, x1, ub));      /* x̄ = rollout(dynamics, x1, ū); initialize_*! */
    CHECK(ilqr_solve(h));                           /* solve!(solver) */
    CHECK(ilqr_synchronize(h));

    double* x = malloc((size_t)B * T * nx * sizeof(double));
    double* u = malloc((size_t)B * (T - 1) * nu * sizeof(double));
    ilqr_stats* st = malloc((size_t)B * sizeof(ilqr_stats));
    CHECK(ilqr_get_trajectory(h, x, u));            /* get_trajectory(solver) */
    CHECK(ilqr_get_stats(h, st));
    double ms = 0.0;
    int32_t launches = 0;
    CHECK(ilqr_timing_get(h, &ms, &launches));

    int ok = 0;
    long iters = 0;
    const double goal[4] = {3.14159265358979323846, 0.0, 0.0, 0.0};
    for (int b = 0; b < B; ++b) {
        double e = 0.0;
        for (int i = 0; i < nx; ++i) e = fmax(e, fabs(x[((size_t)b * T + (T - 1)) * nx + i] - goal[i]));
        ok += e < opt.constraint_tolerance;         /* test/acrobot.jl:114 */
        iters += st[b].iterations;
    }
    printf("acrobot T=%d B=%d: %d/%d reached the goal, mean inner iterations %.1f, solve kernel %.2f ms\n",
           T, B, ok, B, (double)iters / B, ms);
    CHECK(ilqr_destroy(h));
    free(x1); free(ub); free(x); free(u); free(st);
    return ok >= (int)(0.99 * B) ? 0 : 2;
}
