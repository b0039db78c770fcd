/* Plain-C host that DEFINES a model through the C-ABI (no Python): examples/particle.jl of the reference
 * (x+ = [x1 + x2, x2 + u], l = 0.1 x.x + 0.1 u.u, terminal equality x - [1, 0], T = 11) handed over as C source of the
 * reference's in-place callables, compiled for gfx950 by the library (ilqr_compile_model), and solved for a small batch.
 *
 *   gcc -O2 -Iinclude examples/particle_compile.c -o particle_compile \
 *       -Literativelqr.jl_amd/lib -lilqr_hip -Wl,-rpath,$PWD/iterativelqr.jl_amd/lib -lm -lpthread
 *   ./particle_compile
 * It then runs the SAME problem with the library's built-in "particle" model on a second handle, from a second host thread,
 * concurrently on the same device (one handle per thread: the documented multi-handle pattern), and compares.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ilqr_hip.h"

static const char* PARTICLE_SOURCE =
    "ILQR_MODEL_FN void dynamics(double* y, const double* x, const double* u, const double* w) { y[0] = x[0] + x[1]; y[1] = x[1] + u[0]; }\n"
    "ILQR_MODEL_FN void dynamics_jacobian_state(double* fx, const double* x, const double* u, const double* w) { fx[0] = 1.0; fx[2] = 1.0; fx[3] = 1.0; }\n"
    "ILQR_MODEL_FN void dynamics_jacobian_action(double* fu, const double* x, const double* u, const double* w) { fu[1] = 1.0; }\n"
    "ILQR_MODEL_FN void cost_stage(double* l, const double* x, const double* u, const double* w) { l[0] = 0.1 * (x[0] * x[0] + x[1] * x[1]) + 0.1 * (u[0] * u[0]); }\n"
    "ILQR_MODEL_FN void cost_stage_gradient_state(double* g, const double* x, const double* u, const double* w) { g[0] = 0.2 * x[0]; g[1] = 0.2 * x[1]; }\n"
    "ILQR_MODEL_FN void cost_stage_gradient_action(double* g, const double* x, const double* u, const double* w) { g[0] = 0.2 * u[0]; }\n"
    "ILQR_MODEL_FN void cost_stage_hessian_state_state(double* h, const double* x, const double* u, const double* w) { h[0] = 0.2; h[3] = 0.2; }\n"
    "ILQR_MODEL_FN void cost_stage_hessian_action_action(double* h, const double* x, const double* u, const double* w) { h[0] = 0.2; }\n"
    "ILQR_MODEL_FN void cost_stage_hessian_action_state(double* h, const double* x, const double* u, const double* w) { }\n"
    "ILQR_MODEL_FN void cost_terminal(double* l, const double* x, const double* u, const double* w) { l[0] = 0.1 * (x[0] * x[0] + x[1] * x[1]); }\n"
    "ILQR_MODEL_FN void cost_terminal_gradient_state(double* g, const double* x, const double* u, const double* w) { g[0] = 0.2 * x[0]; g[1] = 0.2 * x[1]; }\n"
    "ILQR_MODEL_FN void cost_terminal_hessian_state_state(double* h, const double* x, const double* u, const double* w) { h[0] = 0.2; h[3] = 0.2; }\n"
    "ILQR_MODEL_FN void constraint_terminal(double* c, const double* x, const double* u, const double* w) { c[0] = x[0] - 1.0; c[1] = x[1]; }\n"
    "ILQR_MODEL_FN void constraint_terminal_jacobian_state(double* cx, const double* x, const double* u, const double* w) { cx[0] = 1.0; cx[3] = 1.0; }\n";

enum { T = 11, B = 6, NX = 2, NU = 1 };

typedef struct {
    const char* model;
    const char* library;
    const double* x1;
    const double* u;
    double x[B * T * NX];
    ilqr_stats st[B];
    int rc;
    char err[512];
} job;

static void* run(void* arg) {
    job* j = (job*)arg;
    ilqr_problem_desc d = {j->model, j->library, T, B, 0, 1};
    ilqr_handle* h = NULL;
    j->rc = ilqr_create(&d, &h);
    if (j->rc == ILQR_OK) {
        ilqr_options o;
        ilqr_default_options(&o);
        o.verbose = 0;
        ilqr_set_options(h, &o);
        j->rc = ilqr_initialize_rollout(h, j->x1, j->u);
        for (int rep = 0; rep < 20 && j->rc == ILQR_OK; ++rep) {          /* keep both threads busy on the device for a while */
            if (rep) { j->rc = ilqr_reset(h); if (j->rc == ILQR_OK) j->rc = ilqr_initialize_rollout(h, j->x1, j->u); }
            if (j->rc == ILQR_OK) j->rc = ilqr_solve(h);
        }
        if (j->rc == ILQR_OK) j->rc = ilqr_get_trajectory(h, j->x, NULL);
        if (j->rc == ILQR_OK) j->rc = ilqr_get_stats(h, j->st);
    }
    if (j->rc != ILQR_OK) snprintf(j->err, sizeof(j->err), "%s", ilqr_last_error());    /* thread-local message */
    ilqr_destroy(h);
    return NULL;
}

int main(void) {
    char name[128], path[1024];
    ilqr_model_source src = {"particle_c", NX, NU, 0, 0, 2, 0, 0, PARTICLE_SOURCE};
    if (ilqr_compile_model(&src, name, sizeof(name), path, sizeof(path)) != ILQR_OK) {
        fprintf(stderr, "ilqr_compile_model failed: %s\n", ilqr_last_error());
        return 1;
    }
    printf("compiled and registered '%s'\n  module %s\n", name, path);
    double x1[B * NX] = {0}, u[B * (T - 1) * NU];
    for (int i = 0; i < B * (T - 1); ++i) u[i] = 0.1 * sin(0.7 * (double)i + 0.3);
    static job a, b;
    a.model = name; a.library = path; a.x1 = x1; a.u = u;
    b.model = "particle"; b.library = NULL; b.x1 = x1; b.u = u;
    pthread_t ta, tb;
    pthread_create(&ta, NULL, run, &a);
    pthread_create(&tb, NULL, run, &b);
    pthread_join(ta, NULL);
    pthread_join(tb, NULL);
    if (a.rc != ILQR_OK || b.rc != ILQR_OK) {
        fprintf(stderr, "solve failed: user model rc=%d (%s), built-in rc=%d (%s)\n", a.rc, a.err, b.rc, b.err);
        return 1;
    }
    double worst = 0.0, goal = 0.0;
    for (int i = 0; i < B * T * NX; ++i) worst = fmax(worst, fabs(a.x[i] - b.x[i]));
    for (int i = 0; i < B; ++i) {
        goal = fmax(goal, fabs(a.x[(i * T + T - 1) * NX] - 1.0));
        goal = fmax(goal, fabs(a.x[(i * T + T - 1) * NX + 1]));
        if (a.st[i].iterations != b.st[i].iterations) { fprintf(stderr, "iteration counts differ on instance %d\n", i); return 1; }
    }
    printf("two handles on two host threads: max |x(user model) - x(built-in)| = %.3e, |x_T - goal|_inf = %.3e, iterations %d\n",
           worst, goal, a.st[0].iterations);
    if (worst > 1e-9 || goal > 5e-3) return 1;
    printf("user-defined model matches the built-in one\n");
    return 0;
}
