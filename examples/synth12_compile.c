/* Plain-C host that defines a LARGE-path model (nx = 12, nu = 5: zero-padded 16x16 MFMA tiles, compact Jacobian / Hessian
 * rows) through the C-ABI: the callables of examples/synth12_model.c are read as text, compiled for gfx950 by the library
 * (ilqr_compile_model) and solved for a small batch, once on one handle and once on a handle sharded over a device list.
 *
 *   gcc -O2 -Iinclude examples/synth12_compile.c -o synth12_compile -Literativelqr.jl_amd/lib -lilqr_hip \
 *       -Wl,-rpath,$PWD/iterativelqr.jl_amd/lib -lm
 *   ./synth12_compile examples/synth12_model.c
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ilqr_hip.h"

enum { T = 41, B = 8, NX = 12, NU = 5 };

static int solve(ilqr_handle* h, const double* x1, const double* u, double* x, ilqr_stats* st) {
    ilqr_options o;
    ilqr_default_options(&o);
    o.verbose = 0; o.max_iterations = 15; o.max_dual_updates = 3;       /* as tests/test_gpu_parity.py: parity, not convergence */
    int rc = ilqr_set_options(h, &o);
    if (rc == ILQR_OK) rc = ilqr_initialize_rollout(h, x1, u);
    if (rc == ILQR_OK) rc = ilqr_solve(h);
    if (rc == ILQR_OK) rc = ilqr_get_trajectory(h, x, NULL);
    if (rc == ILQR_OK) rc = ilqr_get_stats(h, st);
    return rc;
}

int main(int argc, char** argv) {
    const char* file = argc > 1 ? argv[1] : "examples/synth12_model.c";
    FILE* f = fopen(file, "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", file); return 1; }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* text = (char*)malloc((size_t)len + 1);
    if (fread(text, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read\n"); return 1; }
    text[len] = 0;
    fclose(f);
    char name[128], path[1024];
    ilqr_model_source src = {"synth12_c", NX, NU, 0, 2 * NU, 3, (1ull << (2 * NU)) - 1, 0, text};
    if (ilqr_compile_model(&src, name, sizeof(name), path, sizeof(path)) != ILQR_OK) {
        fprintf(stderr, "ilqr_compile_model failed: %s\n", ilqr_last_error());
        return 1;
    }
    printf("compiled and registered '%s'\n  module %s\n", name, path);
    static double x1[B * NX], u[B * (T - 1) * NU], xa[B * T * NX], xb[B * T * NX];
    for (int i = 0; i < B * NX; ++i) x1[i] = 0.5 * sin(1.3 * (double)i + 0.2);
    for (int i = 0; i < B * (T - 1) * NU; ++i) u[i] = 0.1 * cos(0.9 * (double)i);
    ilqr_stats sa[B], sb[B];
    ilqr_problem_desc d = {name, path, T, B, 0, 1};
    ilqr_handle *one = NULL, *many = NULL;
    const int32_t devices[2] = {0, 0};                                     /* two ranges on the one GPU of the box */
    if (ilqr_create(&d, &one) != ILQR_OK || ilqr_create_sharded(&d, devices, 2, &many) != ILQR_OK ||
        solve(one, x1, u, xa, sa) != ILQR_OK || solve(many, x1, u, xb, sb) != ILQR_OK) {
        fprintf(stderr, "failed: %s\n", ilqr_last_error());
        return 1;
    }
    double worst = 0.0, box = 0.0;
    for (int i = 0; i < B * T * NX; ++i) worst = fmax(worst, fabs(xa[i] - xb[i]));
    for (int b = 0; b < B; ++b) {
        if (sa[b].iterations != sb[b].iterations) { fprintf(stderr, "iteration counts differ on instance %d\n", b); return 1; }
        box = fmax(box, sa[b].max_violation);
    }
    printf("one handle vs handle sharded over {0, 0}: max |dx| = %.3e; iterations of instance 0: %d; worst max_violation %.3e\n",
           worst, sa[0].iterations, box);
    ilqr_destroy(one);
    ilqr_destroy(many);
    free(text);
    return worst == 0.0 ? 0 : 1;
}
