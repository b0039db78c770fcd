/* The reference's callable contract (src/dynamics.jl:55-60, src/costs.jl:1-15, src/constraints.jl:54-64) for BASELINE config 5's
 * model: synth32, nx = 32, nu = 8, x+ = x + h (A x + B u + 0.1 sin x), action box as 16 stage inequalities — the twin of
 * iterativelqr.jl_amd/models.py:synth32 (generated Model_synth32) and of oracle/models.cpp "synth32". Handed to
 * ilqr_compile_model as text: the Jacobians below are dense loops, the library finds the 1248 constant entries of the 1280 and
 * the 40 structurally non-zero Hessian entries of the 1344 by probing them on the host. `out` arrives zeroed. */
#define S32_N 32
#define S32_M 8
#define S32_H 0.05
static ILQR_MODEL_FN double s32_A(int i, int j) { return (i == j ? -1.0 : 0.0) + 0.3 * cos((double)((i + 1) + 2 * (j + 1))) / (double)S32_N; }
static ILQR_MODEL_FN double s32_B(int i, int j) { return sin((double)(3 * (i + 1) + (j + 1))) / sqrt((double)S32_N); }
ILQR_MODEL_FN void dynamics(double* y, const double* x, const double* u, const double* w) {
    for (int i = 0; i < S32_N; ++i) {
        double acc = 0.0;
        for (int j = 0; j < S32_N; ++j) acc += s32_A(i, j) * x[j];
        for (int j = 0; j < S32_M; ++j) acc += s32_B(i, j) * u[j];
        acc += 0.1 * sin(x[i]);
        y[i] = x[i] + S32_H * acc;
    }
}
ILQR_MODEL_FN void dynamics_jacobian_state(double* fx, const double* x, const double* u, const double* w) {   /* column-major n x n */
    for (int j = 0; j < S32_N; ++j)
        for (int i = 0; i < S32_N; ++i)
            fx[j * S32_N + i] = (i == j ? 1.0 : 0.0) + S32_H * (s32_A(i, j) + (i == j ? 0.1 * cos(x[i]) : 0.0));
}
ILQR_MODEL_FN void dynamics_jacobian_action(double* fu, const double* x, const double* u, const double* w) {  /* column-major n x m */
    for (int j = 0; j < S32_M; ++j)
        for (int i = 0; i < S32_N; ++i) fu[j * S32_N + i] = S32_H * s32_B(i, j);
}
ILQR_MODEL_FN void cost_stage(double* l, const double* x, const double* u, const double* w) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < S32_N; ++i) a += (x[i] - 0.5) * (x[i] - 0.5);
    for (int j = 0; j < S32_M; ++j) b += u[j] * u[j];
    l[0] = 0.1 * a + 0.01 * b;
}
ILQR_MODEL_FN void cost_stage_gradient_state(double* g, const double* x, const double* u, const double* w) { for (int i = 0; i < S32_N; ++i) g[i] = 0.2 * (x[i] - 0.5); }
ILQR_MODEL_FN void cost_stage_gradient_action(double* g, const double* x, const double* u, const double* w) { for (int j = 0; j < S32_M; ++j) g[j] = 0.02 * u[j]; }
ILQR_MODEL_FN void cost_stage_hessian_state_state(double* h, const double* x, const double* u, const double* w) { for (int i = 0; i < S32_N; ++i) h[i * S32_N + i] = 0.2; }
ILQR_MODEL_FN void cost_stage_hessian_action_action(double* h, const double* x, const double* u, const double* w) { for (int j = 0; j < S32_M; ++j) h[j * S32_M + j] = 0.02; }
ILQR_MODEL_FN void cost_stage_hessian_action_state(double* h, const double* x, const double* u, const double* w) { }
ILQR_MODEL_FN void cost_terminal(double* l, const double* x, const double* u, const double* w) {
    double a = 0.0;
    for (int i = 0; i < S32_N; ++i) a += (x[i] - 0.5) * (x[i] - 0.5);
    l[0] = 10.0 * a;
}
ILQR_MODEL_FN void cost_terminal_gradient_state(double* g, const double* x, const double* u, const double* w) { for (int i = 0; i < S32_N; ++i) g[i] = 20.0 * (x[i] - 0.5); }
ILQR_MODEL_FN void cost_terminal_hessian_state_state(double* h, const double* x, const double* u, const double* w) { for (int i = 0; i < S32_N; ++i) h[i * S32_N + i] = 20.0; }
/* stage: the action box -1 <= u <= 1 as 2 m inequalities */
ILQR_MODEL_FN void constraint_stage(double* c, const double* x, const double* u, const double* w) {
    for (int j = 0; j < S32_M; ++j) { c[j] = -1.0 - u[j]; c[S32_M + j] = u[j] - 1.0; }
}
ILQR_MODEL_FN void constraint_stage_jacobian_state(double* cx, const double* x, const double* u, const double* w) { }
ILQR_MODEL_FN void constraint_stage_jacobian_action(double* cu, const double* x, const double* u, const double* w) {   /* column-major 2m x m */
    for (int j = 0; j < S32_M; ++j) { cu[j * 2 * S32_M + j] = -1.0; cu[j * 2 * S32_M + S32_M + j] = 1.0; }
}
