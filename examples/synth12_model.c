/* The reference's callable contract (src/dynamics.jl:55-60, src/costs.jl:1-15, src/constraints.jl:54-64) for a model that takes
 * the LARGE path (nx = 12, nu = 5): x+ = x + h (A x + B u + 0.1 sin x + 0.02 x_i u_{i mod 5}), the twin of
 * iterativelqr.jl_amd/models.py:synth12 and oracle/models.cpp "synth12". Handed to ilqr_compile_model as text
 * (examples/synth12_compile.c reads this file; tests/test_gpu_parity.py feeds it through ctypes). `out` arrives zeroed. */
#define S12_N 12
#define S12_M 5
#define S12_H 0.05
static ILQR_MODEL_FN double s12_A(int i, int j) { return (i == j ? -1.0 : 0.0) + 0.3 * cos((double)((i + 1) + 2 * (j + 1))) / 12.0; }
static ILQR_MODEL_FN double s12_B(int i, int j) { return sin((double)(3 * (i + 1) + (j + 1))) / sqrt(12.0); }
ILQR_MODEL_FN void dynamics(double* y, const double* x, const double* u, const double* w) {
    for (int i = 0; i < S12_N; ++i) {
        double acc = 0.0;
        for (int j = 0; j < S12_N; ++j) acc += s12_A(i, j) * x[j];
        for (int j = 0; j < S12_M; ++j) acc += s12_B(i, j) * u[j];
        acc += 0.1 * sin(x[i]);
        acc += 0.02 * (x[i] * u[i % S12_M]);
        y[i] = x[i] + S12_H * acc;
    }
}
ILQR_MODEL_FN void dynamics_jacobian_state(double* fx, const double* x, const double* u, const double* w) {   /* column-major n x n */
    for (int j = 0; j < S12_N; ++j)
        for (int i = 0; i < S12_N; ++i)
            fx[j * S12_N + i] = (i == j ? 1.0 : 0.0) + S12_H * (s12_A(i, j) + (i == j ? 0.1 * cos(x[i]) + 0.02 * u[i % S12_M] : 0.0));
}
ILQR_MODEL_FN void dynamics_jacobian_action(double* fu, const double* x, const double* u, const double* w) {  /* column-major n x m */
    for (int j = 0; j < S12_M; ++j)
        for (int i = 0; i < S12_N; ++i)
            fu[j * S12_N + i] = S12_H * (s12_B(i, j) + (i % S12_M == j ? 0.02 * x[i] : 0.0));
}
ILQR_MODEL_FN void cost_stage(double* l, const double* x, const double* u, const double* w) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < S12_N; ++i) a += (x[i] - 0.5) * (x[i] - 0.5);
    for (int j = 0; j < S12_M; ++j) b += u[j] * u[j];
    l[0] = 0.1 * a + 0.01 * b;
}
ILQR_MODEL_FN void cost_stage_gradient_state(double* g, const double* x, const double* u, const double* w) { for (int i = 0; i < S12_N; ++i) g[i] = 0.2 * (x[i] - 0.5); }
ILQR_MODEL_FN void cost_stage_gradient_action(double* g, const double* x, const double* u, const double* w) { for (int j = 0; j < S12_M; ++j) g[j] = 0.02 * u[j]; }
ILQR_MODEL_FN void cost_stage_hessian_state_state(double* h, const double* x, const double* u, const double* w) { for (int i = 0; i < S12_N; ++i) h[i * S12_N + i] = 0.2; }
ILQR_MODEL_FN void cost_stage_hessian_action_action(double* h, const double* x, const double* u, const double* w) { for (int j = 0; j < S12_M; ++j) h[j * S12_M + j] = 0.02; }
ILQR_MODEL_FN void cost_stage_hessian_action_state(double* h, const double* x, const double* u, const double* w) { }
ILQR_MODEL_FN void cost_terminal(double* l, const double* x, const double* u, const double* w) {
    double a = 0.0;
    for (int i = 0; i < S12_N; ++i) a += (x[i] - 0.5) * (x[i] - 0.5);
    l[0] = 10.0 * a;
}
ILQR_MODEL_FN void cost_terminal_gradient_state(double* g, const double* x, const double* u, const double* w) { for (int i = 0; i < S12_N; ++i) g[i] = 20.0 * (x[i] - 0.5); }
ILQR_MODEL_FN void cost_terminal_hessian_state_state(double* h, const double* x, const double* u, const double* w) { for (int i = 0; i < S12_N; ++i) h[i * S12_N + i] = 20.0; }
/* stage: the action box -1 <= u <= 1 as 2 m inequalities; terminal: x_{1:3} = 0.1 */
ILQR_MODEL_FN void constraint_stage(double* c, const double* x, const double* u, const double* w) {
    for (int j = 0; j < S12_M; ++j) { c[j] = -1.0 - u[j]; c[S12_M + j] = u[j] - 1.0; }
}
ILQR_MODEL_FN void constraint_stage_jacobian_state(double* cx, const double* x, const double* u, const double* w) { }
ILQR_MODEL_FN void constraint_stage_jacobian_action(double* cu, const double* x, const double* u, const double* w) {   /* column-major 2m x m */
    for (int j = 0; j < S12_M; ++j) { cu[j * 2 * S12_M + j] = -1.0; cu[j * 2 * S12_M + S12_M + j] = 1.0; }
}
ILQR_MODEL_FN void constraint_terminal(double* c, const double* x, const double* u, const double* w) { for (int i = 0; i < 3; ++i) c[i] = x[i] - 0.1; }
ILQR_MODEL_FN void constraint_terminal_jacobian_state(double* cx, const double* x, const double* u, const double* w) { for (int i = 0; i < 3; ++i) cx[i * 3 + i] = 1.0; }
