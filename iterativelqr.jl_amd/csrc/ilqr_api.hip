// C-ABI implementation (include/ilqr_hip.h): handle management, HBM workspace,
// kernel launches on a private HIP stream, host<->device accessors.
// No CPU fallback: without a HIP device ilqr_create fails loudly.
#include <dlfcn.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "ilqr_device.hpp"

namespace ilqr {
// fresh-solver defaults after the workspace has been zero-filled:
// objective = Inf, step_size = 1 (src/data/solver.jl:37-39), ρ = 1, a = 1
// (src/augmented_lagrangian.jl:17-22).
__global__ void defaults_kernel(KArgs a) {
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    for (int i = threadIdx.x; i < L.C; i += blockDim.x) { g[L.rho + i] = 1.0; g[L.act + i] = 1.0; }
    if (threadIdx.x == 0) {
        g[L.scal + S_OBJECTIVE] = __builtin_huge_val();
        g[L.scal + S_STEP_SIZE] = 1.0;
    }
}

// ilqr_reset of an HBM-resident (large) model in ONE pass: zero the instance block except [keep_lo, keep_hi) — the full
// Jacobian / Hessian / value mirrors, rewritten or zeroed on demand — with 16-byte coalesced stores, then the fresh-solver
// defaults. (Two pitched hipMemset2DAsync calls took 0.43 ms per reset for 512 synth32 instances, a tenth of a BASELINE step.)
__global__ __launch_bounds__(256) void reset_large_kernel(KArgs a, int keep_lo, int keep_hi) {
    const int b = blockIdx.x / 8, part = blockIdx.x % 8;
    if (b >= a.B) return;
    const Layout& L = a.L;
    double2* g = reinterpret_cast<double2*>(a.ws + (size_t)b * (size_t)L.stride);
    const int lo2 = keep_lo / 2, hi2 = (keep_hi + 1) / 2, n2 = L.stride / 2;          // offsets are even (pad2), the stride a multiple of 16
    const int Cp = pad2(L.C);
    // the blocks of an instance run in any order: the ranges that get defaults (rho, act, scalars) are left to part 0 alone
    auto special = [&](int i) {
        const int d = 2 * i;
        return (d >= L.rho && d < L.rho + Cp) || (d >= L.act && d < L.act + Cp) || (d >= L.scal && d < L.scal + S_COUNT);
    };
    for (int i = part * 256 + threadIdx.x; i < n2; i += 8 * 256)
        if ((i < lo2 || i >= hi2) && !special(i)) g[i] = double2{0.0, 0.0};
    if (part == 0) {
        double* gd = a.ws + (size_t)b * (size_t)L.stride;
        for (int i = threadIdx.x; i < Cp; i += blockDim.x) { gd[L.rho + i] = i < L.C ? 1.0 : 0.0; gd[L.act + i] = i < L.C ? 1.0 : 0.0; }
        for (int i = threadIdx.x; i < S_COUNT; i += blockDim.x)
            gd[L.scal + i] = i == S_OBJECTIVE ? __builtin_huge_val() : (i == S_STEP_SIZE ? 1.0 : 0.0);
    }
}

}  // namespace ilqr

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ILQR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

std::vector<const ilqr_model_vtable*>& registry() {
    static std::vector<const ilqr_model_vtable*> r;
    return r;
}

const ilqr_model_vtable* find_model(const char* name) {
    for (auto* vt : registry())
        if (!std::strcmp(vt->name, name)) return vt;
    return nullptr;
}

struct BufferDesc { const char* name; int offset; int len; };

}  // namespace

struct ilqr_handle {
    const ilqr_model_vtable* vt;
    ilqr::Layout L;
    int B, device, constrained;
    ilqr_options opt;
    double* ws;
    size_t ws_bytes, lds_bytes;
    hipStream_t stream;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing;
    double* d_x1;
    double* d_u;   // staging for host-pointer initialize_rollout
    double* trace;
    int trace_cap;
    int handover;         // straggler hand-over of the packed kernel: -1 auto (by head count), 0 off, k > 1 = instances entering outer iteration k
    int handover_live;    // head-count rule: survivors of the batch at which they all leave (-1 auto, 0 off)
    int* done_counter;    // device counter of finished instances for that rule
    int* pool;            // one-wave form of the packed kernel: queue of handed-over instances its workgroups finish themselves (POOL_Q + B ints)
    int handover_mark;    // rejected line-search trials above the batch's mean at which an instance is marked a straggler (-1 auto, 0 never)
    int* cu_slots = nullptr;   // latency kernel: critical waves per SIMD (ilqr::KArgs::cu_slots), CU_SLOT_CUS x 4 counters
    bool pool_valid = false;   // the queue words hold the counts of THIS handle's last solve (false after a solve that did not use the queue)
    int variant;          // 0 auto, 1 latency kernel (all-LDS, 2 waves per instance; large models: four waves per instance), 2 throughput kernel (slim), 3 packed kernel (4 instances per wave, no LDS), 4 one wave per instance of a large model with nx, nu <= 16
    bool lds_fits;        // the LDS-resident kernels can hold this horizon (otherwise only the packed kernel runs it)
    int num_simds;
    double* qv;           // optional action-value buffers Qx, Qu, Qxx, Quu, Qux (allocated on first use by a getter)
    ilqr::QLayout QL;
    // large (HBM-resident) models: the kernels stream a COMPACT form of the Jacobians and Hessians (Layout::fv, ::hc, see
    // ilqr_device_large.hpp); the reference's full jacobian_* / hessian_* arrays are a mirror written on demand
    bool full_stale;      // the full fx, fu, gxx, guu, gux arrays do not reflect the compact form (a getter materialises them first)
    bool P_dirty;         // P, p hold pre-reset values (only the backward-pass STAGE kernel writes them): zeroed lazily
    std::vector<BufferDesc> buffers;
    std::vector<BufferDesc> qbuffers;
    // ilqr_create_sharded: this handle owns no device memory itself but one sub-handle per entry of the device list, each with
    // its own contiguous range of instances [lo[i], lo[i] + shards[i]->B), workspace and stream on its device
    std::vector<ilqr_handle*> shards;
    std::vector<int> lo;
    // ilqr_set_stage_selectors: the one-hot selector table [T][n_sel] of a lowered problem (distinct per-step objects); it occupies
    // the LAST n_sel parameter columns of every instance, the caller's ilqr_set_parameters fills the first nw - n_sel
    std::vector<double> sel;
    int n_sel = 0;
};

namespace {

ilqr::KArgs make_args(const ilqr_handle* h) {
    ilqr::KArgs a;
    a.ws = h->ws; a.L = h->L; a.B = h->B; a.constrained = h->constrained; a.stage = 0; a.opt = h->opt;
    a.x1 = nullptr; a.u_in = nullptr;
    a.trace = h->trace; a.trace_cap = h->trace_cap;
    a.qv = h->qv; a.QL = h->QL;
    a.stage_param = 0.0; a.stage_flag = 0;
    a.handover_outer = 0; a.resume = 0; a.handover_live = 0; a.done_counter = h->done_counter;
    a.pool = nullptr; a.pool_mark = 0; a.pool_lds = 0; a.pool_ctl = 0; a.pool_cu = 0; a.cu_slots = nullptr; a.cu_expect = 0;
    return a;
}

void fill_buffers(ilqr_handle* h) {
    const ilqr::Layout& L = h->L;
    const int T = L.T, N = T - 1, n = L.nx, m = L.nu;
    h->buffers = {
        {"nominal_states", L.xb, T * n}, {"nominal_actions", L.ub, N * m},
        {"states", L.x, T * n}, {"actions", L.u, N * m},
        {"jacobian_state", L.fx, N * n * n}, {"jacobian_action", L.fu, N * n * m},
        {"gradient_state", L.gx, T * n}, {"gradient_action", L.gu, N * m},
        {"hessian_state_state", L.gxx, T * n * n}, {"hessian_action_action", L.guu, N * m * m},
        {"hessian_action_state", L.gux, N * m * n},
        {"K", L.K, N * m * n}, {"k", L.k, N * m}, {"P", L.P, T * n * n}, {"p", L.p, T * n},
        {"gradient_state_lagrangian", L.Lx, N * n}, {"gradient_action_lagrangian", L.Lu, N * m},
        {"violations", L.c, L.C}, {"constraint_dual", L.lam, L.C},
        {"constraint_penalty", L.rho, L.C}, {"active_set", L.act, L.C},
        {"parameters", L.w, T * L.nw},
        {"_scalars", L.scal, ilqr::S_COUNT},
    };
    const ilqr::QLayout& Q = h->QL;
    h->qbuffers = {
        {"Qx", Q.Qx, N * n}, {"Qu", Q.Qu, N * m}, {"Qxx", Q.Qxx, N * n * n}, {"Quu", Q.Quu, N * m * m}, {"Qux", Q.Qux, N * m * n},
    };
}

const BufferDesc* find_qbuffer(const ilqr_handle* h, const char* name) {
    for (auto& b : h->qbuffers)
        if (!std::strcmp(b.name, name)) return &b;
    return nullptr;
}

// the deferred parts of ilqr_reset / of a solve on a large model (see ilqr_handle::full_stale / P_dirty)
int settle_reset(ilqr_handle* h) {
    const ilqr::Layout& L = h->L;
    const size_t pitch = (size_t)L.stride * 8;
    if (h->P_dirty) {
        HIP_TRY(hipMemset2DAsync((char*)h->ws + (size_t)L.P * 8, pitch, 0, (size_t)(L.scal - L.P) * 8, (size_t)h->B, h->stream));
        h->P_dirty = false;
    }
    return ILQR_OK;
}
// full jacobian_* / hessian_* arrays <- compact form (dir 0) or the reverse (dir 1)
int mirror(ilqr_handle* h, int dir) {
    if (!h->vt->launch_mirror) return ILQR_OK;
    ilqr::KArgs a = make_args(h);
    if (h->vt->launch_mirror(&a, dir, h->stream) != 0) return fail(ILQR_ERR_HIP, "mirror kernel launch failed");
    if (dir == 0) h->full_stale = false;
    return ILQR_OK;
}
bool is_mirrored(const ilqr_handle* h, const BufferDesc* bd) {
    return h->vt->launch_mirror != nullptr && bd->offset >= h->L.fx && bd->offset < h->L.P && bd->offset != h->L.ring;
}

const BufferDesc* find_buffer(const ilqr_handle* h, const char* name) {
    for (auto& b : h->buffers)
        if (!std::strcmp(b.name, name)) return &b;
    return nullptr;
}

int copy_out(ilqr_handle* h, const BufferDesc* bd, double* out) {
    if (bd->len == 0) return ILQR_OK;
    HIP_TRY(hipSetDevice(h->device));
    if (bd->offset >= h->L.P && bd->offset < h->L.scal) { const int rc = settle_reset(h); if (rc != ILQR_OK) return rc; }
    if (is_mirrored(h, bd) && h->full_stale) { const int rc = mirror(h, 0); if (rc != ILQR_OK) return rc; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy2D(out, (size_t)bd->len * 8, h->ws + bd->offset, (size_t)h->L.stride * 8,
                        (size_t)bd->len * 8, (size_t)h->B, hipMemcpyDeviceToHost));
    return ILQR_OK;
}

int copy_in(ilqr_handle* h, const BufferDesc* bd, const double* in) {
    if (bd->len == 0) return ILQR_OK;
    HIP_TRY(hipSetDevice(h->device));
    if (bd->offset >= h->L.P && bd->offset < h->L.scal) { const int rc = settle_reset(h); if (rc != ILQR_OK) return rc; }
    const bool mirrored = is_mirrored(h, bd);
    if (mirrored && h->full_stale) { const int rc = mirror(h, 0); if (rc != ILQR_OK) return rc; }     // the OTHER full arrays must be current before the gather below
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy2D(h->ws + bd->offset, (size_t)h->L.stride * 8, in, (size_t)bd->len * 8,
                        (size_t)bd->len * 8, (size_t)h->B, hipMemcpyHostToDevice));
    // any direct write may break states == nominal: drop the shortcut flag
    std::vector<double> zeros(h->B, 0.0);
    HIP_TRY(hipMemcpy2D(h->ws + h->L.scal + ilqr::S_STATES_EQ_NOMINAL, (size_t)h->L.stride * 8, zeros.data(), 8, 8,
                        (size_t)h->B, hipMemcpyHostToDevice));
    if (mirrored) {
        if (bd->offset == h->L.fx || bd->offset == h->L.fu) {       // host-written Jacobians count as evaluated
            std::vector<double> ones(h->B, 1.0);
            HIP_TRY(hipMemcpy2D(h->ws + h->L.scal + ilqr::S_JAC_VALID, (size_t)h->L.stride * 8, ones.data(), 8, 8, (size_t)h->B, hipMemcpyHostToDevice));
        }
        const int rc = mirror(h, 1);
        if (rc != ILQR_OK) return rc;
        // The kernels work on the compact rows: what the host wrote at a CONSTANT Jacobian position or outside the Hessian pattern
        // is not representable there and is dropped by the gather. The full arrays count as stale from here on, so that a getter
        // shows what the kernels will use (the constant, a structural zero), not what was written.
        h->full_stale = true;
    }
    return ILQR_OK;
}

// Sharded handles: run f(sub-handle, first instance of its range) for every shard. Launch-type calls are asynchronous and go
// one after the other from the calling thread; calls that move data (blocking copies) run on one host thread per device.
template <class F>
int each_shard(ilqr_handle* h, F f, bool threaded = false) {
    const size_t G = h->shards.size();
    if (!threaded || G == 1) {
        for (size_t i = 0; i < G; ++i) { const int rc = f(h->shards[i], (size_t)h->lo[i]); if (rc != ILQR_OK) return rc; }
        return ILQR_OK;
    }
    std::vector<int> rcs(G, ILQR_OK);
    std::vector<std::string> msgs(G);
    std::vector<std::thread> th;
    for (size_t i = 0; i < G; ++i)
        th.emplace_back([&, i] { rcs[i] = f(h->shards[i], (size_t)h->lo[i]); if (rcs[i] != ILQR_OK) msgs[i] = g_err; });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < G; ++i)
        if (rcs[i] != ILQR_OK) return fail(rcs[i], "device " + std::to_string(h->shards[i]->device) + ": " + msgs[i]);
    return ILQR_OK;
}
#define SHARDED(h) ((h) && !(h)->shards.empty())

// test hook (ilqr_device_math): the scalar routines of ilqr_math.hpp as the device executes them
__global__ void device_math_kernel(int which, const double* x, double* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = 0.0, s, c;
    switch (which) {
        case 0: r = ilqr::recip_fast(v); break;
        case 1: r = ilqr::rsqrt_fast(v); break;
        case 2: ilqr::sqrt_rsqrt_fast(v, r, s); break;
        case 3: ilqr::sincos_fast(v, r, c); break;
        default: ilqr::sincos_fast(v, s, r); break;
    }
    y[i] = r;
}

}  // namespace

extern "C" {

const char* ilqr_last_error(void) { return g_err.c_str(); }

int ilqr_device_math(const char* fn, const double* x, double* y, int32_t n) {
    static const char* names[] = {"recip_fast", "rsqrt_fast", "sqrt_fast", "sin_fast", "cos_fast"};
    int which = -1;
    for (int i = 0; i < 5; ++i)
        if (fn && !std::strcmp(fn, names[i])) which = i;
    if (which < 0 || !x || !y || n < 0) return fail(ILQR_ERR_INVALID, "ilqr_device_math: unknown function or null argument");
    if (ilqr_device_count() < 1) return fail(ILQR_ERR_NO_DEVICE, "no HIP device");
    if (n == 0) return ILQR_OK;
    double *dx = nullptr, *dy = nullptr;
    hipError_t e;
    auto bail = [&](hipError_t err, const char* what) { if (dx) (void)hipFree(dx); if (dy) (void)hipFree(dy); return fail(ILQR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(err)); };
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);                       // the caller's current device is put back below
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev_dev};
    if ((e = hipSetDevice(0)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipMalloc(&dx, (size_t)n * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(&dy, (size_t)n * 8)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpy(dx, x, (size_t)n * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy");
    hipLaunchKernelGGL(device_math_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, which, dx, dy, n);
    if ((e = hipGetLastError()) != hipSuccess) return bail(e, "launch");
    if ((e = hipMemcpy(y, dy, (size_t)n * 8, hipMemcpyDeviceToHost)) != hipSuccess) return bail(e, "hipMemcpy");
    (void)hipFree(dx); (void)hipFree(dy);
    return ILQR_OK;
}

// the synthetic inputs as a pure function of (seed, instance, timestep, component)
int ilqr_synthetic_inputs(const char* model, int32_t T, uint64_t seed, int64_t first, int32_t B, double* x1, double* ub) {
    if (!model || !x1 || !ub || T < 2 || B < 0 || first < 0) return fail(ILQR_ERR_INVALID, "ilqr_synthetic_inputs: bad argument");
    struct Kind { const char* name; int n, m, kind; };
    static const Kind kinds[] = {{"particle", 2, 1, 0}, {"acrobot", 4, 1, 1}, {"car", 3, 2, 2}, {"car_goal", 3, 2, 2}, {"car_obs", 3, 2, 2},
                                 {"synth32", 32, 8, 3}, {"synth12", 12, 5, 4}};
    const Kind* k = nullptr;
    for (const Kind& q : kinds) if (!std::strcmp(q.name, model)) k = &q;
    if (!k) return fail(ILQR_ERR_MODEL, std::string("ilqr_synthetic_inputs: no workload for model '") + model + "'");
    auto mix = [](uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    auto unif = [](uint64_t h) { return ((double)(h >> 11) + 0.5) / 9007199254740992.0; };          // (0, 1)
    auto key = [&](int64_t b, int t, int j) { return seed ^ (uint64_t)(b * (1ll << 20) + (int64_t)t * 16 + j); };
    auto U = [&](int64_t b, int t, int j) { return unif(mix(key(b, t, j))); };
    auto Z = [&](int64_t b, int t, int j) { const uint64_t h1 = mix(key(b, t, j)), h2 = mix(h1); return std::sqrt(-2.0 * std::log(unif(h1))) * std::cos(6.283185307179586 * unif(h2)); };
    const int n = k->n, m = k->m, N = T - 1;
    for (int i = 0; i < B; ++i) {
        const int64_t b = first + i;
        double* x = x1 + (size_t)i * n;
        double* u = ub + (size_t)i * N * m;
        for (int j = 0; j < n; ++j) x[j] = 0.0;
        for (int e = 0; e < N * m; ++e) u[e] = 0.0;
        switch (k->kind) {
            case 0: for (int t = 0; t < N; ++t) u[t] = 0.1 * Z(b, t, 0); break;
            case 1: for (int t = 0; t < N; ++t) u[t] = Z(b, t, 0); break;
            case 2: {
                const double sc = b > 0 ? 1.0 + 0.5 * (2.0 * U(b, 0, 8) - 1.0) : 1.0;
                for (int t = 0; t < N; ++t) { u[t * 2] = 1.0e-2 * sc; u[t * 2 + 1] = 1.0e-3 * sc; }
                if (b > 0) { x[0] = 0.05 * Z(b, 0, 9); x[1] = 0.05 * Z(b, 0, 10); }
                break;
            }
            case 3: break;                                                  // x1 below; u = 0
            default:
                for (int t = 0; t < N; ++t) for (int j = 0; j < m; ++j) u[t * m + j] = 0.1 * Z(b, t, j);
        }
        if (k->kind >= 3) for (int j = 0; j < n; ++j) x[j] = 0.5 * Z(b, T + j / 16, j % 16);     // component j of x1: key slot (T + j / 16, j % 16), behind the horizon's
    }
    return ILQR_OK;
}

int ilqr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ilqr_default_options(ilqr_options* o) {
    if (!o) return fail(ILQR_ERR_INVALID, "null options");
    // src/options.jl:1-15
    o->line_search = 1; o->max_iterations = 100; o->max_dual_updates = 10;
    o->min_step_size = 1.0e-5; o->objective_tolerance = 1.0e-3; o->lagrangian_gradient_tolerance = 1.0e-3;
    o->constraint_tolerance = 5.0e-3; o->constraint_norm = INFINITY; o->initial_constraint_penalty = 1.0;
    o->scaling_penalty = 10.0; o->max_penalty = 1.0e8; o->reset_cache = 0; o->verbose = 1;
    return ILQR_OK;
}

int ilqr_register_model(const ilqr_model_vtable* vt) {
    if (!vt) return ILQR_ERR_INVALID;
    // a module built against other headers would be handed a KArgs / Layout it misreads: refuse it
    if (vt->abi_version != ILQR_MODEL_ABI_VERSION || vt->kargs_bytes != (int)sizeof(ilqr::KArgs))
        return fail(ILQR_ERR_MODEL, "model module was built against a different library version (ABI mismatch): rebuild it");
    if (!vt->name) return ILQR_ERR_INVALID;
    auto& r = registry();
    for (auto& e : r)
        if (!std::strcmp(e->name, vt->name)) { e = vt; return ILQR_OK; }
    r.push_back(vt);
    return ILQR_OK;
}
int ilqr_model_count(void) { return (int)registry().size(); }

// ---- structure of a large model's callables, found by running them on the HOST (ilqr_compile_model)
// The reference gets sparse, partly constant Jacobians and Hessians for free: Symbolics differentiates the user's function and
// emits code for the non-trivial entries only (src/dynamics.jl:16-34, src/costs.jl:17-44). A C host hands over opaque callables;
// what the large path streams per timestep (ilqr_device_large.hpp: state-dependent Jacobian entries, structurally non-zero
// Hessian entries) is therefore found by PROBING: the source is compiled once more with the host compiler, every Jacobian /
// Hessian / constraint-Jacobian callable is evaluated at three pseudo-random points, and an entry that comes out bitwise equal
// at all of them is a constant (zero or not). A Hessian entry counts as structurally non-zero if a cost Hessian or a Gauss-Newton
// term cxᵀ Iρ cx (src/gradients.jl:63-79) can put something there. Failing any step (no host compiler, source that does not
// compile as host C++) is not an error: the dense tables are used, as before.
struct ModelStructure {
    bool found = false;
    std::vector<double> fxc, fuc;          // constant Jacobian entries (0 where state-dependent)
    std::vector<int> jac_var;              // indices into [fx | fu] of the state-dependent ones
    std::vector<int> hess_idx, tile_start; // compact Hessian row: [gxx by 16x16 tile | guu | gux], indices inside each matrix
    int nxx = 0, nuu = 0, nux = 0;
    std::string note;
};

static int run_child(const std::vector<std::string>& args, const std::string& log) {
    std::vector<const char*> argv;
    for (auto& a : args) argv.push_back(a.c_str());
    argv.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    extern char** environ;
    const int sp = posix_spawn(&pid, args[0].c_str(), &fa, nullptr, const_cast<char* const*>(argv.data()), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (sp != 0) return -1;
    int status = 0;
    if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status)) return -1;
    return WEXITSTATUS(status);
}

// selector columns of a lowered model (ilqr_compile_model_stages): the probe must visit every kind of every category
struct ProbeHints { int sel[3] = {-1, -1, -1}; int kinds[3] = {0, 0, 0}; };

static ModelStructure probe_model_structure(const ilqr_model_source* src, const std::string& dir, const std::string& tag, const ProbeHints& hints) {
    ModelStructure ms;
    const int n = src->nx, m = src->nu, nw = src->nw, ncs = src->nc_stage, nct = src->nc_term;
    if (src->flags & ILQR_MODEL_DENSE_TABLES) { ms.note = "disabled for this model (ILQR_MODEL_DENSE_TABLES)"; return ms; }
    if (std::getenv("ILQR_NO_STRUCTURE_PROBE")) { ms.note = "disabled by ILQR_NO_STRUCTURE_PROBE"; return ms; }
    std::string cxx;
    const char* cand[] = {std::getenv("ILQR_HOSTCXX"), "/usr/bin/g++", "/usr/bin/c++", "/usr/bin/clang++", "/opt/rocm/lib/llvm/bin/clang++"};
    for (const char* c : cand)
        if (c && access(c, X_OK) == 0) { cxx = c; break; }
    if (cxx.empty()) { ms.note = "no host C++ compiler (set ILQR_HOSTCXX)"; return ms; }
    static std::atomic<unsigned> probe_seq{0};          // two threads of one process compiling the same model must not share temp files
    const std::string pid = std::to_string((long)getpid()) + "_" + std::to_string(probe_seq.fetch_add(1));
    const std::string cpp = dir + "/probe_" + tag + "." + pid + ".cpp", so = dir + "/probe_" + tag + "." + pid + ".so", log = dir + "/probe_" + tag + "." + pid + ".log";
    FILE* f = std::fopen(cpp.c_str(), "w");
    if (!f) { ms.note = "cannot write " + cpp; return ms; }
    std::fprintf(f, "// GENERATED by ilqr_compile_model: the user's callables compiled for the host, to find constant / zero entries\n"
                    "#include <cmath>\n#include <math.h>\n#define ILQR_MODEL_FN inline\nnamespace ilqr_user {\nusing namespace std;\n%s\n}\n"
                    "extern \"C\" void ilqr_probe(int which, double* o, const double* x, const double* u, const double* w) {\n    switch (which) {\n"
                    "        case 0: ilqr_user::dynamics_jacobian_state(o, x, u, w); break;\n        case 1: ilqr_user::dynamics_jacobian_action(o, x, u, w); break;\n"
                    "        case 2: ilqr_user::cost_stage_hessian_state_state(o, x, u, w); break;\n        case 3: ilqr_user::cost_stage_hessian_action_action(o, x, u, w); break;\n"
                    "        case 4: ilqr_user::cost_stage_hessian_action_state(o, x, u, w); break;\n        case 5: ilqr_user::cost_terminal_hessian_state_state(o, x, u, w); break;\n",
                 src->source);
    if (ncs > 0) std::fprintf(f, "        case 6: ilqr_user::constraint_stage_jacobian_state(o, x, u, w); break;\n        case 7: ilqr_user::constraint_stage_jacobian_action(o, x, u, w); break;\n");
    if (nct > 0) std::fprintf(f, "        case 8: ilqr_user::constraint_terminal_jacobian_state(o, x, u, w); break;\n");
    std::fprintf(f, "        default: break;\n    }\n}\n");
    std::fclose(f);
    const int rc = run_child({cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-w", cpp, "-o", so, "-lm"}, log);
    auto cleanup = [&]() { std::remove(cpp.c_str()); std::remove(so.c_str()); std::remove(log.c_str()); };
    if (rc != 0) { ms.note = "the source does not compile for the host (" + cxx + ", see " + log + ")"; std::remove(cpp.c_str()); return ms; }
    void* hl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    typedef void (*probe_fn)(int, double*, const double*, const double*, const double*);
    probe_fn pf = hl ? (probe_fn)dlsym(hl, "ilqr_probe") : nullptr;
    if (!pf) { ms.note = "cannot load the host build of the callables"; if (hl) dlclose(hl); cleanup(); return ms; }
    // 48 points: magnitudes 1e-3, 0.03, 1, 30, 1e3 per point (every argument of a point at the same scale, so that small and large
    // regimes are each seen whole) and mixed scales per component on the rest, both signs; x, u and w alike. What this cannot
    // see is a piecewise callable whose pieces have the same derivative on all of them: see ilqr_hip.h (ILQR_MODEL_DENSE_TABLES).
    const int P = 48;
    const int sizes[9] = {n * n, n * m, n * n, m * m, m * n, n * n, ncs * n, ncs * m, nct * n};
    std::vector<std::vector<double>> val[9];
    unsigned long long seed = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (double)(seed >> 11) / 9007199254740992.0; };
    const double scales[5] = {1.0e-3, 3.0e-2, 1.0, 30.0, 1.0e3};
    for (int p = 0; p < P; ++p) {
        std::vector<double> x(n), u(m), w(nw > 0 ? nw : 1);
        auto draw = [&]() { const double sc = p < 30 ? scales[p % 5] : scales[(int)(uni() * 5.0) % 5]; return (uni() < 0.5 ? -1.0 : 1.0) * sc * (0.35 + 1.3 * uni()); };
        for (auto& v : x) v = draw();
        for (auto& v : u) v = draw();
        for (auto& v : w) v = draw();
        for (int c = 0; c < 3; ++c)                                  // one-hot selectors: every kind of every category in turn
            for (int k = 0; k < hints.kinds[c]; ++k) w[hints.sel[c] + k] = (k == (p / (c == 0 ? 1 : (c == 1 ? 2 : 3))) % hints.kinds[c]) ? 1.0 : 0.0;
        for (int k = 0; k < 9; ++k) {
            std::vector<double> o((size_t)(sizes[k] > 0 ? sizes[k] : 1), 0.0);
            const bool have = k < 6 || (k < 8 ? ncs > 0 : nct > 0);
            if (have && sizes[k] > 0) pf(k, o.data(), x.data(), u.data(), w.data());       // (terminal objects see u of the stage size: ignored by them)
            val[k].push_back(o);
        }
    }
    dlclose(hl);
    cleanup();
    // a NaN / Inf compares bitwise equal to itself: such an entry is never a constant
    auto same = [&](int k, int e) { for (int p = 0; p < P; ++p) if (!std::isfinite(val[k][p][e]) || std::memcmp(&val[k][p][e], &val[k][0][e], 8) != 0) return false; return true; };
    auto nonzero = [&](int k, int e) { for (int p = 0; p < P; ++p) if (val[k][p][e] != 0.0 || val[k][p][e] != val[k][p][e]) return true; return false; };
    ms.fxc.assign(n * n, 0.0); ms.fuc.assign(n * m, 0.0);
    for (int e = 0; e < n * n; ++e) { if (same(0, e)) ms.fxc[e] = val[0][0][e]; else ms.jac_var.push_back(e); }
    for (int e = 0; e < n * m; ++e) { if (same(1, e)) ms.fuc[e] = val[1][0][e]; else ms.jac_var.push_back(n * n + e); }
    // Hessian pattern: cost Hessians (stage and terminal share the row) and the Gauss-Newton terms of every constraint row
    std::vector<char> pxx(n * n, 0), puu(m * m, 0), pux(m * n, 0);
    for (int e = 0; e < n * n; ++e) pxx[e] = nonzero(2, e) || nonzero(5, e);
    for (int e = 0; e < m * m; ++e) puu[e] = nonzero(3, e);
    for (int e = 0; e < m * n; ++e) pux[e] = nonzero(4, e);
    auto gauss_newton = [&](int kx, int ku, int nc) {
        for (int i = 0; i < nc; ++i) {
            std::vector<int> sx, su;
            for (int j = 0; j < n; ++j) if (nonzero(kx, j * nc + i)) sx.push_back(j);
            if (ku >= 0) for (int j = 0; j < m; ++j) if (nonzero(ku, j * nc + i)) su.push_back(j);
            for (int a : sx) for (int b : sx) pxx[a * n + b] = 1;
            for (int a : su) for (int b : su) puu[a * m + b] = 1;
            for (int a : sx) for (int b : su) pux[a * m + b] = 1;         // gux(i2, j) at j * m + i2: column j = state, row i2 = action
        }
    };
    if (ncs > 0) gauss_newton(6, 7, ncs);
    if (nct > 0) gauss_newton(8, -1, nct);
    const int TN = (n + 15) / 16;
    for (int tile = 0; tile < TN * TN; ++tile) {
        ms.tile_start.push_back((int)ms.hess_idx.size());
        for (int idx = 0; idx < n * n; ++idx) {
            const int col = idx / n, row = idx % n;
            if (pxx[idx] && (row / 16) * TN + col / 16 == tile) ms.hess_idx.push_back(idx);
        }
    }
    ms.tile_start.push_back((int)ms.hess_idx.size());
    ms.nxx = (int)ms.hess_idx.size();
    for (int e = 0; e < m * m; ++e) if (puu[e]) { ms.hess_idx.push_back(e); ms.nuu++; }
    for (int e = 0; e < m * n; ++e) if (pux[e]) { ms.hess_idx.push_back(e); ms.nux++; }
    ms.found = true;
    return ms;
}

// Dynamics / Cost / Constraint constructors for hosts without Python: C source of the reference's callables -> model module.
// What the probe found, kept beside the modules under the PRE-probe hash (source, dimensions, ABI, kernel headers + flags) and the
// probe hints: ilqr_compile_model[_rows,_stages] is called by every rank of a job and on every start of a host, and the probe —
// a host compilation in a child process, a dlopen, 48 points x 9 callables — used to run every time although the module it
// leads to was cached (advisor finding, round 5). Only a SUCCESSFUL probe is kept: where probing is impossible (no host
// compiler, switched off) the answer is cheap and must not be served on a box where it would work.
static ModelStructure probe_model_structure_cached(const ilqr_model_source* src, const std::string& dir, const std::string& tag, const ProbeHints& hints) {
    if ((src->flags & ILQR_MODEL_DENSE_TABLES) || std::getenv("ILQR_NO_STRUCTURE_PROBE")) return probe_model_structure(src, dir, tag, hints);
    unsigned long long hh = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(hints); ++i) { hh ^= ((const unsigned char*)&hints)[i]; hh *= 1099511628211ull; }
    char hb[24];
    std::snprintf(hb, sizeof(hb), "%08llx", hh & 0xffffffffull);
    const std::string path = dir + "/probe_" + tag + "_" + hb + ".bin";
    const int n = src->nx, m = src->nu;
    static const char magic[8] = {'I', 'L', 'Q', 'R', 'P', 'R', 'B', '1'};
    if (FILE* f = std::fopen(path.c_str(), "rb")) {
        ModelStructure ms;
        char mg[8];
        int hd[8];
        bool ok = std::fread(mg, 1, 8, f) == 8 && std::memcmp(mg, magic, 8) == 0 && std::fread(hd, sizeof(int), 8, f) == 8 &&
                  hd[0] == n && hd[1] == m && hd[5] >= 0 && hd[6] >= 0 && hd[7] >= 0 && hd[5] <= n * n + n * m && hd[6] <= n * n + m * m + m * n && hd[7] <= 64;
        if (ok) {
            ms.nxx = hd[2]; ms.nuu = hd[3]; ms.nux = hd[4];
            ms.fxc.resize((size_t)n * n); ms.fuc.resize((size_t)n * m); ms.jac_var.resize(hd[5]); ms.hess_idx.resize(hd[6]); ms.tile_start.resize(hd[7]);
            ok = std::fread(ms.fxc.data(), 8, ms.fxc.size(), f) == ms.fxc.size() && std::fread(ms.fuc.data(), 8, ms.fuc.size(), f) == ms.fuc.size() &&
                 std::fread(ms.jac_var.data(), sizeof(int), ms.jac_var.size(), f) == ms.jac_var.size() &&
                 std::fread(ms.hess_idx.data(), sizeof(int), ms.hess_idx.size(), f) == ms.hess_idx.size() &&
                 std::fread(ms.tile_start.data(), sizeof(int), ms.tile_start.size(), f) == ms.tile_start.size() &&
                 ms.nxx + ms.nuu + ms.nux == (int)ms.hess_idx.size();
        }
        std::fclose(f);
        if (ok) { ms.found = true; ms.note = "structure from " + path; return ms; }
    }
    ModelStructure ms = probe_model_structure(src, dir, tag, hints);
    if (ms.found && (int)ms.fxc.size() == n * n && (int)ms.fuc.size() == n * m) {
        const std::string tmp = path + "." + std::to_string((long)getpid());
        if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
            const int hd[8] = {n, m, ms.nxx, ms.nuu, ms.nux, (int)ms.jac_var.size(), (int)ms.hess_idx.size(), (int)ms.tile_start.size()};
            bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(hd, sizeof(int), 8, f) == 8 &&
                      std::fwrite(ms.fxc.data(), 8, ms.fxc.size(), f) == ms.fxc.size() && std::fwrite(ms.fuc.data(), 8, ms.fuc.size(), f) == ms.fuc.size() &&
                      std::fwrite(ms.jac_var.data(), sizeof(int), ms.jac_var.size(), f) == ms.jac_var.size() &&
                      std::fwrite(ms.hess_idx.data(), sizeof(int), ms.hess_idx.size(), f) == ms.hess_idx.size() &&
                      std::fwrite(ms.tile_start.data(), sizeof(int), ms.tile_start.size(), f) == ms.tile_start.size();
            ok = (std::fclose(f) == 0) && ok;
            if (ok) std::rename(tmp.c_str(), path.c_str()); else std::remove(tmp.c_str());
        }
    }
    return ms;
}

static int compile_model_impl(const ilqr_model_source* src, const uint64_t* ineq_stage_words, const uint64_t* ineq_term_words,
                              char* registered_name, size_t name_len, char* library_path, size_t path_len, const ProbeHints& hints = ProbeHints());
int ilqr_compile_model(const ilqr_model_source* src, char* registered_name, size_t name_len, char* library_path, size_t path_len) {
    if (src && (src->nc_stage > 64 || src->nc_term > 64))
        return fail(ILQR_ERR_INVALID, "ilqr_compile_model: at most 64 constraint rows per stage fit the 64-bit inequality masks — "
                                      "ilqr_compile_model_rows takes the masks as arrays of words");
    return compile_model_impl(src, nullptr, nullptr, registered_name, name_len, library_path, path_len);
}
int ilqr_compile_model_rows(const ilqr_model_source* src, const uint64_t* ineq_stage_words, const uint64_t* ineq_term_words,
                            char* registered_name, size_t name_len, char* library_path, size_t path_len) {
    if (src && ((src->nc_stage > 64 && !ineq_stage_words) || (src->nc_term > 64 && !ineq_term_words)))
        return fail(ILQR_ERR_INVALID, "ilqr_compile_model_rows: more than 64 rows need their inequality words");
    return compile_model_impl(src, ineq_stage_words, ineq_term_words, registered_name, name_len, library_path, path_len);
}
static int compile_model_impl(const ilqr_model_source* src, const uint64_t* ineq_stage_words, const uint64_t* ineq_term_words,
                              char* registered_name, size_t name_len, char* library_path, size_t path_len, const ProbeHints& hints) {
    if (!src || !src->name || !src->source || !registered_name || !library_path)
        return fail(ILQR_ERR_INVALID, "null argument");
    if (src->nx < 1 || src->nx > 64 || src->nu < 1 || src->nu > 16 || src->nw < 0 || src->nc_stage < 0 || src->nc_stage > ILQR_MAX_CONSTRAINT_ROWS ||
        src->nc_term < 0 || src->nc_term > ILQR_MAX_CONSTRAINT_ROWS)
        return fail(ILQR_ERR_INVALID, "ilqr_compile_model: 1 <= nx <= 64, 1 <= nu <= 16, at most 256 constraint rows per stage");
    // inequality rows as words of 64 (the struct's masks are word 0 when no array is given)
    const int nwords = std::max(1, (std::max(src->nc_stage, src->nc_term) + 63) / 64);
    std::vector<uint64_t> iw_s(nwords, 0), iw_t(nwords, 0);
    for (int k = 0; k < nwords; ++k) {
        iw_s[k] = ineq_stage_words ? (k < (src->nc_stage + 63) / 64 ? ineq_stage_words[k] : 0) : (k == 0 ? src->ineq_stage : 0);
        iw_t[k] = ineq_term_words ? (k < (src->nc_term + 63) / 64 ? ineq_term_words[k] : 0) : (k == 0 ? src->ineq_term : 0);
    }
    for (const char* c = src->name; *c; ++c)
        if (!((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '_'))
            return fail(ILQR_ERR_INVALID, "model name must be a C identifier");
    // where this library lives: <pkg>/lib/libilqr_hip.so, kernels in <pkg>/csrc (ILQR_CSRC_DIR overrides)
    Dl_info di;
    if (!dladdr((const void*)&ilqr_compile_model, &di) || !di.dli_fname) return fail(ILQR_ERR_MODEL, "cannot locate libilqr_hip.so");
    std::string libdir(di.dli_fname);
    libdir = libdir.substr(0, libdir.find_last_of('/'));
    const char* env_csrc = std::getenv("ILQR_CSRC_DIR");
    const std::string csrc = env_csrc ? env_csrc : libdir + "/../csrc";
    const char* env_hipcc = std::getenv("ILQR_HIPCC");
    const std::string hipcc = env_hipcc ? env_hipcc : "/opt/rocm/bin/hipcc";
    // tag = FNV-1a over everything that ends up in the module: source, dimensions, ABI version and the hash of the kernel
    // headers + compiler flags this library was built from (ILQR_BUILD_HASH, computed by the Makefile): a kernel fix without an
    // ABI bump must not be served a module compiled against the old kernels
    unsigned long long hsh = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { for (size_t i = 0; i < n; ++i) { hsh ^= ((const unsigned char*)p)[i]; hsh *= 1099511628211ull; } };
    mix(src->source, std::strlen(src->source)); mix(src->name, std::strlen(src->name));
    const long long dims[8] = {src->nx, src->nu, src->nw, src->nc_stage, src->nc_term, (long long)iw_s[0], (long long)iw_t[0],
                               ILQR_MODEL_ABI_VERSION * 1000 + (long long)sizeof(ilqr::KArgs)};
    mix(dims, sizeof(dims));
    if (nwords > 1) { mix(iw_s.data(), sizeof(uint64_t) * nwords); mix(iw_t.data(), sizeof(uint64_t) * nwords); }
#ifdef ILQR_BUILD_HASH
    mix(ILQR_BUILD_HASH, std::strlen(ILQR_BUILD_HASH));
#endif
    // The tables a large model is built with are part of its identity: the probe runs BEFORE the name is formed and what it
    // found (or that it found nothing: no host compiler, source that is not host C++, probe switched off) goes into the hash —
    // a module cached on a box where probing was impossible is not served where it works, and the reverse.
    const std::string dir = libdir + "/models";
    mkdir(dir.c_str(), 0755);
    const bool large = src->nx > 4 || src->nu > 4;
    ModelStructure ms;
    if (large) {
        char ptag[32];
        std::snprintf(ptag, sizeof(ptag), "%016llx", hsh);
        ms = probe_model_structure_cached(src, dir, ptag, hints);
        mix(ms.found ? "probed" : "dense", 6);
        if (ms.found) {
            mix(ms.fxc.data(), ms.fxc.size() * 8); mix(ms.fuc.data(), ms.fuc.size() * 8);
            if (!ms.jac_var.empty()) mix(ms.jac_var.data(), ms.jac_var.size() * sizeof(int));
            if (!ms.hess_idx.empty()) mix(ms.hess_idx.data(), ms.hess_idx.size() * sizeof(int));
        }
    }
    char tag[32];
    std::snprintf(tag, sizeof(tag), "%016llx", hsh);
    const std::string uname = std::string(src->name) + "_c" + tag;
    // source and log are written under process-unique names: ranks compiling the same model at the same time must not truncate
    // each other's input under a running hipcc (the module itself is moved into place atomically)
    const std::string pid = std::to_string((long)getpid());
    const std::string so = dir + "/libilqr_model_" + uname + ".so", hip = dir + "/model_" + uname + "." + pid + ".hip",
                      log = dir + "/model_" + uname + "." + pid + ".log";
    if (uname.size() + 1 > name_len || so.size() + 1 > path_len) return fail(ILQR_ERR_INVALID, "output buffers too small");
    struct stat st;
    if (stat(so.c_str(), &st) != 0) {
        FILE* f = std::fopen(hip.c_str(), "w");
        if (!f) return fail(ILQR_ERR_MODEL, "cannot write " + hip);
        const bool cs = src->nc_stage > 0, ct = src->nc_term > 0;
        std::fprintf(f, "// GENERATED by ilqr_compile_model — user callables wrapped for the kernels (ilqr_model_adapter.hpp)\n"
                        "#include \"ilqr_model_adapter.hpp\"\nnamespace user_%s {\n%s\n}\n", tag, src->source);
        std::fprintf(f, "struct Fns_%s {\n", tag);
        const char* names[] = {"dynamics", "dynamics_jacobian_state", "dynamics_jacobian_action", "cost_stage", "cost_stage_gradient_state",
                               "cost_stage_gradient_action", "cost_stage_hessian_state_state", "cost_stage_hessian_action_action",
                               "cost_stage_hessian_action_state", "cost_terminal", "cost_terminal_gradient_state",
                               "cost_terminal_hessian_state_state"};
        for (const char* nm : names)
            std::fprintf(f, "    ILQR_MODEL_FN void %s(double* o, const double* x, const double* u, const double* w) { user_%s::%s(o, x, u, w); }\n", nm, tag, nm);
        const char* cnames[] = {"constraint_stage", "constraint_stage_jacobian_state", "constraint_stage_jacobian_action",
                                "constraint_terminal", "constraint_terminal_jacobian_state"};
        for (int i = 0; i < 5; ++i) {
            const bool have = i < 3 ? cs : ct;
            if (have) std::fprintf(f, "    ILQR_MODEL_FN void %s(double* o, const double* x, const double* u, const double* w) { user_%s::%s(o, x, u, w); }\n", cnames[i], tag, cnames[i]);
            else std::fprintf(f, "    ILQR_MODEL_FN void %s(double*, const double*, const double*, const double*) {}\n", cnames[i]);
        }
        std::fprintf(f, "};\n");
        // nx > 4 or nu > 4: the compact forms of the large path — with the constant / zero entries found by probing the callables
        // on the host (probe_model_structure), dense when that is not possible
        std::string tables;
        if (large) {
            if (ms.found) {
                const int nn = src->nx, mm = src->nu, TNt = (nn + 15) / 16;
                auto ilist = [](const std::vector<int>& v, size_t lo, size_t hi) { std::string o; for (size_t i = lo; i < hi; ++i) o += std::to_string(v[i]) + ","; if (hi == lo) o = "0"; return o; };
                auto dlist = [](const std::vector<double>& v) { std::string o; char b[40]; for (double d : v) { std::snprintf(b, sizeof(b), "%.17g,", d); o += b; } return o; };
                const int hs = (int)ms.hess_idx.size(), jv = (int)ms.jac_var.size();
                std::fprintf(f, "// structure found on the host: %d of %d Jacobian entries state-dependent, %d + %d + %d of %d Hessian entries structurally non-zero\n"
                                "struct Tables_%s {\n    static constexpr int TN = %d, NXX = %d, NUU = %d, NUX = %d, HS = %d, JV = %d;\n"
                                "    struct Tab { int hess_idx[%d]; int tile_start[%d]; int jac_idx[%d]; double fxc[%d]; double fuc[%d]; };\n"
                                "    static constexpr Tab tab = {{%s}, {%s}, {%s}, {%s}, {%s}};\n};\n",
                             jv, nn * nn + nn * mm, ms.nxx, ms.nuu, ms.nux, nn * nn + mm * mm + mm * nn, tag, TNt, ms.nxx, ms.nuu, ms.nux, hs, jv,
                             hs > 0 ? hs : 1, TNt * TNt + 1, jv > 0 ? jv : 1, nn * nn, nn * mm,
                             ilist(ms.hess_idx, 0, ms.hess_idx.size()).c_str(), ilist(ms.tile_start, 0, ms.tile_start.size()).c_str(),
                             ilist(ms.jac_var, 0, ms.jac_var.size()).c_str(), dlist(ms.fxc).c_str(), dlist(ms.fuc).c_str());
                tables = std::string(", Tables_") + tag;
            } else {
                // dense tables: every Jacobian entry evaluated and stored per timestep, every Hessian entry streamed (correct, slow:
                // kilobytes of per-thread arrays in the linearisation beyond nx = 16)
                std::fprintf(f, "// no structure found (%s): dense tables\n", ms.note.c_str());
            }
        }
        std::string words;       // more than 64 rows: the masks as word arrays (ilqr::IneqMask picks them up)
        if (nwords > 1) {
            auto wl = [](const std::vector<uint64_t>& v) { std::string o; char b[32]; for (uint64_t x : v) { std::snprintf(b, sizeof(b), "0x%llxull,", (unsigned long long)x); o += b; } return o; };
            words = "    static constexpr int INEQ_WORDS = " + std::to_string(nwords) + ";\n    static constexpr unsigned long long INEQ_S_W[" +
                    std::to_string(nwords) + "] = {" + wl(iw_s) + "}, INEQ_T_W[" + std::to_string(nwords) + "] = {" + wl(iw_t) + "};\n";
        }
        std::fprintf(f, "struct Model_%s : ilqr::%s<Fns_%s, %d, %d, %d, %d, %d, 0x%llxull, 0x%llxull%s> {\n"
                        "    static constexpr const char* NAME = \"%s\";\n%s};\nILQR_DEFINE_MODEL(Model_%s)\n",
                     uname.c_str(), large ? "AdaptedLargeModel" : "AdaptedModel", tag, src->nx, src->nu, src->nw, src->nc_stage, src->nc_term,
                     (unsigned long long)iw_s[0], (unsigned long long)iw_t[0], tables.c_str(), uname.c_str(), words.c_str(), uname.c_str());
        std::fclose(f);
        // hipcc as a child process (no shell): same flags as the built-in models
        const std::string tmp = so + ".tmp" + pid;
        const std::string inc = "-I" + csrc, lflag = "-L" + libdir, rpath = "-Wl,-rpath," + libdir;
        std::vector<const char*> argv = {hipcc.c_str(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                                         "-amdgpu-mfma-vgpr-form", "-Wno-unused-parameter", inc.c_str(), hip.c_str(), "-o", tmp.c_str(),
                                         lflag.c_str(), "-lilqr_hip", rpath.c_str(), nullptr};
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        posix_spawn_file_actions_adddup2(&fa, 1, 2);
        pid_t pid = 0;
        extern char** environ;
        const int sp = posix_spawn(&pid, hipcc.c_str(), &fa, nullptr, const_cast<char* const*>(argv.data()), environ);
        posix_spawn_file_actions_destroy(&fa);
        if (sp != 0) return fail(ILQR_ERR_MODEL, "cannot start " + hipcc + " (set ILQR_HIPCC)");
        int status = 0;
        if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) {
            std::string msg = "hipcc failed for model '" + std::string(src->name) + "', see " + log;
            if (FILE* lf = std::fopen(log.c_str(), "r")) {
                char buf[1500];
                const size_t nr = std::fread(buf, 1, sizeof(buf) - 1, lf);
                buf[nr] = 0;
                std::fclose(lf);
                msg += ":\n" + std::string(buf);
            }
            return fail(ILQR_ERR_MODEL, msg);
        }
        if (std::rename(tmp.c_str(), so.c_str()) != 0) return fail(ILQR_ERR_MODEL, "cannot move the model module into place");
        if (!std::getenv("ILQR_KEEP_MODEL_SOURCE")) std::remove(hip.c_str());      // (kept on request: the module's .hip next to its .so)
        std::remove(log.c_str());
    }
    if (!dlopen(so.c_str(), RTLD_NOW | RTLD_GLOBAL)) return fail(ILQR_ERR_MODEL, std::string("dlopen failed: ") + dlerror());
    if (!find_model(uname.c_str())) return fail(ILQR_ERR_MODEL, "the compiled module did not register '" + uname + "' (ABI mismatch?)");
    std::snprintf(registered_name, name_len, "%s", uname.c_str());
    std::snprintf(library_path, path_len, "%s", so.c_str());
    return ILQR_OK;
}
const char* ilqr_model_name(int32_t i) {
    return (i >= 0 && i < (int)registry().size()) ? registry()[i]->name : nullptr;
}
int ilqr_model_compact_sizes(const char* model, int32_t* jac_nvar, int32_t* hess_nnz) {
    const ilqr_model_vtable* vt = model ? find_model(model) : nullptr;
    if (!vt) return fail(ILQR_ERR_MODEL, "unknown model");
    if (jac_nvar) *jac_nvar = vt->jac_nvar;
    if (hess_nnz) *hess_nnz = vt->hess_nnz;
    return ILQR_OK;
}

// ---- Vectors of distinct per-step objects / time-varying dimensions, lowered onto the one-stage template (ilqr_hip.h).
// What the reference does by indexing a Vector of objects with t (src/solver.jl:28-46, src/gradients.jl:1-21, src/rollout.jl:22-30)
// becomes: one combined callable per category that branches on one-hot selector parameters, constraint kinds stacked row-wise,
// every kind zero-padded to the largest dimensions of the horizon.
int ilqr_plan_stages(const ilqr_stage_kinds* k, ilqr_stage_plan* plan, double* selectors, size_t selectors_len,
                     int32_t* state_dims, int32_t* action_dims) {
    if (!k || !plan) return fail(ILQR_ERR_INVALID, "null argument");
    const int T = k->horizon, N = T - 1;
    if (T < 2) return fail(ILQR_ERR_INVALID, "horizon must be >= 2");
    if (k->n_dynamics < 1 || k->n_dynamics > ILQR_MAX_STAGE_KINDS || k->n_costs < 1 || k->n_costs > ILQR_MAX_STAGE_KINDS ||
        k->n_constraints < 0 || k->n_constraints > ILQR_MAX_STAGE_KINDS)
        return fail(ILQR_ERR_INVALID, "1 .. 16 dynamics kinds, 1 .. 16 stage-cost kinds, 0 .. 16 stage-constraint kinds");
    if (!k->dynamics_nx || !k->dynamics_nu || !k->dynamics_nx_next || !k->dynamics_of_step || !k->cost_nx || !k->cost_nu || !k->cost_of_step ||
        (k->n_constraints > 0 && (!k->constraint_nc || !k->constraint_nx || !k->constraint_nu || !k->constraint_ineq || !k->constraint_of_step)))
        return fail(ILQR_ERR_INVALID, "null kind table");
    if (k->num_parameter < 0 || k->nc_term < 0 || k->nc_term > ILQR_MAX_CONSTRAINT_ROWS) return fail(ILQR_ERR_INVALID, "num_parameter >= 0, 0 <= nc_term <= 256");
    std::memset(plan, 0, sizeof(*plan));
    // dimensions along the horizon (src/data/problem.jl:32-38) and the consistency the reference's broadcasts would enforce
    std::vector<int> nt(T), mt(N);
    for (int t = 0; t < N; ++t) {
        const int d = k->dynamics_of_step[t], c = k->cost_of_step[t];
        if (d < 0 || d >= k->n_dynamics || c < 0 || c >= k->n_costs) return fail(ILQR_ERR_INVALID, "kind index out of range at step " + std::to_string(t));
        nt[t] = k->dynamics_nx[d]; mt[t] = k->dynamics_nu[d];
        if (nt[t] < 1 || mt[t] < 1 || k->dynamics_nx_next[d] < 1) return fail(ILQR_ERR_INVALID, "dimensions must be >= 1");
        if (t > 0 && k->dynamics_nx_next[k->dynamics_of_step[t - 1]] != nt[t])
            return fail(ILQR_ERR_INVALID, "dynamics[" + std::to_string(t - 1) + "] does not produce the state of step " + std::to_string(t));
        if (k->cost_nx[c] != nt[t] || k->cost_nu[c] != mt[t]) return fail(ILQR_ERR_INVALID, "cost[" + std::to_string(t) + "] dimensions are not its step's");
        if (k->n_constraints > 0) {
            const int q = k->constraint_of_step[t];
            if (q < 0 || q >= k->n_constraints) return fail(ILQR_ERR_INVALID, "constraint kind index out of range at step " + std::to_string(t));
            if (k->constraint_nc[q] < 0 || k->constraint_nc[q] > ILQR_MAX_CONSTRAINT_ROWS) return fail(ILQR_ERR_INVALID, "at most 256 rows per constraint kind");
            if (k->constraint_nc[q] > 0 && (k->constraint_nx[q] != nt[t] || k->constraint_nu[q] != mt[t]))
                return fail(ILQR_ERR_INVALID, "constraint[" + std::to_string(t) + "] dimensions are not its step's");
        }
    }
    nt[N] = k->dynamics_nx_next[k->dynamics_of_step[N - 1]];
    if (k->nx_term != nt[N]) return fail(ILQR_ERR_INVALID, "the terminal objects' num_state is not the last dynamics' num_next_state");
    int n = 0, m = 0;
    for (int t = 0; t < T; ++t) n = std::max(n, nt[t]);
    for (int t = 0; t < N; ++t) m = std::max(m, mt[t]);
    plan->nx = n; plan->nu = m; plan->nc_term = k->nc_term;
    // stacked stage constraint: kind q owns rows [row0[q], row0[q] + nc_q)
    int rows = 0;
    for (int q = 0; q < k->n_constraints; ++q) {
        plan->constraint_row0[q] = rows;
        for (int i = 0; i < k->constraint_nc[q] && i < ILQR_MAX_CONSTRAINT_ROWS; ++i)
            if ((k->constraint_ineq[4 * q + i / 64] >> (i % 64)) & 1ull) {
                const int r = rows + i;
                if (r < ILQR_MAX_CONSTRAINT_ROWS) plan->ineq_stage_words[r / 64] |= 1ull << (r % 64);
            }
        rows += k->constraint_nc[q];
    }
    if (rows > ILQR_MAX_CONSTRAINT_ROWS) return fail(ILQR_ERR_INVALID, "at most 256 stage constraint rows over all kinds");
    plan->nc_stage = rows;
    // selector blocks: one per category that really varies, in the order dynamics, cost, constraint, behind the user's parameters
    int off = k->num_parameter;
    plan->sel_dynamics = plan->sel_cost = plan->sel_constraint = -1;
    const bool uniform = k->n_dynamics == 1 && k->n_costs == 1 && k->n_constraints <= 1;
    if (!uniform) {
        if (k->n_dynamics > 1) { plan->sel_dynamics = off; off += k->n_dynamics; }
        if (k->n_costs > 1) { plan->sel_cost = off; off += k->n_costs; }
        if (k->n_constraints > 1) { plan->sel_constraint = off; off += k->n_constraints; }
    }
    plan->nw = off; plan->n_selectors = off - k->num_parameter;
    const int S = plan->n_selectors;
    if (selectors) {
        if (selectors_len < (size_t)T * (size_t)S) return fail(ILQR_ERR_INVALID, "selector buffer too small");
        std::fill(selectors, selectors + (size_t)T * S, 0.0);
        for (int t = 0; t < N; ++t) {
            double* row = selectors + (size_t)t * S - k->num_parameter;          // indexed by parameter column
            if (plan->sel_dynamics >= 0) row[plan->sel_dynamics + k->dynamics_of_step[t]] = 1.0;
            if (plan->sel_cost >= 0) row[plan->sel_cost + k->cost_of_step[t]] = 1.0;
            if (plan->sel_constraint >= 0) row[plan->sel_constraint + k->constraint_of_step[t]] = 1.0;
        }
    }
    if (state_dims) for (int t = 0; t < T; ++t) state_dims[t] = nt[t];
    if (action_dims) for (int t = 0; t < N; ++t) action_dims[t] = mt[t];
    return ILQR_OK;
}

// The combined callables of the template, as C source around the kinds' own callables (which live in namespace kinds).
static std::string compose_stage_source(const ilqr_stage_kinds* k, const ilqr_stage_plan& pl, const char* user_source) {
    const int n = pl.nx, m = pl.nu;
    std::string o = "namespace kinds {\n" + std::string(user_source) + "\n}\n";
    const std::string sig = "(double* o, const double* x, const double* u, const double* w)";
    auto S = [](int v) { return std::to_string(v); };
    // `if (w[sel + k] > 0.5)` chain of a category; a category with one kind calls it unconditionally
    auto branch = [&](int sel, int kinds, int kk) {
        if (sel < 0 || kinds <= 1) return std::string("    {\n");
        return std::string(kk == 0 ? "    if" : "    else if") + " (w[" + S(sel + kk) + "] > 0.5) {\n";
    };
    // call NAME into a (rows x cols, leading dimension ld_k) scratch and copy into o with leading dimension ld at row offset r0;
    // direct when the layouts coincide
    auto mat = [&](const std::string& name, int rows_k, int cols_k, int ld, int r0) {
        if (rows_k == ld && r0 == 0) return "        kinds::" + name + "(o, x, u, w);\n";
        if (rows_k * cols_k == 0) return std::string();
        std::string c = "        double t[" + S(rows_k * cols_k) + "];\n        for (int i = 0; i < " + S(rows_k * cols_k) + "; ++i) t[i] = 0.0;\n";
        c += "        kinds::" + name + "(t, x, u, w);\n";
        c += "        for (int c = 0; c < " + S(cols_k) + "; ++c) for (int r = 0; r < " + S(rows_k) + "; ++r) o[c * " + S(ld) + " + " + S(r0) + " + r] = t[c * " + S(rows_k) + " + r];\n";
        return c;
    };
    // ---- Dynamics (src/dynamics.jl:36-50): jacobian_state is num_next_state x num_state, jacobian_action num_next_state x num_action
    const char* dsuf[3] = {"", "_jacobian_state", "_jacobian_action"};
    for (int f = 0; f < 3; ++f) {
        o += "ILQR_MODEL_FN void dynamics" + std::string(dsuf[f]) + sig + " {\n";
        for (int q = 0; q < k->n_dynamics; ++q) {
            const std::string nm = "dynamics_" + S(q) + dsuf[f];
            o += branch(pl.sel_dynamics, k->n_dynamics, q);
            if (f == 0) o += "        kinds::" + nm + "(o, x, u, w);\n";
            else o += mat(nm, k->dynamics_nx_next[q], f == 1 ? k->dynamics_nx[q] : k->dynamics_nu[q], n, 0);
            o += "    }\n";
        }
        o += "}\n";
    }
    // ---- stage Cost (src/costs.jl:48-84); padded actions cost u^2 / 2
    const char* csuf[6] = {"", "_gradient_state", "_gradient_action", "_hessian_state_state", "_hessian_action_action", "_hessian_action_state"};
    for (int f = 0; f < 6; ++f) {
        o += "ILQR_MODEL_FN void cost_stage" + std::string(csuf[f]) + sig + " {\n";
        for (int q = 0; q < k->n_costs; ++q) {
            const std::string nm = "cost_stage_" + S(q) + csuf[f];
            const int n0 = k->cost_nx[q], m0 = k->cost_nu[q];
            o += branch(pl.sel_cost, k->n_costs, q);
            if (f == 0) {
                o += "        kinds::" + nm + "(o, x, u, w);\n";
                if (m0 < m) {
                    std::string sum;
                    for (int j = m0; j < m; ++j) sum += (j > m0 ? " + " : "") + ("u[" + S(j) + "] * u[" + S(j) + "]");
                    o += "        o[0] = o[0] + (" + sum + ") / 2.0;\n";
                }
            } else if (f == 1) o += "        kinds::" + nm + "(o, x, u, w);\n";
            else if (f == 2) {
                o += "        kinds::" + nm + "(o, x, u, w);\n";
                for (int j = m0; j < m; ++j) o += "        o[" + S(j) + "] = u[" + S(j) + "];\n";
            } else if (f == 3) o += mat(nm, n0, n0, n, 0);
            else if (f == 4) {
                o += mat(nm, m0, m0, m, 0);
                for (int j = m0; j < m; ++j) o += "        o[" + S(j * m + j) + "] = 1.0;\n";
            } else o += mat(nm, m0, n0, m, 0);
            o += "    }\n";
        }
        o += "}\n";
    }
    // ---- terminal Cost: num_state = nx_term
    o += "ILQR_MODEL_FN void cost_terminal" + sig + " { kinds::cost_terminal(o, x, u, w); }\n";
    o += "ILQR_MODEL_FN void cost_terminal_gradient_state" + sig + " { kinds::cost_terminal_gradient_state(o, x, u, w); }\n";
    o += "ILQR_MODEL_FN void cost_terminal_hessian_state_state" + sig + " {\n    {\n" + mat("cost_terminal_hessian_state_state", k->nx_term, k->nx_term, n, 0) + "    }\n}\n";
    // ---- stage Constraint (src/constraints.jl:66-87): kinds stacked, kind q at rows row0[q] ..
    if (pl.nc_stage > 0) {
        const char* ksuf[3] = {"", "_jacobian_state", "_jacobian_action"};
        for (int f = 0; f < 3; ++f) {
            o += "ILQR_MODEL_FN void constraint_stage" + std::string(ksuf[f]) + sig + " {\n";
            bool first = true;
            for (int q = 0; q < k->n_constraints; ++q) {
                if (k->constraint_nc[q] == 0) continue;
                const std::string nm = "constraint_stage_" + S(q) + ksuf[f];
                if (pl.sel_constraint < 0) o += "    {\n";
                else o += std::string(first ? "    if" : "    else if") + " (w[" + S(pl.sel_constraint + q) + "] > 0.5) {\n";
                first = false;
                if (f == 0) o += "        kinds::" + nm + "(o + " + S(pl.constraint_row0[q]) + ", x, u, w);\n";
                else o += mat(nm, k->constraint_nc[q], f == 1 ? k->constraint_nx[q] : k->constraint_nu[q], pl.nc_stage, pl.constraint_row0[q]);
                o += "    }\n";
            }
            o += "}\n";
        }
    }
    // ---- terminal Constraint: nc_term x nx_term, column-major with leading dimension nc_term — the template's nc_term x nx has the same
    if (k->nc_term > 0) {
        o += "ILQR_MODEL_FN void constraint_terminal" + sig + " { kinds::constraint_terminal(o, x, u, w); }\n";
        o += "ILQR_MODEL_FN void constraint_terminal_jacobian_state" + sig + " { kinds::constraint_terminal_jacobian_state(o, x, u, w); }\n";
    }
    return o;
}

int ilqr_compile_model_stages(const char* name, const ilqr_stage_kinds* kinds, const char* source, ilqr_stage_plan* plan,
                              double* selectors, size_t selectors_len, int32_t* state_dims, int32_t* action_dims,
                              char* registered_name, size_t name_len, char* library_path, size_t path_len) {
    if (!name || !kinds || !source || !plan) return fail(ILQR_ERR_INVALID, "null argument");
    const int rc = ilqr_plan_stages(kinds, plan, selectors, selectors_len, state_dims, action_dims);
    if (rc != ILQR_OK) return rc;
    const std::string combined = compose_stage_source(kinds, *plan, source);
    ilqr_model_source ms;
    ms.name = name; ms.nx = plan->nx; ms.nu = plan->nu; ms.nw = plan->nw; ms.nc_stage = plan->nc_stage; ms.nc_term = plan->nc_term;
    ms.ineq_stage = plan->ineq_stage_words[0]; ms.ineq_term = kinds->ineq_term[0]; ms.source = combined.c_str(); ms.flags = 0;
    ProbeHints hints;
    hints.sel[0] = plan->sel_dynamics; hints.kinds[0] = plan->sel_dynamics >= 0 ? kinds->n_dynamics : 0;
    hints.sel[1] = plan->sel_cost; hints.kinds[1] = plan->sel_cost >= 0 ? kinds->n_costs : 0;
    hints.sel[2] = plan->sel_constraint; hints.kinds[2] = plan->sel_constraint >= 0 ? kinds->n_constraints : 0;
    return compile_model_impl(&ms, plan->ineq_stage_words, kinds->ineq_term, registered_name, name_len, library_path, path_len, hints);
}

int ilqr_create(const ilqr_problem_desc* d, ilqr_handle** out) {
    if (!d || !out || !d->model) return fail(ILQR_ERR_INVALID, "null descriptor/model");
    if (d->horizon < 2 || d->batch < 1) return fail(ILQR_ERR_INVALID, "horizon must be >= 2 and batch >= 1");
    if (d->model_library && d->model_library[0]) {
        if (!dlopen(d->model_library, RTLD_NOW | RTLD_GLOBAL))
            return fail(ILQR_ERR_MODEL, std::string("dlopen failed: ") + dlerror());
    }
    const ilqr_model_vtable* vt = find_model(d->model);
    if (!vt) return fail(ILQR_ERR_MODEL, std::string("unknown model '") + d->model + "'");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(ILQR_ERR_NO_DEVICE, "no HIP device available: the batched solver has no CPU fallback");
    if (d->device < 0 || d->device >= ndev) return fail(ILQR_ERR_INVALID, "device ordinal out of range");
    ilqr_handle* h = new ilqr_handle();
    h->vt = vt; h->B = d->batch; h->device = d->device; h->constrained = d->constrained ? 1 : 0;
    h->L = ilqr::make_layout(vt->nx, vt->nu, vt->nw, vt->ncs, vt->nct, d->horizon, vt->jac_nvar, vt->hess_nnz);
    h->lds_bytes = ilqr::is_large_model(vt->nx, vt->nu) ? (size_t)ilqr::large_lds_doubles(vt->nx, vt->nu, vt->hess_nnz) * 8
                                                          : (size_t)h->L.lds_doubles * 8;
    h->ws = nullptr; h->d_x1 = nullptr; h->d_u = nullptr; h->stream = nullptr;
    h->trace = nullptr; h->trace_cap = 0; h->variant = 0; h->num_simds = 1024; h->handover = -1; h->handover_live = -1; h->done_counter = nullptr; h->pool = nullptr; h->handover_mark = -1;
    h->qv = nullptr; h->QL = ilqr::make_qlayout(vt->nx, vt->nu, d->horizon); h->full_stale = false; h->P_dirty = false;
    ilqr_default_options(&h->opt);
    fill_buffers(h);
    h->lds_fits = h->lds_bytes <= 160 * 1024;
    if (!h->lds_fits && vt->launch_solve_packed == nullptr) {
        // the reference has no horizon limit (src/data/problem.jl:25-46); here only the streaming (packed) kernel is free of one
        delete h;
        return fail(ILQR_ERR_LDS, "per-instance working set exceeds the 160 KiB LDS of a gfx950 CU and this model has no "
                                  "streaming (packed) kernel; reduce the horizon");
    }
    // from here on every failure must release the handle
    auto bail = [&](hipError_t e, const char* what) {
        const int rc = fail(ILQR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        ilqr_destroy(h);
        return rc;
    };
    hipError_t e;
    if ((e = hipSetDevice(h->device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) h->num_simds = 4 * prop.multiProcessorCount;
    }
    h->ws_bytes = (size_t)h->B * (size_t)h->L.stride * 8;
    if ((e = hipMalloc((void**)&h->ws, h->ws_bytes)) != hipSuccess) return bail(e, "hipMalloc(workspace)");
    if (vt->launch_solve_packed != nullptr && (e = hipMalloc((void**)&h->done_counter, sizeof(int))) != hipSuccess) return bail(e, "hipMalloc(hand-over counter)");
    if (vt->launch_solve_packed != nullptr && (e = hipMalloc((void**)&h->pool, sizeof(int) * (size_t)(ilqr::POOL_Q + h->B + ilqr::POOL_CUS))) != hipSuccess) return bail(e, "hipMalloc(hand-over queue)");
    if ((e = hipMemsetAsync(h->ws, 0, h->ws_bytes, h->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if (!ilqr::is_large_model(vt->nx, vt->nu) && (e = hipMalloc((void**)&h->cu_slots, sizeof(int) * ilqr::CU_SLOT_INTS * ilqr::CU_SLOT_CUS)) != hipSuccess) return bail(e, "hipMalloc(role table)");
    if (h->pool && (e = hipMemsetAsync(h->pool, 0, sizeof(int) * (size_t)(ilqr::POOL_Q + h->B + ilqr::POOL_CUS), h->stream)) != hipSuccess) return bail(e, "hipMemsetAsync(hand-over queue)");
    *out = h;
    int rc = ilqr_reset(h);
    if (rc != ILQR_OK) { ilqr_destroy(h); *out = nullptr; return rc; }
    return ILQR_OK;
}

// Solver(...) over a device list — SURVEY §8(b)/(e): one reference-style handle whose batch is split into contiguous ranges of
// ceil(B / G) instances, range i on devices[i] (a device may be listed more than once). Every other entry point accepts the
// handle and scatters / gathers instance-major host arrays over the ranges.
int ilqr_create_sharded(const ilqr_problem_desc* d, const int32_t* devices, int32_t n_devices, ilqr_handle** out) {
    if (!d || !out || !d->model || !devices) return fail(ILQR_ERR_INVALID, "null descriptor/model/device list");
    if (n_devices < 1) return fail(ILQR_ERR_INVALID, "empty device list");
    if (d->horizon < 2 || d->batch < 1) return fail(ILQR_ERR_INVALID, "horizon must be >= 2 and batch >= 1");
    if (n_devices > d->batch) return fail(ILQR_ERR_INVALID, "more devices than instances");
    ilqr_handle* h = new ilqr_handle();
    h->vt = nullptr; h->ws = nullptr; h->d_x1 = nullptr; h->d_u = nullptr; h->stream = nullptr; h->trace = nullptr; h->qv = nullptr; h->done_counter = nullptr; h->pool = nullptr;
    h->B = d->batch; h->device = devices[0]; h->constrained = d->constrained ? 1 : 0; h->trace_cap = 0; h->variant = 0;
    const int per = (d->batch + n_devices - 1) / n_devices;
    for (int i = 0, lo = 0; i < n_devices && lo < d->batch; ++i, lo += per) {
        ilqr_problem_desc sd = *d;
        sd.device = devices[i];
        sd.batch = (d->batch - lo) < per ? (d->batch - lo) : per;
        ilqr_handle* sub = nullptr;
        const int rc = ilqr_create(&sd, &sub);
        if (rc != ILQR_OK) { const std::string msg = g_err; ilqr_destroy(h); return fail(rc, msg); }
        h->shards.push_back(sub);
        h->lo.push_back(lo);
    }
    const ilqr_handle* s0 = h->shards[0];
    h->vt = s0->vt; h->L = s0->L; h->QL = s0->QL; h->opt = s0->opt; h->lds_bytes = s0->lds_bytes; h->lds_fits = s0->lds_fits; h->num_simds = s0->num_simds;
    h->full_stale = false; h->P_dirty = false;
    fill_buffers(h);
    *out = h;
    return ILQR_OK;
}

int ilqr_destroy(ilqr_handle* h) {
    if (!h) return ILQR_OK;
    if (!h->shards.empty() || h->vt == nullptr) {
        for (ilqr_handle* s : h->shards) ilqr_destroy(s);
        delete h;
        return ILQR_OK;
    }
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (auto& p : h->timing) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (h->ws) hipFree(h->ws);
    if (h->d_x1) hipFree(h->d_x1);
    if (h->done_counter) hipFree(h->done_counter);
    if (h->pool) hipFree(h->pool);
    if (h->cu_slots) hipFree(h->cu_slots);
    if (h->d_u) hipFree(h->d_u);
    if (h->trace) hipFree(h->trace);
    if (h->qv) hipFree(h->qv);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return ILQR_OK;
}

int ilqr_set_options(ilqr_handle* h, const ilqr_options* opt) {
    if (!h || !opt) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) { h->opt = *opt; return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_set_options(s, opt); }); }
    h->opt = *opt;
    return ILQR_OK;
}

int ilqr_get_dims(const ilqr_handle* h, int32_t* nx, int32_t* nu, int32_t* nw, int32_t* ncs, int32_t* nct,
                  int32_t* horizon, int32_t* batch) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (nx) *nx = h->vt->nx;
    if (nu) *nu = h->vt->nu;
    if (nw) *nw = h->vt->nw - h->n_sel;          // the USER's parameters per timestep (selector columns are the library's)
    if (ncs) *ncs = h->vt->ncs;
    if (nct) *nct = h->vt->nct;
    if (horizon) *horizon = h->L.T;
    if (batch) *batch = h->B;
    return ILQR_OK;
}

int ilqr_reset(ilqr_handle* h) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_reset(s); });
    HIP_TRY(hipSetDevice(h->device));
    if (ilqr::is_large_model(h->vt->nx, h->vt->nu) && h->vt->nw == 0) {
        // HBM-resident models: zero the trajectories, gradients, gains, duals, scalars and the compact Jacobian / Hessian rows
        // now; the megabyte-sized full Jacobian / Hessian mirrors are rewritten from the compact form when a getter asks
        // for them, the value arrays P, p are zeroed when a getter, setter or stage call could observe them
        ilqr::KArgs ra = make_args(h);
        hipLaunchKernelGGL(ilqr::reset_large_kernel, dim3(h->B * 8), dim3(256), 0, h->stream, ra, h->L.fx, h->L.scal);
        HIP_TRY(hipGetLastError());
        h->full_stale = true; h->P_dirty = true;
        return ILQR_OK;
    } else if (h->vt->nw == 0) {
        HIP_TRY(hipMemsetAsync(h->ws, 0, h->ws_bytes, h->stream));
    } else {
        h->full_stale = ilqr::is_large_model(h->vt->nx, h->vt->nu);
        // keep the parameters θ (they belong to the problem, not to the solver state)
        const size_t pitch = (size_t)h->L.stride * 8, w0 = (size_t)h->L.w * 8, w1 = (size_t)h->L.zslot * 8;
        HIP_TRY(hipMemset2DAsync(h->ws, pitch, 0, w0, (size_t)h->B, h->stream));
        HIP_TRY(hipMemset2DAsync((char*)h->ws + w1, pitch, 0, pitch - w1, (size_t)h->B, h->stream));
    }
    ilqr::KArgs a = make_args(h);
    hipLaunchKernelGGL(ilqr::defaults_kernel, dim3(h->B), dim3(64), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    return ILQR_OK;
}

int ilqr_initialize_controls(ilqr_handle* h, const double* u) {
    if (!h || !u) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_initialize_controls(s, u + lo * (size_t)(h->L.T - 1) * h->L.nu); }, true);
    return copy_in(h, find_buffer(h, "nominal_actions"), u);
}
int ilqr_initialize_states(ilqr_handle* h, const double* x) {
    if (!h || !x) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_initialize_states(s, x + lo * (size_t)h->L.T * h->L.nx); }, true);
    return copy_in(h, find_buffer(h, "nominal_states"), x);
}

// the full parameter block [B][T][nw] of a handle with selectors: the user's columns (NULL = zeros) followed by the table
static int write_parameters_with_selectors(ilqr_handle* h, const double* w_user) {
    const int T = h->L.T, nw = h->vt->nw, S = h->n_sel, nwu = nw - S;
    std::vector<double> full((size_t)h->B * T * nw, 0.0);
    if (!w_user && nwu > 0) {        // selectors alone (ilqr_set_stage_selectors): the user's columns keep what they hold
        const int rc = copy_out(h, find_buffer(h, "parameters"), full.data());
        if (rc != ILQR_OK) return rc;
    }
    for (int b = 0; b < h->B; ++b)
        for (int t = 0; t < T; ++t) {
            double* row = &full[((size_t)b * T + t) * nw];
            if (w_user) for (int j = 0; j < nwu; ++j) row[j] = w_user[((size_t)b * T + t) * nwu + j];
            for (int j = 0; j < S; ++j) row[nwu + j] = h->sel[(size_t)t * S + j];
        }
    return copy_in(h, find_buffer(h, "parameters"), full.data());
}

int ilqr_set_stage_selectors(ilqr_handle* h, const double* selectors, int32_t n_selectors) {
    if (!h || n_selectors < 0 || (n_selectors > 0 && !selectors)) return fail(ILQR_ERR_INVALID, "null argument");
    if (n_selectors > h->vt->nw) return fail(ILQR_ERR_INVALID, "more selector columns than the model has parameters");
    if (SHARDED(h)) {
        const int rc = each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_set_stage_selectors(s, selectors, n_selectors); }, true);
        if (rc == ILQR_OK) { h->n_sel = n_selectors; h->sel.assign(selectors, selectors + (size_t)h->L.T * n_selectors); }
        return rc;
    }
    h->n_sel = n_selectors;
    h->sel.assign(selectors, selectors + (size_t)h->L.T * n_selectors);
    if (n_selectors == 0) return ILQR_OK;
    return write_parameters_with_selectors(h, nullptr);
}

int ilqr_set_parameters(ilqr_handle* h, const double* w) {
    if (!h || !w) return fail(ILQR_ERR_INVALID, "null argument");
    const size_t nwu = (size_t)(h->L.nw - h->n_sel);
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_set_parameters(s, w + lo * (size_t)h->L.T * nwu); }, true);
    if (nwu == 0) return fail(ILQR_ERR_INVALID, "this model has no parameters (num_parameter == 0)");
    if (h->n_sel > 0) return write_parameters_with_selectors(h, w);
    return copy_in(h, find_buffer(h, "parameters"), w);
}

int ilqr_initialize_rollout_device(ilqr_handle* h, const double* d_x1, const double* d_u) {
    if (!h || !d_x1 || !d_u) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return fail(ILQR_ERR_INVALID, "device pointers belong to one device: call ilqr_initialize_rollout (host pointers) on a sharded handle");
    HIP_TRY(hipSetDevice(h->device));
    ilqr::KArgs a = make_args(h);
    a.x1 = d_x1; a.u_in = d_u;
    if (h->vt->launch_init(&a, h->stream) != 0) return fail(ILQR_ERR_HIP, "init_rollout launch failed");
    return ILQR_OK;
}

int ilqr_initialize_rollout(ilqr_handle* h, const double* x1, const double* u) {
    if (!h || !x1 || !u) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) {
        return ilqr_initialize_rollout(s, x1 + lo * (size_t)h->L.nx, u + lo * (size_t)(h->L.T - 1) * h->L.nu); }, true);
    HIP_TRY(hipSetDevice(h->device));
    const size_t bx = (size_t)h->B * h->vt->nx * 8, bu = (size_t)h->B * (h->L.T - 1) * h->vt->nu * 8;
    if (!h->d_x1) HIP_TRY(hipMalloc((void**)&h->d_x1, bx));
    if (!h->d_u) HIP_TRY(hipMalloc((void**)&h->d_u, bu));
    HIP_TRY(hipMemcpyAsync(h->d_x1, x1, bx, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_u, u, bu, hipMemcpyHostToDevice, h->stream));
    int rc = ilqr_initialize_rollout_device(h, h->d_x1, h->d_u);
    if (rc != ILQR_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));   // host buffers may be reused by the caller
    return ILQR_OK;
}

// Large models whose matrices are single tiles (nx, nu <= 16): the four-wave kernel holds 2 instances per CU, the one-wave kernel SIX
// (248 VGPRs and 25 KB of LDS per wave: profiles/r04_synth12_b4096_mid_rocprofv3.txt).
// An instance alone is faster on four waves (its windows run their tiles side by side: 3.9 k against 5.3 k clk per Riccati step on
// synth12, the rollout beside the sensitivity sweep instead of behind it), so auto takes one wave per instance only where residency
// wins: beyond 8 instances per CU the four-wave kernel works in more than four rounds (tools/mid_bench.py: equal at 2048 instances
// on 256 CUs, 1.28x at 4096, 1.47x at 8192).
static bool use_mid(const ilqr_handle* h) {
    if (h->vt->launch_solve_mid == nullptr) return false;
    return h->variant == 4 || (h->variant == 0 && h->B > 2 * h->num_simds);
}

int ilqr_initialize_rollout_resident(ilqr_handle* h) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_initialize_rollout_resident(s); });
    if (!h->d_x1 || !h->d_u) return fail(ILQR_ERR_INVALID, "no resident inputs: ilqr_initialize_rollout (host pointers) has to come first");
    return ilqr_initialize_rollout_device(h, h->d_x1, h->d_u);
}

int ilqr_solve(ilqr_handle* h) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_solve(s); });      // asynchronous on every device's stream
    HIP_TRY(hipSetDevice(h->device));
    ilqr::KArgs a = make_args(h);
    a.qv = nullptr;
    if (h->trace)      // rows of an earlier, longer solve must not survive
        HIP_TRY(hipMemsetAsync(h->trace, 0, (size_t)h->B * h->trace_cap * ilqr::TRACE_W * 8, h->stream));
    if (h->vt->launch_mirror) h->full_stale = true;   // the kernel works on the compact Jacobian / Hessian rows
    h->pool_valid = false;                            // (set again below when this solve zeroes and uses the hand-over queue)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto drop = [&](int rc) { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); return rc; };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, h->stream) != hipSuccess)
        return drop(fail(ILQR_ERR_HIP, "hipEventCreate/Record failed"));
    // auto: the latency kernel while the batch fits the chip (one instance per SIMD); larger batches take the packed
    // kernel (four instances per wave, workspace streamed from HBM / L2) when the model has one (nx, nu <= 4),
    // the throughput kernel otherwise.
    // Horizons whose LDS-resident set exceeds 160 KiB run on the packed kernel only.
    const bool can_pack = h->vt->launch_solve_packed != nullptr;
    const bool packed = can_pack && (h->variant == 3 || h->variant == 5 || h->variant == 6 || !h->lds_fits || (h->variant == 0 && h->B > h->num_simds));
    const bool slim = !packed && h->vt->launch_solve_slim != nullptr &&
                      (h->variant == 2 || (h->variant == 0 && h->B > h->num_simds));
    if (!packed && !h->lds_fits) return drop(fail(ILQR_ERR_LDS, "this horizon only runs on the packed kernel"));
    if (packed) {
        // straggler hand-over: the survivors of a batch leave the packed kernel and are finished by the latency kernel (two waves
        // per instance, LDS-resident state; a rejected line-search trial costs it one rollout where it costs the packed kernel a
        // whole cycle of the wave) in a launch that follows on the stream. By head count (the default) — once no more than `live`
        // instances of the batch are still running, each of them leaves at the next head of an inner or outer iteration;
        // auto: live = min(1024, B / 4), what the latency kernel holds at full speed. By outer iteration (ilqr_set_handover(k >= 2))
        // — an instance entering outer iteration k leaves at that boundary. Both kernels do the same arithmetic, so which
        // instances change kernels, and when, never shows in a result.
        // (Measured and dropped: a pool of latency workgroups BESIDE the packed kernel taking leavers from a device queue — at
        // 8192 instances the packed kernel's 2048 waves fill every CU, a pool workgroup only starts once they retire)
        int ho = h->handover < 0 ? 0 : h->handover;
        int live = h->handover < 0 ? (h->handover_live < 0 ? std::min(1024, h->B / 4) : h->handover_live) : 0;
        const bool can = h->constrained && h->lds_fits && h->done_counter != nullptr;
        if (!can || ho < 2 || ho > h->opt.max_dual_updates) ho = 0;
        if (!can) live = 0;
        a.handover_outer = ho; a.handover_live = live;
        if (live > 0) HIP_TRY(hipMemsetAsync(h->done_counter, 0, sizeof(int), h->stream));
        // the one-wave form's workgroups finish the instances handed over themselves (ilqr_device_packed.hpp: solve_kernel_packed);
        // under the head-count rule an instance whose rejected line-search trials exceed the batch's mean by `mark` leaves at once
        if ((ho > 0 || live > 0) && h->pool != nullptr) {
            HIP_TRY(hipMemsetAsync(h->pool, 0, sizeof(int) * (size_t)(ilqr::POOL_Q + h->B + ilqr::POOL_CUS), h->stream));
            // a marked straggler gets its CU to itself (config 4, shard 6: 126.6 -> 118.0 ms) — while the launch is ONE round of
            // workgroups (four of them per CU): in a launch of several rounds a workgroup that waits holds the slots the next round
            // needs, so there nobody waits (pool_cu = 0: no CU is vacated, an idle worker leaves as soon as the queue is empty)
            a.pool_cu = ((h->B + 3) / 4 + 1) / 2 <= h->num_simds ? 1 : 0;
            a.pool = h->pool; a.pool_lds = (int)h->lds_bytes;
            h->pool_valid = true;
            a.pool_mark = live > 0 ? (h->handover_mark < 0 ? 6 : h->handover_mark) : 0;
            if (!a.pool_cu) a.pool_mark = 0;      // (and nobody is marked: the batch's mean says nothing while half the batch has not started)
        }
#ifdef ILQR_PK_DEBUG_HOOK      // phase-timing hook of tools/packed_phases.py (see ilqr_device_packed.hpp); never compiled into the product library
        if (const char* dbg = std::getenv("ILQR_PK_DEBUG")) a.stage = std::atoi(dbg);
#endif
        // two waves per pack (a linearisation server beside the solver wave) while the batch leaves every SIMD at most two waves and
        // a CU's LDS holds the second chunk buffers: up to 4 workgroups per CU (variant 5 = one wave per pack, 6 = two where they fit)
        const int packs = (h->B + 3) / 4, cus = std::max(1, h->num_simds / 4), per_cu = (packs + cus - 1) / cus;
        a.stage_flag = (h->variant != 5 && packed2_fits(h->vt->packed2_lds_bytes, per_cu)) ? 2 : 0;
        a.stage_param = (double)per_cu;
        {   // two-wave form: one solver wave per SIMD (KArgs::cu_slots; the one-wave form ignores it)
            static const bool role_slots = !(std::getenv("ILQR_ROLE_SLOTS") && std::getenv("ILQR_ROLE_SLOTS")[0] == '0');
            if (h->cu_slots != nullptr && role_slots && a.stage_flag == 2) {
                HIP_TRY(hipMemsetAsync(h->cu_slots, 0, sizeof(int) * ilqr::CU_SLOT_INTS * ilqr::CU_SLOT_CUS, h->stream));
                a.cu_slots = h->cu_slots; a.cu_expect = std::min(4, per_cu);
            }
        }
        if (h->vt->launch_solve_packed(&a, h->stream) != 0) return drop(fail(ILQR_ERR_HIP, "solve (packed variant) launch failed"));
        a.cu_slots = nullptr; a.cu_expect = 0;
        a.stage_flag = 0; a.stage_param = 0.0; a.pool = nullptr; a.pool_mark = 0;
        if (ho > 0 || live > 0) {
            ilqr::KArgs r = a;
            r.resume = 1; r.stage = 0;
            if (h->vt->launch_solve(&r, h->lds_bytes, h->stream) != 0) return drop(fail(ILQR_ERR_HIP, "solve (hand-over resume) launch failed"));
        }
    } else if (slim) {
        if (h->vt->launch_solve_slim(&a, (size_t)h->L.lds_doubles_slim * 8, h->stream) != 0)
            return drop(fail(ILQR_ERR_HIP, "solve (throughput variant) launch failed"));
    } else if (use_mid(h)) {
        size_t lds_mid = h->lds_bytes;
#ifdef ILQR_DBG_LDS_PAD_HOOK   // residency experiment of tools/mid_bench.py (more LDS per workgroup = fewer workgroups per CU); never in the product library
        if (const char* pad = std::getenv("ILQR_DBG_LDS_PAD")) lds_mid += (size_t)std::atoi(pad);
#endif
        if (h->vt->launch_solve_mid(&a, lds_mid, h->stream) != 0) return drop(fail(ILQR_ERR_HIP, "solve (one-wave variant) launch failed"));
    } else {
        // two-wave latency kernel: one critical wave per SIMD (KArgs::cu_slots; ILQR_ROLE_SLOTS=0 leaves the roles as launched: A/B runs)
        static const bool role_slots = !(std::getenv("ILQR_ROLE_SLOTS") && std::getenv("ILQR_ROLE_SLOTS")[0] == '0');
        if (h->cu_slots != nullptr && role_slots) {
            HIP_TRY(hipMemsetAsync(h->cu_slots, 0, sizeof(int) * ilqr::CU_SLOT_INTS * ilqr::CU_SLOT_CUS, h->stream));
            const int cus = std::max(1, h->num_simds / 4);
            a.cu_slots = h->cu_slots; a.cu_expect = std::min(4, (h->B + cus - 1) / cus);
        }
        if (h->vt->launch_solve(&a, h->lds_bytes, h->stream) != 0) return drop(fail(ILQR_ERR_HIP, "solve launch failed"));
    }
    if (hipEventRecord(e1, h->stream) != hipSuccess) return drop(fail(ILQR_ERR_HIP, "hipEventRecord failed"));
    h->timing.emplace_back(e0, e1);
    if (h->timing.size() > 4096) {     // long-running callers that never read the timing: keep the newest half
        for (size_t i = 0; i < 2048; ++i) { hipEventDestroy(h->timing[i].first); hipEventDestroy(h->timing[i].second); }
        h->timing.erase(h->timing.begin(), h->timing.begin() + 2048);
    }
    return ILQR_OK;
}

int ilqr_run_stage(ilqr_handle* h, int32_t stage) { return ilqr_run_stage_param(h, stage, 0.0, 0); }

int ilqr_run_stage_param(ilqr_handle* h, int32_t stage, double param, int32_t flag) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_run_stage_param(s, stage, param, flag); }, true);
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = settle_reset(h); if (rc != ILQR_OK) return rc; }
    if (!h->lds_fits) return fail(ILQR_ERR_LDS, "stage kernels are LDS-resident: this horizon only runs through ilqr_solve (packed kernel)");
    ilqr::KArgs a = make_args(h);
    a.stage = stage; a.stage_param = param; a.stage_flag = flag;
    if (stage != ILQR_STAGE_BACKWARD_PASS && stage != ILQR_STAGE_ILQR_SOLVE) a.qv = nullptr;
    // the stage runs in the mapping selected by ilqr_set_kernel_variant (2 = throughput: one wave per instance)
    if (h->variant == 2 && h->vt->launch_stage_slim != nullptr) {
        if (h->vt->launch_stage_slim(&a, (size_t)h->L.lds_doubles_slim * 8, h->stream) != 0)
            return fail(ILQR_ERR_HIP, "stage (throughput variant) launch failed");
    } else if (h->variant == 4 && h->vt->launch_stage_mid != nullptr) {
        if (h->vt->launch_stage_mid(&a, h->lds_bytes, h->stream) != 0) return fail(ILQR_ERR_HIP, "stage (one-wave variant) launch failed");
    } else if (h->vt->launch_stage(&a, h->lds_bytes, h->stream) != 0) return fail(ILQR_ERR_HIP, "stage launch failed");
    if (h->vt->launch_mirror) h->full_stale = true;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ILQR_OK;
}

// solve! with one step size per inner iteration for the whole (multi-rank) batch: forward_pass!'s Armijo loop on the host over the
// summed merit (see ilqr_hip.h). The loop is the reference's (src/solve.jl:88-129 around :1-54, src/forward_pass.jl:26-52) with
// the per-instance phases as stage launches.
int ilqr_solve_shared_step(ilqr_handle* h, ilqr_allreduce_sum_fn reduce, void* ctx, double* steps, int32_t steps_cap, int32_t* n_steps) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (!h->constrained) return fail(ILQR_ERR_INVALID, "shared-step mode: constrained solvers only");
    const int B = h->B, NS = ilqr::S_COUNT;
    std::vector<double> sc((size_t)B * NS);
    auto scalars = [&]() { return ilqr_get_buffer(h, "_scalars", sc.data()); };
    auto allsum = [&](double* v, int n) -> int {
        if (!reduce) return ILQR_OK;
        return reduce(v, n, ctx) == 0 ? ILQR_OK : fail(ILQR_ERR_INVALID, "shared-step mode: the reduction callback failed");
    };
    const ilqr_options opt = h->opt;
    int count = 0, rc;
    if ((rc = ilqr_run_stage(h, ILQR_STAGE_AL_BEGIN)) != ILQR_OK) return rc;
    for (int outer = 0; outer < opt.max_dual_updates; ++outer) {
        if ((rc = ilqr_run_stage(h, ILQR_STAGE_SS_INNER_BEGIN)) != ILQR_OK) return rc;
        for (int it = 0; it < opt.max_iterations; ++it) {
            if ((rc = scalars()) != ILQR_OK) return rc;
            std::vector<char> active(B);
            double n_active = 0.0;
            for (int b = 0; b < B; ++b) {
                active[b] = sc[(size_t)b * NS + ilqr::S_DONE] == 0.0 && sc[(size_t)b * NS + ilqr::S_INNER_DONE] == 0.0;
                n_active += active[b];
            }
            if ((rc = allsum(&n_active, 1)) != ILQR_OK) return rc;
            if (n_active == 0.0) break;
            double alpha = 1.0;
            int first = 1, accepted = 0, trials = 1;
            while (alpha >= opt.min_step_size && trials <= 25) {                          // src/forward_pass.jl:28-29
                if ((rc = ilqr_run_stage_param(h, ILQR_STAGE_SS_TRIAL, alpha, first)) != ILQR_OK) return rc;
                if ((rc = scalars()) != ILQR_OK) return rc;
                double sums[3] = {0.0, 0.0, 0.0};
                for (int b = 0; b < B; ++b)
                    if (active[b]) {
                        sums[0] += sc[(size_t)b * NS + ilqr::S_OBJECTIVE]; sums[1] += sc[(size_t)b * NS + ilqr::S_J_PREV];
                        sums[2] += sc[(size_t)b * NS + ilqr::S_DELTA];
                    }
                if ((rc = allsum(sums, 3)) != ILQR_OK) return rc;                          // the data-path collective: three doubles
                if (sums[0] <= sums[1] + 1.0e-4 * alpha * sums[2]) { accepted = 1; break; }  // (:44) NaN ⇒ reject
                alpha *= 0.5;                                                            // (:51)
                first = 0;
                ++trials;
            }
            if ((rc = ilqr_run_stage_param(h, ILQR_STAGE_SS_FINISH, alpha, accepted)) != ILQR_OK) return rc;
            if (steps && count < steps_cap) steps[count] = accepted ? alpha : 0.0;
            ++count;
        }
        if ((rc = ilqr_run_stage(h, ILQR_STAGE_SS_OUTER)) != ILQR_OK) return rc;
        if ((rc = scalars()) != ILQR_OK) return rc;
        double n_open = 0.0;
        for (int b = 0; b < B; ++b) n_open += sc[(size_t)b * NS + ilqr::S_DONE] == 0.0;
        if ((rc = allsum(&n_open, 1)) != ILQR_OK) return rc;
        if (n_open == 0.0) break;
    }
    if (n_steps) *n_steps = count;
    return ILQR_OK;
}

int ilqr_synchronize(ilqr_handle* h) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_synchronize(s); });
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ILQR_OK;
}

int ilqr_get_trajectory(ilqr_handle* h, double* x, double* u) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) {
        return ilqr_get_trajectory(s, x ? x + lo * (size_t)h->L.T * h->L.nx : nullptr, u ? u + lo * (size_t)(h->L.T - 1) * h->L.nu : nullptr); }, true);
    int rc = ILQR_OK;
    if (x) rc = copy_out(h, find_buffer(h, "nominal_states"), x);
    if (rc == ILQR_OK && u) rc = copy_out(h, find_buffer(h, "nominal_actions"), u);
    return rc;
}

int ilqr_get_policy(ilqr_handle* h, double* K, double* k) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) {
        const size_t N = (size_t)(h->L.T - 1);
        return ilqr_get_policy(s, K ? K + lo * N * h->L.nu * h->L.nx : nullptr, k ? k + lo * N * h->L.nu : nullptr); }, true);
    int rc = ILQR_OK;
    if (K) rc = copy_out(h, find_buffer(h, "K"), K);
    if (rc == ILQR_OK && k) rc = copy_out(h, find_buffer(h, "k"), k);
    return rc;
}

int ilqr_get_stats(ilqr_handle* h, ilqr_stats* st) {
    if (!h || !st) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_get_stats(s, st + lo); }, true);
    std::vector<double> s((size_t)h->B * ilqr::S_COUNT);
    int rc = copy_out(h, find_buffer(h, "_scalars"), s.data());
    if (rc != ILQR_OK) return rc;
    for (int b = 0; b < h->B; ++b) {
        const double* v = &s[(size_t)b * ilqr::S_COUNT];
        st[b].objective = v[ilqr::S_OBJECTIVE]; st[b].gradient_norm = v[ilqr::S_GRADIENT_NORM];
        st[b].max_violation = v[ilqr::S_MAX_VIOLATION]; st[b].step_size = v[ilqr::S_STEP_SIZE];
        st[b].iterations = (int32_t)v[ilqr::S_ITERATIONS]; st[b].outer_iterations = (int32_t)v[ilqr::S_OUTER_ITERATIONS];
        st[b].status = (int32_t)v[ilqr::S_STATUS]; st[b].potrf_info = (int32_t)v[ilqr::S_POTRF_INFO];
        st[b].rollouts = (int32_t)v[ilqr::S_ROLLOUTS]; st[b].reserved = 0;
    }
    return ILQR_OK;
}

int ilqr_buffer_len(const ilqr_handle* h, const char* name, size_t* len) {
    if (!h || !name || !len) return fail(ILQR_ERR_INVALID, "null argument");
    const BufferDesc* bd = find_buffer(h, name);
    if (!bd) bd = find_qbuffer(h, name);
    if (!bd) return fail(ILQR_ERR_INVALID, std::string("unknown buffer '") + name + "'");
    *len = (size_t)bd->len;
    return ILQR_OK;
}
int ilqr_enable_action_value_buffers(ilqr_handle* h) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_enable_action_value_buffers(s); });
    if (h->qv) return ILQR_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = (size_t)h->B * (size_t)h->QL.stride * 8;
    HIP_TRY(hipMalloc((void**)&h->qv, bytes));
    HIP_TRY(hipMemsetAsync(h->qv, 0, bytes, h->stream));
    return ILQR_OK;
}

int ilqr_scalar_slot(const char* name) {
    static const struct { const char* n; int i; } slots[] = {
        {"objective", ilqr::S_OBJECTIVE}, {"max_violation", ilqr::S_MAX_VIOLATION}, {"step_size", ilqr::S_STEP_SIZE},
        {"status", ilqr::S_STATUS}, {"iterations", ilqr::S_ITERATIONS}, {"gradient_norm", ilqr::S_GRADIENT_NORM},
        {"outer_iterations", ilqr::S_OUTER_ITERATIONS}, {"potrf_info", ilqr::S_POTRF_INFO}, {"rollouts", ilqr::S_ROLLOUTS},
        {"states_eq_nominal", ilqr::S_STATES_EQ_NOMINAL}, {"profile", ilqr::S_PROF}, {"done", ilqr::S_DONE},
        {"delta_grad_product", ilqr::S_DELTA}, {"trace_len", ilqr::S_TRACE_LEN}, {"count", ilqr::S_COUNT},
        {"obj_prev", ilqr::S_OBJ_PREV}, {"inner_done", ilqr::S_INNER_DONE}, {"j_prev", ilqr::S_J_PREV}, {"inner_it", ilqr::S_INNER_IT},
        {"resume", ilqr::S_RESUME}, {"literal_backward_passes", ilqr::S_LITERAL_PASSES},
        {"t_start", ilqr::S_T_START}, {"t_end", ilqr::S_T_END}, {"hw_id_wave0", ilqr::S_HW0}, {"hw_id_wave1", ilqr::S_HW1}, {"hw_id_wave2", ilqr::S_HW2}, {"hw_id_wave3", ilqr::S_HW3},
    };
    if (!name) return -1;
    for (auto& s_ : slots)
        if (!std::strcmp(s_.n, name)) return s_.i;
    return -1;
}

int ilqr_get_buffer(ilqr_handle* h, const char* name, double* out) {
    if (!h || !name || !out) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) {
        size_t len = 0;
        const int rc = ilqr_buffer_len(h, name, &len);
        if (rc != ILQR_OK) return rc;
        return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_get_buffer(s, name, out + lo * len); }, true);
    }
    if (const BufferDesc* qd = find_qbuffer(h, name)) {
        if (!h->qv) return fail(ILQR_ERR_INVALID, "action-value buffers are off: call ilqr_enable_action_value_buffers first");
        if (qd->len == 0) return ILQR_OK;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy2D(out, (size_t)qd->len * 8, h->qv + qd->offset, (size_t)h->QL.stride * 8,
                            (size_t)qd->len * 8, (size_t)h->B, hipMemcpyDeviceToHost));
        return ILQR_OK;
    }
    const BufferDesc* bd = find_buffer(h, name);
    if (!bd) return fail(ILQR_ERR_INVALID, std::string("unknown buffer '") + name + "'");
    return copy_out(h, bd, out);
}
int ilqr_set_buffer(ilqr_handle* h, const char* name, const double* in) {
    if (!h || !name || !in) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) {
        size_t len = 0;
        const int rc = ilqr_buffer_len(h, name, &len);
        if (rc != ILQR_OK) return rc;
        return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_set_buffer(s, name, in + lo * len); }, true);
    }
    const BufferDesc* bd = find_buffer(h, name);
    if (!bd) return fail(ILQR_ERR_INVALID, std::string("unknown buffer '") + name + "'");
    return copy_in(h, bd, in);
}

int ilqr_set_kernel_variant(ilqr_handle* h, int32_t variant) {
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_set_kernel_variant(s, variant); });
    if (!h || variant < 0 || variant > 6) return fail(ILQR_ERR_INVALID, "variant must be 0 (auto), 1 (latency), 2 (throughput), 3 (packed), 4 (one wave per instance of a large model), 5 / 6 (packed with one / two waves per pack)");
    if (variant == 4 && h->vt->launch_solve_mid == nullptr)
        return fail(ILQR_ERR_INVALID, "the one-wave variant exists for large models with nx, nu <= 16 only");
    if ((variant == 3 || variant == 5 || variant == 6) && h->vt->launch_solve_packed == nullptr)
        return fail(ILQR_ERR_INVALID, "the packed variant exists for small models (nx, nu <= 4) only");
    if ((variant == 1 || variant == 2) && !h->lds_fits)
        return fail(ILQR_ERR_LDS, "this horizon exceeds the LDS-resident kernels: only the packed variant can run it");
    if (variant == 2 && h->vt->launch_solve_slim == nullptr)
        return fail(ILQR_ERR_INVALID, "the throughput variant exists for small models (nx, nu <= 4) only");
    h->variant = variant;
    return ILQR_OK;
}

// what ilqr_solve launches for this handle as it stands (variant, batch, horizon): the auto rules of ilqr_solve, stated once more
int ilqr_resolved_kernel_variant(ilqr_handle* h, int32_t* variant) {
    if (!h || !variant) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return ilqr_resolved_kernel_variant(h->shards[0], variant);
    const bool can_pack = h->vt->launch_solve_packed != nullptr;
    const bool packed = can_pack && (h->variant == 3 || h->variant == 5 || h->variant == 6 || !h->lds_fits || (h->variant == 0 && h->B > h->num_simds));
    const bool slim = !packed && h->vt->launch_solve_slim != nullptr && (h->variant == 2 || (h->variant == 0 && h->B > h->num_simds));
    if (packed) {
        const int packs = (h->B + 3) / 4, cus = std::max(1, h->num_simds / 4), per_cu = (packs + cus - 1) / cus;
        *variant = (h->variant != 5 && packed2_fits(h->vt->packed2_lds_bytes, per_cu)) ? 6 : 5;      // the launcher's own rule (packed2_fits)
    } else if (slim) *variant = 2;
    else if (use_mid(h)) *variant = 4;
    else *variant = 1;
    return ILQR_OK;
}

int ilqr_set_handover(ilqr_handle* h, int32_t outer) {
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_set_handover(s, outer); });
    if (!h || outer < -1 || outer == 1) return fail(ILQR_ERR_INVALID, "hand-over: -1 (auto), 0 (off) or the outer iteration (>= 2) from which stragglers leave the packed kernel");
    h->handover = outer;
    return ILQR_OK;
}

int ilqr_set_handover_live(ilqr_handle* h, int32_t live) {
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_set_handover_live(s, live); });
    if (!h || live < -1) return fail(ILQR_ERR_INVALID, "hand-over by head count: -1 (auto), 0 (off) or the number of surviving instances at which they leave the packed kernel");
    h->handover_live = live;
    return ILQR_OK;
}

int ilqr_set_handover_mark(ilqr_handle* h, int32_t rejected) {
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_set_handover_mark(s, rejected); });
    if (!h || rejected < -1) return fail(ILQR_ERR_INVALID, "straggler mark: -1 (auto), 0 (never) or the number of rejected line-search trials above the batch's mean at which an instance leaves the packed kernel at once");
    h->handover_mark = rejected;
    return ILQR_OK;
}

int ilqr_get_handover_stats(ilqr_handle* h, int32_t* queued, int32_t* marked) {
    if (!h || !queued || !marked) return fail(ILQR_ERR_INVALID, "null argument");
    *queued = 0; *marked = 0;
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) {
        int32_t q = 0, m = 0;
        const int rc = ilqr_get_handover_stats(s, &q, &m);
        *queued += q; *marked += m;
        return rc; });
    if (h->pool == nullptr || !h->pool_valid) return ILQR_OK;      // the last solve did not go through the queue: 0 / 0
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int words[ilqr::POOL_Q];
    HIP_TRY(hipMemcpy(words, h->pool, sizeof(words), hipMemcpyDeviceToHost));
    *queued = words[ilqr::POOL_TAIL]; *marked = words[ilqr::POOL_MARKED];
    return ILQR_OK;
}

int ilqr_enable_trace(ilqr_handle* h, int32_t capacity) {
    if (!h || capacity < 0) return fail(ILQR_ERR_INVALID, "bad argument");
    if (SHARDED(h)) { h->trace_cap = capacity; return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_enable_trace(s, capacity); }); }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->trace) { HIP_TRY(hipFree(h->trace)); h->trace = nullptr; }
    h->trace_cap = capacity;
    if (capacity > 0) {
        const size_t bytes = (size_t)h->B * capacity * ilqr::TRACE_W * 8;
        HIP_TRY(hipMalloc((void**)&h->trace, bytes));
        HIP_TRY(hipMemset(h->trace, 0, bytes));
    }
    return ILQR_OK;
}

int ilqr_get_trace(ilqr_handle* h, double* out) {
    if (!h || !out) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t lo) { return ilqr_get_trace(s, out + lo * (size_t)h->trace_cap * ilqr::TRACE_W); }, true);
    if (!h->trace) return fail(ILQR_ERR_INVALID, "trace not enabled");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->trace, (size_t)h->B * h->trace_cap * ilqr::TRACE_W * 8, hipMemcpyDeviceToHost));
    return ILQR_OK;
}

int ilqr_get_stream(ilqr_handle* h, void** s) {
    if (!h || !s) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) return fail(ILQR_ERR_INVALID, "a sharded handle has one stream per device: synchronise with ilqr_synchronize");
    *s = (void*)h->stream;
    return ILQR_OK;
}

int ilqr_timing_reset(ilqr_handle* h) {
    if (!h) return fail(ILQR_ERR_INVALID, "null handle");
    if (SHARDED(h)) return each_shard(h, [&](ilqr_handle* s, size_t) { return ilqr_timing_reset(s); });
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& p : h->timing) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    h->timing.clear();
    return ILQR_OK;
}

int ilqr_timing_get(ilqr_handle* h, double* ms_avg, int32_t* launches) {
    if (!h || !ms_avg) return fail(ILQR_ERR_INVALID, "null argument");
    if (SHARDED(h)) {          // the devices run side by side: the slowest shard's kernel time
        double worst = 0.0; int32_t nl = 0;
        const int rc = each_shard(h, [&](ilqr_handle* s, size_t) {
            double ms = 0.0; int32_t n_ = 0;
            const int r = ilqr_timing_get(s, &ms, &n_);
            if (ms > worst) worst = ms;
            nl = n_;
            return r; });
        *ms_avg = worst;
        if (launches) *launches = nl;
        return rc;
    }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double total = 0.0;
    for (auto& p : h->timing) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        total += ms;
    }
    *ms_avg = h->timing.empty() ? 0.0 : total / (double)h->timing.size();
    if (launches) *launches = (int32_t)h->timing.size();
    return ILQR_OK;
}

}  // extern "C"
