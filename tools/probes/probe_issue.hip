// Probe: what ONE wave can issue per shader clock on gfx950, by instruction class, alone on its SIMD and next to other waves.
// Each body is 32 INDEPENDENT instructions (distinct destination registers, constant sources) in inline assembly, repeated n times;
// time from s_memtime. The solve kernels' critical wave is a single instruction stream: this is its speed limit per class.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/probe_issue.hip -o tools/probes/probe_issue && tools/probes/probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// every variant: 32 instructions per loop trip; registers v[10:41] as 16 fp64 destinations, sources v[2:9]
#define KERNEL(name, asm32)                                                                                        \
    __global__ void name(double* out, int n) {                                                                     \
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;                                              \
        asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v4, %2\n v_mov_b32 v5, %3\n"                 \
                     "v_mov_b32 v6, %4\n v_mov_b32 v7, %5\n"                                                       \
                     :: "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(__double2loint(b)), "v"(__double2hiint(b)),      \
                        "v"(__double2loint(c)), "v"(__double2hiint(c)) : "v2", "v3", "v4", "v5", "v6", "v7");      \
        __shared__ double lds_[64]; lds_[threadIdx.x & 63] = 0.0;                                                  \
        asm volatile("v_mov_b32 v8, 0" ::: "v8");                                                                  \
        long long t0 = clock64();                                                                                  \
        for (int i = 0; i < n; ++i) {                                                                              \
            asm volatile(asm32 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", \
                         "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40",    \
                         "v41", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc");                    \
        }                                                                                                          \
        long long t1 = clock64();                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (32.0 * n);                          \
    }

#define FMA64_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[6:7]\n"
#define MUL64_4 "v_mul_f64 v[10:11], v[2:3], v[4:5]\n v_mul_f64 v[12:13], v[2:3], v[4:5]\n v_mul_f64 v[14:15], v[2:3], v[4:5]\n v_mul_f64 v[16:17], v[2:3], v[4:5]\n"
#define ADD64_4 "v_add_f64 v[10:11], v[2:3], v[4:5]\n v_add_f64 v[12:13], v[2:3], v[4:5]\n v_add_f64 v[14:15], v[2:3], v[4:5]\n v_add_f64 v[16:17], v[2:3], v[4:5]\n"
#define FMA32_4 "v_fma_f32 v10, v2, v4, v6\n v_fma_f32 v11, v2, v4, v6\n v_fma_f32 v12, v2, v4, v6\n v_fma_f32 v13, v2, v4, v6\n"
#define MOV32_4 "v_mov_b32 v10, v2\n v_mov_b32 v11, v3\n v_mov_b32 v12, v4\n v_mov_b32 v13, v5\n"
#define MOV64_4 "v_mov_b64 v[10:11], v[2:3]\n v_mov_b64 v[12:13], v[4:5]\n v_mov_b64 v[14:15], v[6:7]\n v_mov_b64 v[16:17], v[2:3]\n"
#define CND_4 "v_cndmask_b32 v10, v2, v4, vcc\n v_cndmask_b32 v11, v3, v5, vcc\n v_cndmask_b32 v12, v2, v4, vcc\n v_cndmask_b32 v13, v3, v5, vcc\n"
#define DPP_4 "v_mov_b32_dpp v10, v2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v11, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v12, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v13, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define RDL_4 "v_readlane_b32 s20, v2, 1\n v_readlane_b32 s21, v3, 2\n v_readlane_b32 s22, v4, 3\n v_readlane_b32 s23, v5, 4\n"
#define SMOV_4 "s_mov_b32 s20, s21\n s_mov_b32 s22, s23\n s_mov_b32 s24, s25\n s_mov_b32 s26, s27\n"
#define SNOP_4 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define XOR_4 "v_xor_b32 v10, v2, v4\n v_xor_b32 v11, v3, v5\n v_xor_b32 v12, v2, v6\n v_xor_b32 v13, v3, v7\n"
// mixes, 4 instructions each
#define FMA_SMOV_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n s_mov_b32 s20, s21\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n s_mov_b32 s22, s23\n"
#define FMA_MOV_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_mov_b32 v20, v2\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_mov_b32 v21, v3\n"
#define FMA_CND_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_cndmask_b32 v20, v2, v4, vcc\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_cndmask_b32 v21, v3, v5, vcc\n"
#define FMA_RDL_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_readlane_b32 s20, v2, 1\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_readlane_b32 s21, v3, 2\n"
#define FMA_SGPRSRC_4 "v_fma_f64 v[10:11], s[20:21], v[4:5], v[6:7]\n v_fma_f64 v[12:13], s[20:21], v[4:5], v[6:7]\n v_fma_f64 v[14:15], s[20:21], v[4:5], v[6:7]\n v_fma_f64 v[16:17], s[20:21], v[4:5], v[6:7]\n"
#define FMA_LIT_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], 0.5\n v_fma_f64 v[12:13], v[2:3], v[4:5], 1.0\n v_fma_f64 v[14:15], v[2:3], v[4:5], 2.0\n v_fma_f64 v[16:17], v[2:3], v[4:5], 4.0\n"
// dependent chains for reference
#define FMA64_DEP_4 "v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n"
#define MOV32_DEP_4 "v_mov_b32 v10, v10\n v_mov_b32 v10, v10\n v_mov_b32 v10, v10\n v_mov_b32 v10, v10\n"
#define CND_DEP_4 "v_cndmask_b32 v10, v10, v4, vcc\n v_cndmask_b32 v10, v10, v4, vcc\n v_cndmask_b32 v10, v10, v4, vcc\n v_cndmask_b32 v10, v10, v4, vcc\n"
#define FMA32_DEP_4 "v_fma_f32 v10, v10, v4, v6\n v_fma_f32 v10, v10, v4, v6\n v_fma_f32 v10, v10, v4, v6\n v_fma_f32 v10, v10, v4, v6\n"
#define MUL64_DEP_4 "v_mul_f64 v[10:11], v[10:11], v[4:5]\n v_mul_f64 v[10:11], v[10:11], v[4:5]\n v_mul_f64 v[10:11], v[10:11], v[4:5]\n v_mul_f64 v[10:11], v[10:11], v[4:5]\n"
#define ADD64_DEP_4 "v_add_f64 v[10:11], v[10:11], v[4:5]\n v_add_f64 v[10:11], v[10:11], v[4:5]\n v_add_f64 v[10:11], v[10:11], v[4:5]\n v_add_f64 v[10:11], v[10:11], v[4:5]\n"
// fp64 result consumed by a 32-bit op and back (the select / sign fix-up pattern)
#define FMA_THEN_CND_4 "v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n v_cndmask_b32 v10, v10, v4, vcc\n v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n v_cndmask_b32 v11, v11, v5, vcc\n"

// more mixes: scalar work hidden behind vector work?  (4 instructions each)
#define FMA3_SMOV_4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n s_mov_b32 s20, s21\n"
#define FMADEP_SMOV_4 "v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n s_mov_b32 s20, s21\n v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n s_mov_b32 s22, s23\n"
#define FMADEP_SNOP_4 "v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n s_nop 0\n v_fma_f64 v[10:11], v[10:11], v[4:5], v[6:7]\n s_nop 0\n"
#define MFMA_4 "v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_mfma_f64_4x4x4_4b_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_mfma_f64_4x4x4_4b_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n v_mfma_f64_4x4x4_4b_f64 v[16:17], v[2:3], v[4:5], v[6:7]\n"
#define MFMA_DEPC_4 "v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n"
#define MFMA_DEPB_4 "v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[10:11], 0\n s_nop 7\n v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[10:11], 0\n s_nop 7\n"
#define MFMA_FMA_4 "v_mfma_f64_4x4x4_4b_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_mfma_f64_4x4x4_4b_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[6:7]\n"
#define DSW_4 "ds_write_b64 v8, v[2:3]\n ds_write_b64 v8, v[4:5] offset:8\n ds_write_b64 v8, v[6:7] offset:16\n ds_write_b64 v8, v[2:3] offset:24\n"
#define DSR_4 "ds_read_b64 v[10:11], v8\n ds_read_b64 v[12:13], v8 offset:8\n ds_read_b64 v[14:15], v8 offset:16\n ds_read_b64 v[16:17], v8 offset:24\n"
#define DSR_FMA_4 "ds_read_b64 v[10:11], v8\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[6:7]\n"
#define BCAST64_4 "v_mov_b64_dpp v[10:11], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[12:13], v[4:5] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[14:15], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[16:17], v[2:3] row_newbcast:9 row_mask:0xf bank_mask:0xf\n"

#define X8(s) s s s s s s s s
KERNEL(k_fma64, X8(FMA64_4))
KERNEL(k_mul64, X8(MUL64_4))
KERNEL(k_add64, X8(ADD64_4))
KERNEL(k_fma32, X8(FMA32_4))
KERNEL(k_mov32, X8(MOV32_4))
KERNEL(k_mov64, X8(MOV64_4))
KERNEL(k_cnd, X8(CND_4))
KERNEL(k_dpp, X8(DPP_4))
KERNEL(k_rdl, X8(RDL_4))
KERNEL(k_smov, X8(SMOV_4))
KERNEL(k_snop, X8(SNOP_4))
KERNEL(k_xor, X8(XOR_4))
KERNEL(k_fma_smov, X8(FMA_SMOV_4))
KERNEL(k_fma_mov, X8(FMA_MOV_4))
KERNEL(k_fma_cnd, X8(FMA_CND_4))
KERNEL(k_fma_rdl, X8(FMA_RDL_4))
KERNEL(k_fma_sgpr, X8(FMA_SGPRSRC_4))
KERNEL(k_fma_lit, X8(FMA_LIT_4))
KERNEL(k_fma64_dep, X8(FMA64_DEP_4))
KERNEL(k_mul64_dep, X8(MUL64_DEP_4))
KERNEL(k_add64_dep, X8(ADD64_DEP_4))
KERNEL(k_mov32_dep, X8(MOV32_DEP_4))
KERNEL(k_cnd_dep, X8(CND_DEP_4))
KERNEL(k_fma32_dep, X8(FMA32_DEP_4))
KERNEL(k_fma_then_cnd, X8(FMA_THEN_CND_4))

KERNEL(k_fma3_smov, X8(FMA3_SMOV_4))
KERNEL(k_fmadep_smov, X8(FMADEP_SMOV_4))
KERNEL(k_fmadep_snop, X8(FMADEP_SNOP_4))
KERNEL(k_mfma, X8(MFMA_4))
KERNEL(k_mfma_depc, X8(MFMA_DEPC_4))
KERNEL(k_mfma_depb, X8(MFMA_DEPB_4))
KERNEL(k_mfma_fma, X8(MFMA_FMA_4))
KERNEL(k_dsw, X8(DSW_4))
KERNEL(k_dsr, X8(DSR_4) "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_dsr_fma, X8(DSR_FMA_4) "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_bcast64, X8(BCAST64_4))

typedef void (*kfn)(double*, int);
int main() {
    double* d; hipMalloc(&d, 1024);
    struct { const char* name; kfn f; } ks[] = {
        {"v_fma_f64 independent", k_fma64}, {"v_mul_f64 independent", k_mul64}, {"v_add_f64 independent", k_add64},
        {"v_fma_f64 literal addend", k_fma_lit}, {"v_fma_f64 SGPR source", k_fma_sgpr},
        {"v_fma_f32 independent", k_fma32}, {"v_mov_b32 independent", k_mov32}, {"v_mov_b64 independent", k_mov64},
        {"v_cndmask_b32 independent", k_cnd}, {"v_xor_b32 independent", k_xor}, {"v_mov_b32_dpp independent", k_dpp},
        {"v_readlane_b32 independent", k_rdl}, {"s_mov_b32", k_smov}, {"s_nop 0", k_snop},
        {"mix fma64 / s_mov", k_fma_smov}, {"mix fma64 / v_mov_b32", k_fma_mov}, {"mix fma64 / v_cndmask", k_fma_cnd}, {"mix fma64 / v_readlane", k_fma_rdl},
        {"v_fma_f64 dependent", k_fma64_dep}, {"v_mul_f64 dependent", k_mul64_dep}, {"v_add_f64 dependent", k_add64_dep},
        {"v_fma_f32 dependent", k_fma32_dep}, {"v_mov_b32 dependent", k_mov32_dep}, {"v_cndmask_b32 dependent", k_cnd_dep},
        {"fma64 -> cndmask -> fma64 dependent", k_fma_then_cnd},
        {"mix 3 fma64 / 1 s_mov", k_fma3_smov}, {"mix dependent fma64 / salu", k_fmadep_smov}, {"mix dependent fma64 / s_nop 0", k_fmadep_snop},
        {"v_mfma_f64_4x4x4 independent", k_mfma}, {"v_mfma_f64_4x4x4 dependent (C)", k_mfma_depc},
        {"v_mfma_f64_4x4x4 dependent (B) + s_nop 7", k_mfma_depb}, {"mix mfma / fma64", k_mfma_fma},
        {"ds_write_b64", k_dsw}, {"ds_read_b64 (32 then wait)", k_dsr}, {"mix ds_read / 3 fma64", k_dsr_fma},
        {"v_mov_b64_dpp row_newbcast", k_bcast64}};
    const int n = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("clk per instruction seen by ONE wave (s_memtime); and wall time of the launch per instruction of a wave; columns: 1 wave on the chip | 1 | 2 | 4 waves per SIMD\n");
    for (auto& k : ks) {
        printf("%-40s", k.name); fflush(stdout);
        for (int blocks : {1, 1024, 2048, 4096}) {
            k.f<<<blocks, 64>>>(d, n);                                    // warm
            hipEventRecord(e0);
            k.f<<<blocks, 64>>>(d, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf(" %6.2f (%6.2f ns)", h, ms * 1e6 / (32.0 * n));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
