// Issue cost of single instructions on gfx950 at full occupancy (8 waves per SIMD): SIMD cycles per wave-instruction,
// relative to v_add_f32.  Development aid for the K1 look-up loop (which instruction mix is cheapest), not product code.
//   hipcc --offload-arch=gfx950 -O2 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

#define KERNEL(name, body, ...)                                                                   \
    __global__ __launch_bounds__(256) void name(float* out, int iters)                             \
    {                                                                                              \
        float a = threadIdx.x * 1.0f, b = 1.5f, c = 2.5f, d = 3.5f;                                \
        unsigned long long m0 = 0, m1 = 0, m2 = 0;                                                 \
        for (int i = 0; i < iters; i++) {                                                          \
            asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(m0), "+s"(m1), "+s"(m2) : : __VA_ARGS__); \
        }                                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (float)(m0 + m1 + m2);        \
    }

// operands: %0..%3 vgprs a,b,c,d ; %4..%6 sgpr pairs
KERNEL(k_add_dep,    "v_add_f32 %0, %0, %1\n\t", "vcc")
KERNEL(k_add_ind,    "v_add_f32 %0, %1, %2\n\t v_add_f32 %3, %1, %2\n\t", "vcc")
KERNEL(k_min3,       "v_min3_u32 %0, %0, %1, %2\n\t", "vcc")
KERNEL(k_cmpx_add,   "v_cmpx_eq_u32 %1, %1\n\t v_add_f32 %0, %0, %1\n\t", "vcc")
KERNEL(k_cmpx_add_smov, "v_cmpx_eq_u32 %1, %1\n\t v_add_f32 %0, %0, %1\n\t s_mov_b64 exec, %4\n\t", "vcc")
KERNEL(k_cmp_sgpr,   "v_cmp_eq_u32_e64 %5, %1, %1\n\t", "vcc")
KERNEL(k_cmp_vcc,    "v_cmp_eq_u32_e32 vcc, %1, %1\n\t", "vcc")
KERNEL(k_cnd_sgpr,   "v_cndmask_b32_e64 %0, 0, %1, %5\n\t", "vcc")
KERNEL(k_cmp_cnd_add, "v_cmp_eq_u32_e64 %5, %1, %1\n\t v_cmp_eq_u32_e64 %6, %2, %2\n\t s_nop 0\n\t v_cndmask_b32_e64 %3, 0, %1, %5\n\t v_add_f32 %0, %0, %3\n\t", "vcc")
KERNEL(k_dpp_nop,    "s_nop 1\n\t v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t", "vcc")
KERNEL(k_dpp_fill,   "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t v_add_f32 %1, %1, %2\n\t v_add_f32 %3, %3, %2\n\t", "vcc")
KERNEL(k_snop,       "s_nop 0\n\t", "vcc")
KERNEL(k_smov,       "s_mov_b64 %5, %4\n\t", "vcc")
KERNEL(k_smov_exec,  "s_mov_b64 exec, %4\n\t", "vcc")
KERNEL(k_add_smov,   "v_add_f32 %0, %0, %1\n\t s_mov_b64 %5, %4\n\t", "vcc")
KERNEL(k_add_2smov,  "v_add_f32 %0, %0, %1\n\t s_mov_b64 %5, %4\n\t s_mov_b64 %6, %4\n\t", "vcc")
KERNEL(k_mul_legacy, "v_mul_legacy_f32 %0, %0, %1\n\t", "vcc")
KERNEL(k_cvt_rpi,    "v_cvt_rpi_i32_f32 %0, %0\n\t", "vcc")
KERNEL(k_mad24,      "v_mad_i32_i24 %0, %0, %1, %2\n\t", "vcc")
KERNEL(k_readlane,   "v_readlane_b32 s60, %0, 63\n\t", "vcc", "s60")
KERNEL(k_exec_iter,  "s_mov_b64 exec, %4\n\t v_min3_u32 %3, %0, %1, %2\n\t v_cmpx_eq_u32 %3, %0\n\t v_add_f32 %0, %0, %1\n\t s_mov_b64 exec, %4\n\t v_cmpx_eq_u32 %3, %1\n\t v_add_f32 %1, %1, %2\n\t s_mov_b64 exec, %4\n\t v_cmpx_eq_u32 %3, %2\n\t v_add_f32 %2, %2, %0\n\t", "vcc")
KERNEL(k_vcc_iter,   "v_min3_u32 %3, %0, %1, %2\n\t v_cmp_eq_u32_e64 %4, %3, %0\n\t v_cmp_eq_u32_e64 %5, %3, %1\n\t v_cmp_eq_u32_e64 %6, %3, %2\n\t v_cndmask_b32_e64 v40, 0, %1, %4\n\t v_cndmask_b32_e64 v41, 0, %2, %5\n\t v_cndmask_b32_e64 v42, 0, %0, %6\n\t v_add_f32 %0, %0, v40\n\t v_add_f32 %1, %1, v41\n\t v_add_f32 %2, %2, v42\n\t", "vcc", "v40", "v41", "v42")
KERNEL(k_rcp,        "v_rcp_f32 %0, %0\n\t", "vcc")
KERNEL(k_fma,        "v_fma_f32 %0, %0, %1, %2\n\t", "vcc")
KERNEL(k_pk_fma,     "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[40:41]\n\t", "vcc", "v40", "v41", "v42", "v43", "v44", "v45")
KERNEL(k_pk_mul,     "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n\t", "vcc", "v40", "v41", "v42", "v43", "v44", "v45")
KERNEL(k_pk_add,     "v_pk_add_f32 v[40:41], v[40:41], v[44:45]\n\t", "vcc", "v40", "v41", "v42", "v43", "v44", "v45")
KERNEL(k_exp,        "v_exp_f32 %0, %0\n\t", "vcc")
KERNEL(k_floor,      "v_floor_f32 %0, %0\n\t", "vcc")
KERNEL(k_dsread128,  "ds_read_b128 v[40:43], %0\n\t s_waitcnt lgkmcnt(0)\n\t", "vcc", "v40", "v41", "v42", "v43")
KERNEL(k_branch,     "s_cmp_eq_u32 s60, 0\n\t s_cbranch_scc1 1f\n\t 1:\n\t", "vcc", "s60", "scc")
KERNEL(k_branch_taken, "s_branch 1f\n\t s_nop 0\n\t 1:\n\t", "vcc")

struct T { const char* name; void (*k)(float*, int); int per; };

int main()
{
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    const int iters = 2000, blocks = 256 * 8;           // 8 blocks of 256 threads per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<T> tests = {
        {"v_pk_fma_f32", k_pk_fma, 1}, {"v_pk_mul_f32", k_pk_mul, 1}, {"v_pk_add_f32", k_pk_add, 1}, {"v_exp_f32", k_exp, 1}, {"v_floor_f32", k_floor, 1},
        {"v_add_f32 dependent", k_add_dep, 1}, {"v_add_f32 x2 independent", k_add_ind, 2}, {"v_min3_u32", k_min3, 1},
        {"v_cmpx + v_add (2)", k_cmpx_add, 2}, {"v_cmpx + v_add + s_mov exec (3)", k_cmpx_add_smov, 3},
        {"v_cmp_e64 -> sgpr", k_cmp_sgpr, 1}, {"v_cmp_e32 -> vcc", k_cmp_vcc, 1}, {"v_cndmask_e64 sgpr mask", k_cnd_sgpr, 1},
        {"cmp,cmp,nop,cnd,add (5)", k_cmp_cnd_add, 5},
        {"s_nop1 + dpp (2)", k_dpp_nop, 2}, {"dpp + 2 fillers (3)", k_dpp_fill, 3}, {"s_nop 0", k_snop, 1}, {"s_mov_b64", k_smov, 1},
        {"s_mov_b64 exec", k_smov_exec, 1}, {"v_add + s_mov (2)", k_add_smov, 2}, {"v_add + 2 s_mov (3)", k_add_2smov, 3},
        {"v_mul_legacy_f32", k_mul_legacy, 1}, {"v_cvt_rpi_i32_f32", k_cvt_rpi, 1}, {"v_mad_i32_i24", k_mad24, 1}, {"v_readlane", k_readlane, 1},
        {"EXEC-form iteration (7V+3S)", k_exec_iter, 10}, {"VCC-form iteration (10V)", k_vcc_iter, 10},
        {"v_rcp_f32", k_rcp, 1}, {"v_fma_f32", k_fma, 1}, {"s_cmp + s_cbranch not taken (2)", k_branch, 2}, {"s_branch taken + skipped nop (1)", k_branch_taken, 1},
    };
    double base = 0;
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD = 8 waves * iters * 16 * per ; report ns per "body" per SIMD
        double body_ns = ms * 1e6 / (8.0 * iters * 16);
        if (base == 0) base = body_ns;
        printf("%-36s %7.3f ms  %6.2f ns/body/SIMD  = %5.2f x v_add  (%.2f per instruction)\n", t.name, ms, body_ns, body_ns / base, body_ns / base / t.per);
    }
    return 0;
}
