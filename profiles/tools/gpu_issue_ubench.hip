// issue-cost microbenchmark for a lone wavefront on gfx950 (one wave per workgroup, few workgroups)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // dependent v_fma_f32 chain
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(b), "v"(c));
        } else if (MODE == 1) {  // 4 independent chains v_fma
            asm volatile(REP8(REP8("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if (MODE == 2) {  // dependent v_pk_fma_f32 chain
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n\t") : "+v"(p0) : "v"(pb), "v"(pc));
        } else if (MODE == 3) {  // 4 independent pk chains
            asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (MODE == 4) {  // dependent v_add_f32_dpp with s_nop 1
            asm volatile(REP64("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(a0));
        } else if (MODE == 5) {  // 4 independent add_dpp, no nops
            asm volatile(REP8(REP8("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (MODE == 6) {  // s_nop 0 only
            asm volatile(REP64("s_nop 0\n\t"));
        } else if (MODE == 7) {  // s_nop 1 only
            asm volatile(REP64("s_nop 1\n\t"));
        } else if (MODE == 8) {  // dependent v_cndmask chain (vcc)
            asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(a0) : "v"(b) : "vcc");
        } else if (MODE == 9) {  // v_mov_dpp + v_add (dependent)
            asm volatile(REP8(REP8("s_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32 %0, %0, %1\n\t")) : "+v"(a0), "+v"(a1));
        } else if (MODE == 10) {  // pk_fma dependent alternating with independent scalar fma
            asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5\n\t")) : "+v"(p0), "+v"(a2) : "v"(pb), "v"(pc), "v"(b), "v"(c));
        } else if (MODE == 11) {  // pk_mul dependent
            asm volatile(REP64("v_pk_mul_f32 %0, %0, %1\n\t") : "+v"(p0) : "v"(pb));
        } else if (MODE == 12) {  // pk_fma with op_sel_hi broadcast, 4 independent
            asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %4, %5 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %2, %2, %4, %5 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %3, %3, %4, %5 op_sel_hi:[1,0,1]\n\t")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (MODE == 13) {  // v_rcp dependent
            asm volatile(REP64("v_rcp_f32 %0, %0\n\t") : "+v"(a0));
        } else if (MODE == 14) {  // s_and_b64 dependent (SALU)
            asm volatile(REP64("s_and_b64 s[20:21], s[20:21], exec\n\t") ::: "s20", "s21");
        } else if (MODE == 15) {  // alternating SALU / VALU independent
            asm volatile(REP8(REP8("s_and_b64 s[20:21], s[20:21], exec\n\tv_fma_f32 %0, %0, %1, %2\n\t")) : "+v"(a0) : "v"(b), "v"(c) : "s20", "s21");
        } else if (MODE == 16) {  // 2 independent pk chains
            asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\t")) : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));
        } else if (MODE == 17) {  // v_max3 dependent
            asm volatile(REP64("v_max3_f32 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(b), "v"(c));
        } else if (MODE == 18) {  // v_accvgpr_write/read pair
            asm volatile(REP8(REP8("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0\n\t")) : "+v"(a0) :: "a0");
        } else if (MODE == 19) {  // v_cmp + cndmask e64 dependent
            asm volatile(REP8(REP8("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\t")) : "+v"(a0) : "v"(b) : "s20", "s21");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p2.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int ninstr_per_iter, int waves_per_block = 1) {
    const int blocks = 64, iters = 200;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 64 * 4 * waves_per_block); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double n = double(ninstr_per_iter) * iters;
    printf("%-44s ns/instr %.3f  (clock64 ticks/instr %.3f)\n", name, ms * 1e6 / n, h[0] / n);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_fma_f32 dependent", 64);
    run<1>("v_fma_f32 4 independent chains", 256);
    run<2>("v_pk_fma_f32 dependent", 64);
    run<16>("v_pk_fma_f32 2 independent chains", 128);
    run<3>("v_pk_fma_f32 4 independent chains", 256);
    run<12>("v_pk_fma_f32 op_sel bcast, 4 indep", 256);
    run<10>("pk_fma dep + scalar fma dep interleaved", 128);
    run<11>("v_pk_mul_f32 dependent", 64);
    run<4>("s_nop 1 + v_add_f32_dpp dependent (2 instr)", 128);
    run<5>("v_add_f32_dpp 4 independent", 256);
    run<9>("s_nop1 + v_mov_dpp + v_add dependent (3)", 192);
    run<6>("s_nop 0", 64);
    run<7>("s_nop 1", 64);
    run<8>("v_cndmask vcc dependent", 64);
    run<19>("v_cmp_e64 + v_cndmask_e64 dependent (2)", 128);
    run<13>("v_rcp_f32 dependent", 64);
    run<17>("v_max3_f32 dependent", 64);
    run<14>("s_and_b64 dependent", 64);
    run<15>("s_and_b64 + v_fma alternating (2)", 128);
    run<18>("accvgpr write+read (2)", 128);
    return 0;
}
