// Issue-rate probe for the instruction classes of the marching cost-volume kernel (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o tools/probes/valu_rates && tools/probes/valu_rates
// Every kernel runs a long dependent-free stream of ONE instruction class on 8 independent register chains per lane;
// reported: cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD (s_memtime around the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float k = 1.0001f, c = 0.5f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (KIND == 1) {  // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(c));)
        } else if (KIND == 2) {  // v_add_f32_dpp wave_shr:1 (source of each add is another chain, written 4 instructions earlier)
            REP16(asm volatile("s_nop 1\n v_add_f32_dpp %0, %4, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %1, %5, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %2, %6, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %3, %7, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %0, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %5, %1, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %6, %2, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %7, %3, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {  // v_add_f32_dpp row_shr:1 (within rows of 16 lanes)
            REP16(asm volatile("s_nop 1\n v_add_f32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %1, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %2, %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %3, %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %5, %1, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %6, %2, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %7, %3, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) {  // v_pk_add_f32 on register pairs (a0,a1) ... : 4 instructions = 8 float adds
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
            const f2 kk = {k, k};
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));)
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 5) {  // v_pk_fma_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
            const f2 kk = {k, k}, cc = {c, c};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk), "v"(cc));)
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == 6) {  // v_rcp_f32
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                               "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) {  // v_div_fixup_f32 (VOP3, 3 sources)
            REP16(asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n"
                               "v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(c));)
        } else if (KIND == 8) {  // v_mov_b32_dpp wave_shr:1
            REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %4, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %5, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %6, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %7, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 9) {  // v_cndmask_b32 (VCC) + v_cmp: pairs
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");)
        } else if (KIND == 10) {  // v_mul_f32 then dependent chain of 1: latency-bound single chain
            REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(k));)
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int insts_per_iter, float* out, long long* cyc) {
    const int iters = 200;
    std::printf("%-28s", name);
    for (int wps : {1, 2, 4}) {                       // waves per SIMD: 256-thread blocks = 4 waves = 1 per SIMD
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (long long v : h) s += (double)v;
        const double per_wave = s / blocks / ((double)iters * insts_per_iter);      // cycles per instruction as one wave sees it
        std::printf("  %dw/SIMD: %6.2f cyc/inst/wave = %5.2f cyc/inst/SIMD", wps, per_wave, per_wave / wps);
    }
    std::printf("\n");
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    hipMalloc(&cyc, 1024 * sizeof(long long));
    run<0>("v_add_f32", 128, out, cyc);
    run<1>("v_fma_f32", 128, out, cyc);
    run<2>("v_add_f32_dpp wave_sh*:1", 128, out, cyc);
    run<3>("v_add_f32_dpp row_sh*:1", 128, out, cyc);
    run<4>("v_pk_add_f32 (2 floats)", 128, out, cyc);
    run<5>("v_pk_fma_f32 (2 floats)", 128, out, cyc);
    run<6>("v_rcp_f32", 128, out, cyc);
    run<7>("v_div_fixup_f32", 128, out, cyc);
    run<8>("v_mov_b32_dpp wave_sh*:1", 128, out, cyc);
    run<9>("v_cmp+v_cndmask pair", 128, out, cyc);
    run<10>("dependent v_add chain", 64, out, cyc);
    return 0;
}
