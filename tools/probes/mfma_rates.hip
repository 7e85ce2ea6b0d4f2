// Issue-rate probe for v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 (round 6; run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rates.hip -o tools/probes/mfma_rates && tools/probes/mfma_rates
// Why: r06_s2 ablated every LDS read and every DMA out of conv_mfma_kernel's sweep and the sweep stayed at ~50 shader cycles per MFMA and SIMD
// (two waves per SIMD) where the instruction's issue time is 32.  Which property of the stream costs the difference?
//   KIND 0  one accumulator, vDst = srcC                   (dependent-accumulator latency)
//   KIND 1  two accumulators alternating, vDst = srcC     (the 1 x 1 register tile's DUAL sums)
//   KIND 2  four accumulators, vDst = srcC
//   KIND 3  eight accumulators, vDst = srcC
//   KIND 4  two chains whose destination ROTATES (x = mfma(.., y); y = mfma(.., x)): vDst != srcC, what hipcc emitted in the product loop
//   KIND 5  KIND 1 + the product loop's fillers (one v_add_u32 on the B-address register and two SALU per MFMA)
//   KIND 6  KIND 2 + the same fillers
//   KIND 7  v_mfma_f32_32x32x2_f32, two accumulators
//   KIND 8  KIND 4 with four chains
// Every kind at 1 and 2 waves per SIMD (256 / 512 threads, one workgroup per CU through 100 KB of dynamic LDS), 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define M1(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define MR(dst, src) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(dst) : "v"(a), "v"(b), "v"(src));
#define VA asm volatile("v_add_u32 %0, %1, %0" : "+v"(addr) : "s"(s0));
#define VF asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fx) : "v"(a));
#define SA asm volatile("s_add_i32 %0, %0, 4" : "+s"(s0));
#define DS asm volatile("ds_read_b32 %0, %1" : "=v"(lv) : "v"(laddr));
#define BM(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ba), "v"(bb));
#define FILL asm volatile("v_add_u32 %0, %1, %0\n s_add_i32 %2, %2, 4\n s_add_i32 %3, %3, %2" : "+v"(addr), "+s"(s0), "+s"(s1) : "s"(s0));

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f + 1.f, b = 1.f - threadIdx.x * 1e-3f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    f32x4 d0 = c0, d1 = c0, d2 = c0, d3 = c0;
    f32x16 e0 = {0}, e1 = {0};
    int addr = threadIdx.x, s0 = 1, s1 = 2;
    float fx = a, lv = 0.f;
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    const unsigned hsh = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    const i32x4v ba = {(int)(0x3f803f00u ^ (hsh & 0x007f007f)), (int)(0xbf003e80u ^ ((hsh >> 3) & 0x007f007f)), (int)(0x3e003f40u ^ ((hsh >> 5) & 0x007f007f)), (int)(0xbe803f20u ^ ((hsh >> 7) & 0x007f007f))};
    const i32x4v bb = {(int)(0x3f003f80u ^ ((hsh >> 2) & 0x007f007f)), (int)(0x3e80bf00u ^ ((hsh >> 4) & 0x007f007f)), (int)(0x3f403e00u ^ ((hsh >> 6) & 0x007f007f)), (int)(0x3f20be80u ^ ((hsh >> 8) & 0x007f007f))};
    const unsigned laddr = (threadIdx.x & 63) * 4;
    if (iters < 0) lds[threadIdx.x] = a;          // keeps the allocation
    __syncthreads();
    const long long t0 = clock64();
    int n = 0;                                      // MFMAs per iteration
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(M1(c0)) n = 16; }
        if (KIND == 1) { REP4(M1(c0) M1(c1) M1(c0) M1(c1)) n = 16; }
        if (KIND == 2) { REP4(M1(c0) M1(c1) M1(c2) M1(c3)) n = 16; }
        if (KIND == 3) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) M1(c4) M1(c5) M1(c6) M1(c7)) n = 32; }
        if (KIND == 4) { REP4(MR(d0, c0) MR(d1, c1) MR(c0, d0) MR(c1, d1)) n = 16; }
        if (KIND == 5) { REP4(M1(c0) FILL M1(c1) FILL M1(c0) FILL M1(c1) FILL) n = 16; }
        if (KIND == 6) { REP4(M1(c0) FILL M1(c1) FILL M1(c2) FILL M1(c3) FILL) n = 16; }
        if (KIND == 7) {
            REP4(asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n v_mfma_f32_32x32x2_f32 %1, %2, %3, %1" : "+v"(e0), "+v"(e1) : "v"(a), "v"(b));)
            n = 8;
        }
        if (KIND == 8) { REP4(MR(d0, c0) MR(d1, c1) MR(d2, c2) MR(d3, c3) MR(c0, d0) MR(c1, d1) MR(c2, d2) MR(c3, d3)) n = 32; }
        // ---- second set: what does a VALU / SALU / LDS instruction cost by PLACEMENT (4 accumulators, groups of 4 MFMAs) ----
        if (KIND == 9) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) VA VA VA VA) n = 16; }                    // 4 VALU grouped behind 4 MFMAs
        if (KIND == 10) { REP4(M1(c0) VA M1(c1) VA M1(c2) VA M1(c3) VA) n = 16; }                   // the same 4 VALU, one behind every MFMA
        if (KIND == 11) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) VA) n = 16; }                            // 1 VALU per 4 MFMAs
        if (KIND == 12) { REP4(M1(c0) SA SA M1(c1) SA SA M1(c2) SA SA M1(c3) SA SA) n = 16; }       // SALU only, 2 behind every MFMA
        if (KIND == 13) { REP4(M1(c0) DS M1(c1) DS M1(c2) DS M1(c3) DS) n = 16; }                   // one ds_read_b32 behind every MFMA (no wait inside the loop body)
        if (KIND == 14) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) M1(c0) M1(c1) M1(c2) M1(c3) VA VA VA) n = 32; }   // 3 VALU grouped per 8 MFMAs
        if (KIND == 15) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) VA VA VA VA VA VA VA VA) n = 16; }       // 8 VALU grouped per 4 MFMAs
        if (KIND == 16) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA) n = 16; }   // 16 VALU grouped per 4 MFMAs
        if (KIND == 17) { REP4(M1(c0) VF M1(c1) VF M1(c2) VF M1(c3) VF) n = 16; }                   // v_fma_f32 instead of v_add_u32, interleaved
        if (KIND == 18) { REP4(M1(c0) M1(c1) M1(c2) M1(c3) VF VF VF VF VF VF VF VF VF VF VF VF VF VF VF VF) n = 16; }   // 16 v_fma_f32 grouped per 4 MFMAs
        if (KIND == 20) {   // v_mfma_f32_16x16x32_bf16, 12 accumulators (conv_b8_kernel's 3 x 4 register tile), operands with varied bits
            REP4(BM(c0) BM(c1) BM(c2) BM(c3) BM(c4) BM(c5) BM(c6) BM(c7) BM(d0) BM(d1) BM(d2) BM(d3)) n = 48;
        }
        if (KIND == 21) {   // v_mfma_f32_32x32x16_bf16, 2 accumulators
            REP4(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(e0), "+v"(e1) : "v"(ba), "v"(bb));)
            n = 8;
        }
        // ---- third set: the same placements on the bf16 instruction of conv_b8_kernel: does VALU work hide under IT? ----
        if (KIND == 22) { REP4(BM(c0) VA BM(c1) VA BM(c2) VA BM(c3) VA) n = 16; }                   // one v_add_u32 behind every bf16 MFMA (kind 10)
        if (KIND == 23) { REP4(BM(c0) BM(c1) BM(c2) BM(c3) VA VA VA VA VA VA VA VA) n = 16; }       // 8 VALU grouped per 4 bf16 MFMAs (kind 15)
        if (KIND == 24) { REP4(BM(c0) VF BM(c1) VF BM(c2) VF BM(c3) VF) n = 16; }                   // v_fma_f32 behind every bf16 MFMA (kind 17)
        if (KIND == 25) { REP4(BM(c0) BM(c1) BM(c2) BM(c3)) n = 16; }                               // 4 accumulators alone: the reference for 22-24
        if (KIND == 19) { REP4(M1(c0) DS DS VA M1(c1) DS DS VA M1(c2) DS DS VA M1(c3) DS DS VA) n = 16; }   // the un-specialised sweep's mix: 2 LDS reads + 1 VALU per MFMA
    }
    const long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + d0 + d1 + d2 + d3;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * THREADS + threadIdx.x] = fx + lv + s[0] + s[1] + s[2] + s[3] + e0[0] + e1[5] + (float)addr + (float)(s0 + s1);
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = (long long)n * iters; }
}

template <int KIND, int THREADS>
void run(const char* what, float* out, long long* cyc) {
    const int iters = KIND >= 20 ? 20000 : 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<KIND, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, THREADS>), dim3(blocks), dim3(THREADS), 100 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, THREADS>), dim3(blocks), dim3(THREADS), 100 * 1024, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 2, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < blocks; ++i) c += (double)h[2 * i];
    c /= blocks;
    const double per_wave = (double)h[1];
    const int wps = THREADS / 256;
    const int flop = KIND == 7 ? 4096 : ((KIND == 20 || KIND >= 22) ? 16384 : (KIND == 21 ? 32768 : 2048));
    printf("KIND %d %-58s %d wave(s)/SIMD: %6.1f cycles per MFMA and SIMD  (%.0f MFMAs/wave, %.3f ms -> %.1f TF, clock ~%.0f MHz)\n", KIND, what, wps,
           c / (per_wave * wps), per_wave, ms, (double)blocks * THREADS / 64 * per_wave * flop / (ms * 1e-3) / 1e12, c / (ms * 1e3));
}

#define BOTH(K, what) run<K, 256>(what, out, cyc); run<K, 512>(what, out, cyc);

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 2 * 8);
    BOTH(0, "1 accumulator, in place")
    BOTH(1, "2 accumulators, in place")
    BOTH(2, "4 accumulators, in place")
    BOTH(3, "8 accumulators, in place")
    BOTH(4, "2 chains, rotating destination (vDst != srcC)")
    BOTH(8, "4 chains, rotating destination")
    BOTH(5, "2 accumulators + 1 VALU + 2 SALU per MFMA")
    BOTH(6, "4 accumulators + 1 VALU + 2 SALU per MFMA")
    BOTH(7, "32x32x2, 2 accumulators")
    BOTH(20, "bf16 16x16x32, 12 accumulators")
    BOTH(21, "bf16 32x32x16, 2 accumulators")
    BOTH(25, "bf16 16x16x32, 4 accumulators")
    BOTH(22, "4 x (bf16 16x16x32, v_add_u32) interleaved")
    BOTH(23, "4 bf16 16x16x32 then 8 VALU grouped")
    BOTH(24, "4 x (bf16 16x16x32, v_fma_f32) interleaved")
    BOTH(9, "4 MFMA then 4 VALU grouped")
    BOTH(10, "4 x (MFMA, VALU) interleaved")
    BOTH(11, "4 MFMA then 1 VALU")
    BOTH(14, "8 MFMA then 3 VALU grouped")
    BOTH(15, "4 MFMA then 8 VALU grouped")
    BOTH(16, "4 MFMA then 16 VALU grouped")
    BOTH(17, "4 x (MFMA, v_fma_f32) interleaved")
    BOTH(18, "4 MFMA then 16 v_fma_f32 grouped")
    BOTH(12, "4 x (MFMA, 2 SALU)")
    BOTH(13, "4 x (MFMA, ds_read_b32)")
    BOTH(19, "4 x (MFMA, 2 ds_read_b32, VALU)")
    return 0;
}
