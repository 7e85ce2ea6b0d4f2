// Probe (GPU box): what does one wave pay to *issue* N back-to-back memory instructions, and how long until
// the data is there?  One workgroup per CU (256 CUs), W waves each issuing N instructions of 1 KiB (64 lanes x 16 B)
// from an L2-resident 8 MiB buffer.
//   lds_m0  : global_load_lds_dwordx4, M0 rewritten per instruction (what conv_mfma.hip does)
//   lds_fix : global_load_lds_dwordx4, M0 written once (all land on the same 1 KiB - timing only)
//   regs    : global_load_dwordx4 into VGPRs
// hipcc --offload-arch=gfx950 -O3 tools/probes/dma_issue.hip -o tools/probes/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int N = 16;
typedef float f4 __attribute__((ext_vector_type(4)));
// cold variant: every workgroup streams its own never-touched 64 KiB x rounds from a 4 GiB buffer (L2 and MALL miss)
__global__ __launch_bounds__(256) void kcold(const float* src, unsigned long long* out, int rounds, size_t wg_stride_floats, size_t istride, size_t rstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* g = src + (size_t)blockIdx.x * wg_stride_floats + wave * N * istride + lane * 4;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + wave * N * 1024);
    __syncthreads();
    const unsigned long long c0 = clock64();
    unsigned long long iss = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long a0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            unsigned keep;
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(lbase + i * 1024), "v"(g + (size_t)r * rstride + i * istride) : "memory");
        }
        iss += clock64() - a0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long c2 = clock64();
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 4 + 0] = iss; out[(blockIdx.x * 8 + wave) * 4 + 1] = c2 - c0; }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* src, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const float* g = src + ((size_t)blockIdx.x * 8192 + wave * N * 256 + lane * 4) % (2u << 20);
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + wave * N * 1024);
    __syncthreads();
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
    f4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            unsigned keep;
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(lbase + i * 1024), "v"(g + i * 256) : "memory");
        }
    } else if (MODE == 1) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lbase) : "memory");
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(g + i * 256) : "memory");
    } else {
        f4 v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __builtin_nontemporal_load((const f4*)(g + i * 256));
        asm volatile("" ::: "memory");
        const unsigned long long c1 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c2 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i) acc += v[i];
        if (lane == 0) { out[(blockIdx.x * 8 + wave) * 4 + 0] = c1 - c0; out[(blockIdx.x * 8 + wave) * 4 + 1] = c2 - c0; }
        if (acc.x == 123.4f) sink[0] = acc.y;
        return;
    }
    const unsigned long long c1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c2 = clock64();
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 4 + 0] = c1 - c0; out[(blockIdx.x * 8 + wave) * 4 + 1] = c2 - c0; }
    __syncthreads();
    if (lds[threadIdx.x] == 123.4f) sink[0] = 1.f;
    (void)t0; (void)nw;
}
template <int MODE> void run(const char* name, int waves, const float* src, unsigned long long* out, float* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) k<MODE><<<256, waves * 64, 8 * N * 1024, 0>>>(src, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, done;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { iss.push_back((double)h[(b * 8 + w) * 4]); done.push_back((double)h[(b * 8 + w) * 4 + 1]); }
    std::sort(iss.begin(), iss.end()); std::sort(done.begin(), done.end());
    printf("%-8s waves/WG %d: issue of %d instr: median %.0f clk (%.0f per instr), all data landed: median %.0f clk  [%.1f B/clk/CU]\n", name, waves, N,
           iss[iss.size() / 2], iss[iss.size() / 2] / N, done[done.size() / 2], waves * N * 1024.0 / done[done.size() / 2]);
}
int main() {
    float *src, *sink; unsigned long long* out;
    hipMalloc(&src, 16u << 20); hipMemset(src, 0, 16u << 20); hipMalloc(&sink, 64); hipMalloc(&out, 256 * 8 * 4 * 8);
    {
        float* big; const size_t bytes = 6ull << 30;
        if (hipMalloc(&big, bytes) == hipSuccess) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&kcold), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            unsigned long long* o2; hipMalloc(&o2, 1024 * 8 * 4 * 8);
            const int rounds = 4;
            for (int mode = 0; mode < 3; ++mode)
            for (int wgs : {128, 256}) {
                // mode 0: each instruction 1 KiB contiguous after the previous (one page per workgroup round)
                // mode 1: every instruction in its own 512 KiB region (a conv tile row block of another channel plane)
                // mode 2: every instruction in its own 2 MiB + 4 KiB region
                const size_t istride = mode == 0 ? 256 : (mode == 1 ? (512u << 10) / 4 : ((2u << 20) + 4096) / 4);
                const size_t wgstride = mode == 0 ? (size_t)rounds * 4 * N * 256 : 1024;      // tiles of one plane
                const size_t rstride = mode == 0 ? (size_t)4 * N * 256 : (size_t)4 * N * istride;
                const size_t need = (size_t)rounds * rstride + (size_t)wgs * wgstride + 4 * N * istride;
                if (need * 4 > bytes) { printf("skip mode %d\n", mode); continue; }
                kcold<<<wgs, 256, 4 * N * 1024, 0>>>(big, o2, rounds, wgstride, istride, rstride);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(wgs * 8 * 4);
                hipMemcpy(h.data(), o2, h.size() * 8, hipMemcpyDeviceToHost);
                std::vector<double> iss, tot;
                for (int b = 0; b < wgs; ++b) for (int w = 0; w < 4; ++w) { iss.push_back((double)h[(b * 8 + w) * 4]); tot.push_back((double)h[(b * 8 + w) * 4 + 1]); }
                std::sort(iss.begin(), iss.end()); std::sort(tot.begin(), tot.end());
                const double clk = tot[tot.size() / 2], bytes_wg = rounds * 4.0 * N * 1024;
                printf("cold mode %d %3d WGs x 4 waves, %d rounds of 64 KiB: issue %.0f clk/round/wave (%.0f per instr), round trip %.0f clk/round -> %.1f B/clk/WG\n",
                       mode, wgs, rounds, iss[iss.size() / 2] / rounds, iss[iss.size() / 2] / rounds / N, clk / rounds, bytes_wg / clk);
            }
        } else printf("cold: hipMalloc failed\n");
    }
    for (int w : {1, 2, 4, 8}) { run<0>("lds_m0", w, src, out, sink); run<1>("lds_fix", w, src, out, sink); run<2>("regs", w, src, out, sink); }
    return 0;
}
