// Probe (GPU box): how fast does the chip take the conv epilogue's store pattern?  48 x 256 x 512 fp32 (25 MB),
// 512 workgroups of 256 threads, each owning an 8 x 32 pixel tile of all 48 channels (mask.dec3.1's shape).
//   A: as the MFMA D fragment holds it: one instruction = 4 channels x 16 consecutive pixels (4 x 64 B)
//   B: after an LDS transpose: one instruction = 8 (channel,row) pairs x 32 pixels as float4 (8 x 128 B)
//   C: contiguous float4 stream (upper bound)
// hipcc --offload-arch=gfx950 -O3 tools/probes/store_patterns.hip -o tools/probes/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 48, H = 256, W = 512;
__global__ __launch_bounds__(256) void pat_a(float* dst) {
    const int tile = blockIdx.x, ty = tile / 16, tx = tile % 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = 0; m < 3; ++m)
        for (int i = 0; i < 4; ++i) {
            const int pb = wave * 4 + i, row = pb >> 1, col = (pb & 1) * 16 + (lane & 15);
            for (int r = 0; r < 4; ++r) {
                const int ch = m * 16 + (lane >> 4) * 4 + r;
                dst[((long long)ch * H + ty * 8 + row) * W + tx * 32 + col] = (float)(lane + r);
            }
        }
}
__global__ __launch_bounds__(256) void pat_b(float* dst) {
    const int tile = blockIdx.x, ty = tile / 16, tx = tile % 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 48 ch x 8 rows = 384 (ch,row) pairs of 128 B; a wave instruction covers 8 pairs; 4 waves -> 12 instructions each
    for (int k = 0; k < 12; ++k) {
        const int pair = (k * 4 + wave) * 8 + (lane >> 3), ch = pair >> 3, row = pair & 7;
        float4 v = {(float)lane, 1.f, 2.f, 3.f};
        *(float4*)&dst[((long long)ch * H + ty * 8 + row) * W + tx * 32 + (lane & 7) * 4] = v;
    }
}
__global__ __launch_bounds__(256) void pat_c(float* dst) {
    const long long base = (long long)blockIdx.x * (C * 8 * 32);
    for (int k = 0; k < 12; ++k) {
        float4 v = {(float)threadIdx.x, 1.f, 2.f, 3.f};
        *(float4*)&dst[base + (k * 256 + threadIdx.x) * 4] = v;
    }
}
template <class F> float run(F f, float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f<<<512, 256>>>(d);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) f<<<512, 256>>>(d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 50, gb = (double)C * H * W * 4 / 1e9;
    printf("%s: %.2f us per launch, %.0f GB/s\n", name, us, gb / (us * 1e-6));
    return ms;
}
int main() {
    float* d; hipMalloc(&d, (size_t)C * H * W * 4 * 2);
    run(pat_a, d, "A fragment-order 4x64B "); run(pat_b, d, "B float4 rows 8x128B  "); run(pat_c, d, "C contiguous float4    ");
    run(pat_a, d, "A again                "); run(pat_b, d, "B again                ");
    return 0;
}
