// Small HBM-bound kernels of the MonoRec inference path (gfx950): everything that is not a
// convolution and could not be folded into a convolution's staging/epilogue.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/monorec_hip.h"

namespace {

// nn.MaxPool2d(3, 2, 1) of the torchvision ResNet stem (monorec_model.py:124). Padding acts as -inf.
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int planes, int H, int W, int Ho, int Wo) {
    const long long total = (long long)planes * Ho * Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const long long p = i / ((long long)Wo * Ho);
        const float* s = src + p * H * W;
        float m = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = 2 * oy - 1 + dy;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int x = 2 * ox - 1 + dx;
                if (x < 0 || x >= W) continue;
                m = fmaxf(m, s[y * W + x]);
            }
        }
        dst[i] = m;
    }
}

// torch.max over the F per-frame encoder outputs (monorec_model.py:365); 16 B per lane.
__global__ __launch_bounds__(256) void max_over_frames_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                              int F, long long count4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count4; i += (long long)gridDim.x * 256) {
        float4 m = src[i];
        for (int f = 1; f < F; ++f) {
            const float4 v = src[f * count4 + i];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        dst[i] = m;
    }
}

// cost_volume = (1 - cv_mask) * cost_volume (monorec_model.py:713); 16 B per lane.
__global__ __launch_bounds__(256) void apply_mask_kernel(const float4* cv, const float4* __restrict__ mask,
                                                         float4* dst, int B, int D, long long plane4) {
    const long long total = (long long)B * D * plane4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % plane4;
        const long long b = i / (plane4 * D);
        const float4 m = mask[b * plane4 + p];
        const float4 v = cv[i];
        float4 o;
        o.x = (1.0f - m.x) * v.x; o.y = (1.0f - m.y) * v.y; o.z = (1.0f - m.z) * v.z; o.w = (1.0f - m.w) * v.w;
        dst[i] = o;
    }
}

// nn.MaxPool2d(2) between the MaskModule encoder stages (monorec_model.py:304-316); two outputs per lane.
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ src, float2* __restrict__ dst,
                                                         long long planes, int H, int W) {
    const int Ho = H >> 1, Wo2 = W >> 2;                 // Wo2 = output float2 per row
    const long long total = planes * Ho * Wo2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x2 = (int)(i % Wo2), oy = (int)((i / Wo2) % Ho);
        const long long p = i / ((long long)Wo2 * Ho);
        const float4 a = *(const float4*)(src + (p * H + 2 * oy) * W + 4 * x2);
        const float4 b = *(const float4*)(src + (p * H + 2 * oy + 1) * W + 4 * x2);
        float2 o;
        o.x = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
        o.y = fmaxf(fmaxf(a.z, a.w), fmaxf(b.z, b.w));
        dst[i] = o;
    }
}

// ResnetEncoder input normalisation ((x + 0.5) - 0.45) / 0.225 (monorec_model.py:691 + :120); 16 B per lane.
__global__ __launch_bounds__(256) void resnet_normalize_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                               long long count4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count4; i += (long long)gridDim.x * 256) {
        const float4 v = src[i];
        float4 o;
        o.x = ((v.x + 0.5f) - 0.45f) / 0.225f; o.y = ((v.y + 0.5f) - 0.45f) / 0.225f;
        o.z = ((v.z + 0.5f) - 0.45f) / 0.225f; o.w = ((v.w + 0.5f) - 0.45f) / 0.225f;
        dst[i] = o;
    }
}

inline unsigned grid_for(long long work_items) {
    long long blocks = (work_items + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;   // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)blocks;
}

}  // namespace

extern "C" int mr_maxpool3x3s2_f32(const float* src, float* dst, int32_t planes, int32_t in_h, int32_t in_w, void* stream) {
    if (!src || !dst || planes < 1 || in_h < 1 || in_w < 1) return MR_ERR_BAD_ARGUMENT;
    const int Ho = (in_h + 2 - 3) / 2 + 1, Wo = (in_w + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((long long)planes * Ho * Wo)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, planes, in_h, in_w, Ho, Wo);
    return (int)hipGetLastError();
}

extern "C" int mr_maxpool2x2_f32(const float* src, float* dst, int64_t planes, int32_t in_h, int32_t in_w, void* stream) {
    if (!src || !dst || planes < 1 || in_h < 2 || in_w < 4 || (in_h & 1) || (in_w & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(planes * (in_h / 2) * (in_w / 4))), dim3(256), 0,
                       (hipStream_t)stream, src, (float2*)dst, (long long)planes, in_h, in_w);
    return (int)hipGetLastError();
}

extern "C" int mr_resnet_normalize_f32(const float* src, float* dst, int64_t count, void* stream) {
    if (!src || !dst || count < 4 || (count & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(resnet_normalize_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, (long long)(count / 4));
    return (int)hipGetLastError();
}

extern "C" int mr_max_over_frames_f32(const float* src, float* dst, int32_t num_frames, int64_t count, void* stream) {
    if (!src || !dst || num_frames < 1 || count < 4 || (count & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(max_over_frames_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, num_frames, (long long)(count / 4));
    return (int)hipGetLastError();
}

extern "C" int mr_apply_mask_f32(const float* cv, const float* mask, float* dst, int32_t batch, int32_t num_depths,
                                 int64_t plane, void* stream) {
    if (!cv || !mask || !dst || batch < 1 || num_depths < 1 || plane < 4 || (plane & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(apply_mask_kernel, dim3(grid_for((long long)batch * num_depths * plane / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)cv, (const float4*)mask, (float4*)dst, batch, num_depths,
                       (long long)(plane / 4));
    return (int)hipGetLastError();
}

extern "C" int mr_abi_version(void) { return MR_ABI_VERSION; }

extern "C" const char* mr_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case MR_ERR_BAD_ARGUMENT: return "monorec_hip: bad argument";
        case MR_ERR_UNSUPPORTED: return "monorec_hip: unsupported configuration";
        case MR_ERR_LDS_BUDGET: return "monorec_hip: launch would exceed the 160 KiB LDS of a gfx950 CU";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "monorec_hip: unknown error";
    }
}
