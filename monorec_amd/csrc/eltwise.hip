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

// SimpleMaskModule input (monorec_model.py:448-449): sum over the F single-frame volumes / max(#non-zero entries, 1); 16 B per lane.
__global__ __launch_bounds__(256) void nonzero_mean_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                           int F, long long count4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count4; i += (long long)gridDim.x * 256) {
        float4 s = src[i];
        float4 n = make_float4(s.x != 0.0f, s.y != 0.0f, s.z != 0.0f, s.w != 0.0f);
        for (int f = 1; f < F; ++f) {
            const float4 v = src[f * count4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            n.x += v.x != 0.0f; n.y += v.y != 0.0f; n.z += v.z != 0.0f; n.w += v.w != 0.0f;
        }
        dst[i] = make_float4(s.x / fmaxf(n.x, 1.0f), s.y / fmaxf(n.y, 1.0f), s.z / fmaxf(n.z, 1.0f), s.w / fmaxf(n.w, 1.0f));
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

// The MaskModule encoder needs both of a stage's output: its 2x2 max-pool per frame (input of the next stage,
// monorec_model.py:304-316) and its maximum over the frames (cv_feats, :365).  One pass reads the stage output once:
// thread = 2 rows x 4 columns of one (sample, channel) plane, all F frames.
__global__ __launch_bounds__(256) void pool2x2_framemax_kernel(const float* __restrict__ src, float2* __restrict__ pooled,
                                                               float* __restrict__ fmax, int F, long long planes, int H, int W) {
    const int Ho = H >> 1, Wo2 = W >> 2;
    const long long per_frame = planes * Ho * Wo2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < per_frame; i += (long long)gridDim.x * 256) {
        const int x2 = (int)(i % Wo2), oy = (int)((i / Wo2) % Ho);
        const long long p = i / ((long long)Wo2 * Ho);
        float4 ma, mb;
        for (int f = 0; f < F; ++f) {
            const float* s = src + (((long long)f * planes + p) * H + 2 * oy) * W + 4 * x2;
            const float4 a = *(const float4*)s;
            const float4 b = *(const float4*)(s + W);
            float2 o;
            o.x = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
            o.y = fmaxf(fmaxf(a.z, a.w), fmaxf(b.z, b.w));
            pooled[(long long)f * per_frame + i] = o;
            if (f == 0) { ma = a; mb = b; }
            else {
                ma.x = fmaxf(ma.x, a.x); ma.y = fmaxf(ma.y, a.y); ma.z = fmaxf(ma.z, a.z); ma.w = fmaxf(ma.w, a.w);
                mb.x = fmaxf(mb.x, b.x); mb.y = fmaxf(mb.y, b.y); mb.z = fmaxf(mb.z, b.z); mb.w = fmaxf(mb.w, b.w);
            }
        }
        float* d = fmax + (p * H + 2 * oy) * W + 4 * x2;
        *(float4*)d = ma;
        *(float4*)(d + W) = mb;
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

// Fused sparse depth metrics (reference model/metric_functions/sparse_metrics.py:136-252 + utils/util.py:36-118):
// one pass over prediction / ground-truth inverse depth gives, per sample, the 8 sums all seven metrics need:
//   [0] #valid  [1] sum |dp-dg|/dg  [2] sum (dp-dg)^2/dg  [3] sum (dp-dg)^2  [4] sum (log dp - log dg)^2
//   [5..7] #(max(dg/dp, dp/dg) < 1.25^k), k = 1,2,3
// mask (get_mask): gt == 0 or gt < 1/max_distance; depths: relu, clamp_min(1/max_distance), reciprocal.
// One 1024-thread workgroup per sample, fp32 per element, fp64 accumulation, fixed reduction order.
__global__ __launch_bounds__(1024) void sparse_metric_sums_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                  int H, int W, int y0, int y1, int x0, int x1,
                                                                  float inv_max_dist, double* __restrict__ out) {
    __shared__ double red[8][16];
    const int b = blockIdx.x;
    const float* p = pred + (long long)b * H * W;
    const float* g = gt + (long long)b * H * W;
    const int rw = x1 - x0, n = (y1 - y0) * rw;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float t1 = 1.25f, t2 = (float)(1.25 * 1.25), t3 = (float)(1.25 * 1.25 * 1.25);
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int y = y0 + i / rw, x = x0 + i % rw;
        float gi = g[y * W + x], pi = p[y * W + x];
        const bool masked = gi == 0.f || (inv_max_dist > 0.f && gi < inv_max_dist);
        if (masked) continue;
        pi = fmaxf(pi, 0.f); gi = fmaxf(gi, 0.f);                               // get_positive_depth
        if (inv_max_dist > 0.f) { pi = fmaxf(pi, inv_max_dist); gi = fmaxf(gi, inv_max_dist); }   // clamp_min
        const float dp = 1.0f / pi, dg = 1.0f / gi;                             // get_absolute_depth
        const float d = dp - dg;
        const float lg = logf(dp) - logf(dg);
        const float th = fmaxf(dg / dp, dp / dg);
        acc[0] += 1.0;
        acc[1] += (double)(fabsf(d) / dg);
        acc[2] += (double)((d * d) / dg);
        acc[3] += (double)(d * d);
        acc[4] += (double)(lg * lg);
        acc[5] += th < t1 ? 1.0 : 0.0;
        acc[6] += th < t2 ? 1.0 : 0.0;
        acc[7] += th < t3 ? 1.0 : 0.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double v = 0;
        for (int w = 0; w < 16; ++w) v += red[threadIdx.x][w];
        out[b * 8 + threadIdx.x] = v;
    }
}

inline unsigned grid_for(long long work_items) {
    long long blocks = (work_items + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;   // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)blocks;
}

// forward() hands the caller tensors it owns (monorec_model.py:713-727: the reference's outputs are fresh tensors): all outputs of
// one forward leave the slot's resident buffers in ONE launch - blockIdx.y = segment, grid-stride over its 16-byte units.
typedef float copy_f32x4 __attribute__((ext_vector_type(4)));
struct CopyArgs {
    const copy_f32x4* src[MR_MAX_COPY_SEGMENTS];
    copy_f32x4* dst[MR_MAX_COPY_SEGMENTS];
    long long units[MR_MAX_COPY_SEGMENTS];
};

__global__ __launch_bounds__(256) void copy_segments_kernel(const CopyArgs a) {
    const int s = blockIdx.y;
    const copy_f32x4* __restrict__ src = a.src[s];
    copy_f32x4* __restrict__ dst = a.dst[s];
    const long long n = a.units[s];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = __builtin_nontemporal_load(src + i);
}

// The 4x4 pose / intrinsics matrices of one forward (2 + 2 F tensors of 16 B floats each, anywhere in device memory) -> one pinned host
// buffer the device can write (hipHostMalloc): the host-side pose algebra of MonoRecModel needs them back (model.host_geometry).  One
// launch instead of an ATen stack + a copy-engine transfer - and its completion is a compute-queue marker, which the host can wait
// for with far less wake-up latency than for the end of a small SDMA copy (measured: ~1 ms after a long wait).
struct GatherArgs {
    const float* src[MR_MAX_GATHER];
    float* dst;
    int count;
};

__global__ __launch_bounds__(64) void gather_small_kernel(const GatherArgs a) {
    const float* __restrict__ s = a.src[blockIdx.x];
    float* d = a.dst + (long long)blockIdx.x * a.count;
    for (int i = threadIdx.x; i < a.count; i += 64) d[i] = s[i];
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

extern "C" int mr_pool2x2_framemax_f32(const float* src, float* pooled, float* frame_max, int32_t num_frames, int64_t planes,
                                       int32_t in_h, int32_t in_w, void* stream) {
    if (!src || !pooled || !frame_max || num_frames < 1 || planes < 1 || in_h < 2 || in_w < 4 || (in_h & 1) || (in_w & 3))
        return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(pool2x2_framemax_kernel, dim3(grid_for(planes * (in_h / 2) * (in_w / 4))), dim3(256), 0,
                       (hipStream_t)stream, src, (float2*)pooled, frame_max, num_frames, (long long)planes, in_h, in_w);
    return (int)hipGetLastError();
}

extern "C" int mr_resnet_normalize_f32(const float* src, float* dst, int64_t count, void* stream) {
    if (!src || !dst || count < 4 || (count & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(resnet_normalize_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, (long long)(count / 4));
    return (int)hipGetLastError();
}

extern "C" int mr_sparse_metric_sums_f32(const float* prediction, const float* target, int32_t batch, int32_t height,
                                         int32_t width, const int32_t* roi, float max_distance, double* sums, void* stream) {
    if (!prediction || !target || !sums || batch < 1 || height < 1 || width < 1) return MR_ERR_BAD_ARGUMENT;
    int y0 = 0, y1 = height, x0 = 0, x1 = width;
    if (roi) { y0 = roi[0]; y1 = roi[1]; x0 = roi[2]; x1 = roi[3]; }
    if (y0 < 0 || x0 < 0 || y1 > height || x1 > width || y1 <= y0 || x1 <= x0) return MR_ERR_BAD_ARGUMENT;
    const float inv = max_distance > 0.f ? 1.0f / max_distance : 0.f;
    hipLaunchKernelGGL(sparse_metric_sums_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, prediction, target,
                       height, width, y0, y1, x0, x1, inv, sums);
    return (int)hipGetLastError();
}

extern "C" int mr_max_over_frames_f32(const float* src, float* dst, int32_t num_frames, int64_t count, void* stream) {
    if (!src || !dst || num_frames < 1 || count < 4 || (count & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(max_over_frames_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, num_frames, (long long)(count / 4));
    return (int)hipGetLastError();
}

extern "C" int mr_nonzero_mean_over_frames_f32(const float* src, float* dst, int32_t num_frames, int64_t count, void* stream) {
    if (!src || !dst || num_frames < 1 || count < 4 || (count & 3)) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(nonzero_mean_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream,
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

extern "C" int mr_copy_segments(const mr_copy_segment* segments, int32_t num_segments, void* stream) {
    if (!segments || num_segments < 1 || num_segments > MR_MAX_COPY_SEGMENTS) return MR_ERR_BAD_ARGUMENT;
    CopyArgs a;
    long long most = 0;
    for (int s = 0; s < num_segments; ++s) {
        const mr_copy_segment& g = segments[s];
        if (!g.src || !g.dst || g.bytes < 16 || (g.bytes & 15) || ((unsigned long long)g.src & 15) || ((unsigned long long)g.dst & 15))
            return MR_ERR_BAD_ARGUMENT;
        a.src[s] = (const copy_f32x4*)g.src;
        a.dst[s] = (copy_f32x4*)g.dst;
        a.units[s] = g.bytes / 16;
        if (a.units[s] > most) most = a.units[s];
    }
    const long long blocks = (most + 255) / 256;
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), (unsigned)num_segments), dim3(256), 0,
                       (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mr_gather_small_f32(const float* const* srcs, int32_t num, int32_t floats_each, float* dst, void* stream) {
    if (!srcs || !dst || num < 1 || num > MR_MAX_GATHER || floats_each < 1) return MR_ERR_BAD_ARGUMENT;
    GatherArgs a;
    for (int s = 0; s < MR_MAX_GATHER; ++s) {
        a.src[s] = s < num ? srcs[s] : nullptr;
        if (s < num && !a.src[s]) return MR_ERR_BAD_ARGUMENT;
    }
    a.dst = dst;
    a.count = floats_each;
    hipLaunchKernelGGL(gather_small_kernel, dim3((unsigned)num), dim3(64), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// One host call for a run of consecutive convolution launches of a plan (engine.Plan.run_stage): the ~90 convolution-type launches of a
// keyframe cost the Python host ~4.5 us each as separate ctypes calls, ~2 us as entries of a list walked here.  An item only names an
// entry point of this ABI and its descriptor; descriptors are read at launch time (the plan re-binds input pointers between forwards).
extern "C" int mr_run_launches(const mr_launch_item* items, int32_t num, void* stream, int32_t* failed_index) {
    if (!items || num < 0) return MR_ERR_BAD_ARGUMENT;
    for (int i = 0; i < num; ++i) {
        int rc;
        switch (items[i].kind) {
            case MR_LAUNCH_CONV2D: rc = mr_conv2d_f32((const mr_conv_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_WINO3X3: rc = mr_conv3x3_winograd_f32((const mr_wino_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_WINO_T: rc = mr_convt4x4s2_winograd_f32((const mr_wino_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_WINO_1D: rc = mr_conv1d3_winograd_f32((const mr_wino_desc*)items[i].desc, items[i].arg, stream); break;
            case MR_LAUNCH_UPCONV: rc = mr_upconv2x2_winograd_f32((const mr_wino_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_WINO44: rc = mr_conv3x3_winograd44_f32((const mr_wino_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_WINO44S: rc = mr_conv3x3_winograd44s_f32((const mr_wino_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_WINO44W: rc = mr_conv3x3_winograd44w_f32((const mr_wino_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_CONV_B8: rc = mr_conv2d_b8((const mr_b8_conv_desc*)items[i].desc, stream); break;
            case MR_LAUNCH_COOKTOOM_1D:
                rc = mr_conv1d_cooktoom_f32((const mr_wino_desc*)items[i].desc, items[i].arg & 15, (items[i].arg >> 4) & 15, (items[i].arg >> 8) & 15, stream);
                break;
            default: rc = MR_ERR_BAD_ARGUMENT;
        }
        if (rc != 0) {
            if (failed_index) *failed_index = i;
            return rc;
        }
    }
    return 0;
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

#ifdef MR_DIAGNOSTIC_FORMS
// exported by the diagnostic build only (bit 0: the F(2,7) instantiations of csrc/conv1d_wino.hip are present)
extern "C" int mr_diagnostic_forms(void) { return 1; }
#endif
