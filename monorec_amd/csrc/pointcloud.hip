// the point-cloud path of create_pointcloud.py on the device.
//
//   static_mask_kernel      cv_mask >= threshold, dilated by a (mask_fill+1)^2 box, inverted
//                           (create_pointcloud.py:76-77: F.conv2d(mask, ones(33x33), padding=16) < 1)
//   pointcloud_append_kernel  vote over the buffered masks (:90), depth *= mask (:91-92), PLYSaver.add_depthmap
//                           (utils/ply_utils.py:34-53): 1/depth, range / roi / dropout mask, back-projection
//                           (model/layers.py:56-61), extrinsics @ coords, colours, boolean-mask compaction.
// The reference pulls every frame's points to the host with .cpu().tolist(); here the records are appended in the
// same (pixel row-major) order to a device-resident buffer behind a device-side cursor - no host synchronisation
// per keyframe, one copy when the .ply is written.
//
// Arithmetic follows the reference operation by operation (the matmuls are ascending-k fmaf chains like the CPU
// sgemm / bmm it runs through); compiled with -ffp-contract=off like cost_volume.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/monorec_hip.h"

namespace {

constexpr int SM_TH = 16, SM_TW = 64;      // output tile of one workgroup (256 threads: 4 outputs each)

__global__ __launch_bounds__(256) void static_mask_kernel(const float* __restrict__ cv_mask, float* __restrict__ out,
                                                          int H, int W, float threshold, int r) {
    extern __shared__ unsigned char sm[];                 // [TH + 2r][TW + 2r] moving flags, then [TH + 2r][TW] row-any
    const int b = blockIdx.z, y0 = blockIdx.y * SM_TH, x0 = blockIdx.x * SM_TW;
    const int IH = SM_TH + 2 * r, IW = SM_TW + 2 * r;
    unsigned char* flag = sm;
    unsigned char* rowany = sm + IH * IW;
    const float* src = cv_mask + (long long)b * H * W;
    for (int p = threadIdx.x; p < IH * IW; p += 256) {
        const int iy = p / IW, ix = p - iy * IW;
        const int gy = y0 - r + iy, gx = x0 - r + ix;
        flag[p] = (gy >= 0 && gy < H && gx >= 0 && gx < W && src[(long long)gy * W + gx] >= threshold) ? 1 : 0;   // zero padding
    }
    __syncthreads();
    for (int p = threadIdx.x; p < IH * SM_TW; p += 256) {
        const int iy = p / SM_TW, ox = p - iy * SM_TW;
        unsigned char any = 0;
        for (int k = 0; k <= 2 * r; ++k) any |= flag[iy * IW + ox + k];
        rowany[p] = any;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < SM_TH * SM_TW; p += 256) {
        const int oy = p / SM_TW, ox = p - oy * SM_TW;
        const int gy = y0 + oy, gx = x0 + ox;
        if (gy >= H || gx >= W) continue;
        unsigned char any = 0;
        for (int k = 0; k <= 2 * r; ++k) any |= rowany[(oy + k) * SM_TW + ox];
        out[((long long)b * H + gy) * W + gx] = any ? 0.f : 1.f;       // "< 1" of the box sum
    }
}

struct PcArgs {
    const float* inv_depth;
    const float* masks[MR_MAX_VOTE_MASKS];
    int num_masks;
    float vote_above;          // keep where sum(masks) > vote_above
    const float* image;
    const float* kinv;         // B x 9
    const float* pose;         // B x 16
    const float* uniform;      // B x H x W or null
    float dropout, min_d, max_d;
    int roi[4];                // y0, y1, x0, x1 (full image when no roi)
    int B, H, W;
    float* records;
    long long capacity;
    long long* cursor;
};

__global__ __launch_bounds__(1024) void pointcloud_append_kernel(const PcArgs a) {
    __shared__ int wave_tot[16];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long plane = (long long)a.H * a.W;
    if (tid == 0) s_base = *a.cursor;
    __syncthreads();
    long long base = s_base;
    for (int b = 0; b < a.B; ++b) {
        const float* K = a.kinv + b * 9;
        const float* T = a.pose + b * 16;
        for (long long p0 = 0; p0 < plane; p0 += 1024) {
            const long long p = p0 + tid;
            bool keep = false;
            float depth = 0.f;
            int y = 0, x = 0;
            if (p < plane) {
                y = (int)(p / a.W); x = (int)(p - (long long)y * a.W);
                float d = a.inv_depth[b * plane + p];
                if (a.num_masks > 0) {                                   // torch.sum(torch.stack(mask_buffer), dim=0) > n - min_hits
                    float s = a.masks[0][b * plane + p];
                    for (int k = 1; k < a.num_masks; ++k) s += a.masks[k][b * plane + p];
                    d *= (s > a.vote_above) ? 1.f : 0.f;                 // depth *= mask
                }
                depth = 1.0f / d;                                        // ply_utils.py:36 (1/0 = inf fails the range test)
                keep = a.min_d <= depth && depth <= a.max_d;             // :38
                keep = keep && y >= a.roi[0] && y < a.roi[1] && x >= a.roi[2] && x < a.roi[3];          // :39-43
                if (a.uniform) keep = keep && a.uniform[b * plane + p] > a.dropout;                    // :44-45
            }
            // ordered compaction: ballot within the wave, wave totals through LDS
            const unsigned long long bal = __ballot(keep);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = __popcll(bal);
            __syncthreads();
            int off = 0, total = 0;
            for (int w = 0; w < 16; ++w) { const int t = wave_tot[w]; if (w < wave) off += t; total += t; }
            if (keep) {
                const long long rec = base + off + before;
                if (rec < a.capacity) {
                    const float fx = (float)x, fy = (float)y;
                    // cam_p_norm = inv_K[:3,:3] @ (x, y, 1)  (layers.py:57), ascending-k fma chain
                    float n0 = K[0] * fx; n0 = fmaf(K[1], fy, n0); n0 = fmaf(K[2], 1.0f, n0);
                    float n1 = K[3] * fx; n1 = fmaf(K[4], fy, n1); n1 = fmaf(K[5], 1.0f, n1);
                    float n2 = K[6] * fx; n2 = fmaf(K[7], fy, n2); n2 = fmaf(K[8], 1.0f, n2);
                    const float c0 = depth * n0, c1 = depth * n1, c2 = depth * n2;                      // :58
                    float* o = a.records + rec * 6;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {                                                       // extrinsics @ coords (ply_utils.py:48)
                        float v = T[i * 4 + 0] * c0;
                        v = fmaf(T[i * 4 + 1], c1, v);
                        v = fmaf(T[i * 4 + 2], c2, v);
                        v = fmaf(T[i * 4 + 3], 1.0f, v);
                        o[i] = v;
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) o[3 + c] = (a.image[(b * 3 + c) * plane + p] + 0.5f) * 255.0f;   // :37
                }
            }
            base += total;
            __syncthreads();                                             // wave_tot is reused
        }
    }
    if (tid == 0) *a.cursor = base;                                      // may exceed capacity: the host checks
}

}  // namespace

extern "C" int mr_static_mask_f32(const float* cv_mask, float* out, int32_t batch, int32_t height, int32_t width,
                                  float threshold, int32_t mask_fill, void* stream) {
    if (!cv_mask || !out || batch < 1 || height < 1 || width < 1 || mask_fill < 0 || (mask_fill & 1)) return MR_ERR_BAD_ARGUMENT;
    const int r = mask_fill / 2;
    const size_t lds = (size_t)(SM_TH + 2 * r) * (SM_TW + 2 * r) + (size_t)(SM_TH + 2 * r) * SM_TW;
    if (lds > 64 * 1024) return MR_ERR_LDS_BUDGET;
    dim3 grid((width + SM_TW - 1) / SM_TW, (height + SM_TH - 1) / SM_TH, batch);
    hipLaunchKernelGGL(static_mask_kernel, grid, dim3(256), lds, (hipStream_t)stream, cv_mask, out, height, width, threshold, r);
    return (int)hipGetLastError();
}

extern "C" int mr_pointcloud_append_f32(const float* inv_depth, const float* const* static_masks, int32_t num_masks,
                                        float vote_above, const float* image, const float* kinv, const float* pose,
                                        const float* uniform, float dropout, float min_d, float max_d, const int32_t* roi,
                                        int32_t batch, int32_t height, int32_t width, float* records, int64_t capacity_records,
                                        int64_t* cursor, void* stream) {
    if (!inv_depth || !image || !kinv || !pose || !records || !cursor || batch < 1 || height < 1 || width < 1) return MR_ERR_BAD_ARGUMENT;
    if (num_masks < 0 || num_masks > MR_MAX_VOTE_MASKS || (num_masks > 0 && !static_masks) || capacity_records < 0) return MR_ERR_BAD_ARGUMENT;
    PcArgs a;
    a.inv_depth = inv_depth;
    for (int k = 0; k < MR_MAX_VOTE_MASKS; ++k) a.masks[k] = k < num_masks ? static_masks[k] : nullptr;
    for (int k = 0; k < num_masks; ++k) if (!a.masks[k]) return MR_ERR_BAD_ARGUMENT;
    a.num_masks = num_masks; a.vote_above = vote_above;
    a.image = image; a.kinv = kinv; a.pose = pose; a.uniform = uniform;
    a.dropout = dropout; a.min_d = min_d; a.max_d = max_d;
    a.roi[0] = 0; a.roi[1] = height; a.roi[2] = 0; a.roi[3] = width;
    if (roi) {                                                          // python slicing semantics of mask[:, :, :r0] etc.
        for (int i = 0; i < 4; ++i) {
            int v = roi[i];
            const int n = i < 2 ? height : width;
            if (v < 0) v += n;
            a.roi[i] = v < 0 ? 0 : (v > n ? n : v);
        }
    }
    a.B = batch; a.H = height; a.W = width;
    a.records = records; a.capacity = capacity_records; a.cursor = (long long*)cursor;
    hipLaunchKernelGGL(pointcloud_append_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
