// the per-frame image preprocessing of KittiOdometryDataset.preprocess_image
// (data_loader/kitti_odometry_dataset.py:120-134) on the device:
//     img.crop(box) -> img.resize((W, H), Image.BILINEAR) -> float32 / 255 - .5 -> CHW   (grey: 3 stacked copies)
// The resize is Pillow's (src/libImaging/Resample.c): triangle filter widened by the scale factor, weights in 22-bit
// fixed point, horizontal pass into an 8-bit intermediate, then the vertical pass - integer work, reproduced bit for
// bit.  mr_resample_coeffs_bilinear is the host part (double arithmetic exactly as precompute_coeffs /
// normalize_coeffs_8bpc); the kernel runs both passes for a 16 x 64 output tile through an LDS intermediate.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/monorec_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int PT_H = 16, PT_W = 64;

struct PreArgs {
    const unsigned char* src;
    long long row_stride;
    int channels, x0, y0;                   // crop origin in the source image
    int out_h, out_w;
    const int* hb; const int* hk; int hks;   // bounds (first, count) per output column, coefficients [out_w][hks]
    const int* vb; const int* vk; int vks;
    int max_rows;                            // LDS rows per tile
    float* dst;
};

__device__ __forceinline__ unsigned char clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void preprocess_kernel(const PreArgs a) {
    extern __shared__ unsigned char tmp[];                 // [rows][PT_W][channels] horizontal-pass output
    const int ox0 = blockIdx.x * PT_W, oy0 = blockIdx.y * PT_H;
    const int oy_last = min(oy0 + PT_H, a.out_h) - 1;
    const int r_first = a.vb[2 * oy0];
    const int r_end = a.vb[2 * oy_last] + a.vb[2 * oy_last + 1];
    const int rows = r_end - r_first, C = a.channels;
    const int cols = min(PT_W, a.out_w - ox0);
    for (int e = threadIdx.x; e < rows * cols; e += 256) {
        const int r = e / cols, xx = e - r * cols;
        const int first = a.hb[2 * (ox0 + xx)], n = a.hb[2 * (ox0 + xx) + 1];
        const int* k = a.hk + (long long)(ox0 + xx) * a.hks;
        const unsigned char* p = a.src + (long long)(a.y0 + r_first + r) * a.row_stride + (long long)(a.x0 + first) * C;
        for (int c = 0; c < C; ++c) {
            int acc = 1 << (PRECISION_BITS - 1);
            for (int t = 0; t < n; ++t) acc += (int)p[t * C + c] * k[t];
            tmp[(r * PT_W + xx) * C + c] = clip8(acc);
        }
    }
    __syncthreads();
    const long long plane = (long long)a.out_h * a.out_w;
    for (int e = threadIdx.x; e < PT_H * cols; e += 256) {
        const int yy = e / cols, xx = e - yy * cols;
        const int oy = oy0 + yy;
        if (oy >= a.out_h) continue;
        const int first = a.vb[2 * oy] - r_first, n = a.vb[2 * oy + 1];
        const int* k = a.vk + (long long)oy * a.vks;
        float v[3];
        for (int c = 0; c < C; ++c) {
            int acc = 1 << (PRECISION_BITS - 1);
            for (int t = 0; t < n; ++t) acc += (int)tmp[((first + t) * PT_W + xx) * C + c] * k[t];
            v[c] = (float)clip8(acc) / 255.0f - 0.5f;      // kitti_odometry_dataset.py:128
        }
        const long long o = (long long)oy * a.out_w + ox0 + xx;
        if (C == 1) { a.dst[o] = v[0]; a.dst[plane + o] = v[0]; a.dst[2 * plane + o] = v[0]; }     // :130
        else { a.dst[o] = v[0]; a.dst[plane + o] = v[1]; a.dst[2 * plane + o] = v[2]; }            // :132
    }
}

// ---- sparse lidar ground truth: preprocess_depth_annotated_lidar (kitti_odometry_dataset.py:184-211) ----------------
// Every non-zero pixel of the 16-bit depth PNG (depth * 256) inside the crop box becomes one inverse-depth sample
// 256 / value at the nearest cell (np.around, half to even) of the target grid; when several samples fall into one
// cell numpy's fancy assignment keeps the last one in row-major source order.  Pass 1 elects that sample per cell
// (atomicMax of the source index), pass 2 writes it - same double arithmetic as numpy, deterministic.
__global__ __launch_bounds__(256) void lidar_elect_kernel(const unsigned short* __restrict__ png, int H, int W, int x0, int y0,
                                                          int x1, int y1, int out_h, int out_w, int* __restrict__ owner) {
    const long long n = (long long)H * W;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (png[i] == 0) continue;
        const int r = (int)(i / W), c = (int)(i - (long long)r * W);
        if (r < y0 || r >= y1 || c < x0 || c >= x1) continue;
        double ty = (double)(r - y0) / (double)(y1 - y0) * (double)out_h;          // :205
        double tx = (double)(c - x0) / (double)(x1 - x0) * (double)out_w;          // :206
        ty = fmin(fmax(ty, 0.0), (double)(out_h - 1));
        tx = fmin(fmax(tx, 0.0), (double)(out_w - 1));
        const int cy = (int)rint(ty), cx = (int)rint(tx);                          // np.around
        atomicMax(owner + cy * out_w + cx, (int)i);
    }
}

__global__ __launch_bounds__(256) void lidar_write_kernel(const unsigned short* __restrict__ png, const int* __restrict__ owner,
                                                          int cells, float* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const int o = owner[i];
        out[i] = o >= 0 ? (float)(256.0 / (double)png[o]) : 0.f;                   // :191, then float32 (:211)
    }
}

// ---- sparse D(V)SO ground truth: preprocess_depth_dso (kitti_odometry_dataset.py:156-182) -----------------------------
// Same scatter, but the source coordinates are first rescaled to the original image size (:160-161, a no-op up to double
// rounding when the depth PNG has the image's size), the crop test runs on those doubles (:169) and the stored value is
// the inverse depth w * value / (0.54 * f_x * 65535) (:164).
__global__ __launch_bounds__(256) void dso_elect_kernel(const unsigned short* __restrict__ png, int H, int W, int orig_h, int orig_w,
                                                        int x0, int y0, int x1, int y1, int out_h, int out_w, int* __restrict__ owner) {
    const long long n = (long long)H * W;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (png[i] == 0) continue;
        const int r = (int)(i / W), c = (int)(i - (long long)r * W);
        const double ry = fmin(fmax((double)r / (double)H * (double)orig_h, 0.0), (double)(orig_h - 1));   // :160
        const double rx = fmin(fmax((double)c / (double)W * (double)orig_w, 0.0), (double)(orig_w - 1));   // :161
        if (!((double)y0 <= ry && ry < (double)y1 && (double)x0 <= rx && rx < (double)x1)) continue;       // :169
        double ty = (ry - (double)y0) / (double)(y1 - y0) * (double)out_h;         // :178
        double tx = (rx - (double)x0) / (double)(x1 - x0) * (double)out_w;         // :179
        ty = fmin(fmax(ty, 0.0), (double)(out_h - 1));
        tx = fmin(fmax(tx, 0.0), (double)(out_w - 1));
        const int cy = (int)rint(ty), cx = (int)rint(tx);                          // np.around
        atomicMax(owner + cy * out_w + cx, (int)i);
    }
}

__global__ __launch_bounds__(256) void dso_write_kernel(const unsigned short* __restrict__ png, const int* __restrict__ owner,
                                                        int cells, double orig_w, double denom, float* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const int o = owner[i];
        out[i] = o >= 0 ? (float)(orig_w * (double)png[o] / denom) : 0.f;           // :164, then float32 (:182)
    }
}

double triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace

extern "C" int32_t mr_resample_ksize_bilinear(int32_t in0, int32_t in1, int32_t out_size) {
    if (out_size < 1 || in1 <= in0) return MR_ERR_BAD_ARGUMENT;
    double filterscale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int32_t)ceil(1.0 * filterscale) * 2 + 1;
}

extern "C" int mr_resample_coeffs_bilinear(int32_t in_size, int32_t in0, int32_t in1, int32_t out_size,
                                           int32_t* bounds, int32_t* coeffs) {
    if (!bounds || !coeffs || in_size < 1 || out_size < 1 || in1 <= in0 || in0 < 0 || in1 > in_size) return MR_ERR_BAD_ARGUMENT;
    const double scale = (double)(in1 - in0) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = triangle((x + xmin - center + 0.5) * ss); ww += w[x]; }
        int32_t* k = coeffs + (long long)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return 0;
}

extern "C" int mr_preprocess_image_u8_f32(const uint8_t* src, int32_t src_h, int32_t src_w, int32_t channels,
                                          int64_t row_stride_bytes, const int32_t* box, int32_t out_h, int32_t out_w,
                                          const int32_t* hbounds, const int32_t* hcoeffs, int32_t hksize,
                                          const int32_t* vbounds, const int32_t* vcoeffs, int32_t vksize,
                                          int32_t max_tile_rows, float* dst, void* stream) {
    if (!src || !box || !hbounds || !hcoeffs || !vbounds || !vcoeffs || !dst) return MR_ERR_BAD_ARGUMENT;
    if ((channels != 1 && channels != 3) || out_h < 1 || out_w < 1 || hksize < 1 || vksize < 1 || max_tile_rows < 1) return MR_ERR_BAD_ARGUMENT;
    if (box[0] < 0 || box[1] < 0 || box[2] > src_w || box[3] > src_h || box[2] <= box[0] || box[3] <= box[1]) return MR_ERR_BAD_ARGUMENT;
    if (row_stride_bytes < (int64_t)src_w * channels) return MR_ERR_BAD_ARGUMENT;
    const size_t lds = (size_t)max_tile_rows * PT_W * channels;
    if (lds > 64 * 1024) return MR_ERR_LDS_BUDGET;
    PreArgs a;
    a.src = src; a.row_stride = row_stride_bytes; a.channels = channels; a.x0 = box[0]; a.y0 = box[1];
    a.out_h = out_h; a.out_w = out_w;
    a.hb = hbounds; a.hk = hcoeffs; a.hks = hksize; a.vb = vbounds; a.vk = vcoeffs; a.vks = vksize;
    a.max_rows = max_tile_rows; a.dst = dst;
    dim3 grid((out_w + PT_W - 1) / PT_W, (out_h + PT_H - 1) / PT_H);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mr_lidar_inverse_depth_u16_f32(const uint16_t* depth_png, int32_t src_h, int32_t src_w, const int32_t* box,
                                              int32_t out_h, int32_t out_w, int32_t* owner_scratch, float* dst, void* stream) {
    if (!depth_png || !owner_scratch || !dst || src_h < 1 || src_w < 1 || out_h < 1 || out_w < 1) return MR_ERR_BAD_ARGUMENT;
    if ((long long)src_h * src_w >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
    int x0 = 0, y0 = 0, x1 = src_w, y1 = src_h;
    if (box) { x0 = box[0]; y0 = box[1]; x1 = box[2]; y1 = box[3]; }
    if (x1 <= x0 || y1 <= y0) return MR_ERR_BAD_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const int cells = out_h * out_w;
    hipError_t e = hipMemsetAsync(owner_scratch, 0xff, (size_t)cells * sizeof(int), st);      // -1 everywhere
    if (e != hipSuccess) return (int)e;
    const long long n = (long long)src_h * src_w;
    hipLaunchKernelGGL(lidar_elect_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st,
                       depth_png, src_h, src_w, x0, y0, x1, y1, out_h, out_w, owner_scratch);
    hipLaunchKernelGGL(lidar_write_kernel, dim3((unsigned)((cells + 255) / 256 < 2048 ? (cells + 255) / 256 : 2048)), dim3(256), 0, st,
                       depth_png, owner_scratch, cells, dst);
    return (int)hipGetLastError();
}

extern "C" int mr_dso_inverse_depth_u16_f32(const uint16_t* depth_png, int32_t src_h, int32_t src_w, int32_t orig_h, int32_t orig_w,
                                            double focal_x, const int32_t* box, int32_t out_h, int32_t out_w,
                                            int32_t* owner_scratch, float* dst, void* stream) {
    if (!depth_png || !owner_scratch || !dst || src_h < 1 || src_w < 1 || orig_h < 1 || orig_w < 1 || out_h < 1 || out_w < 1 ||
        !(focal_x > 0.0)) return MR_ERR_BAD_ARGUMENT;
    if ((long long)src_h * src_w >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
    int x0 = 0, y0 = 0, x1 = orig_w, y1 = orig_h;
    if (box) { x0 = box[0]; y0 = box[1]; x1 = box[2]; y1 = box[3]; }
    if (x1 <= x0 || y1 <= y0) return MR_ERR_BAD_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const int cells = out_h * out_w;
    hipError_t e = hipMemsetAsync(owner_scratch, 0xff, (size_t)cells * sizeof(int), st);      // -1 everywhere
    if (e != hipSuccess) return (int)e;
    const long long n = (long long)src_h * src_w;
    hipLaunchKernelGGL(dso_elect_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st,
                       depth_png, src_h, src_w, orig_h, orig_w, x0, y0, x1, y1, out_h, out_w, owner_scratch);
    hipLaunchKernelGGL(dso_write_kernel, dim3((unsigned)((cells + 255) / 256 < 2048 ? (cells + 255) / 256 : 2048)), dim3(256), 0, st,
                       depth_png, owner_scratch, cells, (double)orig_w, 0.54 * focal_x * 65535.0, dst);
    return (int)hipGetLastError();
}
