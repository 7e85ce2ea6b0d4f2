// Direct convolution on the fp32 matrix cores of gfx950 (MI355X).
//
// Stands in for every nn.Conv2d / ConvTranspose2d phase on the MonoRec inference path
// (reference: model/layers.py:241-252,289-356,380-400; model/monorec/monorec_model.py:118-129,
// 345-385,526-557).  See conv_layout.h for the GEMM view and the packed weight stream.
//
// Workgroup = 256 threads = 4 waves.  It owns a TH x (TWB*16) tile of one output plane and MB
// consecutive 16-channel output blocks.  Per K chunk (<=16 input channels) the haloed input tile is
// staged once into LDS (zero padding, nearest-upsample, 2x2 max-pool and the ResNet input
// normalisation are applied while staging, so those ops never touch HBM); every wave then walks
// taps x channel-quads, reading its B fragments from LDS (conflict-free: plane stride = 16 mod 32)
// and its A fragments straight from the L2-resident packed weight stream (register double buffer),
// and issues MB*NB v_mfma_f32_16x16x4_f32 per k-step.  fp32 MFMA is an exact fmaf chain, so results
// differ from the oneDNN CPU reference only by summation order.
//
// Epilogue: bias (eval-BatchNorm folded by the host), residual add, activation, scatter with an
// output step/offset (ConvTranspose2d phases) into a channel slice of the destination.
// split_k > 1 writes raw partial sums to a workspace; splitk_epilogue_kernel finishes them in a fixed
// order (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "../../include/monorec_hip.h"
#include "conv_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvKArgs {
    const float* src[MR_MAX_SOURCES];
    long long src_bstride[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];
    int nsrc;
    int Hs, Ws, Hin, Win;
    int in_mode, in_tf;
    int KH, KW, SH, SW, PT, PL;
    int Ho, Wo;
    float* dst;
    long long dst_bstride;
    int dst_H, dst_W, ch_off;
    int ostep_h, ostep_w, ooff_h, ooff_w;
    int Cout, CB;
    const float* w;
    const float* bias;
    const float* res;
    int act;
    float p0, p1;
    int tiles_x, TH, TWB;
    int IH, IW, PLANE;
    int ksplit, nchunks, batch;
    float* ws;
};

__device__ __forceinline__ float mr_activate(float v, int act, float p0, float p1) {
    switch (act) {
        case MR_ACT_RELU: return v > 0.f ? v : 0.f;
        case MR_ACT_LEAKY_RELU: return v > 0.f ? v : v * p0;
        case MR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case MR_ACT_ABS_TANH_AFFINE: {
            float t = fabsf(tanhf(v));
            return (1.f - t) * p0 + t * p1;   // monorec_model.py:717
        }
        default: return v;
    }
}

__device__ __forceinline__ void mr_store_out(const ConvKArgs& a, int b, int cout, int oy, int ox, float v) {
    if (a.bias) v += a.bias[cout];
    const long long idx = (long long)b * a.dst_bstride +
                          ((long long)(a.ch_off + cout) * a.dst_H + (oy * a.ostep_h + a.ooff_h)) * a.dst_W +
                          (ox * a.ostep_w + a.ooff_w);
    if (a.res) v += a.res[idx];
    a.dst[idx] = mr_activate(v, a.act, a.p0, a.p1);
}

__device__ __forceinline__ float mr_fetch(const float* plane, int gy, int gx, int Ws, int mode) {
    if (mode == MR_IN_DIRECT) return plane[gy * Ws + gx];
    if (mode == MR_IN_UPSAMPLE2) return plane[(gy >> 1) * Ws + (gx >> 1)];
    const float* p = plane + (2 * gy) * Ws + 2 * gx;   // MR_IN_MAXPOOL2
    return fmaxf(fmaxf(p[0], p[1]), fmaxf(p[Ws], p[Ws + 1]));
}

template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int cb0 = blockIdx.y * MB;
    const int b = blockIdx.z / a.ksplit, ks = blockIdx.z % a.ksplit;
    const int oy0 = ty * a.TH, ox0 = tx * a.TWB * 16;
    const int iy_base = oy0 * a.SH - a.PT, ix_base = ox0 * a.SW - a.PL;

    int prow[NB], pcol[NB], lbase[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pb = wave * NB + i;
        prow[i] = pb / a.TWB;
        pcol[i] = (pb % a.TWB) * 16 + (lane & 15);
        lbase[i] = (lane >> 4) * a.PLANE + prow[i] * a.SH * a.IW + pcol[i] * a.SW;
    }
    // clamp cout blocks of a partially filled last group onto valid weights (results are discarded)
    int wcb[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) wcb[m] = min(cb0 + m, a.CB - 1) * 64 + lane;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int q_lo = (ks * a.nchunks) / a.ksplit, q_hi = ((ks + 1) * a.nchunks) / a.ksplit;
    const int T = a.KH * a.KW;
    const int step_stride = a.CB * 64;
    long long woff = 0;
    int q = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const float* sbase = a.src[s] + (long long)b * a.src_bstride[s];
        for (int c0 = 0; c0 < a.src_cpad[s]; c0 += MR_CHUNK_CHANNELS, ++q) {
            const int ck = min(MR_CHUNK_CHANNELS, a.src_cpad[s] - c0);
            const int ck4 = ck >> 2;
            const int nsteps = T * ck4;
            if (q >= q_lo && q < q_hi) {
                __syncthreads();  // all waves finished reading the previous chunk
                // ---- stage the haloed input tile of channels [c0, c0+ck) -------------------------
                for (int c = 0; c < ck; ++c) {
                    const bool cok = (c0 + c) < a.src_c[s];
                    const float* plane = sbase + (long long)(c0 + c) * a.Hs * a.Ws;
                    for (int iy = wave; iy < a.IH; iy += 4) {
                        const int gy = iy_base + iy;
                        const bool yok = cok && gy >= 0 && gy < a.Hin;
                        float* lrow = lds + c * a.PLANE + iy * a.IW;
                        for (int ix = lane; ix < a.IW; ix += 64) {
                            const int gx = ix_base + ix;
                            float v = 0.f;
                            if (yok && gx >= 0 && gx < a.Win) {
                                v = mr_fetch(plane, gy, gx, a.Ws, a.in_mode);
                                if (a.in_tf == MR_TF_RESNET_NORM) v = ((v + 0.5f) - 0.45f) / 0.225f;
                            }
                            lrow[ix] = v;
                        }
                    }
                }
                __syncthreads();
                // ---- MFMA sweep ------------------------------------------------------------------
                const float* wq = a.w + woff;
                float a_cur[MB], a_nxt[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) a_cur[m] = wq[wcb[m]];
                int st = 0;
                for (int kh = 0; kh < a.KH; ++kh) {
                    for (int kw = 0; kw < a.KW; ++kw) {
                        const int tapoff = kh * a.IW + kw;
                        for (int c4 = 0; c4 < ck4; ++c4, ++st) {
                            const int nst = min(st + 1, nsteps - 1);
#pragma unroll
                            for (int m = 0; m < MB; ++m) a_nxt[m] = wq[nst * step_stride + wcb[m]];
                            const int off = c4 * 4 * a.PLANE + tapoff;
                            float bv[NB];
#pragma unroll
                            for (int i = 0; i < NB; ++i) bv[i] = lds[lbase[i] + off];
#pragma unroll
                            for (int m = 0; m < MB; ++m)
#pragma unroll
                                for (int i = 0; i < NB; ++i)
                                    acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m], bv[i], acc[m][i], 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < MB; ++m) a_cur[m] = a_nxt[m];
                        }
                    }
                }
            }
            woff += mr_chunk_weight_floats(ck, T, a.CB);
        }
    }

    // ---- epilogue: D fragment lane l holds pixel (l&15), couts (l>>4)*4 + r ---------------------
    const int CB16 = a.CB * 16;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int oy = oy0 + prow[i], ox = ox0 + pcol[i];
            if (oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = (cb0 + m) * 16 + (lane >> 4) * 4 + r;
                if (a.ksplit > 1) {
                    if (cout < CB16)
                        a.ws[((((long long)ks * a.batch + b) * CB16 + cout) * a.Ho + oy) * a.Wo + ox] = acc[m][i][r];
                } else if (cout < a.Cout) {
                    mr_store_out(a, b, cout, oy, ox, acc[m][i][r]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvKArgs a) {
    const long long plane = (long long)a.Ho * a.Wo;
    const long long total = (long long)a.batch * a.Cout * plane;
    const int CB16 = a.CB * 16;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % a.Wo);
        const int oy = (int)((i / a.Wo) % a.Ho);
        const int cout = (int)((i / plane) % a.Cout);
        const int b = (int)(i / (plane * a.Cout));
        float v = 0.f;
        for (int ks = 0; ks < a.ksplit; ++ks)
            v += a.ws[((((long long)ks * a.batch + b) * CB16 + cout) * a.Ho + oy) * a.Wo + ox];
        mr_store_out(a, b, cout, oy, ox, v);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct Derived {
    ConvKArgs k;
    int max_ck;
    size_t lds_bytes;
    int mb, nb;
    dim3 grid;
};

int derive(const mr_conv_desc* d, Derived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES) return MR_ERR_BAD_ARGUMENT;
    if (d->batch < 1 || d->kh < 1 || d->kw < 1 || d->stride_h < 1 || d->stride_w < 1) return MR_ERR_BAD_ARGUMENT;
    if (d->stride_w > 2) return MR_ERR_UNSUPPORTED;
    if (d->out_h < 1 || d->out_w < 1 || d->out_channels < 1 || !d->dst || !d->packed_weights) return MR_ERR_BAD_ARGUMENT;
    const int mb = d->cout_blocks_per_wg, nb = d->pixel_blocks_per_wave;
    if (!(mb == 1 || mb == 2 || mb == 3 || mb == 4 || mb == 6)) return MR_ERR_BAD_ARGUMENT;
    if (!(nb == 1 || nb == 2 || nb == 4)) return MR_ERR_BAD_ARGUMENT;
    if (d->split_k < 1 || (d->split_k > 1 && !d->workspace)) return MR_ERR_BAD_ARGUMENT;
    ConvKArgs& k = out->k;
    memset(&k, 0, sizeof(k));
    k.nsrc = d->num_src;
    k.Hs = d->src_h;
    k.Ws = d->src_w;
    switch (d->in_mode) {
        case MR_IN_DIRECT: k.Hin = k.Hs; k.Win = k.Ws; break;
        case MR_IN_UPSAMPLE2: k.Hin = 2 * k.Hs; k.Win = 2 * k.Ws; break;
        case MR_IN_MAXPOOL2: k.Hin = k.Hs / 2; k.Win = k.Ws / 2; break;
        default: return MR_ERR_BAD_ARGUMENT;
    }
    k.in_mode = d->in_mode;
    k.in_tf = d->in_transform;
    int nchunks = 0, max_ck = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        k.src[s] = d->src[s];
        k.src_c[s] = d->src_channels[s];
        k.src_cpad[s] = mr_pad4(d->src_channels[s]);
        k.src_bstride[s] = (long long)d->src_channels[s] * d->src_h * d->src_w;
        nchunks += mr_chunks_of(d->src_channels[s]);
        const int ck_s = k.src_cpad[s] < MR_CHUNK_CHANNELS ? k.src_cpad[s] : MR_CHUNK_CHANNELS;
        if (ck_s > max_ck) max_ck = ck_s;
    }
    if (d->split_k > nchunks) return MR_ERR_BAD_ARGUMENT;
    k.KH = d->kh; k.KW = d->kw; k.SH = d->stride_h; k.SW = d->stride_w; k.PT = d->pad_top; k.PL = d->pad_left;
    k.Ho = d->out_h; k.Wo = d->out_w;
    k.dst = d->dst;
    k.dst_H = d->dst_plane_h; k.dst_W = d->dst_plane_w;
    k.dst_bstride = (long long)d->dst_total_channels * d->dst_plane_h * d->dst_plane_w;
    k.ch_off = d->dst_channel_offset;
    k.ostep_h = d->out_step_h; k.ostep_w = d->out_step_w; k.ooff_h = d->out_off_h; k.ooff_w = d->out_off_w;
    if (k.ostep_h < 1 || k.ostep_w < 1) return MR_ERR_BAD_ARGUMENT;
    if ((k.Ho - 1) * k.ostep_h + k.ooff_h >= k.dst_H || (k.Wo - 1) * k.ostep_w + k.ooff_w >= k.dst_W) return MR_ERR_BAD_ARGUMENT;
    if (d->dst_channel_offset < 0 || d->dst_channel_offset + d->out_channels > d->dst_total_channels) return MR_ERR_BAD_ARGUMENT;
    k.Cout = d->out_channels;
    k.CB = mr_ceil_div(d->out_channels, 16);
    k.w = d->packed_weights; k.bias = d->bias; k.res = d->residual;
    k.act = d->activation; k.p0 = d->act_p0; k.p1 = d->act_p1;
    k.TWB = d->out_w >= 32 ? 2 : 1;
    k.TH = 4 * nb / k.TWB;
    k.tiles_x = mr_ceil_div(d->out_w, k.TWB * 16);
    const int tiles_y = mr_ceil_div(d->out_h, k.TH);
    k.IH = (k.TH - 1) * k.SH + k.KH;
    k.IW = (k.TWB * 16 - 1) * k.SW + k.KW;
    int plane = k.IH * k.IW;
    if (k.SW == 1) { while ((plane & 31) != 16) ++plane; } else { plane |= 1; }
    k.PLANE = plane;
    k.ksplit = d->split_k; k.nchunks = nchunks; k.batch = d->batch; k.ws = d->workspace;
    out->max_ck = max_ck;
    out->lds_bytes = (size_t)max_ck * plane * sizeof(float);
    if (out->lds_bytes > 160 * 1024) return MR_ERR_LDS_BUDGET;
    out->mb = mb; out->nb = nb;
    out->grid = dim3((unsigned)(k.tiles_x * tiles_y), (unsigned)mr_ceil_div(k.CB, mb), (unsigned)(d->batch * d->split_k));
    return 0;
}

template <int MB, int NB>
int launch(const Derived& dv, hipStream_t stream) {
    static bool attr_set = false;  // raise the dynamic-LDS ceiling once per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<MB, NB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB>), dv.grid, dim3(256), dv.lds_bytes, stream, dv.k);
    return (int)hipGetLastError();
}

template <int MB>
int launch_nb(const Derived& dv, hipStream_t stream) {
    switch (dv.nb) {
        case 1: return launch<MB, 1>(dv, stream);
        case 2: return launch<MB, 2>(dv, stream);
        default: return launch<MB, 4>(dv, stream);
    }
}

}  // namespace

extern "C" size_t mr_conv_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                               int32_t kh, int32_t kw) {
    const int cb = mr_ceil_div(out_channels, 16);
    size_t n = 0;
    for (int s = 0; s < num_src; ++s) {
        const int cpad = mr_pad4(src_channels[s]);
        for (int c0 = 0; c0 < cpad; c0 += MR_CHUNK_CHANNELS) {
            const int ck = cpad - c0 < MR_CHUNK_CHANNELS ? cpad - c0 : MR_CHUNK_CHANNELS;
            n += (size_t)mr_chunk_weight_floats(ck, kh * kw, cb);
        }
    }
    return n;
}

extern "C" int mr_conv_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels,
                                        int32_t num_src, int32_t kh, int32_t kw, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES) return MR_ERR_BAD_ARGUMENT;
    const int cb_n = mr_ceil_div(out_channels, 16);
    const int taps = kh * kw;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    size_t o = 0;
    int cin_off = 0;
    for (int s = 0; s < num_src; ++s) {
        const int cpad = mr_pad4(src_channels[s]);
        for (int c0 = 0; c0 < cpad; c0 += MR_CHUNK_CHANNELS) {
            const int ck = cpad - c0 < MR_CHUNK_CHANNELS ? cpad - c0 : MR_CHUNK_CHANNELS;
            for (int tap = 0; tap < taps; ++tap)
                for (int c4 = 0; c4 < ck / 4; ++c4)
                    for (int cb = 0; cb < cb_n; ++cb)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int cout = cb * 16 + (lane & 15);
                            const int cl = c0 + c4 * 4 + (lane >> 4);
                            float v = 0.f;
                            if (cout < out_channels && cl < src_channels[s])
                                v = weight[((size_t)cout * cin_total + (cin_off + cl)) * taps + tap];
                            dst[o++] = v;
                        }
        }
        cin_off += src_channels[s];
    }
    return 0;
}

extern "C" int64_t mr_conv2d_lds_bytes(const mr_conv_desc* desc) {
    Derived dv;
    const int rc = derive(desc, &dv);
    if (rc != 0) return rc;
    return (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv2d_f32(const mr_conv_desc* desc, void* stream_) {
    Derived dv;
    int rc = derive(desc, &dv);
    if (rc != 0) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    switch (dv.mb) {
        case 1: rc = launch_nb<1>(dv, stream); break;
        case 2: rc = launch_nb<2>(dv, stream); break;
        case 3: rc = launch_nb<3>(dv, stream); break;
        case 4: rc = launch_nb<4>(dv, stream); break;
        default: rc = launch_nb<6>(dv, stream); break;
    }
    if (rc != 0) return rc;
    if (dv.k.ksplit > 1) {
        const long long total = (long long)dv.k.batch * dv.k.Cout * dv.k.Ho * dv.k.Wo;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, dv.k);
        rc = (int)hipGetLastError();
    }
    return rc;
}
