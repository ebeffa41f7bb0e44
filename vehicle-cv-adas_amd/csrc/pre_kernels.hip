// pre_kernels.hip -- frame pre-processing on the device: one BGR u8 frame per stream is
// uploaded once (2.76 MB at 1280x720) and both networks' input tensors are produced from it in HBM.
//
//   adas_preprocess_yolo  <- YoloDetector.__prepare_input (yoloDetector.py:96-102) = Scaler.process_image
//                            (utils.py:42-63: letterbox, cv2.resize INTER_LINEAR, canvas 114) +
//                            cv2.dnn.blobFromImage(1/255, swapRB=True) -> (N,3,H,W) fp32
//   adas_preprocess_ufld  <- UltrafastLaneDetectorV2.__prepare_input (ultrafastLaneDetectorV2.py:96-112):
//                            BGR->RGB, resize to (W, int(H/crop_ratio)), keep the bottom H rows,
//                            ((x/255 - mean)/std) with the reference's float32/float64 promotion -> fp32
//
// cv2.resize is third-party arithmetic (opencv-python==4.5.4.60, requirements.txt:1) and not available here:
// the bilinear step restates OpenCV's 8-bit INTER_LINEAR reference path (imgproc/src/resize.cpp: float source
// coordinate (dx+0.5)*scale-0.5, 11-bit fixed-point coefficients, two-pass HResize/VResize rounding
// ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2).  Parity with a real cv2 build is UNPINNED; identity-size
// resizes are exact by construction.  HBM-bound streaming kernels: one thread per output pixel.
#include "common.h"
#include <stdlib.h>
#include "elem16.h"
#include <math.h>

namespace {

struct ResizeGeom {
    int sh, sw;        // source
    int rh, rw;        // resized image
    double scale_y, scale_x;
};

__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int* s0, int* s1, int* c0, int* c1, bool horizontal) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {  // resize.cpp: out-of-range taps collapse onto the border pixel with weight 1
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    *c0 = (int)rintf((1.f - f) * 2048.f);
    *c1 = (int)rintf(f * 2048.f);
    int a = s, b = s + 1;
    a = a < 0 ? 0 : (a > ssize - 1 ? ssize - 1 : a);
    b = b < 0 ? 0 : (b > ssize - 1 ? ssize - 1 : b);
    *s0 = a;
    *s1 = b;
}

// resized pixel (ry, rx) of frame `src` (HWC u8, 3 channels) -> 3 u8 values (source channel order)
__device__ __forceinline__ void resize_px(const uint8_t* __restrict__ src, const ResizeGeom& g, int ry, int rx, int out[3]) {
    if (g.rh == g.sh && g.rw == g.sw) {  // cv2.resize with dsize == ssize copies
        const uint8_t* p = src + ((size_t)ry * g.sw + rx) * 3;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
        return;
    }
    int x0, x1, a0, a1, y0, y1, b0, b1;
    lin_coef(rx, g.scale_x, g.sw, &x0, &x1, &a0, &a1, true);
    lin_coef(ry, g.scale_y, g.sh, &y0, &y1, &b0, &b1, false);
    const uint8_t* r0 = src + (size_t)y0 * g.sw * 3;
    const uint8_t* r1 = src + (size_t)y1 * g.sw * 3;
    if (x1 == x0 + 1 && x0 * 3 + 8 <= g.sw * 3) {
        // both taps of a row are 6 consecutive bytes: one (unaligned) 8-byte load per row instead of six byte loads
        unsigned long long w0, w1;
        __builtin_memcpy(&w0, r0 + x0 * 3, 8);
        __builtin_memcpy(&w1, r1 + x0 * 3, 8);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = (int)((w0 >> (8 * c)) & 0xff) * a0 + (int)((w0 >> (8 * (3 + c))) & 0xff) * a1;
            const int h1 = (int)((w1 >> (8 * c)) & 0xff) * a0 + (int)((w1 >> (8 * (3 + c))) & 0xff) * a1;
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
        const int h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
}

// Row-wise form of resize_px for the streaming kernels below: a workgroup owns a few output rows of one frame, the column taps and
// coefficients (identical for every row) are tabulated once per workgroup in LDS, the row taps are workgroup-uniform.  Same
// arithmetic as resize_px, value for value (lin_coef is evaluated with the same arguments): the outputs are bit-identical; what
// goes away is two integer divisions (one 64-bit) and two double-precision coordinate evaluations per pixel.
struct ColTap {
    short x0, x1, a0, a1;   // source columns, 11-bit coefficients (<= 2048)
};
// output rows per workgroup: every workgroup tabulates the column taps (double-precision coordinates) and the normalisation LUT
// (double divisions) before its first pixel, but fewer, larger workgroups cost more than they save -- measured at 64 frames
// (tools/bench_pre.py, bit-identical outputs): detector 107-111 us at 3-4 rows, 120 at 8, 165 at 32, 549 at 160; lane 154-155 us at
// 3-16 rows except 8 (162), 210 at 32.  ADAS_PRE_ROWS overrides both (A/B measurements).
static int pre_rows(int dflt) {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("ADAS_PRE_ROWS");
        v = e ? atoi(e) : -1;
        if (v < 1 || v > 1024) v = -1;
    }
    return v > 0 ? v : dflt;
}
constexpr int PRE_MAXW = 7680;    // widest resized row the (dynamic LDS) table holds: 60 KB + the kernels' <= 3 KB of static LDS stay under
                                  // the 64 KB a launch gets without hipFuncSetAttribute(MaxDynamicSharedMemorySize)
__device__ __forceinline__ void col_table(ColTap* tab, const ResizeGeom& g, int first, int count) {
    for (int t = threadIdx.x; t < count; t += blockDim.x) {
        int x0, x1, a0, a1;
        lin_coef(first + t, g.scale_x, g.sw, &x0, &x1, &a0, &a1, true);
        tab[t] = ColTap{(short)x0, (short)x1, (short)a0, (short)a1};
    }
}
// resized pixel of row taps (r0, r1, b0, b1) and column entry c; identity resizes are handled by the caller
__device__ __forceinline__ void resize_row_px(const uint8_t* __restrict__ r0, const uint8_t* __restrict__ r1, int b0, int b1, const ColTap c,
                                              int sw, int out[3]) {
    const int x0 = c.x0, x1 = c.x1, a0 = c.a0, a1 = c.a1;
    if (x1 == x0 + 1 && x0 * 3 + 8 <= sw * 3) {
        unsigned long long w0, w1;
        __builtin_memcpy(&w0, r0 + x0 * 3, 8);
        __builtin_memcpy(&w1, r1 + x0 * 3, 8);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int h0 = (int)((w0 >> (8 * ch)) & 0xff) * a0 + (int)((w0 >> (8 * (3 + ch))) & 0xff) * a1;
            const int h1 = (int)((w1 >> (8 * ch)) & 0xff) * a0 + (int)((w1 >> (8 * (3 + ch))) & 0xff) * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out[ch] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        return;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int h0 = r0[x0 * 3 + ch] * a0 + r0[x1 * 3 + ch] * a1;
        const int h1 = r1[x0 * 3 + ch] * a0 + r1[x1 * 3 + ch] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out[ch] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
}

// PACK: write the frame as the fused first layer stages it anyway -- (c0, c1, c2, 0) bf16 pixels, 8 B each, NHWC -- instead of
// three fp32 planes: the same round-to-nearest-even conversion conv_stem.hip applies to the fp32 seam tensor, so the network sees
// identical values while the tensor between pre-processing and stem shrinks from 12 to 8 bytes per pixel.
__device__ __forceinline__ uint32_t pre_bf16(float f) {  // round to nearest even on the bits (finite inputs): what v_cvt_pk_bf16_f32 does;
    const uint32_t u = __float_as_uint(f);                 // written out so that the float result above is materialised first and the
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;         // double -> float -> bf16 roundings cannot be merged into one
}
__device__ __forceinline__ uint32_t pre_pack2(float a, float b) { return pre_bf16(a) | (pre_bf16(b) << 16); }

struct YoloPreDev {
    const uint8_t* src;
    float* dst;
    ResizeGeom g;
    int n, dh, dw, padh, padw;
    int rows;   // output rows per workgroup
};
template <int PACK>   // 0: fp32 NCHW planes; 1: packed bf16 pixels; 2: packed fp16 pixels
__global__ void preprocess_yolo_kernel(YoloPreDev d) {
    const size_t plane = (size_t)d.dh * d.dw;
    // a u8 value has 256 images under the normalisation: tabulate them once per workgroup with the reference's arithmetic
    // (blobFromImage: float32(v) * (1/255.0) evaluated in double) instead of redoing the double-precision step per pixel
    __shared__ float lut[256];
    extern __shared__ ColTap tab[];   // [resized width]
    for (int t = threadIdx.x; t < 256; t += blockDim.x) lut[t] = (float)((double)t * (1.0 / 255.0));
    const bool identity = d.g.rh == d.g.sh && d.g.rw == d.g.sw;   // cv2.resize with dsize == ssize copies
    if (!identity) col_table(tab, d.g, 0, d.g.rw);
    __syncthreads();
    const int PRE_ROWS = d.rows;
    const int rows_per_img = (d.dh + PRE_ROWS - 1) / PRE_ROWS;
    const int b = blockIdx.x / rows_per_img, yb = (blockIdx.x - b * rows_per_img) * PRE_ROWS;
    const uint8_t* src = d.src + (size_t)b * d.g.sh * d.g.sw * 3;
    for (int yy = 0; yy < PRE_ROWS && yb + yy < d.dh; ++yy) {
        const int y = yb + yy, ry = y - d.padh;
        const bool row_in = ry >= 0 && ry < d.g.rh;
        int y0 = 0, y1 = 0, b0 = 0, b1 = 0;
        if (row_in && !identity) lin_coef(ry, d.g.scale_y, d.g.sh, &y0, &y1, &b0, &b1, false);
        const uint8_t* r0 = src + (size_t)y0 * d.g.sw * 3;
        const uint8_t* r1 = src + (size_t)y1 * d.g.sw * 3;
        for (int x = threadIdx.x; x < d.dw; x += blockDim.x) {
            int v[3] = {114, 114, 114};  // canvas (utils.py:54)
            const int rx = x - d.padw;
            if (row_in && rx >= 0 && rx < d.g.rw) {
                if (identity) {
                    const uint8_t* q = src + ((size_t)ry * d.g.sw + rx) * 3;
                    v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
                } else {
                    resize_row_px(r0, r1, b0, b1, tab[rx], d.g.sw, v);
                }
            }
            // swapRB: plane 0 = R = source channel 2
            const float c0 = lut[v[2]], c1 = lut[v[1]], c2 = lut[v[0]];
            const size_t p = (size_t)y * d.dw + x, i = (size_t)b * plane + p;
            if (PACK == 2) {
                reinterpret_cast<uint2*>(d.dst)[i] = make_uint2(adas::Fp16::pack2(c0, c1), adas::Fp16::pack2(c2, 0.f));
            } else if (PACK == 1) {
                reinterpret_cast<uint2*>(d.dst)[i] = make_uint2(pre_pack2(c0, c1), pre_pack2(c2, 0.f));
            } else {
                float* o = d.dst + (size_t)b * 3 * plane + p;
                o[0] = c0;
                o[plane] = c1;
                o[2 * plane] = c2;
            }
        }
    }
}

// EfficientdetDetector.__prepare_input (efficientdetDetector.py:57-65): the letterboxed u8 canvas, then (image / 255 - mean) / std
// on the BGR channels in place (no swap), all in float64 (uint8 array / int -> float64), cast to float32 at the end.
__global__ void preprocess_effdet_kernel(YoloPreDev d) {
    const size_t plane = (size_t)d.dh * d.dw;
    const size_t total = (size_t)d.n * plane;
    const double mean[3] = {0.406, 0.456, 0.485}, stdv[3] = {0.225, 0.224, 0.229};
    __shared__ float lut[3][256];
    for (int t = threadIdx.x; t < 768; t += blockDim.x) {
        const int c = t >> 8, u = t & 255;
        lut[c][u] = (float)(((double)u / 255.0 - mean[c]) / stdv[c]);
    }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / plane);
        const int p = (int)(i - (size_t)b * plane);
        const int y = p / d.dw, x = p - y * d.dw;
        int v[3] = {114, 114, 114};  // canvas (utils.py:54)
        const int ry = y - d.padh, rx = x - d.padw;
        if (ry >= 0 && ry < d.g.rh && rx >= 0 && rx < d.g.rw) resize_px(d.src + (size_t)b * d.g.sh * d.g.sw * 3, d.g, ry, rx, v);
        float* o = d.dst + (size_t)b * 3 * plane + p;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(size_t)c * plane] = lut[c][v[c]];
    }
}

struct UfldPreDev {
    const uint8_t* src;
    float* dst;
    ResizeGeom g;
    int n, ih, iw, row0;
    int rows;   // output rows per workgroup
};
template <int PACK>
__global__ void preprocess_ufld_kernel(UfldPreDev d) {
    const size_t plane = (size_t)d.ih * d.iw;
    const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
    // 256 possible u8 values per channel: the reference's float32 / float64 normalisation is tabulated once per workgroup
    // (three double-precision divisions per pixel otherwise: the kernel was ALU-bound at 1.9 TB/s)
    __shared__ float lut[3][256];
    extern __shared__ ColTap tab[];   // [input width]
    for (int t = threadIdx.x; t < 768; t += blockDim.x) {
        const int c = t >> 8, u = t & 255;
        const float q = (float)u / 255.0f;                                 // float32 array / Python float stays float32
        lut[c][u] = (float)(((double)q - mean[c]) / stdv[c]);             // - list, / list promote to float64
    }
    const bool identity = d.g.rh == d.g.sh && d.g.rw == d.g.sw;
    if (!identity) col_table(tab, d.g, 0, d.iw);
    __syncthreads();
    const int PRE_ROWS = d.rows;
    const int rows_per_img = (d.ih + PRE_ROWS - 1) / PRE_ROWS;
    const int b = blockIdx.x / rows_per_img, yb = (blockIdx.x - b * rows_per_img) * PRE_ROWS;
    const uint8_t* src = d.src + (size_t)b * d.g.sh * d.g.sw * 3;
    for (int yy = 0; yy < PRE_ROWS && yb + yy < d.ih; ++yy) {
        const int y = yb + yy, ry = d.row0 + y;
        int y0 = 0, y1 = 0, b0 = 0, b1 = 0;
        if (!identity) lin_coef(ry, d.g.scale_y, d.g.sh, &y0, &y1, &b0, &b1, false);
        const uint8_t* r0 = src + (size_t)y0 * d.g.sw * 3;
        const uint8_t* r1 = src + (size_t)y1 * d.g.sw * 3;
        for (int x = threadIdx.x; x < d.iw; x += blockDim.x) {
            int v[3];
            if (identity) {
                const uint8_t* q = src + ((size_t)ry * d.g.sw + x) * 3;
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
            } else {
                resize_row_px(r0, r1, b0, b1, tab[x], d.g.sw, v);
            }
            float cv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) cv[c] = lut[c][v[2 - c]];  // RGB plane c = BGR source channel 2-c
            const size_t p = (size_t)y * d.iw + x, i = (size_t)b * plane + p;
            if (PACK == 2) {
                reinterpret_cast<uint2*>(d.dst)[i] = make_uint2(adas::Fp16::pack2(cv[0], cv[1]), adas::Fp16::pack2(cv[2], 0.f));
            } else if (PACK == 1) {
                reinterpret_cast<uint2*>(d.dst)[i] = make_uint2(pre_pack2(cv[0], cv[1]), pre_pack2(cv[2], 0.f));
            } else {
                float* o = d.dst + (size_t)b * 3 * plane + p;
#pragma unroll
                for (int c = 0; c < 3; ++c) o[(size_t)c * plane] = cv[c];
            }
        }
    }
}

int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 16384 ? b : 16384);
}

}  // namespace

extern "C" {

static int preprocess_yolo_impl(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int dst_h, int dst_w,
                                int keep_ratio, void* stream, int pack /* 0 fp32 planes | 1 bf16 pixels | 2 fp16 pixels | 3 EfficientDet normalisation */) {
    ADAS_REQUIRE(d_frames_bgr && d_out_nchw && n > 0 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, ADAS_ERR_INVALID,
                 "adas_preprocess_yolo: bad argument");
    adas_yolo_post_params lb;
    int rc = adas_letterbox_params(src_h, src_w, dst_h, dst_w, keep_ratio, &lb);
    if (rc) return rc;
    YoloPreDev d;
    d.rows = 0;
    d.src = d_frames_bgr; d.dst = d_out_nchw; d.n = n; d.dh = dst_h; d.dw = dst_w; d.padh = lb.pad_h; d.padw = lb.pad_w;
    // resized extent = Scaler._new_shape (utils.py:43-52): recover it from the ratios' definition
    int newh = dst_h, neww = dst_w;
    if (keep_ratio && src_h != src_w) {
        const double hw = (double)src_h / (double)src_w;
        if (hw > 1) neww = (int)((double)dst_w / hw);
        else newh = (int)((double)dst_h * hw) + 1;
    }
    d.g.sh = src_h; d.g.sw = src_w; d.g.rh = newh; d.g.rw = neww;
    d.g.scale_y = 1.0 / ((double)newh / (double)src_h);
    d.g.scale_x = 1.0 / ((double)neww / (double)src_w);
    if (pack == 3) hipLaunchKernelGGL(preprocess_effdet_kernel, dim3(grid_for((size_t)n * dst_h * dst_w)), dim3(256), 0, (hipStream_t)stream, d);
    else {
        ADAS_REQUIRE(neww <= PRE_MAXW && src_w < 32768 && src_h < 32768, ADAS_ERR_INVALID, "adas_preprocess_yolo: resized width %d exceeds %d", neww, PRE_MAXW);
        d.rows = pre_rows(4);
        const dim3 grid((unsigned)(n * ((dst_h + d.rows - 1) / d.rows)));
        const size_t tab_bytes = (size_t)neww * sizeof(ColTap);
        if (pack == 2) hipLaunchKernelGGL(preprocess_yolo_kernel<2>, grid, dim3(256), tab_bytes, (hipStream_t)stream, d);
        else if (pack == 1) hipLaunchKernelGGL(preprocess_yolo_kernel<1>, grid, dim3(256), tab_bytes, (hipStream_t)stream, d);
        else hipLaunchKernelGGL(preprocess_yolo_kernel<0>, grid, dim3(256), tab_bytes, (hipStream_t)stream, d);
    }
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_preprocess_yolo(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int dst_h, int dst_w,
                         int keep_ratio, void* stream) {
    return preprocess_yolo_impl(d_frames_bgr, n, src_h, src_w, d_out_nchw, dst_h, dst_w, keep_ratio, stream, 0);
}
int adas_preprocess_effdet(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int dst_h, int dst_w, int keep_ratio,
                           void* stream) {
    return preprocess_yolo_impl(d_frames_bgr, n, src_h, src_w, d_out_nchw, dst_h, dst_w, keep_ratio, stream, 3);
}
int adas_preprocess_yolo_packed(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int dst_h, int dst_w,
                                int keep_ratio, void* stream) {
    return preprocess_yolo_impl(d_frames_bgr, n, src_h, src_w, reinterpret_cast<float*>(d_out_nhwc4), dst_h, dst_w, keep_ratio, stream, 1);
}
int adas_preprocess_yolo_packed_prec(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int dst_h, int dst_w,
                                     int keep_ratio, int precision, void* stream) {
    ADAS_REQUIRE(precision == ADAS_PREC_BF16 || precision == ADAS_PREC_FP16, ADAS_ERR_INVALID, "packed pixels are bf16 or fp16 (precision %d)", precision);
    return preprocess_yolo_impl(d_frames_bgr, n, src_h, src_w, reinterpret_cast<float*>(d_out_nhwc4), dst_h, dst_w, keep_ratio, stream,
                                precision == ADAS_PREC_FP16 ? 2 : 1);
}

static int preprocess_ufld_impl(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int in_h, int in_w,
                                double crop_ratio, void* stream, int pack) {
    ADAS_REQUIRE(d_frames_bgr && d_out_nchw && n > 0 && src_h > 0 && src_w > 0 && in_h > 0 && in_w > 0 && crop_ratio > 0.0 &&
                     crop_ratio <= 1.0, ADAS_ERR_INVALID, "adas_preprocess_ufld: bad argument");
    UfldPreDev d;
    d.src = d_frames_bgr; d.dst = d_out_nchw; d.n = n; d.ih = in_h; d.iw = in_w;
    const int rh = (int)((double)in_h / crop_ratio);  // ultrafastLaneDetectorV2.py:101
    ADAS_REQUIRE(rh >= in_h, ADAS_ERR_INVALID, "resized height %d is smaller than the network input %d", rh, in_h);
    d.row0 = rh - in_h;                               // img_input[-input_height:, :, :]
    d.g.sh = src_h; d.g.sw = src_w; d.g.rh = rh; d.g.rw = in_w;
    d.g.scale_y = 1.0 / ((double)rh / (double)src_h);
    d.g.scale_x = 1.0 / ((double)in_w / (double)src_w);
    ADAS_REQUIRE(in_w <= PRE_MAXW && src_w < 32768 && src_h < 32768, ADAS_ERR_INVALID, "adas_preprocess_ufld: input width %d exceeds %d", in_w, PRE_MAXW);
    d.rows = pre_rows(12);
    const dim3 grid((unsigned)(n * ((in_h + d.rows - 1) / d.rows)));
    const size_t tab_bytes = (size_t)in_w * sizeof(ColTap);
    if (pack == 2) hipLaunchKernelGGL(preprocess_ufld_kernel<2>, grid, dim3(256), tab_bytes, (hipStream_t)stream, d);
    else if (pack == 1) hipLaunchKernelGGL(preprocess_ufld_kernel<1>, grid, dim3(256), tab_bytes, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(preprocess_ufld_kernel<0>, grid, dim3(256), tab_bytes, (hipStream_t)stream, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_preprocess_ufld(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int in_h, int in_w,
                         double crop_ratio, void* stream) {
    return preprocess_ufld_impl(d_frames_bgr, n, src_h, src_w, d_out_nchw, in_h, in_w, crop_ratio, stream, 0);
}
int adas_preprocess_ufld_packed(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int in_h, int in_w,
                                double crop_ratio, void* stream) {
    return preprocess_ufld_impl(d_frames_bgr, n, src_h, src_w, reinterpret_cast<float*>(d_out_nhwc4), in_h, in_w, crop_ratio, stream, 1);
}
int adas_preprocess_ufld_packed_prec(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int in_h, int in_w,
                                     double crop_ratio, int precision, void* stream) {
    ADAS_REQUIRE(precision == ADAS_PREC_BF16 || precision == ADAS_PREC_FP16, ADAS_ERR_INVALID, "packed pixels are bf16 or fp16 (precision %d)", precision);
    return preprocess_ufld_impl(d_frames_bgr, n, src_h, src_w, reinterpret_cast<float*>(d_out_nhwc4), in_h, in_w, crop_ratio, stream,
                                precision == ADAS_PREC_FP16 ? 2 : 1);
}

}  // extern "C"
