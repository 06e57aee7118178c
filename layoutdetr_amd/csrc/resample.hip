// Page-background preprocessing of a dataset item on the device:
//     background = np.array(PIL_image.resize((S, S), PIL.Image.ANTIALIAS))           training/dataset_layoutganpp.py:333
//     background = (background.astype(np.float32) / 255.0 - rgb_mean) / rgb_std      :335
//     background = background.transpose(2, 0, 1)                                     :336
// The resize is Pillow's 8-bit two-pass resampling (src/libImaging/Resample.c, Pillow==9.3.0 in the reference's environment.yaml:46)
// with the Lanczos-3 window: per output coordinate a window of weights (double, normalised, 22-bit fixed point, round half away
// from zero), out = clip8((2^21 + sum px * k) >> 22); horizontal pass into a uint8 intermediate, then the vertical pass.  Integer
// arithmetic end to end, so the device result is bit-identical to Pillow's; the fp32 normalisation uses correctly rounded
// division / subtraction in the reference's order, so the float tensor is bit-identical too.
//
// Byte work whose algorithmic traffic is tiny (per image: read H*W*3 bytes once, write and re-read the H*S*3 intermediate, write
// 3*S*S floats); the passes are bound by instruction issue instead: ~75 integer multiply-adds + 75 LDS byte reads per output pixel
// of the horizontal pass (ablation: without the LDS reads 85 of 144 us remain, the bare multiply-add loop is 57 us).  Horizontal pass: one block per (row, 256 output columns), the row segment staged in LDS with 4-byte loads;
// vertical pass: plane-major threads so the float stores are fully coalesced.  Weights are stored tap-major ([ksize][out]) so
// neighbouring lanes read neighbouring weights.
#include <math.h>

#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;
constexpr int RS_LDS_BYTES = 48 * 1024;

struct ResampleParams {
    const unsigned char* src; long src_bytes;
    unsigned char* tmp; unsigned char* out_u8; float* out_chw;
    const int* hb; const int* hk; int hks;
    const int* vb; const int* vk; int vks;
    int H, W, OH, OW;
    int lds_bytes;
    float mean[3], stdv[3];
};

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> RS_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (ceil(OW / 256), H, images)
__global__ __launch_bounds__(256) void resample_h_kernel(ResampleParams p) {
    extern __shared__ unsigned char rs_row[];
    const int x0 = blockIdx.x * 256, x1 = min(x0 + 255, p.OW - 1);
    const int y = blockIdx.y;
    const long img = blockIdx.z;
    const int first = p.hb[2 * x0], last = p.hb[2 * x1] + p.hb[2 * x1 + 1];          // input columns [first, last) feed this block
    const long row_off = ((img * p.H + y) * (long)p.W + first) * 3;                    // byte offset of column `first`
    const int span = (last - first) * 3;
    const bool staged = span + 8 <= p.lds_bytes;
    const int lead = (int)(row_off & 3);                                               // staged copy starts at the aligned word below
    if (staged) {
        const long base = row_off - lead;
        const int words = (lead + span + 3) >> 2;
        for (int i = threadIdx.x; i < words; i += 256) {
            const long o = base + 4L * i;
            unsigned int w;
            if (o + 4 <= p.src_bytes) w = *reinterpret_cast<const unsigned int*>(p.src + o);
            else {                                                                     // last word of the buffer: byte-wise
                w = 0;
                for (int b = 0; b < 4; b++) if (o + b < p.src_bytes) w |= (unsigned int)p.src[o + b] << (8 * b);
            }
            reinterpret_cast<unsigned int*>(rs_row)[i] = w;
        }
        __syncthreads();
    }
    const int xx = x0 + threadIdx.x;
    if (xx >= p.OW) return;
    const int xmin = p.hb[2 * xx], cnt = p.hb[2 * xx + 1];
    int a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    if (staged) {
        const unsigned char* px = rs_row + lead + (xmin - first) * 3;
        for (int t = 0; t < cnt; t++) {
            const int k = p.hk[(long)t * p.OW + xx];
            a0 += px[3 * t] * k; a1 += px[3 * t + 1] * k; a2 += px[3 * t + 2] * k;
        }
    } else {
        const unsigned char* px = p.src + ((img * p.H + y) * (long)p.W + xmin) * 3;
        for (int t = 0; t < cnt; t++) {
            const int k = p.hk[(long)t * p.OW + xx];
            a0 += px[3 * t] * k; a1 += px[3 * t + 1] * k; a2 += px[3 * t + 2] * k;
        }
    }
    unsigned char* o = p.tmp + ((img * p.H + y) * (long)p.OW + xx) * 3;
    o[0] = (unsigned char)clip8(a0); o[1] = (unsigned char)clip8(a1); o[2] = (unsigned char)clip8(a2);
}

// grid (ceil(OW / 256), OH, images * 3): one colour plane per block so the float stores are contiguous
__global__ __launch_bounds__(256) void resample_v_norm_kernel(ResampleParams p) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= p.OW) return;
    const int yy = blockIdx.y;
    const long img = blockIdx.z / 3; const int c = blockIdx.z % 3;
    const int ymin = p.vb[2 * yy], cnt = p.vb[2 * yy + 1];
    const unsigned char* col = p.tmp + ((img * p.H + ymin) * (long)p.OW + x) * 3 + c;
    const long pitch = (long)p.OW * 3;
    int acc = 1 << (RS_PRECISION_BITS - 1);
    if (p.vks <= 32) {                          // every tap's byte load in flight at once (8-bit x 23-bit: the 24-bit multiply is full rate)
        int v[32];
#pragma unroll
        for (int t = 0; t < 32; t++) v[t] = t < cnt ? (int)col[t * pitch] : 0;
#pragma unroll
        for (int t = 0; t < 32; t++)
            if (t < cnt) acc += __mul24(v[t], p.vk[(long)t * p.OH + yy]);
    } else {
        for (int t = 0; t < cnt; t++) acc += __mul24((int)col[t * pitch], p.vk[(long)t * p.OH + yy]);
    }
    const int u8 = clip8(acc);
    if (p.out_u8) p.out_u8[((img * p.OH + yy) * (long)p.OW + x) * 3 + c] = (unsigned char)u8;
    if (p.out_chw) {
        const float v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u8, 255.0f), p.mean[c]), p.stdv[c]);
        p.out_chw[((img * 3 + c) * (long)p.OH + yy) * p.OW + x] = v;
    }
}

static double rs_sinc(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double rs_lanczos3(double x) { return (-3.0 <= x && x < 3.0) ? rs_sinc(x) * rs_sinc(x / 3.0) : 0.0; }

}  // namespace ldetr

using namespace ldetr;

extern "C" int ldetr_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int64_t kk_capacity, int* ksize_out) {
    LDETR_CHECK(in_size > 0 && out_size > 0 && ksize_out, "resample_coeffs: bad arguments");
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!bounds && !kk) return LDETR_OK;                        // size query
    LDETR_CHECK(bounds && kk && kk_capacity >= (int64_t)ksize * out_size, "resample_coeffs: weight buffer too small (need ksize * out_size ints)");
    const double ss = 1.0 / filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; xx++) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; x++) { w[x] = rs_lanczos3((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < ksize; x++) {
            int q = 0;
            if (x < xmax) {
                const double v = ww != 0.0 ? w[x] / ww : w[x];
                q = v < 0 ? (int)(-0.5 + v * (1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (1 << RS_PRECISION_BITS));
            }
            kk[(int64_t)x * out_size + xx] = q;                 // tap-major
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return LDETR_OK;
}

extern "C" int ldetr_resize_normalize_u8(const uint8_t* src, int64_t images, int H, int W, int out_h, int out_w, const int32_t* hbounds,
                                         const int32_t* hweights, int hksize, const int32_t* vbounds, const int32_t* vweights, int vksize,
                                         uint8_t* tmp, uint8_t* out_u8, float* out_chw, float mean0, float mean1, float mean2, float std0,
                                         float std1, float std2, void* stream) {
    LDETR_CHECK(src && tmp && hbounds && hweights && vbounds && vweights, "resize_normalize: null pointer");
    LDETR_CHECK(out_u8 || out_chw, "resize_normalize: no output requested");
    LDETR_CHECK(images >= 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && hksize > 0 && vksize > 0, "resize_normalize: bad shape");
    LDETR_CHECK(images * 3 <= 65535 && H <= 65535 && out_h <= 65535, "resize_normalize: grid limit (images * 3, rows <= 65535)");
    LDETR_CHECK((((uintptr_t)src) & 3) == 0, "resize_normalize: source must be 4-byte aligned");
    LDETR_CHECK(std0 != 0.f && std1 != 0.f && std2 != 0.f, "resize_normalize: zero std");
    if (images == 0) return LDETR_OK;
    ResampleParams p; memset(&p, 0, sizeof(p));
    p.src = src; p.src_bytes = images * (long)H * W * 3; p.tmp = tmp; p.out_u8 = out_u8; p.out_chw = out_chw;
    p.hb = hbounds; p.hk = hweights; p.hks = hksize; p.vb = vbounds; p.vk = vweights; p.vks = vksize;
    p.H = H; p.W = W; p.OH = out_h; p.OW = out_w;
    p.mean[0] = mean0; p.mean[1] = mean1; p.mean[2] = mean2; p.stdv[0] = std0; p.stdv[1] = std1; p.stdv[2] = std2;
    hipStream_t st = (hipStream_t)stream;
    // LDS for the widest row segment a block of 256 output columns can need (falls back to direct reads beyond 48 KiB)
    const double scale = (double)W / out_w, support = 3.0 * (scale < 1.0 ? 1.0 : scale);
    long lds = ((long)(256.0 * scale + 2.0 * support) + 4) * 3 + 16;
    p.lds_bytes = (int)(lds > RS_LDS_BYTES ? RS_LDS_BYTES : (lds + 15) / 16 * 16);
    hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(out_w, 256), H, (unsigned)images), 256, p.lds_bytes, st, p);
    int rc = check_launch("resample_h"); if (rc) return rc;
    hipLaunchKernelGGL(resample_v_norm_kernel, dim3(cdiv(out_w, 256), out_h, (unsigned)(images * 3)), 256, 0, st, p);
    return check_launch("resample_v_norm");
}
