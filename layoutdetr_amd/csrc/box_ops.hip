// Pairwise box IoU / union / generalised IoU (reference: detr_util/box_ops.py:19-71 -- box_cxcywh_to_xyxy, box_iou,
// generalized_box_iou; north_star's "bbox Hungarian-matched regression head") as ONE launch: one thread per (batch, i, j) pair, the
// 9 x 9 (<= 64 x 64) matrices of a whole batch at once, optionally written a second time as the float64 cost matrix
// (cost = sign * GIoU) that ldetr_lsap_f64 consumes -- the DETR matcher's cost_giou = -generalized_box_iou(...) feeds the Hungarian
// solve without a host round trip or a dtype-conversion launch.
// Arithmetic: the reference's fp32 operations in the reference's order, one IEEE operation each (no fma contraction, correctly rounded
// division), so the matrices are bit-identical to the reference's CPU results -- which is what makes the assignment indices
// bit-exact, ties included.
#pragma clang fp contract(off)
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct BoxPairParams {
    const float* b1; const float* b2;   // [B, N, 4], [B, M, 4]
    int B, N, M, cxcywh;
    float* iou; float* uni; float* giou;   // [B, N, M] each, any may be null
    double* cost; double cost_sign;        // [B, N, M] float64, may be null
};

__device__ __forceinline__ void load_xyxy(const float* p, int cxcywh, float& x0, float& y0, float& x1, float& y1) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    if (cxcywh) {   // box_ops.py:19-23: (x_c - 0.5 w, y_c - 0.5 h, x_c + 0.5 w, y_c + 0.5 h)
        x0 = v.x - 0.5f * v.z; y0 = v.y - 0.5f * v.w; x1 = v.x + 0.5f * v.z; y1 = v.y + 0.5f * v.w;
    } else { x0 = v.x; y0 = v.y; x1 = v.z; y1 = v.w; }
}

__global__ __launch_bounds__(256) void box_pair_kernel(BoxPairParams p) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.B * p.N * p.M;
    if (t >= total) return;
    const int j = (int)(t % p.M); const long bi = t / p.M; const int i = (int)(bi % p.N); const int b = (int)(bi / p.N);
    float ax0, ay0, ax1, ay1, bx0, by0, bx1, by1;
    load_xyxy(p.b1 + ((long)b * p.N + i) * 4, p.cxcywh, ax0, ay0, ax1, ay1);
    load_xyxy(p.b2 + ((long)b * p.M + j) * 4, p.cxcywh, bx0, by0, bx1, by1);
    const float area1 = (ax1 - ax0) * (ay1 - ay0), area2 = (bx1 - bx0) * (by1 - by0);      // torchvision box_area
    const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f), ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);   // (rb - lt).clamp(min=0)
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = inter / uni;
    const float cw = fmaxf(fmaxf(ax1, bx1) - fminf(ax0, bx0), 0.f), ch = fmaxf(fmaxf(ay1, by1) - fminf(ay0, by0), 0.f);
    const float area = cw * ch;
    const float giou = iou - (area - uni) / area;
    if (p.iou) p.iou[t] = iou;
    if (p.uni) p.uni[t] = uni;
    if (p.giou) p.giou[t] = giou;
    if (p.cost) p.cost[t] = p.cost_sign * (double)giou;
}

}  // namespace ldetr

extern "C" int ldetr_box_giou_pairwise_f32(const float* boxes1, const float* boxes2, int B, int N, int M, int cxcywh, float* iou, float* uni,
                                           float* giou, double* cost, double cost_sign, void* stream) {
    using namespace ldetr;
    LDETR_CHECK(boxes1 && boxes2, "box_giou_pairwise: null boxes");
    LDETR_CHECK(B >= 0 && N >= 0 && M >= 0, "box_giou_pairwise: negative size");
    LDETR_CHECK((((uintptr_t)boxes1 | (uintptr_t)boxes2) & 15) == 0, "box_giou_pairwise: boxes must be 16-byte aligned [.., 4] fp32 rows");
    const long total = (long)B * N * M;
    if (total == 0) return LDETR_OK;
    LDETR_CHECK(total <= 0x7fffffffL, "box_giou_pairwise: too many pairs");
    BoxPairParams p{boxes1, boxes2, B, N, M, cxcywh, iou, uni, giou, cost, cost_sign};
    hipLaunchKernelGGL(box_pair_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("box_giou_pairwise");
}
