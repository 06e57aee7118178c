// Batched linear sum assignment (shortest augmenting path, Crouse 2016) — one problem per lane.
// The reference has no Hungarian kernel of its own: its only assignment arithmetic is
// scipy.optimize.linear_sum_assignment (metrics/metric_layoutnet.py:111,125,240; scipy pinned at
// 1.6.3 in environment.yaml:44).  This restates scipy's published rectangular_lsap algorithm for
// square n <= 64 problems (n <= 16: arrays and visit masks sized for the layouts of the training sets; 17..64: the
// same code with 64-entry arrays and 64-bit masks), including its tie rules (columns scanned from a "remaining" list that is
// filled in reverse, ties broken in favour of unassigned columns), so indices are bit-exact.
#pragma clang fp contract(off)
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

#define LSAP_MAXN 64

namespace ldetr {

template <int MAXN, typename MaskT>
__global__ __launch_bounds__(64) void lsap_kernel(const double* cost_all, int batch, int n, int maximize, int* row_ind, int* col_ind) {
    const int pb = blockIdx.x * blockDim.x + threadIdx.x;
    if (pb >= batch) return;
    const double* cin = cost_all + (long)pb * n * n;
    const double sgn = maximize ? -1.0 : 1.0;
    double u[MAXN], v[MAXN], spc[MAXN];
    int path[MAXN], col4row[MAXN], row4col[MAXN], remaining[MAXN];
    MaskT SR, SC;  // row / column visit sets as bit masks
    for (int i = 0; i < n; i++) { u[i] = 0.0; v[i] = 0.0; path[i] = -1; col4row[i] = -1; row4col[i] = -1; }
    bool infeasible = false;
    for (int cur = 0; cur < n && !infeasible; cur++) {
        double minVal = 0.0;
        int num_remaining = n;
        for (int it = 0; it < n; it++) { remaining[it] = n - it - 1; spc[it] = (double)INFINITY; }
        SR = 0; SC = 0;
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = (double)INFINITY;
            SR |= (MaskT)1 << i;
            for (int it = 0; it < num_remaining; it++) {
                int j = remaining[it];
                double r = minVal + sgn * cin[i * n + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                const double sj = spc[j];
                bool take = sj < lowest;
                if (!take && sj == lowest && row4col[j] == -1) take = true;
                if (take) { lowest = sj; index = it; }
            }
            minVal = lowest;
            if (minVal == (double)INFINITY) { infeasible = true; break; }
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC |= (MaskT)1 << j;
            remaining[index] = remaining[--num_remaining];
        }
        if (infeasible) break;
        u[cur] += minVal;
        for (int r = 0; r < n; r++) if (((SR >> r) & 1) && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < n; j++) if ((SC >> j) & 1) v[j] -= minVal - spc[j];
        int j = sink;
        while (true) {
            int r = path[j];
            row4col[j] = r;
            int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
    for (int i = 0; i < n; i++) {
        row_ind[(long)pb * n + i] = infeasible ? -1 : i;
        col_ind[(long)pb * n + i] = infeasible ? -1 : col4row[i];
    }
}

}  // namespace ldetr

extern "C" int ldetr_lsap_f64(const double* cost, int batch, int n, int maximize, int* row_ind, int* col_ind, void* stream) {
    using namespace ldetr;
    LDETR_CHECK(cost && row_ind && col_ind, "lsap: null pointer");
    LDETR_CHECK(n >= 1 && n <= LSAP_MAXN, "lsap: n must be in [1, %d]", LSAP_MAXN);
    LDETR_CHECK(batch >= 0, "lsap: negative batch");
    if (batch == 0) return LDETR_OK;
    if (n <= 16) hipLaunchKernelGGL((lsap_kernel<16, unsigned>), cdiv(batch, 64), 64, 0, (hipStream_t)stream, cost, batch, n, maximize, row_ind, col_ind);
    else hipLaunchKernelGGL((lsap_kernel<64, unsigned long long>), cdiv(batch, 64), 64, 0, (hipStream_t)stream, cost, batch, n, maximize, row_ind, col_ind);
    return check_launch("lsap");
}
