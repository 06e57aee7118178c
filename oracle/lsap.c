/* ORACLE (test infrastructure only).  Plain-C restatement of the rectangular linear-sum-assignment
 * solver published in SciPy (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp; Crouse 2016,
 * "On implementing 2D rectangular assignment algorithms"), square case.  The reference repo has no
 * Hungarian implementation of its own: it calls scipy.optimize.linear_sum_assignment
 * (metrics/metric_layoutnet.py:15,111,125,240; scipy==1.6.3 pinned in environment.yaml:44).
 * Pinned against scipy itself in tests/test_oracle_golden.py (tests/golden/lsap.npz + live scipy).
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/liblsap_oracle.so oracle/lsap.c
 */
#include <math.h>
#include <stdlib.h>

int lsap_oracle(const double* cost_in, int n, int maximize, int* row_ind, int* col_ind) {
    if (n <= 0) return 0;
    double* cost = (double*)malloc(sizeof(double) * n * n);
    double* u = (double*)calloc(n, sizeof(double));
    double* v = (double*)calloc(n, sizeof(double));
    double* spc = (double*)malloc(sizeof(double) * n);
    int* path = (int*)malloc(sizeof(int) * n);
    int* col4row = (int*)malloc(sizeof(int) * n);
    int* row4col = (int*)malloc(sizeof(int) * n);
    int* remaining = (int*)malloc(sizeof(int) * n);
    char* SR = (char*)malloc(n);
    char* SC = (char*)malloc(n);
    int rc = 0;
    for (int i = 0; i < n * n; i++) cost[i] = maximize ? -cost_in[i] : cost_in[i];
    for (int i = 0; i < n; i++) { path[i] = -1; col4row[i] = -1; row4col[i] = -1; }
    for (int cur = 0; cur < n; cur++) {
        double minVal = 0.0;
        int num_remaining = n;
        for (int it = 0; it < n; it++) { remaining[it] = n - it - 1; SR[it] = 0; SC[it] = 0; spc[it] = INFINITY; }
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; it++) {
                int j = remaining[it];
                double r = minVal + cost[i * n + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INFINITY) { rc = -1; goto done; }
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < n; r++) if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < n; j++) if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        for (;;) {
            int r = path[j];
            row4col[j] = r;
            int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
    for (int i = 0; i < n; i++) { row_ind[i] = i; col_ind[i] = col4row[i]; }
done:
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining); free(SR); free(SC);
    return rc;
}
