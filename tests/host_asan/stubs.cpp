// Host-only stand-ins for the kernel entry points the stack sequencers (csrc/stack_exec.hip) call.  The sanitizer driver
// hands the sequencers HOST buffers sized exactly as include/sh_kernels.h specifies; every stub reads / writes the first and
// the last element of each operand range its real counterpart would touch, so AddressSanitizer sees any pointer the
// sequencer derived past the end of a buffer, and logs the call for the order check.  No GPU, no launches.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/sh_kernels.h"

std::vector<std::string> g_calls;
static void touch_r(const void* p, size_t bytes) {
    if (!p || !bytes) return;
    volatile char c = static_cast<const volatile char*>(p)[0];
    c = static_cast<const volatile char*>(p)[bytes - 1];
    (void)c;
}
static void touch_w(void* p, size_t bytes) {
    if (!p || !bytes) return;
    static_cast<volatile char*>(p)[0] = 1;
    static_cast<volatile char*>(p)[bytes - 1] = 1;
}
// bytes spanned by `rows` rows of a (row stride sv, batch stride sb, C channels) tensor of element size e
static size_t span(int64_t sv, int64_t sb, int rows, int B, int C, size_t e) {
    return (size_t)((int64_t)(rows - 1) * sv + (int64_t)(B - 1) * sb + C) * e;
}
static void log(const char* fmt, int a = 0, int b = 0, int c = 0) {
    char buf[160];
    snprintf(buf, sizeof buf, fmt, a, b, c);
    g_calls.push_back(buf);
}

extern "C" {
int sh_spiral_conv_fwd(const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, const float* w, const float* bias, float* y, int64_t y_sv,
                       int64_t y_sb, int B, int R, int S, int Cin, int Cout, int act, int zero_row, int, sh_stream_t) {
    int n_in = 0;
    for (long i = 0; i < (long)R * S; ++i) n_in = table[i] + 1 > n_in ? table[i] + 1 : n_in;
    touch_r(x, span(x_sv, x_sb, n_in, B, Cin, 4)); touch_r(w, (size_t)Cout * S * Cin * 4); touch_r(bias, bias ? Cout * 4 : 0);
    touch_w(y, span(y_sv, y_sb, R, B, Cout, 4));
    log("conv_fwd R=%d Cin=%d Cout=%d", R, Cin, Cout);
    return 0;
}
int sh_to_p3(const float* x, int64_t x_sv, int64_t x_sb, void* planes, int B, int rows, int C, sh_stream_t);
int sh_spiral_conv_fwd_img(const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, const float* w, const float* bias, float* y, int64_t y_sv,
                           int64_t y_sb, void* y_planes, int B, int R, int S, int Cin, int Cout, int act, int zero_row, int mma, sh_stream_t st) {
    const int rc = sh_spiral_conv_fwd(x, x_sv, x_sb, table, w, bias, y, y_sv, y_sb, B, R, S, Cin, Cout, act, zero_row, mma, st);
    return (rc == 0 && y_planes) ? sh_to_p3(y, y_sv, y_sb, y_planes, B, R, Cout, st) : rc;
}
int sh_spiral_conv_fwd_bf16(const void* x, int xd, int64_t x_sv, int64_t x_sb, const int32_t* table, const void* wfrag, const float* bias, void* y,
                            int yd, int64_t y_sv, int64_t y_sb, int B, int R, int S, int Cin, int Cout, int act, int zero_row, sh_stream_t) {
    int n_in = 0;
    for (long i = 0; i < (long)R * S; ++i) n_in = table[i] + 1 > n_in ? table[i] + 1 : n_in;
    touch_r(x, span(x_sv, x_sb, n_in, B, Cin, xd == SH_DTYPE_BF16 ? 2 : 4)); touch_r(wfrag, sh_conv_wfrag_bytes(S, Cin, Cout));
    touch_w(y, span(y_sv, y_sb, R, B, Cout, yd == SH_DTYPE_BF16 ? 2 : 4));
    log("conv_fwd_bf16 R=%d Cin=%d Cout=%d", R, Cin, Cout);
    return 0;
}
static int spmm_common(const char* name, const int32_t* rowptr, const int32_t* col, const float* val, const void* x, int64_t x_sv, int64_t x_sb,
                       void* y, int64_t y_sv, int64_t y_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int B, int rows, int C, size_t e) {
    int cols = 0;
    for (int i = 0; i < rowptr[rows]; ++i) cols = col[i] + 1 > cols ? col[i] + 1 : cols;
    touch_r(val, (size_t)rowptr[rows] * 4);
    if (cols) touch_r(x, span(x_sv, x_sb, cols, B, C, e));
    touch_w(y, span(y_sv, y_sb, rows, B, C, e));
    if (yprev) touch_r(yprev, span(yp_sv, yp_sb, rows, B, C, e));
    log(name, rows, C);
    return 0;
}
int sh_spmm(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t x_sv, int64_t x_sb, float* y, int64_t y_sv,
            int64_t y_sb, const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int rows, int C, sh_stream_t) {
    return spmm_common("spmm rows=%d C=%d", rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb, yprev, yp_sv, yp_sb, B, rows, C, 4);
}
size_t sh_p3_bytes(int rows, int B, int C);
int sh_spmm_p3(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t x_sv, int64_t x_sb, float* y, int64_t y_sv,
               int64_t y_sb, void* y_planes, const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int rows, int C,
               sh_stream_t) {
    if (y_planes) touch_w(y_planes, sh_p3_bytes(rows, B, C));
    return spmm_common(y_planes ? "spmm+image rows=%d C=%d" : "spmm rows=%d C=%d", rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb, yprev, yp_sv,
                       yp_sb, B, rows, C, 4);
}
int sh_spmm_bf16(const int32_t* rowptr, const int32_t* col, const float* val, const void* x, int64_t x_sv, int64_t x_sb, void* y, int64_t y_sv,
                 int64_t y_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int rows, int C, sh_stream_t) {
    return spmm_common("spmm_bf16 rows=%d C=%d", rowptr, col, val, x, x_sv, x_sb, y, y_sv, y_sb, yprev, yp_sv, yp_sb, B, rows, C, 2);
}
int sh_act_backward(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dp, int64_t dp_sv, int64_t dp_sb,
                    int B, int R, int C, int act, int zero_row, sh_stream_t) {
    touch_r(dy, span(dy_sv, dy_sb, R, B, C, 4)); touch_r(y, span(y_sv, y_sb, R, B, C, 4)); touch_w(dp, span(dp_sv, dp_sb, R, B, C, 4));
    log("act_backward R=%d C=%d", R, C);
    return 0;
}
int sh_act_backward_tr(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dp, int64_t dp_sv, int64_t dp_sb,
                       int B, int R, int C, int act, int zero_row, int n, const float* const* w, float* const* wt, const int* S, const int* Ci,
                       const int* Co, sh_stream_t) {
    for (int i = 0; i < n; ++i) { touch_r(w[i], (size_t)S[i] * Ci[i] * Co[i] * 4); touch_w(wt[i], (size_t)S[i] * Ci[i] * Co[i] * 4); }
    touch_r(dy, span(dy_sv, dy_sb, R, B, C, 4)); touch_r(y, span(y_sv, y_sb, R, B, C, 4)); touch_w(dp, span(dp_sv, dp_sb, R, B, C, 4));
    log("act_backward R=%d C=%d transposes=%d", R, C, n);
    return 0;
}
size_t sh_p3_bytes(int rows, int B, int C);
int sh_act_backward_tr_img(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dp, int64_t dp_sv,
                           int64_t dp_sb, void* dpre_planes, int B, int R, int C, int act, int zero_row, int n, const float* const* w,
                           float* const* wt, const int* S, const int* Ci, const int* Co, sh_stream_t st) {
    if (dpre_planes) touch_w(dpre_planes, sh_p3_bytes(R, B, C));
    return sh_act_backward_tr(dy, dy_sv, dy_sb, y, y_sv, y_sb, dp, dp_sv, dp_sb, B, R, C, act, zero_row, n, w, wt, S, Ci, Co, st);
}
int sh_act_backward_bf16(const void* dy, int64_t dy_sv, int64_t dy_sb, const void* y, int64_t y_sv, int64_t y_sb, void* dp, int64_t dp_sv, int64_t dp_sb,
                         int B, int R, int C, int act, int zero_row, sh_stream_t) {
    touch_r(dy, span(dy_sv, dy_sb, R, B, C, 2)); touch_r(y, span(y_sv, y_sb, R, B, C, 2)); touch_w(dp, span(dp_sv, dp_sb, R, B, C, 2));
    log("act_backward_bf16 R=%d C=%d", R, C);
    return 0;
}
size_t sh_spiral_conv_bwd_wgt_workspace(int B, int R, int S, int Cin, int Cout) { return (size_t)7 * ((size_t)Cout * S * Cin + Cout) * 4; }
size_t sh_spiral_conv_bwd_wgt_workspace_bf16(int B, int R, int S, int Cin, int Cout) { return (size_t)5 * ((size_t)Cout * S * Cin + Cout) * 4; }
int sh_spiral_conv_bwd_wgt(const float* dpre, int64_t dp_sv, int64_t dp_sb, const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, float* dW,
                           float* db, void* ws, size_t ws_bytes, int B, int R, int S, int Cin, int Cout, int, sh_stream_t) {
    int n_in = 0;
    for (long i = 0; i < (long)R * S; ++i) n_in = table[i] + 1 > n_in ? table[i] + 1 : n_in;
    touch_r(dpre, span(dp_sv, dp_sb, R, B, Cout, 4)); touch_r(x, span(x_sv, x_sb, n_in, B, Cin, 4));
    if (ws_bytes < sh_spiral_conv_bwd_wgt_workspace(B, R, S, Cin, Cout)) return SH_ERR_WORKSPACE;
    touch_w(ws, ws_bytes);
    log("bwd_wgt R=%d Cin=%d Cout=%d", R, Cin, Cout);
    return 0;
}
int sh_spiral_conv_bwd_wgt_presum(const float* dpre, int64_t dp_sv, int64_t dp_sb, const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table,
                                  float* dW, float* db, void* ws, size_t ws_bytes, const int32_t* sum_rowptr, const int32_t* sum_col,
                                  const float* sum_val, float* sum_out, void* sum_out_planes, int sum_rows, int B, int R, int S, int Cin, int Cout, int mma,
                                  sh_stream_t st) {
    // the rider reads and writes exactly what the sh_spmm launch it replaces does; logged in the order the work is visible in
    const int rc = sh_spiral_conv_bwd_wgt(dpre, dp_sv, dp_sb, x, x_sv, x_sb, table, dW, db, ws, ws_bytes, B, R, S, Cin, Cout, mma, st);
    if (rc != 0 || sum_rows == 0) return rc;
    return sh_spmm_p3(sum_rowptr, sum_col, sum_val, dpre, dp_sv, dp_sb, sum_out, dp_sv, dp_sb, sum_out_planes, nullptr, 0, 0, 0, -1, B, sum_rows, Cout, st);
}
// three-plane weight gradient: the real shape rule; reads the two images, writes its slabs, the rider as above
int sh_spiral_conv_bwd_wgt_p3_ok(int B, int R, int S, int Cin, int Cout) {
    return B > 0 && B % 16 == 0 && R > 0 && S > 0 && (Cin == 16 || (Cin > 0 && Cin % 32 == 0)) && Cout > 0 && Cout % 32 == 0;
}
size_t sh_spiral_conv_bwd_wgt_p3_workspace(int B, int R, int S, int Cin, int Cout) {
    return sh_spiral_conv_bwd_wgt_p3_ok(B, R, S, Cin, Cout) ? (size_t)8 * ((size_t)Cout * S * Cin + Cout) * 4 : 0;
}
size_t sh_p3_bytes(int rows, int B, int C);
int sh_spiral_conv_bwd_wgt_p3_presum(const void* dpre_planes, int dpre_zero_row, const void* x_planes, const int32_t* table, void* ws, size_t ws_bytes, const float* dpre,
                                     int64_t dp_sv, int64_t dp_sb, const int32_t* sum_rowptr, const int32_t* sum_col, const float* sum_val, float* sum_out,
                                     void* sum_out_planes, int sum_rows, int B, int R, int S, int Cin, int Cout, sh_stream_t st) {
    if (!sh_spiral_conv_bwd_wgt_p3_ok(B, R, S, Cin, Cout)) return SH_ERR_UNSUPPORTED;
    if (((long)R * (B / 16)) % 2 != 0 && !(dpre_zero_row >= 0 && dpre_zero_row < R)) return SH_ERR_UNSUPPORTED;
    int n_in = 0;
    for (long i = 0; i < (long)R * S; ++i) n_in = table[i] + 1 > n_in ? table[i] + 1 : n_in;
    touch_r(dpre_planes, sh_p3_bytes(R, B, Cout)); touch_r(x_planes, sh_p3_bytes(n_in, B, Cin));
    if (ws_bytes < sh_spiral_conv_bwd_wgt_p3_workspace(B, R, S, Cin, Cout)) return SH_ERR_WORKSPACE;
    touch_w(ws, ws_bytes);
    log("bwd_wgt_p3 R=%d Cin=%d Cout=%d", R, Cin, Cout);
    if (sum_rows == 0) return 0;
    return sh_spmm_p3(sum_rowptr, sum_col, sum_val, dpre, dp_sv, dp_sb, sum_out, dp_sv, dp_sb, sum_out_planes, nullptr, 0, 0, 0, -1, B, sum_rows, Cout, st);
}
int sh_spiral_conv_bwd_wgt_bf16(const void* dpre, int dd, int64_t dp_sv, int64_t dp_sb, const void* x, int xd, int64_t x_sv, int64_t x_sb,
                                const int32_t* table, void* ws, size_t ws_bytes, int B, int R, int S, int Cin, int Cout, sh_stream_t) {
    int n_in = 0;
    for (long i = 0; i < (long)R * S; ++i) n_in = table[i] + 1 > n_in ? table[i] + 1 : n_in;
    touch_r(dpre, span(dp_sv, dp_sb, R, B, Cout, dd == SH_DTYPE_BF16 ? 2 : 4)); touch_r(x, span(x_sv, x_sb, n_in, B, Cin, xd == SH_DTYPE_BF16 ? 2 : 4));
    if (ws_bytes < sh_spiral_conv_bwd_wgt_workspace_bf16(B, R, S, Cin, Cout)) return SH_ERR_WORKSPACE;
    touch_w(ws, ws_bytes);
    log("bwd_wgt_bf16 R=%d Cin=%d Cout=%d", R, Cin, Cout);
    return 0;
}
int sh_spiral_conv_bwd_wgt_thin_ok(int, int, int, int, int, int) { return 0; }       // the general kernels' call sequence is the one checked
int sh_spiral_conv_bwd_wgt_thin(const float*, int64_t, int64_t, const void*, int, int64_t, int64_t, const int32_t*, void*, size_t, const float*, void*,
                                int64_t, int64_t, void*, int, int, int, int, int, int, int, int, int, sh_stream_t) { return SH_ERR_UNSUPPORTED; }
static int bwd_data_common(const char* name, const void* dpre, size_t de, int64_t dp_sv, int64_t dp_sb, const int32_t* table_t, void* dx, size_t xe,
                           int64_t dx_sv, int64_t dx_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int B, int n_in, int S, int Cin, int Cout) {
    int rows = 0;
    for (long i = 0; i < (long)n_in * S; ++i) rows = table_t[i] + 1 > rows ? table_t[i] + 1 : rows;
    touch_r(dpre, span(dp_sv, dp_sb, rows, B, Cout, de));                 // includes the pre-summed extra rows the table points at
    touch_w(dx, span(dx_sv, dx_sb, n_in, B, Cin, xe));
    if (yprev) touch_r(yprev, span(yp_sv, yp_sb, n_in, B, Cin, xe));
    log(name, n_in, Cin, Cout);
    return 0;
}
int sh_spiral_conv_bwd_data(const float* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* table_t, const float* weight_t, float* dx, int64_t dx_sv,
                            int64_t dx_sb, const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin,
                            int Cout, int, sh_stream_t) {
    touch_r(weight_t, (size_t)Cin * S * Cout * 4);
    return bwd_data_common("bwd_data n_in=%d Cin=%d Cout=%d", dpre, 4, dp_sv, dp_sb, table_t, dx, 4, dx_sv, dx_sb, yprev, yp_sv, yp_sb, B, n_in, S, Cin, Cout);
}
int sh_spiral_conv_bwd_data_z(const float* dpre, int64_t dp_sv, int64_t dp_sb, int dpre_zero_row, const int32_t* table_t, const float* weight_t,
                              float* dx, int64_t dx_sv, int64_t dx_sb, const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row,
                              int B, int n_in, int S, int Cin, int Cout, int mma, sh_stream_t st) {
    // the zero row the no-source entries point at is one of dpre's own rows: reading it is covered by the same range
    if (dpre_zero_row >= 0) touch_r(dpre + (size_t)dpre_zero_row * dp_sv, (size_t)Cout * 4);
    return sh_spiral_conv_bwd_data(dpre, dp_sv, dp_sb, table_t, weight_t, dx, dx_sv, dx_sb, yprev, yp_sv, yp_sb, act_prev, zero_row, B, n_in, S, Cin,
                                   Cout, mma, st);
}
int sh_spiral_conv_bwd_data_bf16(const void* dpre, int dd, int64_t dp_sv, int64_t dp_sb, const int32_t* table_t, const void* wfrag_t, void* dx, int xd,
                                 int64_t dx_sv, int64_t dx_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in,
                                 int S, int Cin, int Cout, sh_stream_t) {
    touch_r(wfrag_t, sh_conv_wfrag_bytes(S, Cout, Cin));
    return bwd_data_common("bwd_data_bf16 n_in=%d Cin=%d Cout=%d", dpre, dd == SH_DTYPE_BF16 ? 2 : 4, dp_sv, dp_sb, table_t, dx, xd == SH_DTYPE_BF16 ? 2 : 4,
                           dx_sv, dx_sb, yprev, yp_sv, yp_sb, B, n_in, S, Cin, Cout);
}
int sh_spiral_conv_p3_grp_ok(int B, int S, int Cg, int Nout, int g_L) { return sh_spiral_conv_p3_ok(B, S, Cg, Nout) && Cg % 32 == 0 && g_L > 0 && g_L <= 64; }
int sh_spiral_conv_p3_grp_pays(int B, int n_groups) { return B >= 16 && n_groups > 0; }
int sh_spiral_conv_p3_grp_members(int B, int S, int Cg, int Nout) { return sh_spiral_conv_p3_grp_ok(B, S, Cg, Nout, 1) ? (Nout <= 32 ? 4 : 2) : 0; }
int sh_spiral_conv_p3_grp(const void* xp, const int32_t* g_rows, const uint32_t* g_pos, const int32_t* g_out, int n_groups, int g_L, const void* wfrag3, const float* bias,
                          float* y, int64_t y_sv, int64_t y_sb, void* yp, const float* yprev, int64_t yp_sv, int64_t yp_sb, const void* yprev_planes, int act, int zero_row,
                          int backward, int B, int R, int S, int Cg, int Nout, sh_stream_t) {
    int rows = 0;
    for (long i = 0; i < (long)n_groups * g_L; ++i) {
        rows = g_rows[i] + 1 > rows ? g_rows[i] + 1 : rows;
        for (int m = 0; m < 4; ++m) { const unsigned q = (g_pos[i] >> (8 * m)) & 0xFF; if (q != 0xFF && (int)q >= S) return SH_ERR_INVALID_ARG; }
    }
    for (long i = 0; i < (long)n_groups * 4; ++i) if (g_out[i] >= R) return SH_ERR_INVALID_ARG;
    touch_r(xp, sh_p3_bytes(rows, B, Cg)); touch_r(wfrag3, sh_conv_wfrag3_bytes(S, Cg, Nout));
    if (bias) touch_r(bias, (size_t)Nout * 4);
    if (y) touch_w(y, span(y_sv, y_sb, R, B, Nout, 4));
    if (yp) touch_w(yp, sh_p3_bytes(R, B, Nout));
    if (yprev_planes) touch_r(yprev_planes, sh_p3_bytes(R, B, Nout));
    else if (yprev) touch_r(yprev, span(yp_sv, yp_sb, R, B, Nout, 4));
    log("conv_p3_grp R=%d N=%d bwd=%d", R, Nout, backward);
    return 0;
}
int sh_spiral_conv_bf16_rag_ok(int B, int S, int Cg, int Nout, int rag_L) { return B > 0 && S <= 64 && Cg % 32 == 0 && Nout % 4 == 0 && rag_L > 0 && rag_L <= 64; }
int sh_spiral_conv_bwd_data_bf16_rag(const void* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* rag_rows, const int32_t* rag_pos, int rag_L, const void* wfrag_t,
                                     void* dx, int64_t dx_sv, int64_t dx_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B,
                                     int n_in, int S, int Cin, int Cout, sh_stream_t) {
    int rows = 0;
    for (long i = 0; i < (long)n_in * rag_L; ++i) { rows = rag_rows[i] + 1 > rows ? rag_rows[i] + 1 : rows; if (rag_pos[i] >= S) return SH_ERR_INVALID_ARG; }
    touch_r(dpre, span(dp_sv, dp_sb, rows, B, Cout, 2)); touch_r(wfrag_t, sh_conv_wfrag_bytes(S, Cout, Cin));
    touch_w(dx, span(dx_sv, dx_sb, n_in, B, Cin, 2));
    if (yprev) touch_r(yprev, span(yp_sv, yp_sb, n_in, B, Cin, 2));
    log("bwd_data_bf16_rag n_in=%d Cin=%d L=%d", n_in, Cin, rag_L);
    return 0;
}
int sh_weight_transpose_multi(int n, const float* const* w, float* const* wt, const int* S, const int* Ci, const int* Co, sh_stream_t) {
    for (int i = 0; i < n; ++i) { touch_r(w[i], (size_t)S[i] * Ci[i] * Co[i] * 4); touch_w(wt[i], (size_t)S[i] * Ci[i] * Co[i] * 4); }
    log("weight_transpose n=%d", n);
    return 0;
}
size_t sh_conv_wfrag_bytes(int S, int Cg, int Nout) {
    const int k = Cg == 3 ? 4 * S : S * Cg, nks = (k + 31) / 32, nt = (Nout + 15) / 16;
    const int ntt = nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : (nt + 7) / 8 * 8;
    return (size_t)nks * ntt * 1024;
}
int sh_conv_wfrag_prep_multi(int n, const float* const* w, void* const* wf, const int* S, const int* Ci, const int* Co, const int* tr, sh_stream_t) {
    for (int i = 0; i < n; ++i) {
        touch_r(w[i], (size_t)S[i] * Ci[i] * Co[i] * 4);
        touch_w(wf[i], tr[i] ? sh_conv_wfrag_bytes(S[i], Co[i], Ci[i]) : sh_conv_wfrag_bytes(S[i], Ci[i], Co[i]));
    }
    log("wfrag_prep n=%d", n);
    return 0;
}
static int reduce_common(const char* name, int n, const void* const* ws, float* const* dW, float* const* db, const int* S, const int* Ci, const int* Co) {
    for (int i = 0; i < n; ++i) { touch_r(ws[i], 16); touch_w(dW[i], (size_t)Co[i] * S[i] * Ci[i] * 4); if (db[i]) touch_w(db[i], (size_t)Co[i] * 4); }
    log(name, n);
    return 0;
}
int sh_spiral_conv_bwd_wgt_reduce_multi(int n, const void* const* ws, float* const* dW, float* const* db, const int* B, const int* R, const int* S,
                                        const int* Ci, const int* Co, sh_stream_t) { return reduce_common("reduce n=%d", n, ws, dW, db, S, Ci, Co); }
int sh_spiral_conv_bwd_wgt_reduce_multi_kinds(int n, const void* const* ws, float* const* dW, float* const* db, const int* B, const int* R, const int* S,
                                              const int* Ci, const int* Co, const int* kinds, sh_stream_t) {
    for (int i = 0; i < n; ++i) if (kinds[i] < 0 || kinds[i] > 2) return SH_ERR_INVALID_ARG;
    return reduce_common("reduce n=%d", n, ws, dW, db, S, Ci, Co);
}
int sh_spiral_conv_bwd_wgt_reduce_multi_bf16(int n, const void* const* ws, float* const* dW, float* const* db, const int* B, const int* R, const int* S,
                                             const int* Ci, const int* Co, sh_stream_t) { return reduce_common("reduce_bf16 n=%d", n, ws, dW, db, S, Ci, Co); }
// ---- three-plane form: image sizes are the real formula (include/sh_kernels.h), the kernels touch the image ranges
size_t sh_p3_bytes(int rows, int B, int C) {
    if (rows <= 0 || B <= 0 || B % 16 || !(C == 16 || (C > 0 && C % 32 == 0))) return 0;
    return (size_t)rows * (B / 16) * (C == 16 ? 1536 : (size_t)(C / 32) * 3072);
}
size_t sh_conv_wfrag3_bytes(int S, int Cg, int Nout) { return 3 * sh_conv_wfrag_bytes(S, Cg, Nout); }
int sh_spiral_conv_p3_kind(int B, int S, int Cg, int Nout) { return (B % 16 == 0 && (Cg == 16 || (Cg > 0 && Cg % 32 == 0)) && Nout % 4 == 0) ? 1 : 0; }
int sh_spiral_conv_p3_ok(int B, int S, int Cg, int Nout) { return B % 16 == 0 && (Cg == 16 || (Cg > 0 && Cg % 32 == 0)) && Nout % 4 == 0; }
int sh_to_p3(const float* x, int64_t x_sv, int64_t x_sb, void* planes, int B, int rows, int C, sh_stream_t) {
    touch_r(x, span(x_sv, x_sb, rows, B, C, 4)); touch_w(planes, sh_p3_bytes(rows, B, C));
    log("to_p3 rows=%d C=%d", rows, C);
    return sh_p3_bytes(rows, B, C) ? 0 : SH_ERR_UNSUPPORTED;
}
int sh_spiral_conv_fwd_p3(const void* xp, const int32_t* table, const void* wfrag3, const float* bias, float* y, int64_t y_sv, int64_t y_sb, void* yp,
                          int B, int R, int S, int Cin, int Cout, int act, int zero_row, sh_stream_t) {
    int n_in = 0;
    for (long i = 0; i < (long)R * S; ++i) n_in = table[i] + 1 > n_in ? table[i] + 1 : n_in;
    touch_r(xp, sh_p3_bytes(n_in, B, Cin)); touch_r(wfrag3, sh_conv_wfrag3_bytes(S, Cin, Cout)); touch_r(bias, bias ? Cout * 4 : 0);
    if (y) touch_w(y, span(y_sv, y_sb, R, B, Cout, 4));
    if (yp) touch_w(yp, sh_p3_bytes(R, B, Cout));
    log("conv_fwd_p3 R=%d Cin=%d Cout=%d", R, Cin, Cout);
    return 0;
}
int sh_spiral_conv_bwd_data_p3(const void* dprep, int, const float* dpre_f32, int64_t dp_sv, int64_t dp_sb, int n_img, const int32_t* table_t, const void* wfrag3_t, float* dx, int64_t dx_sv, int64_t dx_sb, void* dxp,
                               const float* yprev, int64_t yp_sv, int64_t yp_sb, const void* yprev_planes, int act_prev, int zero_row, int B, int n_in,
                               int S, int Cin, int Cout, sh_stream_t) {
    int rows = 0;
    for (long i = 0; i < (long)n_in * S; ++i) rows = table_t[i] + 1 > rows ? table_t[i] + 1 : rows;
    touch_r(dprep, sh_p3_bytes(dpre_f32 ? (n_img < rows ? n_img : rows) : rows, B, Cout)); touch_r(wfrag3_t, sh_conv_wfrag3_bytes(S, Cout, Cin));
    if (dpre_f32) touch_r(dpre_f32, span(dp_sv, dp_sb, rows, B, Cout, 4));
    if (dx) touch_w(dx, span(dx_sv, dx_sb, n_in, B, Cin, 4));
    if (dxp) touch_w(dxp, sh_p3_bytes(n_in, B, Cin));
    if (yprev_planes) touch_r(yprev_planes, sh_p3_bytes(n_in, B, Cin));
    else if (yprev) touch_r(yprev, span(yp_sv, yp_sb, n_in, B, Cin, 4));
    log("bwd_data_p3 n_in=%d Cin=%d Cout=%d", n_in, Cin, Cout);
    return 0;
}
int sh_spiral_conv_p3_rag_ok(int B, int S, int Cg, int Nout, int rag_L) { return sh_spiral_conv_p3_ok(B, S, Cg, Nout) && Cg % 32 == 0 && rag_L > 0 && rag_L <= 64; }
int sh_spiral_conv_bwd_data_p3_rag(const void* dprep, const int32_t* rag_rows, const int32_t* rag_pos, int rag_L, const void* wfrag3_t, float* dx, int64_t dx_sv,
                                   int64_t dx_sb, void* dxp, const float* yprev, int64_t yp_sv, int64_t yp_sb, const void* yprev_planes, int act_prev, int zero_row,
                                   int B, int n_in, int S, int Cin, int Cout, sh_stream_t) {
    int rows = 0;
    for (long i = 0; i < (long)n_in * rag_L; ++i) { rows = rag_rows[i] + 1 > rows ? rag_rows[i] + 1 : rows; if (rag_pos[i] >= S) return SH_ERR_INVALID_ARG; }
    touch_r(dprep, sh_p3_bytes(rows, B, Cout)); touch_r(wfrag3_t, sh_conv_wfrag3_bytes(S, Cout, Cin));
    if (dx) touch_w(dx, span(dx_sv, dx_sb, n_in, B, Cin, 4));
    if (dxp) touch_w(dxp, sh_p3_bytes(n_in, B, Cin));
    if (yprev_planes) touch_r(yprev_planes, sh_p3_bytes(n_in, B, Cin));
    else if (yprev) touch_r(yprev, span(yp_sv, yp_sb, n_in, B, Cin, 4));
    log("bwd_data_p3_rag n_in=%d Cin=%d L=%d", n_in, Cin, rag_L);
    return 0;
}
}  // extern "C"
bool sh_mma_mode_valid(int mode) { return mode >= 0 && mode <= 2; }
