// Sanitizer driver for the host-side stack sequencers (semantichuman_amd/csrc/stack_exec.hip compiled host-only with
// -fsanitize=address,undefined, the kernel entry points replaced by stubs.cpp).  It builds a four-step stack with every feature
// the sequencers handle - a conv with list pre-sums of both levels, a folded (extend) up-sampling, a conv behind it, a
// 3-channel last conv with a batch-major output - allocates every buffer EXACTLY as include/sh_kernels.h sizes it (host
// memory standing in for device memory) and runs forward + backward on the fp32 path, the bf16 path and the fp32 path in its
// three-plane form (batch 16: plane images behind the 16-channel tensors).  A pointer the sequencer
// derives past a buffer, a mis-sized pointer table or an uninitialised step field is an ASan / UBSan report = non-zero exit.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/sh_kernels.h"

extern std::vector<std::string> g_calls;
void sh_set_error(const char* fmt, ...) { (void)fmt; }
int sh_env_int(const char*, int dflt, int, int) { return dflt; }      // the sequencers' switches at their defaults

struct Csr { std::vector<int32_t> rp, col; std::vector<float> val; sh_csr_ref ref() const { return {rp.data(), col.data(), val.data()}; } };
static Csr make_csr(int rows, int cols, int per_row) {
    Csr m; m.rp.push_back(0);
    for (int r = 0; r < rows; ++r) { for (int e = 0; e < per_row; ++e) { m.col.push_back((r * 7 + e * 3) % cols); m.val.push_back(1.f); } m.rp.push_back((int)m.col.size()); }
    return m;
}
static std::vector<int32_t> make_table(int rows, int S, int limit) {
    std::vector<int32_t> t((size_t)rows * S);
    for (size_t i = 0; i < t.size(); ++i) t[i] = (int32_t)((i * 5 + 1) % limit);
    t[t.size() - 1] = limit - 1;                                     // the largest index is referenced: the stubs size operands from it
    return t;
}
template <class T> static T* alloc(size_t n) { return static_cast<T*>(calloc(n ? n : 1, sizeof(T))); }      // exact size: ASan guards the end

int main() {
    const int S = 3;
    // step 0: conv 8 -> 16, 6 rows out of 7 in; pre-sums 1 + 2.  step 1: extend, 4 blended rows behind the 6.  step 2: conv 16 -> 8 over
    // the 10 rows of Z, 9 rows out, one pre-sum.  step 3: conv 8 -> 3, 9 -> 9 rows, two pre-sums, batch-major output.
    struct L { int R, n_in, cin, cout, n1, n2; } L0{6, 7, 8, 16, 1, 2}, L2{9, 10, 16, 8, 0, 1}, L3{9, 9, 8, 3, 0, 2};
    const int n_b = 4;
    auto t0 = make_table(L0.R, S, L0.n_in), t2 = make_table(L2.R, S, L2.n_in), t3 = make_table(L3.R, S, L3.n_in);
    auto tt0 = make_table(L0.n_in, S, L0.R + L0.n1 + L0.n2), tt2 = make_table(L2.n_in, S, L2.R + L2.n1 + L2.n2), tt3 = make_table(L3.n_in, S, L3.R + L3.n1 + L3.n2);
    Csr s1_0 = make_csr(L0.n1, L0.R, 4), s2_0 = make_csr(L0.n2, L0.R + L0.n1, 2), s2_2 = make_csr(L2.n2, L2.R, 2), s2_3 = make_csr(L3.n2, L3.R, 3);
    Csr ub = make_csr(n_b, L0.R, 3), mt = make_csr(L0.R, L0.R + n_b, 3);
    sh_stack_step st[4];
    memset(st, 0, sizeof st);
    auto conv = [&](sh_stack_step& s, const L& l, int param, const int32_t* t, const int32_t* tt) {
        s.kind = 0; s.param = param; s.table = t; s.table_t = tt; s.R = l.R; s.S = S; s.n_in = l.n_in; s.cin = l.cin; s.cout = l.cout;
        s.act = SH_ACT_ELU; s.zero_row = l.R - 1; s.n1 = l.n1; s.n2 = l.n2;
    };
    conv(st[0], L0, 0, t0.data(), tt0.data()); st[0].sum1 = s1_0.ref(); st[0].sum2 = s2_0.ref();
    st[1].kind = 1; st[1].param = -1; st[1].m = ub.ref(); st[1].mt = mt.ref(); st[1].m_rows = n_b; st[1].m_cols = L0.R; st[1].extend = 1;
    conv(st[2], L2, 1, t2.data(), tt2.data()); st[2].sum2 = s2_2.ref();
    conv(st[3], L3, 2, t3.data(), tt3.data()); st[3].sum2 = s2_3.ref(); st[3].act = SH_ACT_IDENTITY;
    const L* Ls[3] = {&L0, &L2, &L3};
    int rc = 0;
    for (int pass = 0; pass < 3 && rc == 0; ++pass) {
        const int dtype = pass == 1 ? 1 : 0, B = pass == 2 ? 16 : 4, mma = pass == 2 ? SH_MMA_PLANES3 : SH_MMA_EXACT;
        const size_t e = dtype == SH_DTYPE_BF16 ? 2 : 4;
        // ---- parameters
        float* W[3]; float* bias[3]; float* dW[3]; float* db[3];
        for (int p = 0; p < 3; ++p) {
            const size_t n = (size_t)Ls[p]->cout * S * Ls[p]->cin;
            W[p] = alloc<float>(n); bias[p] = alloc<float>(Ls[p]->cout); dW[p] = alloc<float>(n); db[p] = alloc<float>(Ls[p]->cout);
        }
        // ---- forward buffers (vertex-major inner tensors, batch-major fp32 output)
        char* x = alloc<char>((size_t)L0.n_in * B * L0.cin * e);
        // SH_ASAN_NEGATIVE: leave out the room for the appended rows - the harness must catch the sequencer's write behind the buffer
        char* o0 = alloc<char>((size_t)(L0.R + (getenv("SH_ASAN_NEGATIVE") ? 0 : n_b)) * B * L0.cout * e);      // conv 0's rows + the appended blended rows
        char* o2 = alloc<char>((size_t)L2.R * B * L2.cout * e);
        float* out = alloc<float>((size_t)B * L3.R * 3);
        void* outs[4] = {o0, o0, o2, out};
        void* wf[4] = {0, 0, 0, 0}, *wft[4] = {0, 0, 0, 0};
        char* img0 = nullptr; char* imgg = nullptr; const void* wf3[4] = {0, 0, 0, 0}; const void* wf3t[4] = {0, 0, 0, 0};
        if (dtype == SH_DTYPE_BF16) {
            const int ci[4] = {0, -1, 2, 3};
            for (int i = 0; i < 4; ++i)
                if (ci[i] >= 0) {
                    const L& l = *Ls[i == 0 ? 0 : i - 1];
                    wf[i] = alloc<char>(sh_conv_wfrag_bytes(S, l.cin, l.cout)); wft[i] = alloc<char>(sh_conv_wfrag_bytes(S, l.cout, l.cin));
                }
            rc = sh_stack_forward_bf16(4, st, x, SH_DTYPE_BF16, 0, L0.n_in, L0.cin, B, W, bias, wf, 0, outs, SH_DTYPE_F32, 1, nullptr);
        } else {
            if (pass == 2) {      // images of the 16-channel buffers: conv 0's output with the appended rows (gathered by conv 2), conv 0's dpre
                img0 = alloc<char>(sh_p3_bytes(L0.R + n_b, B, L0.cout)); imgg = alloc<char>(sh_p3_bytes(L0.R + L0.n1 + L0.n2, B, L0.cout));
                wf3[2] = alloc<char>(sh_conv_wfrag3_bytes(S, L2.cin, L2.cout)); wf3t[0] = alloc<char>(sh_conv_wfrag3_bytes(S, L0.cout, L0.cin));
            }
            void* planes[4] = {img0, img0, nullptr, nullptr};
            rc = sh_stack_forward(4, st, reinterpret_cast<const float*>(x), 0, L0.n_in, L0.cin, B, W, bias, reinterpret_cast<float* const*>(outs), 1, mma,
                                  planes, wf3, 1, nullptr);
        }
        if (rc) { printf("forward rc=%d\n", rc); break; }
        // ---- backward buffers
        float* g = alloc<float>((size_t)B * L3.R * 3);
        char* gx = alloc<char>((size_t)L0.n_in * B * L0.cin * e);
        char* g1 = alloc<char>((size_t)(L0.R + L0.n1 + L0.n2) * B * L0.cout * e);        // input of the extend step = conv 0's dpre incl. its pre-sum rows
        char* g2 = alloc<char>((size_t)L2.n_in * B * L2.cin * e);
        char* g3 = alloc<char>((size_t)(L2.R + L2.n1 + L2.n2) * B * L3.cin * e);
        float* dpre_last = alloc<float>((size_t)(L3.R + L3.n1 + L3.n2) * B * 3);
        void* gin[4] = {gx, g1, g2, g3};
        void* ws[4] = {0, 0, 0, 0}; size_t wsb[4] = {0, 0, 0, 0}; float* wt[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i)
            if (st[i].kind == 0) {
                wsb[i] = dtype == SH_DTYPE_BF16 ? sh_spiral_conv_bwd_wgt_workspace_bf16(B, st[i].R, S, st[i].cin, st[i].cout)
                                                : sh_spiral_conv_bwd_wgt_workspace(B, st[i].R, S, st[i].cin, st[i].cout);
                ws[i] = alloc<char>(wsb[i]);
                wt[i] = alloc<float>((size_t)st[i].cin * S * st[i].cout);
            }
        void* gpl[4] = {nullptr, imgg, nullptr, nullptr};
        const void* inpl[4] = {nullptr, nullptr, img0, nullptr};       // image of the input of step 2 = of the buffer steps 0 and 1 wrote
        if (dtype == SH_DTYPE_BF16)
            rc = sh_stack_backward_bf16(4, st, x, SH_DTYPE_BF16, 0, L0.n_in, L0.cin, B, outs, g, SH_DTYPE_F32, 1, W, gin, SH_DTYPE_BF16, dpre_last, wft, 0, ws,
                                        wsb, dW, db, 1, nullptr);
        else
            rc = sh_stack_backward(4, st, reinterpret_cast<const float*>(x), 0, L0.n_in, L0.cin, B, reinterpret_cast<const float* const*>(outs), g, 1, W,
                                   reinterpret_cast<float* const*>(gin), dpre_last, wt, ws, wsb, dW, db, 1, mma, gpl, nullptr, wf3t, inpl, 1, nullptr);
        if (rc) { printf("backward rc=%d\n", rc); break; }
        for (int p = 0; p < 3; ++p) { free(W[p]); free(bias[p]); free(dW[p]); free(db[p]); }
        free(x); free(o0); free(o2); free(out); free(g); free(gx); free(g1); free(g2); free(g3); free(dpre_last);
        for (int i = 0; i < 4; ++i) { free(ws[i]); free(wt[i]); free(wf[i]); free(wft[i]); free(const_cast<void*>(wf3[i])); free(const_cast<void*>(wf3t[i])); }
        free(img0); free(imgg);
    }
    for (const auto& c : g_calls) printf("%s\n", c.c_str());
    printf(rc == 0 ? "SEQUENCERS OK %zu calls\n" : "SEQUENCERS FAILED\n", g_calls.size());
    return rc;
}
