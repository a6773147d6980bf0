// Whole-stack execution (sh_stack_forward / sh_stack_backward): host-side sequencing of the conv / spmm /
// activation-backward / weight-gradient entry points for one encoder or decoder stack.  It mirrors, launch for
// launch, what semantichuman_amd/stack.py does call by call from Python (reference: the layer loops of
// models.py:119-128 and :146-153 plus autograd's backward chain) - the point is host time: one call instead of
// ~25 us of interpreter and ctypes work per launch.  No kernels here.
#include "sh_common.h"

namespace {

struct Lay { long sv, sb; };                                   // element strides of (row, batch)
inline Lay lay(int layout, int rows, int B, int C) {
    return layout == 0 ? Lay{(long)B * C, (long)C} : Lay{(long)C, (long)rows * C};
}
inline int out_rows(const sh_stack_step& s) { return s.kind == 0 ? s.R : s.m_rows; }
inline bool is_last_step(int i, int n) { return i == n - 1; }
inline int in_rows(const sh_stack_step& s) { return s.kind == 0 ? s.n_in : s.m_cols; }

// Every tensor a step gathers from is addressed with 32-bit offsets: element offsets in the fp32 conv kernels, row offsets
// pre-multiplied by the row stride in 16-byte units in the plane convs (csrc/p3_conv.hip: tv[]) and the streaming weight
// gradients (Tl[]).  2^32 ELEMENTS per tensor is the tightest of the three (16 GiB of fp32, 24 GiB of planes against the 64 GiB
// the pre-multiplied forms reach), so that is what is refused here - with a message, not with wrapped gather addresses.
int check_tensor_sizes(int n, const sh_stack_step* st, int rows0, int c0, int B, const char* what) {
    long rows = rows0;
    int c = c0;
    for (int i = 0; i <= n; ++i) {
        long extra = 0;                                       // pre-summed rows behind a conv step's gradient rows
        if (i > 0 && st[i - 1].kind == 0) extra = (long)st[i - 1].n1 + st[i - 1].n2;
        SH_REQUIRE((rows + extra) * (long)B * c < (1L << 32), SH_ERR_UNSUPPORTED,
                   "%s: the tensor %s step %d has %ld rows x %d batch entries x %d channels >= 2^32 elements - split the batch", what,
                   i < n ? "entering" : "leaving", i < n ? i : n - 1, rows + extra, B, c);
        if (i == n) break;
        rows = st[i].kind == 0 ? st[i].R : (st[i].extend ? (long)st[i].m_cols + st[i].m_rows : st[i].m_rows);
        if (st[i].kind == 0) c = st[i].cout;
    }
    return SH_OK;
}

int check_steps(int n, const sh_stack_step* st, int c0, const char* what) {
    SH_REQUIRE(n > 0 && st, SH_ERR_INVALID_ARG, "%s: no steps", what);
    int c = c0;
    for (int i = 0; i < n; ++i) {
        SH_REQUIRE(st[i].kind == 0 || st[i].kind == 1, SH_ERR_INVALID_ARG, "%s: step %d has kind %d", what, i, st[i].kind);
        if (st[i].kind == 0) {
            SH_REQUIRE(st[i].cin == c, SH_ERR_INVALID_ARG, "%s: step %d takes %d channels, its input has %d", what, i, st[i].cin, c);
            SH_REQUIRE(st[i].table && st[i].param >= 0, SH_ERR_INVALID_ARG, "%s: step %d incomplete", what, i);
            c = st[i].cout;
        } else {
            SH_REQUIRE(st[i].m.rowptr && st[i].m.col && st[i].m.val, SH_ERR_INVALID_ARG, "%s: step %d has no matrix", what, i);
        }
    }
    return SH_OK;
}

}  // namespace

namespace {

// ---- three-plane form (SH_MMA_PLANES3): which steps run the ..._p3 kernels.  A conv step's forward takes the plane image of
// its input when the caller supplied the buffers (planes of the producing step's output buffer, three-plane weight fragments)
// and the kernels take the shape; its backward-data pass likewise with the image of its pre-activation gradient.
struct P3Ctx {
    int n, B, mode;
    const sh_stack_step* st;
    void* const* planes;              // forward: per step, image buffer of outs[i] (NULL: none)
    const void* const* wfrag3;        // per conv step: forward operand (sh_stack_forward) / backward-data operand (sh_stack_backward)
};
// does conv step j's FORWARD pass take the plane kernel for this batch?  The kernels' shape test, and one measured rule: a layer with
// <= 16 output channels (one channel tile: 6 MFMAs per gathered 3-KiB fragment) loses to the exact staged kernel once the batch is
// large - per 64 meshes, dec3 (6891 rows, K = 320 -> 16): plane 50.6 / 58.1 / 63.4 / 65.4 / 67.9 us at batch 64 / 128 / 256 / 512 / 1024,
// exact 68.8 / 66.6 / 63.4 / 60.4 / 56.5 (profiles/r05_decode_batch_sweep.txt).  SH_P3_N16_MAXB: the largest batch at which such a layer
// still runs the plane kernel.
inline bool p3_step_shape_ok(int B, const sh_stack_step& s) {
    static const int n16_maxb = sh_env_int("SH_P3_N16_MAXB", 256, 0, 1 << 30);
    if (s.cout <= 16 && B > n16_maxb) return false;
    return sh_spiral_conv_p3_ok(B, s.S, s.cin, s.cout) != 0;
}
inline bool p3_fwd(const P3Ctx& c, int i, bool in_vm) {
    const sh_stack_step& s = c.st[i];
    return c.mode == SH_MMA_PLANES3 && s.kind == 0 && i > 0 && in_vm && c.planes && c.planes[i - 1] && c.wfrag3 && c.wfrag3[i] &&
           p3_step_shape_ok(c.B, s);
}
// Training on the plane images (round 6, keep_fp32 == 2): does the BACKWARD pass of conv step j leave the fp32 rows of the step's
// gathered input (acts[j - 1]) unread?  Yes when its weight gradient runs on the images (wgrad_p3.hip) and the activation
// derivative its backward-data pass applies is evaluated from the image.  This is the part of sh_stack_backward's decisions that is
// known from the steps alone; sh_stack_forward drops fp32 rows by it and sh_stack_backward, told so (acts_fp32 == 2), refuses to
// run such a step in any other form.
inline bool bwd_plane_static(const sh_stack_step* st, int j, int B) {
    static const int on = sh_env_int("SH_P3_BWD", 1, 0, 1) && sh_env_int("SH_P3_WGRAD", 1, 0, 1) && sh_env_int("SH_P3_YPREV_IMG", 1, 0, 1) &&
                          sh_env_int("SH_P3_DROP_FP32", 1, 0, 1);
    const sh_stack_step& s = st[j];
    if (!on || j == 0 || s.kind != 0 || !s.table_t) return false;
    if (!sh_spiral_conv_p3_ok(B, s.S, s.cout, s.cin)) return false;                  // its backward-data pass gathers the image of dpre
    if (s.R == s.n_in && sh_spiral_conv_bwd_wgt_thin_ok(B, s.n_in, s.S, s.cin, s.cout, SH_DTYPE_F32)) return false;
    if (!sh_spiral_conv_bwd_wgt_p3_ok(B, s.R, s.S, s.cin, s.cout)) return false;
    if (((long)s.R * (B / 16)) % 2 != 0 && s.zero_row < 0) return false;
    return sh_p3_bytes(1, B, s.cin) != 0;
}
// ... and does it leave the fp32 rows of its pre-activation gradient (what the step behind it writes to gin[j + 1]) unread?  When,
// besides, no pre-sum launch or rider reads them: ragged source lists, or a table without multiplicities.
inline bool bwd_grad_plane_static(const sh_stack_step* st, int j, int B) {
    static const int rag_on = sh_env_int("SH_P3_RAGGED", 1, 0, 1);
    const sh_stack_step& s = st[j];
    if (!bwd_plane_static(st, j, B)) return false;
    const bool rag = rag_on && s.rag_rows && s.rag_pos && sh_spiral_conv_p3_rag_ok(B, s.S, s.cout, s.cin, s.rag_L);
    return rag || (s.n1 == 0 && s.n2 == 0);
}
// the conv step that gathers the buffer step i writes (through a folded up-sampling that appends to it), or -1
inline int consumer_conv(const P3Ctx& c, int i) {
    int j = i + 1;
    if (j < c.n && c.st[j].kind == 1 && c.st[j].extend) ++j;
    return (j < c.n && c.st[j].kind == 0) ? j : -1;
}

}  // namespace

extern "C" {

int sh_stack_forward(int n_steps, const sh_stack_step* steps, const float* x, int x_layout, int rows0, int c0, int B,
                     const float* const* weights, const float* const* biases, float* const* outs, int out_layout, int mma_mode,
                     void* const* planes, const void* const* wfrag3, int keep_fp32, sh_stream_t stream) {
    int rc = check_steps(n_steps, steps, c0, "sh_stack_forward");
    if (rc != SH_OK) return rc;
    if (B > 0 && (rc = check_tensor_sizes(n_steps, steps, rows0, c0, B, "sh_stack_forward")) != SH_OK) return rc;
    SH_REQUIRE(x && weights && outs && B > 0, SH_ERR_INVALID_ARG, "sh_stack_forward: null pointer or empty batch");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_stack_forward: unknown mma_mode %d", mma_mode);
    SH_REQUIRE(keep_fp32 >= 0 && keep_fp32 <= 2, SH_ERR_INVALID_ARG, "sh_stack_forward: keep_fp32 = %d (0, 1 or 2)", keep_fp32);
    const P3Ctx pc{n_steps, B, mma_mode, steps, planes, wfrag3};
    const float* cur = x;
    Lay cl = lay(x_layout, rows0, B, c0);
    int c = c0;
    for (int i = 0; i < n_steps; ++i) {
        const sh_stack_step& s = steps[i];
        const int co = s.kind == 0 ? s.cout : c;
        const Lay ol = lay(i == n_steps - 1 ? out_layout : 0, out_rows(s), B, co);
        SH_REQUIRE(outs[i], SH_ERR_INVALID_ARG, "sh_stack_forward: no output buffer for step %d", i);
        // does a three-plane conv gather the buffer this step writes?  then its image is written with it
        const int cons = i < n_steps - 1 ? consumer_conv(pc, i) : -1;
        void* img = (cons >= 0 && planes && planes[i] && mma_mode == SH_MMA_PLANES3 && wfrag3 && wfrag3[cons] &&
                     p3_step_shape_ok(B, steps[cons]) && sh_p3_bytes(1, B, co)) ? planes[i] : nullptr;
        // forward only (keep_fp32 == 0: no backward pass will read this pass's activations): rows that are gathered through
        // their plane image alone are written as the image alone - a re-sampling step in front of a plane conv, a plane conv
        // directly in front of another (6 instead of 10 bytes per element; BASELINE config 5's decode)
        // training on the images (keep_fp32 == 2): the same rows, when the backward pass of the consumer leaves them unread too
        const bool img_only = (keep_fp32 == 0 || (keep_fp32 == 2 && cons >= 0 && bwd_plane_static(steps, cons, B))) && img && cons >= 0 &&
                              p3_fwd(pc, cons, true);
        if (s.kind == 0) {
            if (p3_fwd(pc, i, cl.sb == c && cl.sv == (long)B * c)) {
                const bool direct = img_only && cons == i + 1;      // (through a folded up-sampling the fp32 rows feed the blend)
                // grouped lists (round 6): output rows with overlapping spirals share one list of the union - every row gathered once
                static const int grp_on = sh_env_int("SH_P3_GROUPED", 1, 0, 1);
                if (grp_on && s.fg_rows && s.fg_pos && s.fg_out && s.fg_n > 0 && sh_spiral_conv_p3_grp_ok(B, s.S, s.cin, s.cout, s.fg_L) &&
                    sh_spiral_conv_p3_grp_pays(B, s.fg_n))
                    rc = sh_spiral_conv_p3_grp(planes[i - 1], s.fg_rows, s.fg_pos, s.fg_out, s.fg_n, s.fg_L, wfrag3[i], biases ? biases[s.param] : nullptr,
                                               direct ? nullptr : outs[i], ol.sv, ol.sb, img, nullptr, 0, 0, nullptr, s.act, s.zero_row, 0, B, s.R, s.S,
                                               s.cin, s.cout, stream);
                else
                rc = sh_spiral_conv_fwd_p3(planes[i - 1], s.table, wfrag3[i], biases ? biases[s.param] : nullptr, direct ? nullptr : outs[i], ol.sv,
                                           ol.sb, img, B, s.R, s.S, s.cin, s.cout, s.act, s.zero_row, stream);
            } else {
                rc = sh_spiral_conv_fwd_img(cur, cl.sv, cl.sb, s.table, weights[s.param], biases ? biases[s.param] : nullptr, outs[i],
                                            ol.sv, ol.sb, img, B, s.R, s.S, s.cin, s.cout, s.act, s.zero_row, mma_mode, stream);
            }
        } else if (s.extend) {
            SH_REQUIRE(i > 0 && !is_last_step(i, n_steps) && outs[i] == outs[i - 1] && cl.sb == c, SH_ERR_INVALID_ARG,
                       "sh_stack_forward: step %d appends to its input, which must be the vertex-major output buffer of step %d", i, i - 1);
            SH_REQUIRE(!planes || planes[i] == planes[i - 1], SH_ERR_INVALID_ARG, "sh_stack_forward: step %d appends to its input: same image buffer", i);
            float* dst = img_only ? nullptr : outs[i] + (long)s.m_cols * cl.sv;
            rc = sh_spmm_p3(s.m.rowptr, s.m.col, s.m.val, cur, cl.sv, cl.sb, dst, cl.sv, cl.sb,
                            img ? static_cast<char*>(img) + sh_p3_bytes(s.m_cols, B, c) : nullptr, nullptr, 0, 0, 0, -1, B, s.m_rows, c, stream);
        } else {
            rc = sh_spmm_p3(s.m.rowptr, s.m.col, s.m.val, cur, cl.sv, cl.sb, img_only ? nullptr : outs[i], ol.sv, ol.sb, img, nullptr, 0, 0, 0, -1, B,
                            s.m_rows, c, stream);
        }
        if (rc != SH_OK) return rc;
        cur = outs[i]; cl = ol; c = co;
    }
    return SH_OK;
}

int sh_stack_backward(int n_steps, const sh_stack_step* steps, const float* x, int x_layout, int rows0, int c0, int B,
                      const float* const* acts, const float* g, int out_layout, const float* const* weights,
                      float* const* gin, float* dpre_last, float* const* weight_t, void* const* workspace,
                      const size_t* workspace_bytes, float* const* dW, float* const* dbias, int need_x_grad, int mma_mode,
                      void* const* gin_planes, void* dpre_last_planes, const void* const* wfrag3_t, const void* const* in_planes,
                      int acts_fp32, sh_stream_t stream) {
    int rc = check_steps(n_steps, steps, c0, "sh_stack_backward");
    if (rc != SH_OK) return rc;
    if (B > 0 && (rc = check_tensor_sizes(n_steps, steps, rows0, c0, B, "sh_stack_backward")) != SH_OK) return rc;
    SH_REQUIRE(x && acts && g && weights && gin && dW && B > 0, SH_ERR_INVALID_ARG, "sh_stack_backward: null pointer or empty batch");
    SH_REQUIRE(n_steps <= 64, SH_ERR_UNSUPPORTED, "sh_stack_backward: more than 64 steps");
    SH_REQUIRE(sh_mma_mode_valid(mma_mode), SH_ERR_INVALID_ARG, "sh_stack_backward: unknown mma_mode %d", mma_mode);
    SH_REQUIRE(acts_fp32 == 1 || (acts_fp32 == 2 && mma_mode == SH_MMA_PLANES3 && in_planes), SH_ERR_INVALID_ARG,
               "sh_stack_backward: acts_fp32 = %d (1: every activation has its fp32 rows; 2, three-plane form with in_planes: the forward pass "
               "ran with keep_fp32 == 2)", acts_fp32);
    const int last = n_steps - 1;
    int cin_of[64];                                            // channels entering step i
    {
        int c = c0;
        for (int i = 0; i < n_steps; ++i) { cin_of[i] = c; if (steps[i].kind == 0) c = steps[i].cout; }
    }
    // three-plane form: conv step i's backward-data pass gathers the IMAGE of its pre-activation gradient when the caller gave
    // a buffer for it (gin_planes[i + 1], or dpre_last_planes for the last step) and the fragments of the transposed weight
    // SH_P3_BWD: 0 = the backward pass keeps the SPLIT3 kernels everywhere, 1 = three-plane backward-data wherever it can run
    static const int p3_bwd_on = sh_env_int("SH_P3_BWD", 1, 0, 1);
    auto bwd_p3 = [&](int i) -> void* {
        const sh_stack_step& s = steps[i];
        if (!p3_bwd_on) return nullptr;
        if (mma_mode != SH_MMA_PLANES3 || s.kind != 0 || !(i > 0 || need_x_grad) || !s.table_t || !wfrag3_t || !wfrag3_t[i]) return nullptr;
        if (!sh_spiral_conv_p3_ok(B, s.S, s.cout, s.cin)) return nullptr;
        return i == last ? dpre_last_planes : (gin_planes ? gin_planes[i + 1] : nullptr);
    };
    // all weight transposes of the stack: workgroups of the launch that opens the pass (the last step's activation backward),
    // or a launch of their own when the pass opens with a re-sampling step
    const float* tr_w[32]; float* tr_wt[32]; int tr_S[32], tr_Ci[32], tr_Co[32];
    int n_tr = 0;
    for (int i = 0; i < n_steps; ++i) {
        if (steps[i].kind != 0 || !(i > 0 || need_x_grad)) continue;
        if (bwd_p3(i)) {                                       // reads fragments, not the transposed weight
            const sh_stack_step& s = steps[i];
            const bool thin = s.R == s.n_in && sh_spiral_conv_bwd_wgt_thin_ok(B, s.n_in, s.S, s.cin, s.cout, SH_DTYPE_F32);
            if (!thin) continue;
        }
        SH_REQUIRE(weight_t && weight_t[i], SH_ERR_INVALID_ARG, "sh_stack_backward: no weight_t buffer for step %d", i);
        SH_REQUIRE(n_tr < 32, SH_ERR_UNSUPPORTED, "sh_stack_backward: more than 32 conv steps");
        tr_w[n_tr] = weights[steps[i].param]; tr_wt[n_tr] = weight_t[i]; tr_S[n_tr] = steps[i].S; tr_Ci[n_tr] = steps[i].cin; tr_Co[n_tr] = steps[i].cout;
        ++n_tr;
    }
    static const int tr_ride = sh_env_int("SH_TR_RIDE", 1, 0, 1);
    if (n_tr && !(tr_ride && steps[last].kind == 0)) {
        rc = sh_weight_transpose_multi(n_tr, tr_w, tr_wt, tr_S, tr_Ci, tr_Co, stream);
        if (rc != SH_OK) return rc;
        n_tr = 0;
    }
    // gradient entering the last step
    const float* cur; Lay cl;
    void* cur_img = nullptr;                                   // image buffer of `cur` when its consumer gathers planes
    bool cur_img_done = false;                                 // rows [0, R) of it already written by the producer
    {
        const sh_stack_step& s = steps[last];
        if (s.kind == 0) {
            SH_REQUIRE(dpre_last, SH_ERR_INVALID_ARG, "sh_stack_backward: no dpre_last buffer");
            const Lay ol = lay(out_layout, s.R, B, s.cout), dl = lay(0, 0, B, s.cout);
            cur_img = bwd_p3(last);
            // the image of dpre rides in the launch when that is the plain element-wise form (16-byte quads, no tile turning)
            const bool img_in = cur_img && s.cout % 4 == 0 && s.cout > 8 && ((ol.sv | ol.sb) % 4 == 0);
            rc = sh_act_backward_tr_img(g, ol.sv, ol.sb, acts[last], ol.sv, ol.sb, dpre_last, dl.sv, dl.sb, img_in ? cur_img : nullptr, B, s.R,
                                        s.cout, s.act, s.zero_row, n_tr, tr_w, tr_wt, tr_S, tr_Ci, tr_Co, stream);
            if (rc != SH_OK) return rc;
            cur = dpre_last; cl = dl;
            cur_img_done = img_in;
        } else {
            cur = g; cl = lay(out_layout, s.m_rows, B, cin_of[last]);
        }
    }
    const void* job_ws[64]; float* job_dW[64]; float* job_db[64]; int jB[64], jR[64], jS[64], jCi[64], jCo[64], jK[64];
    int njobs = 0;
    // three-plane WEIGHT GRADIENT (csrc/wgrad_p3.hip, round 6): conv step i takes it when the caller kept the image of the step's
    // input alive (in_planes[i] = what sh_stack_forward wrote to planes[i - 1]), the image of its pre-activation gradient exists
    // (the step's backward-data pass gathers it) and the kernel takes the shape; SH_P3_WGRAD=0: the exact fp32 MFMA kernels everywhere
    static const int p3_wgrad_on = sh_env_int("SH_P3_WGRAD", 1, 0, 1);
    for (int i = last; i >= 0; --i) {
        const sh_stack_step& s = steps[i];
        const bool want_in = i > 0 || need_x_grad;
        const float* inp = i == 0 ? x : acts[i - 1];
        const Lay il = i == 0 ? lay(x_layout, rows0, B, c0) : lay(0, 0, B, cin_of[i]);
        float* gi = want_in ? gin[i] : nullptr;
        SH_REQUIRE(!want_in || gi, SH_ERR_INVALID_ARG, "sh_stack_backward: no gradient buffer for the input of step %d", i);
        const Lay gl = i == 0 ? lay(x_layout, rows0, B, c0) : lay(0, 0, B, cin_of[i]);
        // the activation derivative of the layer that produced this step's input is applied by whoever writes gin[i]
        const float* yprev = nullptr; Lay yl{0, 0}; int act_prev = 0, zero_prev = -1;
        if (i > 0 && steps[i - 1].kind == 0) {
            yprev = acts[i - 1]; yl = lay(0, 0, B, steps[i - 1].cout); act_prev = steps[i - 1].act; zero_prev = steps[i - 1].zero_row;
        }
        // gin[i] is the pre-activation gradient of conv step i - 1: does that step's backward-data pass want its image?
        void* gi_img = (want_in && i > 0 && steps[i - 1].kind == 0) ? bwd_p3(i - 1) : nullptr;
        bool gi_img_done = false;
        if (s.kind == 0) {
            SH_REQUIRE(workspace && workspace[i], SH_ERR_INVALID_ARG, "sh_stack_backward: no workspace for step %d", i);
            // a 16 -> 3 channel layer takes its weight gradient in role-swapped form (wgrad_thin.hip): it reads the extended
            // gradient buffer through the transposed table, so it runs after the pre-sum launches below
            const bool thin = want_in && s.table_t && s.R == s.n_in && il.sb == s.cin && il.sv == (long)B * s.cin && cl.sb == s.cout &&
                              cl.sv == (long)B * s.cout && sh_spiral_conv_bwd_wgt_thin_ok(B, s.n_in, s.S, s.cin, s.cout, SH_DTYPE_F32);
            const bool p3 = !thin && cur_img && cl.sb == s.cout && cl.sv == (long)B * s.cout;
            // backward-data over ragged source lists (round 6): every source an image row, no pre-summed rows - neither the launches
            // that fill them nor the rider in the weight gradient (SH_P3_RAGGED=0: the dense table with its pre-sums)
            static const int rag_on = sh_env_int("SH_P3_RAGGED", 1, 0, 1);
            static const int grp_on = sh_env_int("SH_P3_GROUPED", 1, 0, 1);
            // ... as GROUPS of input rows sharing one list where the launch has groups enough (also the layers that gather 16 channels,
            // which the one-row list kernel does not take)
            const bool grp_b = p3 && rag_on && grp_on && want_in && s.bg_rows && s.bg_pos && s.bg_out && s.bg_n > 0 &&
                               sh_spiral_conv_p3_grp_ok(B, s.S, s.cout, s.cin, s.bg_L) && sh_spiral_conv_p3_grp_pays(B, s.bg_n);
            const bool rag = grp_b || (p3 && rag_on && want_in && s.rag_rows && s.rag_pos && sh_spiral_conv_p3_rag_ok(B, s.S, s.cout, s.cin, s.rag_L));
            // the last pre-sum level of this layer rides in the weight-gradient launch (sh_spiral_conv_bwd_wgt_presum); an
            // earlier level (very long lists: two levels) runs first, on its own
            const bool ride = !thin && !rag && want_in && s.table_t && (s.n1 || s.n2);
            // SH_P3_PRESUM_IMG=1: the riders / pre-sum launches write the image of their rows; default 0: they stay plain and the
            // backward-data kernel splits those rows itself from the fp32 buffer (they are ~6 % of what it gathers)
            // pre-summed rows: imaged by their producers (the riders / pre-sum launches), or - LDS-resident plane kernel and at
            // least half as many of them as real rows - left fp32 and split by the backward-data kernel itself (they are ~6 % of
            // what it gathers; the riders' image stores cost their hosts more).  SH_P3_PRESUM_IMG: 0 / 1 force, 2 = that rule
            static const int presum_mode = sh_env_int("SH_P3_PRESUM_IMG", 2, 0, 2);
            const bool presum_img = !p3 ? false : presum_mode == 1 ? true
                                    : !(sh_spiral_conv_p3_kind(B, s.S, s.cout, s.cin) == 1 && (presum_mode == 0 || 2 * (s.n1 + s.n2) >= s.R));
            char* pimg0 = (p3 && presum_img) ? static_cast<char*>(cur_img) : nullptr;
            float* mut0 = const_cast<float*>(cur);
            if (ride && s.n1 && s.n2) {
                rc = sh_spmm_p3(s.sum1.rowptr, s.sum1.col, s.sum1.val, cur, cl.sv, cl.sb, mut0 + (long)s.R * cl.sv, cl.sv, cl.sb,
                                pimg0 ? pimg0 + sh_p3_bytes(s.R, B, s.cout) : nullptr, nullptr, 0, 0, 0, -1, B, s.n1, s.cout, stream);
                if (rc != SH_OK) return rc;
            }
            const bool p3w = p3 && p3_wgrad_on && i > 0 && in_planes && in_planes[i] && il.sb == s.cin && il.sv == (long)B * s.cin &&
                             sh_spiral_conv_bwd_wgt_p3_ok(B, s.R, s.S, s.cin, s.cout) && (((long)s.R * (B / 16)) % 2 == 0 || s.zero_row >= 0) &&
                             workspace_bytes[i] >= sh_spiral_conv_bwd_wgt_p3_workspace(B, s.R, s.S, s.cin, s.cout);
            // the forward pass left the fp32 rows of this step's input unwritten (keep_fp32 == 2) when this much was known from the
            // steps alone: then the step must run on the images, whatever the buffers the caller gave this pass
            const bool in_dropped = acts_fp32 == 2 && bwd_plane_static(steps, i, B);
            SH_REQUIRE(!in_dropped || (p3w && (!yprev || (in_planes[i] && yl.sb == s.cin && yl.sv == (long)B * s.cin))), SH_ERR_INVALID_ARG,
                       "sh_stack_backward: step %d: the forward pass kept only the image of its input (keep_fp32 == 2) but this pass cannot run "
                       "the step on images (gin_planes / wfrag3_t / in_planes / workspace of sh_spiral_conv_bwd_wgt_p3_workspace bytes)", i);
            if (!thin) {
                const sh_csr_ref& lm = s.n2 ? s.sum2 : s.sum1;
                const int ln = ride ? (s.n2 ? s.n2 : s.n1) : 0;
                float* lout = mut0 + (long)(s.R + (s.n2 ? s.n1 : 0)) * cl.sv;
                void* limg = (ln && pimg0) ? pimg0 + sh_p3_bytes(s.R + (s.n2 ? s.n1 : 0), B, s.cout) : nullptr;
                if (p3w) {
                    if (!cur_img_done) {                       // the gradient rows' image, unless their producer wrote it
                        rc = sh_to_p3(cur, cl.sv, cl.sb, cur_img, B, s.R, s.cout, stream);
                        if (rc != SH_OK) return rc;
                        cur_img_done = true;
                    }
                    rc = sh_spiral_conv_bwd_wgt_p3_presum(cur_img, s.zero_row, in_planes[i], s.table, workspace[i], workspace_bytes[i], cur, cl.sv, cl.sb,
                                                          ln ? lm.rowptr : nullptr, ln ? lm.col : nullptr, ln ? lm.val : nullptr,
                                                          ln ? lout : nullptr, limg, ln, B, s.R, s.S, s.cin, s.cout, stream);
                } else {
                    rc = sh_spiral_conv_bwd_wgt_presum(cur, cl.sv, cl.sb, inp, il.sv, il.sb, s.table, nullptr, nullptr, workspace[i],
                                                       workspace_bytes[i], ln ? lm.rowptr : nullptr, ln ? lm.col : nullptr,
                                                       ln ? lm.val : nullptr, ln ? lout : nullptr, limg, ln, B, s.R, s.S, s.cin, s.cout, mma_mode,
                                                       stream);
                }
                if (rc != SH_OK) return rc;
            }
            job_ws[njobs] = workspace[i]; job_dW[njobs] = dW[s.param]; job_db[njobs] = dbias ? dbias[s.param] : nullptr;
            jB[njobs] = B; jR[njobs] = s.R; jS[njobs] = s.S; jCi[njobs] = s.cin; jCo[njobs] = s.cout; jK[njobs] = p3w ? 2 : 0;
            SH_REQUIRE(job_dW[njobs], SH_ERR_INVALID_ARG, "sh_stack_backward: no dW buffer for parameter %d", s.param);
            ++njobs;
            if (want_in) {
                SH_REQUIRE(s.table_t, SH_ERR_INVALID_ARG, "sh_stack_backward: step %d has no transposed table", i);
                float* mut = const_cast<float*>(cur);          // the extra rows behind the R real ones of this step's own buffer
                char* pimg = pimg0;
                if (s.n1 && !ride && !rag) {
                    rc = sh_spmm_p3(s.sum1.rowptr, s.sum1.col, s.sum1.val, cur, cl.sv, cl.sb, mut + (long)s.R * cl.sv, cl.sv, cl.sb,
                                    pimg ? pimg + sh_p3_bytes(s.R, B, s.cout) : nullptr, nullptr, 0, 0, 0, -1, B, s.n1, s.cout, stream);
                    if (rc != SH_OK) return rc;
                }
                if (s.n2 && !ride && !rag) {
                    rc = sh_spmm_p3(s.sum2.rowptr, s.sum2.col, s.sum2.val, cur, cl.sv, cl.sb, mut + (long)(s.R + s.n1) * cl.sv,
                                    cl.sv, cl.sb, pimg ? pimg + sh_p3_bytes(s.R + s.n1, B, s.cout) : nullptr, nullptr, 0, 0, 0, -1, B, s.n2,
                                    s.cout, stream);
                    if (rc != SH_OK) return rc;
                }
                // ... and computes the input gradient from the same staged gradient rows when that buffer is plain vertex-major
                // and the activation to differentiate (if any) produced the layer input itself
                const bool thin_dx = thin && gl.sb == s.cin && gl.sv == (long)B * s.cin && (!yprev || yprev == inp);
                if (thin) {
                    const bool img_out = thin_dx && gi_img && sh_p3_bytes(1, B, s.cin);
                    rc = sh_spiral_conv_bwd_wgt_thin(cur, cl.sv, cl.sb, inp, SH_DTYPE_F32, il.sv, il.sb, s.table_t, workspace[i],
                                                     workspace_bytes[i], weights[s.param], thin_dx ? gi : nullptr, gl.sv, gl.sb,
                                                     img_out ? gi_img : nullptr, yprev ? act_prev : SH_ACT_IDENTITY, zero_prev, B, s.R, s.n_in,
                                                     s.S, s.cin, s.cout, SH_DTYPE_F32, stream);
                    if (rc != SH_OK) return rc;
                    gi_img_done = img_out;
                }
                if (!thin_dx) {
                    if (p3) {
                        // image of the gradient rows this pass gathers: the R real rows unless their producer wrote them, and
                        // the pre-summed rows behind them
                        const int r0 = 0, r1 = cur_img_done ? 0 : s.R;          // (the pre-sum launches wrote the image of their rows)
                        if (r1 > r0) {
                            rc = sh_to_p3(cur + (long)r0 * cl.sv, cl.sv, cl.sb, static_cast<char*>(cur_img) + sh_p3_bytes(r0, B, s.cout), B, r1 - r0,
                                          s.cout, stream);
                            if (rc != SH_OK) return rc;
                        }
                        const bool img_out = gi_img && gl.sb == s.cin && gl.sv == (long)B * s.cin && sh_p3_bytes(1, B, s.cin);
                        // the step that takes this gradient reads it through the image alone (acts_fp32 == 2 holds it to that): image only
                        float* gi_f = (acts_fp32 == 2 && img_out && i > 0 && bwd_grad_plane_static(steps, i - 1, B)) ? nullptr : gi;
                        // the activation to differentiate, from its image when the caller kept the forward images (SH_P3_YPREV_IMG=0: fp32)
                        static const int yimg_on = sh_env_int("SH_P3_YPREV_IMG", 1, 0, 1);
                        const void* yimg = (yimg_on && yprev && in_planes && in_planes[i] && yl.sb == s.cin && yl.sv == (long)B * s.cin &&
                                            sh_p3_bytes(1, B, s.cin)) ? in_planes[i] : nullptr;
                        if (grp_b)
                            rc = sh_spiral_conv_p3_grp(cur_img, s.bg_rows, s.bg_pos, s.bg_out, s.bg_n, s.bg_L, wfrag3_t[i], nullptr, gi_f, gl.sv, gl.sb,
                                                       img_out ? gi_img : nullptr, yprev, yl.sv, yl.sb, yimg, act_prev, zero_prev, 1, B, s.n_in, s.S, s.cout,
                                                       s.cin, stream);
                        else if (rag)
                            rc = sh_spiral_conv_bwd_data_p3_rag(cur_img, s.rag_rows, s.rag_pos, s.rag_L, wfrag3_t[i], gi_f, gl.sv, gl.sb, img_out ? gi_img : nullptr,
                                                                yprev, yl.sv, yl.sb, yimg, act_prev, zero_prev, B, s.n_in, s.S, s.cin, s.cout, stream);
                        else
                        rc = sh_spiral_conv_bwd_data_p3(cur_img, s.zero_row, presum_img ? nullptr : cur, cl.sv, cl.sb, s.R, s.table_t, wfrag3_t[i], gi_f, gl.sv, gl.sb, img_out ? gi_img : nullptr, yprev, yl.sv,
                                                        yl.sb, yimg, act_prev, zero_prev, B, s.n_in, s.S, s.cin, s.cout, stream);
                        gi_img_done = img_out;
                    } else {
                        // the "no source" entries of table_t point at this step's own dummy row of dpre (stack.py ConvStep.finalize),
                        // which its producer forced to zero
                        rc = sh_spiral_conv_bwd_data_z(cur, cl.sv, cl.sb, s.zero_row, s.table_t, weight_t[i], gi, gl.sv, gl.sb, yprev, yl.sv,
                                                       yl.sb, act_prev, zero_prev, B, s.n_in, s.S, s.cin, s.cout, mma_mode, stream);
                    }
                    if (rc != SH_OK) return rc;
                }
            }
        } else if (want_in) {
            SH_REQUIRE(s.mt.rowptr && s.mt.col && s.mt.val, SH_ERR_INVALID_ARG, "sh_stack_backward: step %d has no transposed matrix", i);
            const bool img_out = gi_img && gl.sb == cin_of[i] && gl.sv == (long)B * cin_of[i] && sh_p3_bytes(1, B, cin_of[i]);
            float* gi_f = (acts_fp32 == 2 && img_out && i > 0 && bwd_grad_plane_static(steps, i - 1, B)) ? nullptr : gi;
            rc = sh_spmm_p3(s.mt.rowptr, s.mt.col, s.mt.val, cur, cl.sv, cl.sb, gi_f, gl.sv, gl.sb, img_out ? gi_img : nullptr, yprev, yl.sv, yl.sb,
                            act_prev, zero_prev, B, s.m_cols, cin_of[i], stream);
            if (rc != SH_OK) return rc;
            gi_img_done = img_out;
        }
        if (want_in) { cur = gi; cl = gl; cur_img = gi_img; cur_img_done = gi_img_done; }
    }
    for (int k = 0; k < njobs; k += 16) {
        const int n = njobs - k < 16 ? njobs - k : 16;
        rc = sh_spiral_conv_bwd_wgt_reduce_multi_kinds(n, job_ws + k, job_dW + k, job_db + k, jB + k, jR + k, jS + k, jCi + k, jCo + k, jK + k, stream);
        if (rc != SH_OK) return rc;
    }
    return SH_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// The same two sequencers for the bf16 compute path (BASELINE config 3).  Tensors between steps are bf16 vertex-major;
// the stack input may be fp32 with 3 channels (xyz), the stack output fp32 when it has <= 16 channels (x_hat).  The
// working copies of the conv weights are converted from the fp32 masters by ONE launch at the start of each pass.
static inline long esz_of(int dtype) { return dtype == SH_DTYPE_BF16 ? 2 : 4; }

int sh_stack_forward_bf16(int n_steps, const sh_stack_step* steps, const void* x, int x_dtype, int x_layout, int rows0, int c0, int B,
                          const float* const* weights, const float* const* biases, void* const* wfrag, int wfrag_ready,
                          void* const* outs, int out_dtype, int out_layout, sh_stream_t stream) {
    int rc = check_steps(n_steps, steps, c0, "sh_stack_forward_bf16");
    if (rc != SH_OK) return rc;
    if (B > 0 && (rc = check_tensor_sizes(n_steps, steps, rows0, c0, B, "sh_stack_forward_bf16")) != SH_OK) return rc;
    SH_REQUIRE(x && weights && outs && wfrag && B > 0, SH_ERR_INVALID_ARG, "sh_stack_forward_bf16: null pointer or empty batch");
    SH_REQUIRE(n_steps <= 64, SH_ERR_UNSUPPORTED, "sh_stack_forward_bf16: more than 64 steps");
    {
        const float* w[64]; void* wf[64]; int S[64], Ci[64], Co[64], tr[64];
        int n = 0;
        for (int i = 0; i < n_steps; ++i) {
            if (steps[i].kind != 0) continue;
            SH_REQUIRE(wfrag[i], SH_ERR_INVALID_ARG, "sh_stack_forward_bf16: no weight-fragment buffer for step %d", i);
            w[n] = weights[steps[i].param]; wf[n] = wfrag[i]; S[n] = steps[i].S; Ci[n] = steps[i].cin; Co[n] = steps[i].cout; tr[n] = 0;
            ++n;
        }
        if (n && !wfrag_ready) {
            rc = sh_conv_wfrag_prep_multi(n, w, wf, S, Ci, Co, tr, stream);
            if (rc != SH_OK) return rc;
        }
    }
    const void* cur = x;
    int cd = x_dtype;
    Lay cl = lay(x_layout, rows0, B, c0);
    int c = c0;
    for (int i = 0; i < n_steps; ++i) {
        const sh_stack_step& s = steps[i];
        const bool is_last = i == n_steps - 1;
        const int co = s.kind == 0 ? s.cout : c;
        const int od = is_last ? out_dtype : SH_DTYPE_BF16;
        const Lay ol = lay(is_last ? out_layout : 0, out_rows(s), B, co);
        SH_REQUIRE(outs[i], SH_ERR_INVALID_ARG, "sh_stack_forward_bf16: no output buffer for step %d", i);
        if (s.kind == 0) {
            rc = sh_spiral_conv_fwd_bf16(cur, cd, cl.sv, cl.sb, s.table, wfrag[i], biases ? biases[s.param] : nullptr, outs[i], od, ol.sv,
                                         ol.sb, B, s.R, s.S, s.cin, s.cout, s.act, s.zero_row, stream);
        } else {
            SH_REQUIRE(cd == SH_DTYPE_BF16 && od == SH_DTYPE_BF16, SH_ERR_UNSUPPORTED,
                       "sh_stack_forward_bf16: re-sampling step %d needs bf16 on both sides", i);
            if (s.extend) {
                SH_REQUIRE(i > 0 && !is_last && outs[i] == outs[i - 1] && cl.sb == c, SH_ERR_INVALID_ARG,
                           "sh_stack_forward_bf16: step %d appends to its input, which must be the vertex-major output buffer of step %d", i, i - 1);
                rc = sh_spmm_bf16(s.m.rowptr, s.m.col, s.m.val, cur, cl.sv, cl.sb, static_cast<char*>(outs[i]) + (long)s.m_cols * cl.sv * 2, cl.sv,
                                  cl.sb, nullptr, 0, 0, 0, -1, B, s.m_rows, c, stream);
            } else {
                rc = sh_spmm_bf16(s.m.rowptr, s.m.col, s.m.val, cur, cl.sv, cl.sb, outs[i], ol.sv, ol.sb, nullptr, 0, 0, 0, -1, B, s.m_rows, c,
                                  stream);
            }
        }
        if (rc != SH_OK) return rc;
        cur = outs[i]; cd = od; cl = ol; c = co;
    }
    return SH_OK;
}

int sh_stack_backward_bf16(int n_steps, const sh_stack_step* steps, const void* x, int x_dtype, int x_layout, int rows0, int c0, int B,
                           const void* const* acts, const void* g, int out_dtype, int out_layout, const float* const* weights,
                           void* const* gin, int gx_dtype, void* dpre_last, void* const* wfrag_t, int wfrag_ready,
                           void* const* workspace, const size_t* workspace_bytes, float* const* dW, float* const* dbias, int need_x_grad,
                           sh_stream_t stream) {
    int rc = check_steps(n_steps, steps, c0, "sh_stack_backward_bf16");
    if (rc != SH_OK) return rc;
    if (B > 0 && (rc = check_tensor_sizes(n_steps, steps, rows0, c0, B, "sh_stack_backward_bf16")) != SH_OK) return rc;
    SH_REQUIRE(x && acts && g && weights && gin && dW && B > 0, SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: null pointer or empty batch");
    SH_REQUIRE(n_steps <= 64, SH_ERR_UNSUPPORTED, "sh_stack_backward_bf16: more than 64 steps");
    const int last = n_steps - 1;
    int cin_of[64];
    {
        int c = c0;
        for (int i = 0; i < n_steps; ++i) { cin_of[i] = c; if (steps[i].kind == 0) c = steps[i].cout; }
    }
    {   // backward-data operands of all conv steps: one conversion launch
        const float* w[64]; void* wf[64]; int S[64], Ci[64], Co[64], tr[64];
        int n = 0;
        for (int i = 0; i < n_steps; ++i) {
            if (steps[i].kind != 0 || !(i > 0 || need_x_grad)) continue;
            SH_REQUIRE(wfrag_t && wfrag_t[i], SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: no weight-fragment buffer for step %d", i);
            w[n] = weights[steps[i].param]; wf[n] = wfrag_t[i]; S[n] = steps[i].S; Ci[n] = steps[i].cin; Co[n] = steps[i].cout; tr[n] = 1;
            ++n;
        }
        if (n && !wfrag_ready) {
            rc = sh_conv_wfrag_prep_multi(n, w, wf, S, Ci, Co, tr, stream);
            if (rc != SH_OK) return rc;
        }
    }
    const void* cur; Lay cl; int cd;
    {
        const sh_stack_step& s = steps[last];
        if (s.kind == 0) {
            SH_REQUIRE(dpre_last, SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: no dpre_last buffer");
            const Lay ol = lay(out_layout, s.R, B, s.cout), dl = lay(0, 0, B, s.cout);
            if (out_dtype == SH_DTYPE_F32)
                rc = sh_act_backward(static_cast<const float*>(g), ol.sv, ol.sb, static_cast<const float*>(acts[last]), ol.sv, ol.sb,
                                     static_cast<float*>(dpre_last), dl.sv, dl.sb, B, s.R, s.cout, s.act, s.zero_row, stream);
            else
                rc = sh_act_backward_bf16(g, ol.sv, ol.sb, acts[last], ol.sv, ol.sb, dpre_last, dl.sv, dl.sb, B, s.R, s.cout, s.act, s.zero_row,
                                          stream);
            if (rc != SH_OK) return rc;
            cur = dpre_last; cl = dl; cd = out_dtype;
        } else {
            SH_REQUIRE(out_dtype == SH_DTYPE_BF16, SH_ERR_UNSUPPORTED, "sh_stack_backward_bf16: a re-sampling last step needs a bf16 gradient");
            cur = g; cl = lay(out_layout, s.m_rows, B, cin_of[last]); cd = out_dtype;
        }
    }
    const void* job_ws[64]; float* job_dW[64]; float* job_db[64]; int jB[64], jR[64], jS[64], jCi[64], jCo[64];
    int njobs = 0;
    for (int i = last; i >= 0; --i) {
        const sh_stack_step& s = steps[i];
        const bool want_in = i > 0 || need_x_grad;
        const void* inp = i == 0 ? x : acts[i - 1];
        const int ind = i == 0 ? x_dtype : SH_DTYPE_BF16;
        const Lay il = i == 0 ? lay(x_layout, rows0, B, c0) : lay(0, 0, B, cin_of[i]);
        void* gi = want_in ? gin[i] : nullptr;
        const int gd = i == 0 ? gx_dtype : SH_DTYPE_BF16;
        SH_REQUIRE(!want_in || gi, SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: no gradient buffer for the input of step %d", i);
        const Lay gl = i == 0 ? lay(x_layout, rows0, B, c0) : lay(0, 0, B, cin_of[i]);
        const void* yprev = nullptr; Lay yl{0, 0}; int act_prev = 0, zero_prev = -1;
        if (i > 0 && steps[i - 1].kind == 0) {
            yprev = acts[i - 1]; yl = lay(0, 0, B, steps[i - 1].cout); act_prev = steps[i - 1].act; zero_prev = steps[i - 1].zero_row;
        }
        if (s.kind == 0) {
            SH_REQUIRE(workspace && workspace[i], SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: no workspace for step %d", i);
            // role-swapped weight gradient of a 16 -> 3 channel layer (see sh_stack_backward)
            const bool thin = want_in && s.table_t && cd == SH_DTYPE_F32 && ind == SH_DTYPE_BF16 && s.R == s.n_in && il.sb == s.cin &&
                              il.sv == (long)B * s.cin && cl.sb == s.cout && cl.sv == (long)B * s.cout &&
                              sh_spiral_conv_bwd_wgt_thin_ok(B, s.n_in, s.S, s.cin, s.cout, SH_DTYPE_BF16);
            if (!thin) {
                rc = sh_spiral_conv_bwd_wgt_bf16(cur, cd, cl.sv, cl.sb, inp, ind, il.sv, il.sb, s.table, workspace[i], workspace_bytes[i], B,
                                                 s.R, s.S, s.cin, s.cout, stream);
                if (rc != SH_OK) return rc;
            }
            job_ws[njobs] = workspace[i]; job_dW[njobs] = dW[s.param]; job_db[njobs] = dbias ? dbias[s.param] : nullptr;
            jB[njobs] = B; jR[njobs] = s.R; jS[njobs] = s.S; jCi[njobs] = s.cin; jCo[njobs] = s.cout;
            SH_REQUIRE(job_dW[njobs], SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: no dW buffer for parameter %d", s.param);
            ++njobs;
            if (want_in) {
                SH_REQUIRE(s.table_t, SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: step %d has no transposed table", i);
                // backward-data over ragged source lists (round 6): every source a real row of dpre - neither the pre-sum launches nor
                // the empty slots of the dense transposed table (SH_BF16_RAGGED=0: the dense form)
                static const int rag_on = sh_env_int("SH_BF16_RAGGED", 1, 0, 1);
                const bool rag = rag_on && !thin && cd == SH_DTYPE_BF16 && gd == SH_DTYPE_BF16 && s.rag_rows && s.rag_pos &&
                                 sh_spiral_conv_bf16_rag_ok(B, s.S, s.cout, s.cin, s.rag_L);
                char* mut = static_cast<char*>(const_cast<void*>(cur));      // extra rows behind the R real ones of this step's buffer
                const long rb = cl.sv * esz_of(cd);
                for (int lev = 0; lev < 2; ++lev) {
                    const int n = lev == 0 ? s.n1 : s.n2;
                    if (!n || rag) continue;
                    const sh_csr_ref& m = lev == 0 ? s.sum1 : s.sum2;
                    void* dst = mut + (long)(s.R + (lev == 0 ? 0 : s.n1)) * rb;
                    if (cd == SH_DTYPE_F32)
                        rc = sh_spmm(m.rowptr, m.col, m.val, static_cast<const float*>(cur), cl.sv, cl.sb, static_cast<float*>(dst), cl.sv, cl.sb,
                                     nullptr, 0, 0, 0, -1, B, n, s.cout, stream);
                    else
                        rc = sh_spmm_bf16(m.rowptr, m.col, m.val, cur, cl.sv, cl.sb, dst, cl.sv, cl.sb, nullptr, 0, 0, 0, -1, B, n, s.cout, stream);
                    if (rc != SH_OK) return rc;
                }
                const bool thin_dx = thin && gd == SH_DTYPE_BF16 && gl.sb == s.cin && gl.sv == (long)B * s.cin && (!yprev || yprev == inp);
                if (thin) {
                    rc = sh_spiral_conv_bwd_wgt_thin(static_cast<const float*>(cur), cl.sv, cl.sb, inp, SH_DTYPE_BF16, il.sv, il.sb, s.table_t,
                                                     workspace[i], workspace_bytes[i], weights[s.param], thin_dx ? gi : nullptr, gl.sv, gl.sb,
                                                     nullptr, yprev ? act_prev : SH_ACT_IDENTITY, zero_prev, B, s.R, s.n_in, s.S, s.cin, s.cout,
                                                     SH_DTYPE_BF16, stream);
                    if (rc != SH_OK) return rc;
                }
                if (rag)
                    rc = sh_spiral_conv_bwd_data_bf16_rag(cur, cl.sv, cl.sb, s.rag_rows, s.rag_pos, s.rag_L, wfrag_t[i], gi, gl.sv, gl.sb, yprev, yl.sv,
                                                          yl.sb, act_prev, zero_prev, B, s.n_in, s.S, s.cin, s.cout, stream);
                else if (!thin_dx)
                    rc = sh_spiral_conv_bwd_data_bf16(cur, cd, cl.sv, cl.sb, s.table_t, wfrag_t[i], gi, gd, gl.sv, gl.sb, yprev, yl.sv, yl.sb,
                                                      act_prev, zero_prev, B, s.n_in, s.S, s.cin, s.cout, stream);
                if (rc != SH_OK) return rc;
            }
        } else if (want_in) {
            SH_REQUIRE(s.mt.rowptr && s.mt.col && s.mt.val, SH_ERR_INVALID_ARG, "sh_stack_backward_bf16: step %d has no transposed matrix", i);
            SH_REQUIRE(cd == SH_DTYPE_BF16 && gd == SH_DTYPE_BF16, SH_ERR_UNSUPPORTED,
                       "sh_stack_backward_bf16: re-sampling step %d needs bf16 on both sides", i);
            rc = sh_spmm_bf16(s.mt.rowptr, s.mt.col, s.mt.val, cur, cl.sv, cl.sb, gi, gl.sv, gl.sb, yprev, yl.sv, yl.sb, act_prev, zero_prev, B,
                              s.m_cols, cin_of[i], stream);
            if (rc != SH_OK) return rc;
        }
        if (want_in) { cur = gi; cl = gl; cd = gd; }
    }
    for (int k = 0; k < njobs; k += 16) {
        const int n = njobs - k < 16 ? njobs - k : 16;
        rc = sh_spiral_conv_bwd_wgt_reduce_multi_bf16(n, job_ws + k, job_dW + k, job_db + k, jB + k, jR + k, jS + k, jCi + k, jCo + k, stream);
        if (rc != SH_OK) return rc;
    }
    return SH_OK;
}

}  // extern "C"
