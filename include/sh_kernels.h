/*
 * sh_kernels.h - C ABI of libsh_kernels.so: the MI355X (gfx950) kernels behind the
 * spiral-convolution mesh-autoencoder training path of SemanticHuman.
 *
 * The reference (pure Python/PyTorch) has no FFI layer; its boundary for this path is the
 * nn.Module surface of models.py.  Each entry point below names the reference lines whose
 * ATen dispatches it replaces.  A maintainer binds them with ctypes;
 * semantichuman_amd/_lib.py is exactly that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch caching allocator); tables are
 *     int32, tensors fp32; nothing is allocated, freed or retained by the library
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it
 *     (no synchronisation, no hipMalloc: safe under hipGraph capture)
 *   - activations are addressed with explicit element strides so one kernel serves both
 *       batch-major  [B][rows][C]  (the reference layout):  sv = C,     sb = rows*C
 *       vertex-major [rows][B][C]  (the fast internal layout): sv = B*C, sb = C
 *     element (row r, batch b, channel c) lives at  base + r*sv + b*sb + c
 *   - return value: 0 on success, negative sh_status on failure (never throws);
 *     sh_last_error() gives a thread-local message
 */
#ifndef SH_KERNELS_H
#define SH_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sh_stream_t;

#if defined(__GNUC__)
#define SH_API __attribute__((visibility("default")))
#else
#define SH_API
#endif

enum sh_status {
    SH_OK = 0,
    SH_ERR_INVALID_ARG = -1,   /* null pointer, negative size, misaligned stride */
    SH_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels were built for */
    SH_ERR_WORKSPACE = -3,     /* workspace too small */
    SH_ERR_LAUNCH = -4         /* hipLaunchKernel reported an error */
};

/* activation ids: the strings reference models.py:19-32 accepts */
enum sh_act {
    SH_ACT_IDENTITY = 0,
    SH_ACT_RELU = 1,
    SH_ACT_ELU = 2,         /* alpha = 1 */
    SH_ACT_LEAKY_RELU = 3,  /* slope 0.02 (models.py:24) */
    SH_ACT_SIGMOID = 4,
    SH_ACT_TANH = 5
};

SH_API int sh_version(void);
SH_API const char* sh_last_error(void);
/* 16 hex digits: SHA-256 over the kernel sources this library was compiled from (csrc/Makefile BUILD_ID).  bench.py quotes
 * a committed PMC profile only for the build it was taken on. */
SH_API const char* sh_build_id(void);

/* Arithmetic form of the fp32 path's matrix products (SpiralConv forward / backward-data / weight gradient).  The
 * reference computes them with fp32 FMAs (models.py:45, aten::addmm).
 *   SH_MMA_EXACT    v_mfma_f32_16x16x4_f32: an exact fp32 FMA chain.
 *   SH_MMA_SPLIT3   every fp32 operand split EXACTLY into three bf16 terms (8+8+8 significand bits), the six leading
 *                   partial products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; dropped terms < 2^-24 |w||x|,
 *                   i.e. fp32-level error (the parity tolerances in tests/ are the same for all forms); the split is done
 *                   by every consumer on what it gathered.
 *   SH_MMA_PLANES3  the same arithmetic with the split done ONCE, by the producer of a tensor, into three bf16 planes that
 *                   the consumers gather (the ..._p3 entry points at the end of this header; sh_stack_forward /
 *                   sh_stack_backward run them wherever the plane buffers they are given allow, and the SPLIT3 kernels
 *                   elsewhere).  On the per-layer fp32 entry points it selects the SPLIT3 kernels.
 * `mma_mode` is an ARGUMENT of every entry point whose kernel choice depends on it: there is no process-wide setting (a
 * forward pass and the backward pass of another node may run concurrently on different threads in different forms). */
enum sh_mma_mode { SH_MMA_EXACT = 0, SH_MMA_SPLIT3 = 1, SH_MMA_PLANES3 = 2 };

/* Optional per-kernel timing with HIP events attached to the kernel dispatch itself (begin / end of
 * the kernel's execution, what rocprofv3 --kernel-trace reports; minor helper kernels are bracketed
 * by event records on the stream instead).  Used by bench.py for the roofline figures; off by
 * default, must be off while a hipGraph is being captured.  sh_profile_get synchronises on the events. */
/* Diagnostic: shader clock currently granted.  n_workgroups single-wave workgroups each time `iters` dependent fp32
 * MFMAs with the shader-cycle counter and the 100 MHz counter: out[2*i] = shader cycles, out[2*i+1] = 100 MHz ticks
 * (device memory, 2*n_workgroups entries).  tools/clock_probe.py launches it between training steps. */
SH_API int sh_clock_probe(unsigned long long* out, int n_workgroups, int iters, sh_stream_t stream);
SH_API int sh_profile_enable(int on);                 /* on=1 start recording (clears), on=0 stop */
SH_API int sh_profile_count(void);
SH_API int sh_profile_get(int i, char* name, int name_len, float* ms);

/* ---------------------------------------------------------------------------------------------
 * SpiralConv forward.  Replaces models.py:40-51 (aten::index gather, aten::addmm, activation,
 * dummy-row mask) with one fused kernel; the gathered [B*(N+1), S*Cin] matrix is never
 * materialised.
 *   y[r,b,:] = act( sum_s x[table[r,s], b, :] . W[:, s*Cin:(s+1)*Cin]^T + bias )      r < R
 *   y[zero_row,b,:] = 0   (zero_row < 0: no masking)
 * table: int32 [R][S], values in [0, n_in) - the caller has already mapped the reference's -1
 * to the dummy row (torch negative-index wrap, models.py:42).  R may be smaller than n_in: a
 * row-select down-sampling D (models.py:127) is fused by passing table = spirals[sel].
 * weight: [Cout][S*Cin] row-major = nn.Linear.weight (models.py:17); bias may be NULL.
 */
SH_API int sh_spiral_conv_fwd(const float* x, int64_t x_sv, int64_t x_sb,
                       const int32_t* table, const float* weight, const float* bias,
                       float* y, int64_t y_sv, int64_t y_sb,
                       int B, int R, int S, int Cin, int Cout, int act, int zero_row, int mma_mode,
                       sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SpiralConv backward w.r.t. the input.  Replaces autograd's aten::mm (dG = dY.W) +
 * aten::_index_put_impl_(accumulate=True) scatter-add by a GATHER over the
 * transposed table - no atomics, fixed summation order, bitwise reproducible.  It is the same
 * fused gather+MFMA kernel as the forward pass:
 *   dx[u,b,:] = sum_s dpre[table_t[u,s], b, :] . W[:, s*Cin:(s+1)*Cin]            u < n_in
 * table_t: int32 [n_in][S]; table_t[u,s] = the output row r with table[r,s] == u.  Where no such r
 * exists it must point at a row of dpre that is zero (the masked dummy row); where several exist
 * (irregular vertices, the dummy row) the caller first sums those rows of dpre into an extra row
 * (sh_spmm with unit values) and points table_t at it - semantichuman_amd/stack.py does both.
 * weight_t: [Cin][S*Cout], weight_t[ci][s*Cout+co] = weight[co][s*Cin+ci]  (sh_weight_transpose).
 * Optional epilogue (yprev != NULL): dx is multiplied by act_prev'(yprev) evaluated from the
 * OUTPUT yprev of the layer that produced x, and row zero_row is forced to 0, so dx is directly
 * that layer's pre-activation gradient (aten::elu_backward + mask backward).
 */
SH_API int sh_spiral_conv_bwd_data(const float* dpre, int64_t dp_sv, int64_t dp_sb,
                            const int32_t* table_t, const float* weight_t,
                            float* dx, int64_t dx_sv, int64_t dx_sb,
                            const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row,
                            int B, int n_in, int S, int Cin, int Cout, int mma_mode,
                            sh_stream_t stream);
/* The same with one more piece of knowledge: row `dpre_zero_row` of dpre is all zero and is the row the "no source" entries
 * of table_t point at (>= 0; -1 = unknown: identical to sh_spiral_conv_bwd_data).  With a batch slice of 16 and gathered
 * channel counts that are multiples of 16 the kernels then skip the matrix products of (vertex, spiral position) pairs
 * without a source - exact zeros: the result is bitwise the same; 51-59 % of the entries on down-sampling levels. */
SH_API int sh_spiral_conv_bwd_data_z(const float* dpre, int64_t dp_sv, int64_t dp_sb, int dpre_zero_row, const int32_t* table_t,
                                     const float* weight_t, float* dx, int64_t dx_sv, int64_t dx_sb, const float* yprev,
                                     int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin,
                                     int Cout, int mma_mode, sh_stream_t stream);

/* weight [Cout][S*Cin] -> weight_t [Cin][S*Cout] (see above). */
SH_API int sh_weight_transpose(const float* weight, float* weight_t, int S, int Cin, int Cout, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SpiralConv backward w.r.t. weight and bias.  Replaces aten::mm dW = dY^T.G; the
 * gathered matrix G is re-gathered on the fly.  Two-stage, deterministic: per-block partial
 * slabs in `workspace`, then a fixed-order reduction.
 *   dW[co, s*Cin+ci] = sum_{r,b} dpre[r,b,co] * x[table[r,s], b, ci];   dbias[co] = sum_{r,b} dpre[r,b,co]
 * dbias may be NULL.  Results OVERWRITE dW/dbias.
 */
SH_API size_t sh_spiral_conv_bwd_wgt_workspace(int B, int R, int S, int Cin, int Cout);
SH_API int sh_spiral_conv_bwd_wgt(const float* dpre, int64_t dp_sv, int64_t dp_sb,
                           const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table,
                           float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                           int B, int R, int S, int Cin, int Cout, int mma_mode,
                           sh_stream_t stream);

/* The same launch with a rider: the LAST pre-sum level of the layer's backward-data pass (sh_stack_step.sum1 / sum2: rows
 * R .. of the dpre buffer = sums of rows that several (vertex, position) pairs of the transposed table share) -
 *     sum_out[r][b][:] = sum_e sum_val[e] * dpre[sum_col[e]][b][:]      r < sum_rows, strides of dpre, sh_spmm's arithmetic
 * - computed by extra workgroups of the weight-gradient launch (both only read the first R rows of dpre) instead of a launch of
 * its own, when a second wave of the kernel fits on a SIMD; otherwise the entry point issues sh_spmm itself, first.
 * sum_out_planes != NULL: the three-plane image of those rows (sh_spmm_p3's y_planes) is written with them.
 * sum_rows == 0: exactly sh_spiral_conv_bwd_wgt. */
SH_API int sh_spiral_conv_bwd_wgt_presum(const float* dpre, int64_t dp_sv, int64_t dp_sb, const float* x, int64_t x_sv, int64_t x_sb,
                                         const int32_t* table, float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                                         const int32_t* sum_rowptr, const int32_t* sum_col, const float* sum_val, float* sum_out,
                                         void* sum_out_planes, int sum_rows, int B, int R, int S, int Cin, int Cout, int mma_mode,
                                         sh_stream_t stream);

/* Batched forms for a whole stack of layers (one launch instead of one per layer; host arrays of
 * n_layers entries, passed by value into the kernel arguments -> graph-capturable):
 *  - sh_spiral_conv_bwd_wgt called with dW == NULL only writes its partial slabs into `workspace`;
 *    sh_spiral_conv_bwd_wgt_reduce_multi then reduces the slabs of up to 16 layers (same B,R,S,Cin,Cout
 *    as the producing calls; dbias[i] may be NULL);
 *  - sh_weight_transpose_multi transposes up to 32 weights. */
SH_API int sh_spiral_conv_bwd_wgt_reduce_multi(int n_layers, const void* const* workspaces, float* const* dW,
                                        float* const* dbias, const int* B, const int* R, const int* S,
                                        const int* Cin, const int* Cout, sh_stream_t stream);
SH_API int sh_weight_transpose_multi(int n_layers, const float* const* weight, float* const* weight_t,
                              const int* S, const int* Cin, const int* Cout, sh_stream_t stream);

/* dpre = dy * act'(y) with row zero_row forced to 0 (aten::elu_backward + mask, models.py:46-51). */
SH_API int sh_act_backward(const float* dy, int64_t dy_sv, int64_t dy_sb,
                    const float* y, int64_t y_sv, int64_t y_sb,
                    float* dpre, int64_t dp_sv, int64_t dp_sb,
                    int B, int R, int C, int act, int zero_row, sh_stream_t stream);
/* The same launch with the weight transposes of a stack's backward pass (sh_weight_transpose_multi's arguments) as extra
 * workgroups: both are a few microseconds of work, a launch boundary costs ~3 us whatever follows it.  n_layers == 0:
 * exactly sh_act_backward. */
SH_API int sh_act_backward_tr(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb,
                       float* dpre, int64_t dp_sv, int64_t dp_sb, int B, int R, int C, int act, int zero_row, int n_layers,
                       const float* const* weight, float* const* weight_t, const int* S, const int* Cin, const int* Cout,
                       sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse mesh re-sampling.  Replaces the dense aten::bmm of models.py:127 (D) and :148 (U), and
 * with the transposed CSR their backward:
 *   y[r,b,:] = sum_{e in [rowptr[r], rowptr[r+1])} val[e] * x[col[e], b, :]
 * Optional epilogue as in sh_spiral_conv_bwd_data (yprev != NULL).
 */
SH_API int sh_spmm(const int32_t* rowptr, const int32_t* col, const float* val,
            const float* x, int64_t x_sv, int64_t x_sb,
            float* y, int64_t y_sv, int64_t y_sb,
            const float* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row,
            int B, int rows, int C, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-stack execution.  The reference runs encode / decode as Python loops over layers
 * (models.py:119-128 conv + D per level, :146-153 U + conv per level) and leaves the backward chain to
 * autograd; issued call by call from Python that is ~25 us of host time per launch (measured), i.e. the
 * host, not the GPU, paces an eagerly launched step.  These two entry points issue the launches of a
 * whole stack - forward, or the hand-scheduled backward chain - from one call.  They only sequence the
 * entry points above (same kernels, same results, same launch order as calling them one by one), never
 * allocate and never synchronise: every buffer is the caller's.
 *
 * A step is a SpiralConv (kind 0; a row-select D is already folded into `table`) or a sparse
 * re-sampling (kind 1).  Layouts: 0 = vertex-major [rows][B][C] (all internal tensors), 1 = batch-major
 * [B][rows][C] (allowed for the stack input and the stack output only).
 */
typedef struct sh_csr_ref { const int32_t* rowptr; const int32_t* col; const float* val; } sh_csr_ref;
typedef struct sh_stack_step {
    int kind;                         /* 0 conv, 1 spmm */
    int param;                        /* conv: index into the weights / biases / dW / dbias arrays */
    /* conv */
    const int32_t* table;             /* [R][S] */
    const int32_t* table_t;           /* [n_in][S] transposed table (see sh_spiral_conv_bwd_data) */
    int R, S, n_in, cin, cout, act, zero_row;
    int n1, n2;                       /* extra rows of the pre-activation gradient buffer: list pre-sums, levels 1 and 2 */
    sh_csr_ref sum1, sum2;            /* their unit-valued CSR (n1 / n2 rows) */
    /* spmm */
    sh_csr_ref m, mt;                 /* y = M x and its transpose */
    int m_rows, m_cols;
    /* extend != 0: the step follows a conv whose output buffer has room for m_rows more rows behind its m_cols real ones;
     * M (m_rows x m_cols) holds only the NON-identity rows of the up-sampling and the forward pass appends M x there: the
     * next step reads Z = [x ; M x] (m_cols + m_rows rows) through a table composed with the row map, outs[i] must equal
     * outs[i-1], and mt is the transpose of [I ; M] (m_cols x (m_cols + m_rows)), so the backward pass is the plain one. */
    int extend;
    /* conv, optional (round 6; NULL / 0 = none): the step's backward-data sources as RAGGED lists instead of table_t + pre-summed
     * rows - rag_rows [n_in][rag_L]: for input row u its sources (rows of the pre-activation gradient, all < R), rag_pos [n_in][rag_L]:
     * the spiral position whose weight multiplies each, -1 behind a row's last source (rag_rows then holds any valid row).  A step
     * that has them and whose backward-data pass runs on planes with a resident weight (sh_spiral_conv_p3_rag_ok) takes
     * sh_spiral_conv_bwd_data_p3_rag and needs neither its pre-sum launches nor the extra rows filled. */
    const int32_t* rag_rows;
    const int32_t* rag_pos;
    int rag_L;
    /* conv, optional (round 6; NULL / 0 = none): GROUPED lists - output rows whose source lists overlap share one list of the union
     * (mesh_ops.group_lists; sh_spiral_conv_p3_grp).  fg_*: the forward pass (groups of rows of the step's output, sources = rows of
     * its input, built from `table`); bg_*: the backward-data pass (groups of rows of the step's input, sources = rows of the
     * pre-activation gradient, built from the ragged lists).  x_rows [n][L] rows of the gathered tensor; x_pos [n][L] one byte per
     * member: the spiral position that member reads the row at, 0xFF = it does not (0xFFFFFFFF behind the group's last entry);
     * x_out [n][4] the members' output rows (-1: none; at most sh_spiral_conv_p3_grp_members() of them).  A step that has them and
     * whose pass runs on planes with a resident weight (sh_spiral_conv_p3_grp_ok) takes the grouped kernel: every row of a group's
     * union is gathered once (SH_P3_GROUPED=0: the one-row lists / the table). */
    const int32_t* fg_rows; const uint32_t* fg_pos; const int32_t* fg_out; int fg_n, fg_L;
    const int32_t* bg_rows; const uint32_t* bg_pos; const int32_t* bg_out; int bg_n, bg_L;
} sh_stack_step;

/* outs[i]: output of step i, vertex-major, except outs[n_steps-1] which has layout out_layout.
 * x: [rows0] rows of c0 channels in layout x_layout. */
/* Three-plane form (mma_mode == SH_MMA_PLANES3; ignored otherwise, may be NULL): planes[i] = buffer for the plane image of
 * outs[i] (sh_p3_bytes of the buffer's rows - a step that appends shares its predecessor's buffer AND image - or NULL: the
 * step that gathers it keeps the SPLIT3 kernels); wfrag3[i] = three-plane weight fragments of conv step i, forward operand
 * (sh_conv_wfrag3_prep_multi, transpose 0), already converted from the CURRENT weights.  A conv step whose input has an
 * image and whose shape sh_spiral_conv_p3_ok() takes runs sh_spiral_conv_fwd_p3; images are written by their producers.
 * keep_fp32: 1 = every step writes its fp32 output (a backward pass reads them); 0 = forward only: rows that the next plane
 * conv gathers through their image alone are written as the image alone (outs[i] of such a step is then partly or wholly
 * unwritten; the last step's output is always fp32); 2 (round 6) = training on the images: the same rows, where the BACKWARD
 * pass of the conv that gathers them leaves their fp32 form unread as well - its weight gradient runs on the two images
 * (sh_spiral_conv_bwd_wgt_p3) and the activation derivative is evaluated from the image.  What is known of that from the steps
 * alone decides (shapes, tables, the switches SH_P3_BWD / SH_P3_WGRAD / SH_P3_YPREV_IMG / SH_P3_DROP_FP32); the caller of 2 owes
 * sh_stack_backward the plane buffers (gin_planes, wfrag3_t, in_planes, plane-sized workspaces) and acts_fp32 == 2. */
SH_API int sh_stack_forward(int n_steps, const sh_stack_step* steps, const float* x, int x_layout, int rows0, int c0, int B,
                            const float* const* weights, const float* const* biases, float* const* outs, int out_layout,
                            int mma_mode, void* const* planes, const void* const* wfrag3, int keep_fp32, sh_stream_t stream);

/* acts[i]: what sh_stack_forward wrote to outs[i].  g: gradient w.r.t. the stack output (out_layout).
 * gin[i]: gradient w.r.t. the INPUT of step i - for i >= 1 vertex-major with (n1 + n2 of step i-1, if that is a
 * conv) extra rows behind the real ones, for i == 0 layout x_layout, NULL when need_x_grad == 0; buffers of
 * non-adjacent steps may alias (gin[i] is dead once gin[i-1] has been produced).  dpre_last: as gin[] for the
 * output of the last step when that is a conv ([R + n1 + n2][B][cout]), else unused.  Per conv step i:
 * weight_t[i] ([cin][S*cout], needed when i > 0 or need_x_grad), workspace[i] / workspace_bytes[i]
 * (>= sh_spiral_conv_bwd_wgt_workspace); per parameter index: dW[p], dbias[p] (may be NULL).
 * Three-plane form (SH_MMA_PLANES3; otherwise ignored, may be NULL): gin_planes[i] / dpre_last_planes = buffers for the plane
 * images of gin[i] / dpre_last (all rows, the pre-summed ones included; NULL = that conv step's backward-data pass keeps the
 * SPLIT3 kernels), wfrag3_t[i] = fragments of conv step i's backward-data operand (transpose 1); weight_t[i] may be NULL for
 * a step that runs sh_spiral_conv_bwd_data_p3.  in_planes (round 6; may be NULL): in_planes[i] = the plane image of the INPUT
 * of conv step i - what sh_stack_forward wrote to planes[i - 1], kept alive by the caller - or NULL; with it, the image of its
 * gradient rows and a workspace of at least sh_spiral_conv_bwd_wgt_p3_workspace() bytes, a step whose shape
 * sh_spiral_conv_bwd_wgt_p3_ok() takes computes its WEIGHT gradient from the two images (sh_spiral_conv_bwd_wgt_p3_presum)
 * instead of from the fp32 tensors (sh_spiral_conv_bwd_wgt_presum).  acts_fp32: the keep_fp32 sh_stack_forward ran with (1 or
 * 2).  With 2 a step whose input was left as its image alone must run on the images: SH_ERR_INVALID_ARG when the buffers given
 * do not allow it (never a read of unwritten rows); and the gradient rows this pass itself hands from step to step are written
 * as their image alone where the step that takes them reads nothing else (ragged source lists or a table without
 * multiplicities, plane weight gradient). */
SH_API int sh_stack_backward(int n_steps, const sh_stack_step* steps, const float* x, int x_layout, int rows0, int c0, int B,
                             const float* const* acts, const float* g, int out_layout, const float* const* weights,
                             float* const* gin, float* dpre_last, float* const* weight_t, void* const* workspace,
                             const size_t* workspace_bytes, float* const* dW, float* const* dbias, int need_x_grad,
                             int mma_mode, void* const* gin_planes, void* dpre_last_planes, const void* const* wfrag3_t,
                             const void* const* in_planes, int acts_fp32, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers with one tiny and one huge dimension: the latent nn.Linear pair fc_latent_enc /
 * fc_latent_dec (models.py:85-86, applied at :130 and :144) and their autograd.  Contiguous
 * row-major tensors; weight is nn.Linear.weight [N][K].  All three stream the weight-shaped matrix
 * once (HBM-bound); a reduction that is too long for one workgroup is split over workgroups into
 * partial slabs in `workspace` (>= sh_linear_workspace bytes) and summed in a fixed order.
 *   fwd       y[M,N]  = x[M,K] . weight^T + bias        (bias may be NULL)
 *   bwd_data  dx[M,K] = dy[M,N] . weight
 *   bwd_wgt   dW[N,K] = dy^T . x ;  dbias[N] = column sums of dy (dbias may be NULL)
 * mma_mode (enum sh_mma_mode, round 5): SH_MMA_EXACT = fp32 MFMA; SH_MMA_SPLIT3 / SH_MMA_PLANES3 = both fp32 operands split
 * exactly into three bf16 terms inside the kernel, six partial products on the bf16 MFMA, fp32 accumulation (batch <= 64 and the
 * streaming kernels' shapes; other shapes run the fp32 MFMA kernels whatever the mode).  Nothing is written in bf16.
 */
SH_API size_t sh_linear_workspace(int M, int N, int K);
SH_API int sh_linear_fwd(const float* x, const float* weight, const float* bias, float* y, int M, int N, int K,
                  void* workspace, size_t workspace_bytes, int mma_mode, sh_stream_t stream);
SH_API int sh_linear_bwd_data(const float* dy, const float* weight, float* dx, int M, int N, int K,
                       void* workspace, size_t workspace_bytes, int mma_mode, sh_stream_t stream);
SH_API int sh_linear_bwd_wgt(const float* dy, const float* x, float* dW, float* dbias, int M, int N, int K,
                      void* workspace, size_t workspace_bytes, int mma_mode, sh_stream_t stream);
/* Weight gradient of a latent FC with torch.optim.Adam's update applied to it in the same kernel (round 5): every 64 x 64 tile of
 * dW = dy^T . x is, instead of being stored, used as the gradient `g` of sh_adam_step's update of the same tile of `weight`,
 * `exp_avg` and `exp_avg_sq` (in place; coefficients from the DEVICE scalars `step` = updates applied so far and `lr`, so the
 * launch is replayable in a hipGraph).  Bit-identical to sh_linear_bwd_wgt followed by sh_adam_step on that tensor (one shared
 * update function), without writing and re-reading the gradient: 24 instead of 32 bytes of HBM traffic per weight.  `step` is
 * NOT advanced here (every workgroup reads it): the caller advances it once the launch is queued (sh_adam_bump, or a numel == 0 entry of sh_adam_step).  dbias as in
 * sh_linear_bwd_wgt (the bias keeps its ordinary gradient).  Served shapes: sh_linear_bwd_wgt_adam_ok (M <= 64, N and K
 * multiples of 64; SH_LIN_WGT_ADAM=0 makes it answer 0); others return SH_ERR_UNSUPPORTED and nothing is launched.
 * Replaces, for one parameter, the pair reference `loss.backward()` (models.py:130 / :144 autograd of nn.Linear) +
 * `optimizer.step()` (train_funcs.py:391-392, 509-510).  Only valid when nothing else consumes that gradient between backward
 * and step (no all-reduce, clipping or accumulation over several backward passes).
 * dy / x carry an element type (enum sh_dtype, declared below): bf16 operands (the bf16 path's layer, sh_linear_bwd_wgt_bf16) are
 * widened to fp32 in the kernel and multiplied on the fp32 MFMA - exact products of bf16 values, fp32 accumulation; `weight_bf16`,
 * if not NULL, is the bf16 working copy of the weight and is rewritten with the update (as sh_adam_step_bf16 does). */
SH_API int sh_linear_bwd_wgt_adam_ok(int M, int N, int K);
SH_API int sh_linear_bwd_wgt_adam(const void* dy, int dy_dtype, const void* x, int x_dtype, float* weight, void* weight_bf16, float* exp_avg,
                           float* exp_avg_sq, const float* step, const float* lr, double beta1, double beta2, double eps,
                           double weight_decay, float* dbias, int M, int N, int K, int mma_mode, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Grouped (ragged) dense layers: the 3 x 17 per-part nn.Linear layers of SpiralAutoencoder_multiz_partkps
 * (models.py:200-204, applied one at a time at :236, :252, :269) in ONE launch per direction.  All groups read their
 * inputs from, and write their outputs to, column ranges of two row-major matrices:
 *   fwd       y[m][y_off[g] + n]  = sum_k x[m][x_off[g] + k] * W_g[n][k] + bias_g[n]            n < N[g], k < K[g]
 *   bwd_data  dx[m][x_off[g] + k] = sum_n dy[m][y_off[g] + n] * W_g[n][k]
 *   bwd_wgt   dW_g[n][k] = sum_m dy[m][y_off[g] + n] * x[m][x_off[g] + k];   dbias_g[n] = sum_m dy[m][y_off[g] + n]
 * x / dx have row stride x_rs, y / dy row stride y_rs (elements).  w, bias, dW, dbias, x_off, y_off, N, K are HOST
 * arrays of G entries (device pointers / element offsets / sizes); bias, dbias and their entries may be NULL.
 * W_g is [N[g]][K[g]] row-major (nn.Linear.weight).  Deterministic; results overwrite their outputs.
 */
SH_API int sh_grouped_linear_fwd(int G, const float* x, int64_t x_rs, const int64_t* x_off, const float* const* w,
                          const float* const* bias, float* y, int64_t y_rs, const int64_t* y_off, int M,
                          const int* N, const int* K, sh_stream_t stream);
SH_API int sh_grouped_linear_bwd_data(int G, const float* dy, int64_t y_rs, const int64_t* y_off, const float* const* w,
                               float* dx, int64_t x_rs, const int64_t* x_off, int M, const int* N, const int* K,
                               sh_stream_t stream);
SH_API int sh_grouped_linear_bwd_wgt(int G, const float* dy, int64_t y_rs, const int64_t* y_off, const float* x,
                              int64_t x_rs, const int64_t* x_off, float* const* dW, float* const* dbias, int M,
                              const int* N, const int* K, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Losses / metric.  All reductions are two-stage with a fixed order (no atomics).
 * workspace: at least sh_reduce_workspace() bytes.
 */
SH_API size_t sh_reduce_workspace(void);

/* loss[0] = mean |a - b| over n elements           (F.l1_loss, train_funcs.py:501) */
SH_API int sh_l1_loss_fwd(const float* a, const float* b, int64_t n, float* loss, void* workspace, sh_stream_t stream);
/* grad_b[i] = sign(b[i]-a[i]) * gscale[0] / n      (gscale: device scalar = upstream gradient) */
SH_API int sh_l1_loss_bwd(const float* a, const float* b, int64_t n, const float* gscale, float* grad_b, sh_stream_t stream);

/* out[0] = mean_{b<B, v<N} | scale * (a[b,v,:3] - b[b,v,:3]) |_2  for contiguous [B][N1][3] tensors,
 * rows v >= N (the dummy row) excluded                           (test_funcs.py:41-49) */
SH_API int sh_vertex_l2(const float* a, const float* b, int B, int N1, int N, float scale, float* out,
                 void* workspace, sh_stream_t stream);

/* Edge-length-ratio regulariser (train_funcs.py:12-39, 503-508), batched on device:
 *   loss[0] = mean_{b,f} sum_{e in 3 edges} | |e_rec| / (|e_gt| + 1e-5) - 1 |
 * x_hat, x: contiguous [B][N1][3]; faces int32 [F][3].
 * bwd: corner lists (vptr [N1+1], vcorner [3F], entries f*3+k) give an atomic-free gradient
 *   grad[b,v,:] = gscale[0]/(B*F) * sum_{corners of v} d score / d x_hat[b,v,:]
 * STATED DEVIATION from the reference on one edge case (here and in sh_recon_loss_bwd): a RECONSTRUCTED edge of length exactly 0.
 * The reference differentiates torch.sqrt(torch.sum(d ** 2)) (train_funcs.py:36-38) at d = 0: autograd multiplies the sqrt's
 * infinite derivative by 2 d = 0 and every gradient of that batch entry downstream becomes NaN - the optimizer step then turns
 * the whole model into NaN.  These kernels return the term's subgradient 0 for such an edge (`if (len > 0)`), so a collapsed
 * face contributes its loss value (|0 / t - 1| = 1) and no gradient, and training continues.  Everywhere else (len > 0) the
 * gradient is the reference's.  tests/test_gpu_parity.py::test_zero_length_reconstructed_edge pins both behaviours. */
SH_API int sh_edge_ratio_loss_fwd(const float* x_hat, const float* x, const int32_t* faces, int B, int N1, int F,
                           float* loss, void* workspace, sh_stream_t stream);
SH_API int sh_edge_ratio_loss_bwd(const float* x_hat, const float* x, const int32_t* faces,
                           const int32_t* vptr, const int32_t* vcorner, int B, int N1, int F,
                           const float* gscale, float* grad, sh_stream_t stream);

/* The reconstruction loss of the plain training loop in one piece (train_funcs.py:501-508):
 *   total[0] = parts[0] + edge_w * parts[1],  parts[0] = mean |x - x_hat| (all N1 rows),  parts[1] = the edge-ratio term above;
 * bwd writes grad = gscale[0] * d total / d x_hat.  Three launches instead of eleven; same arithmetic as the separate
 * kernels.  workspace: sh_recon_loss_workspace() bytes.
 * bwd takes the vertex -> neighbour lists instead of the corner lists: vptr [N1+1] counts the corners of every vertex (as
 * above), vnbr int32 [2 * 3F] holds, corner by corner in that order, the two other vertices of the corner's face
 * (faces[f][(k+1)%3], faces[f][(k+2)%3]) - the same sum in the same order without the corner -> face indirection. */
SH_API size_t sh_recon_loss_workspace(void);
SH_API int sh_recon_loss_fwd(const float* x_hat, const float* x, const int32_t* faces, int B, int N1, int F, float edge_w,
                      float* total, float* parts, void* workspace, sh_stream_t stream);
SH_API int sh_recon_loss_bwd(const float* x_hat, const float* x, const int32_t* vptr, const int32_t* vnbr, int B, int N1, int F,
                      float edge_w, const float* gscale, float* grad, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Part-wise pairwise-distance loss of the semantic training loop (train_funcs.py:243-284 and
 * :353-389; utils_distance.calc_euclidean_dist_matrix :366-376; utils_SH.angle_skl :442-478).
 * For every part p, batch entry b and ordered vertex pair (i,j), i != j, of the part:
 *   De = |g_i-g_j| * scale[b,p];  De_r = |r_i-r_j|;  w from the angle between (g_i-g_j) and bone[b,p]
 *   (w_mode 0 all_one, 1 angle/90, 2 sin(angle), 3 angle/90 thresholded at w_threshold; flags[p]&1
 *   forces w = 1); pairs with w*De == 0 are dropped;
 *   term = relat ? |w*De_r/De - w| : |w*De_r - w*De|;   loss = sum_p w_part[p] * mean_{kept pairs} term.
 * x_rec, x_gt: contiguous [B][N1][3]; bone [B][P][3]; scale [B][P] or NULL (= 1); parts as CSR
 * (part_ptr [P+1], part_vert), disjoint; tile_ptr [P+1] = cumulative ceil(n_p / sh_part_pairdist_tile_rows());
 * T = tile_ptr[P]; max_part = largest n_p.  workspace >= B*T*2 floats.  part_sum / part_cnt [P] are
 * outputs of fwd; bwd takes part_cnt, a device scalar gscale, and OVERWRITES grad [B][N1][3]
 * (gradient w.r.t. x_rec; vertices outside every part get 0).  Deterministic (no atomics).
 */
SH_API int sh_part_pairdist_tile_rows(void);
SH_API int sh_part_pairdist_loss_fwd(const float* x_rec, const float* x_gt, const float* bone, const float* scale,
                              const int32_t* part_ptr, const int32_t* part_vert, const int32_t* tile_ptr,
                              const int32_t* flags, const float* w_part, int B, int N1, int P, int T, int max_part,
                              int w_mode, float w_threshold, int relat, float* loss, float* part_sum, float* part_cnt,
                              void* workspace, size_t workspace_bytes, sh_stream_t stream);
SH_API int sh_part_pairdist_loss_bwd(const float* x_rec, const float* x_gt, const float* bone, const float* scale,
                              const int32_t* part_ptr, const int32_t* part_vert, const int32_t* tile_ptr,
                              const int32_t* flags, const float* w_part, int B, int N1, int P, int T, int max_part,
                              int w_mode, float w_threshold, int relat, const float* part_cnt, const float* gscale,
                              float* grad, sh_stream_t stream);
/* The forward pass that also leaves the backward pass's row sums: grad_raw [B][N1][3] (rows of part vertices written, the
 * rest untouched) = pairdist's gradient without the factor 2 gscale w_p / count_p, which only exists once the counts are
 * complete; sh_part_pairdist_loss_bwd_scale applies it (grad: cleared, then the part vertices' rows written;
 * n_part_verts = part_ptr[P]) - the same bits as sh_part_pairdist_loss_bwd, without its second sweep over the pairs.
 * grad_raw == NULL: exactly sh_part_pairdist_loss_fwd. */
SH_API int sh_part_pairdist_loss_fwd_grad(const float* x_rec, const float* x_gt, const float* bone, const float* scale,
                                   const int32_t* part_ptr, const int32_t* part_vert, const int32_t* tile_ptr, const int32_t* flags,
                                   const float* w_part, int B, int N1, int P, int T, int max_part, int w_mode, float w_threshold,
                                   int relat, float* loss, float* part_sum, float* part_cnt, float* grad_raw, void* workspace,
                                   size_t workspace_bytes, sh_stream_t stream);
SH_API int sh_part_pairdist_loss_bwd_scale(const float* grad_raw, const int32_t* part_ptr, const int32_t* part_vert,
                                   const float* w_part, const float* part_cnt, const float* gscale, int B, int N1, int P,
                                   int n_part_verts, float* grad, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Body measurements (utils_SH.py:86-98 cal_length, :144-161 measure_body_quick; numpy twins
 * obj2npy.py:61-79), batched over meshes.
 * Girth of ring p = length of the CLOSED polyline through its n_p points
 *   q_i = v[ring_a[i]] * (1 - ring_f[i]) + v[ring_b[i]] * ring_f[i],   i in [ring_ptr[p], ring_ptr[p+1])
 * (utils_SH.py:155-158: closing segment q_0-q_last plus the n_p-1 consecutive ones).
 * v: [B] meshes of [*][3] floats, batch stride v_sb floats; girth [B][P].
 * Bone length p = | k[bones[3p]] - k[bones[3p+1]] |, or, when bones[3p+2] >= 0,
 *   | k[bones[3p]] - (k[bones[3p+1]] + k[bones[3p+2]]) / 2 |   (utils_SH.py:94-97);
 * kps contiguous [B][K][3]; bones int32 [P][3]; length [B][P].
 */
SH_API int sh_measure_girth(const float* v, int64_t v_sb, const int32_t* ring_ptr, const int32_t* ring_a,
                     const int32_t* ring_b, const float* ring_f, int B, int P, float* girth, sh_stream_t stream);
SH_API int sh_bone_length(const float* kps, const int32_t* bones, int B, int K, int P, float* length, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GPU-resident dataset (autoencoder_dataset.py:26-58; main.py:209-237).  The reference loads and
 * normalises one .npy per sample in DataLoader worker processes; here the packed split is
 * normalised once on device and every batch is a row gather from the resident tensor.
 *
 * sh_dataset_normalize: raw contiguous [n][N][3] -> out contiguous [n][N+dummy_rows][3] (dummy rows
 * zero, :45-48), applying in the reference's order (:29-43) the steps selected by `flags`:
 *   ZEROMEAN  v -= mean_v(v)                      ZEROROOT  v -= sum_v j_root[v] * v   (J_regressor row 0)
 *   ONELENGTH v = v / (max_y - min_y) * 1.5       SMALL     v = v / 1.5
 *   GASS      v = (v - mean[N][3]) / stdv[N][3]   NORMAL    v = (v - center[n][3]) * scale[n][3]
 * then NaN -> 0.  Unused table pointers may be NULL.
 * sh_gather_meshes: out[j][:] = src[idx[j]][:] for j < b, rows of row_elems floats, idx int64 on
 * device (bit-exact copy; the caller guarantees 0 <= idx[j] < rows of src).
 */
enum sh_norm_flags {
    SH_NORM_ZEROMEAN = 1, SH_NORM_ZEROROOT = 2, SH_NORM_ONELENGTH = 4, SH_NORM_SMALL = 8, SH_NORM_GASS = 16, SH_NORM_NORMAL = 32
};
SH_API int sh_dataset_normalize(const float* raw, float* out, int n, int N, int dummy_rows, unsigned flags,
                         const float* j_root, const float* mean, const float* stdv, const float* center,
                         const float* scale, sh_stream_t stream);
SH_API int sh_gather_meshes(const float* src, int64_t row_elems, const int64_t* idx, int b, float* out, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser: torch.optim.Adam with coupled L2 weight decay (main.py:262; steps at
 * train_funcs.py:391-392, 509-510), multi-tensor.  For every tensor i < n_tensors, with t = steps[i][0] + 1:
 *   g' = g + weight_decay * p;  m = lerp(m, g', 1 - beta1);  v = beta2 * v + (1 - beta2) * g'^2;
 *   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps);   then steps[i][0] += 1.
 * params / grads / exp_avg / exp_avg_sq / steps / numel are HOST arrays of length n_tensors holding
 * DEVICE pointers (steps[i]: one float, the number of updates applied so far) and element counts;
 * `lr` is a DEVICE scalar, so a captured launch follows learning-rate schedules and step counts
 * without re-capture.  An entry with numel[i] == 0 only advances steps[i] (its other pointers are not read): a parameter whose
 * update was applied by sh_linear_bwd_wgt_adam during backward.
 */
SH_API int sh_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, float* const* steps, const int64_t* numel, const float* lr,
                 double beta1, double beta2, double eps, double weight_decay, sh_stream_t stream);
/* steps[i][0] += 1 for i < n_tensors (HOST array of DEVICE pointers): the step counts of parameters whose update was applied by
 * sh_linear_bwd_wgt_adam during backward. */
SH_API int sh_adam_bump(int n_tensors, float* const* steps, sh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The small loss terms of the semantic loop as kernels.  x tensors are [B] meshes of rows of 3 floats,
 * batch stride x_bs floats (the dummy row may be present: faces / J never index it).  Deterministic (no atomics).
 *  - joint regression kps[b][j][:] = sum_v J[j][v] x[b][v][:] (train_funcs.py:131,161,230,296,336), J dense [K][N];
 *    sh_joint_l1_loss_fwd also returns loss[0] = mean |kps[:, keep] - target| (F.l1_loss, :231,:342; keep int32 [Kk],
 *    target [B][Kk][3]); _bwd OVERWRITES grad [B][N1][3] (gradient w.r.t. x; rows >= N get 0) from the saved kps.
 *  - part volume ratio (cal_volloss :56-71 averaged over the batch :323-330): parts as CSR over faces (pf_ptr [P+1], pf),
 *    vol [2][B][P] receives the signed volumes of (x_rec, x_gt); loss[0] = (1/P) sum_p mean_b | |vr/vg| - 1 |.
 *    _bwd: face_slot [F] = position of the face's part in the list or -1; vptr / vcorner as for the edge loss.
 *  - sh_zpart_reg (zpartreg :145-152): z [B][P][L], measure [B][M], part_idx / measure_idx int32 [n];
 *    loss[0] = mean | |z_p| / m - 1 | (relat) or | |z_p| - m |; with dz != NULL also writes gscale[0] * d loss / d z
 *    (loss may then be NULL). */
SH_API int sh_joint_regress(const float* x, int64_t x_bs, const float* J, int B, int N, int K, float* kps, sh_stream_t stream);
SH_API int sh_joint_l1_loss_fwd(const float* x, int64_t x_bs, const float* J, const int32_t* keep, const float* target, int B,
                                int N, int K, int Kk, float* kps, float* loss, sh_stream_t stream);
SH_API int sh_joint_l1_loss_bwd(const float* kps, const int32_t* keep, const float* target, const float* J, int B, int N1, int N,
                                int K, int Kk, const float* gscale, float* grad, sh_stream_t stream);
SH_API int sh_part_volume_loss_fwd(const float* x_rec, const float* x_gt, int64_t x_bs, const int32_t* faces,
                                   const int32_t* pf_ptr, const int32_t* pf, int B, int P, float* vol, float* loss,
                                   sh_stream_t stream);
SH_API int sh_part_volume_loss_bwd(const float* x_rec, int64_t x_bs, const int32_t* faces, const int32_t* face_slot,
                                   const int32_t* vptr, const int32_t* vcorner, const float* vol, int B, int P, int N1,
                                   const float* gscale, float* grad, sh_stream_t stream);
SH_API int sh_zpart_reg(const float* z, const float* measure, const int32_t* part_idx, const int32_t* measure_idx, int B, int P,
                        int L, int M, int n, int relat, float* loss, float* dz, const float* gscale, sh_stream_t stream);
/*  - joints <-> bones (utils_SH.py:26-84; inputs of the loop, no gradient).  sh_kps2skl: kps [B][J][3], bone k = joint i0[k] -
 *    (joint i1[k] + joint i2[k]) / 2 (i2 == i1 for two-joint bones), n = |bone|; mode 0: out [B][n_bones][4] = (bone / n, n),
 *    1: (bone, n), 2: [..][3] = bone (the pair loss's bone directions, :449-452), 3: [..][1] = n.  sh_skl2kps: skl
 *    [B][n_bones][4] (mode 0: direction * length; 1: first three) or [..][3] (mode 2); joints rebuilt in list order, joint
 *    tail[k] = joint head[k] - bone k with unassigned joints at the origin (n_joints <= 64), out [B][n_keep][3] = joints keep[].
 *    Operation for operation the arithmetic of the tensor-op forms (same bits).
 *  - sh_weighted_sum: out[0] = w0 t0 + w1 t1 + ... summed in sequence (terms: HOST array of n <= 16 DEVICE scalars, weights:
 *    HOST floats; a weight of exactly 1 leaves its term unscaled) when out != NULL; grads[i] = gscale[0] * w_i when grads !=
 *    NULL - the loop's `loss = loss + w * term` chain and its backward as one launch each. */
SH_API int sh_kps2skl(const float* kps, int B, int J, const int32_t* i0, const int32_t* i1, const int32_t* i2, int n_bones, int mode,
                      float* out, sh_stream_t stream);
SH_API int sh_skl2kps(const float* skl, int B, int n_bones, int mode, const int32_t* head, const int32_t* tail, int n_joints,
                      const int32_t* keep, int n_keep, float* out, sh_stream_t stream);
SH_API int sh_weighted_sum(int n, const float* const* terms, const float* weights, float* out, const float* gscale, float* grads,
                           sh_stream_t stream);

/* =============================================================================================
 * bf16 compute path (BASELINE.json configs[2]: "batch=512 bf16, DDP 8x"; the reference itself is fp32-only,
 * models.py:45).  Same operators as above with bf16 activations and bf16 working copies of the weights,
 * fp32 accumulation inside the kernels (v_mfma_f32_16x16x32_bf16), fp32 bias / activation arithmetic, and
 * fp32 master weights, gradients and Adam state outside them.  Tensors carry an explicit element type:
 */
enum sh_dtype { SH_DTYPE_F32 = 0, SH_DTYPE_BF16 = 1 };

/* Working copy of a SpiralConv weight: bf16, pre-ordered into the 1-KiB MFMA fragments the conv kernel keeps in
 * LDS (frag[k-step][16-channel tile][lane][8]; layout in csrc/sh_bf16.h), zero-padded.  `transpose` = 0 gives the
 * forward operand (Cg = Cin gathered channels, Nout = Cout), 1 the backward-data operand (Cg = Cout, Nout = Cin)
 * of the same fp32 master weight [Cout][S*Cin] (models.py:17).  sh_conv_wfrag_bytes(S, Cg, Nout) sizes the buffer
 * (16-byte aligned).  One launch converts all layers of a stack. */
SH_API size_t sh_conv_wfrag_bytes(int S, int Cg, int Nout);
SH_API int sh_conv_wfrag_prep_multi(int n_layers, const float* const* weight, void* const* wfrag, const int* S,
                                    const int* Cin, const int* Cout, const int* transpose, sh_stream_t stream);

/* sh_spiral_conv_fwd / sh_spiral_conv_bwd_data (models.py:40-51 and its autograd) in bf16.  Strides are in ELEMENTS
 * of the tensor's own type.  x / dpre: bf16 with 16 or a multiple of 32 channels, or fp32 with exactly 3 channels
 * (the xyz input of the first layer, the xyz gradient entering the last one); y / dx: bf16, or fp32 when it has
 * <= 16 channels (x_hat, dL/dx).  bias fp32; yprev bf16.  Other shapes: SH_ERR_UNSUPPORTED (use the fp32 entry
 * points, which take any shape). */
SH_API int sh_spiral_conv_fwd_bf16(const void* x, int x_dtype, int64_t x_sv, int64_t x_sb, const int32_t* table,
                                   const void* wfrag, const float* bias, void* y, int y_dtype, int64_t y_sv, int64_t y_sb,
                                   int B, int R, int S, int Cin, int Cout, int act, int zero_row, sh_stream_t stream);
SH_API int sh_spiral_conv_bwd_data_bf16(const void* dpre, int dp_dtype, int64_t dp_sv, int64_t dp_sb,
                                        const int32_t* table_t, const void* wfrag_t, void* dx, int dx_dtype,
                                        int64_t dx_sv, int64_t dx_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb,
                                        int act_prev, int zero_row, int B, int n_in, int S, int Cin, int Cout,
                                        sh_stream_t stream);

/* Weight gradient of a 16 -> 3 channel layer (the decoder's last conv, models.py:146-153) in role-swapped form:
 *   dW[co][s][ci] = sum_{u,b} x[u,b,ci] * dpre_ext[table_t[u,s], b, co]
 * x is read once and the three-channel gradient is gathered through the TRANSPOSED table - the table and the extended
 * gradient buffer (rows of irregular vertices pre-summed behind the R real rows) that sh_spiral_conv_bwd_data[_bf16] takes,
 * so the caller runs the pre-sum launches first.  dpre_ext fp32 [rows][B][3], x [n_in][B][16] of the path's dtype, both
 * vertex-major and contiguous; needs R == n_in, B % 16 == 0 and S <= 10 (fp32 path: S <= 30, run as two or three launches over
 * shares of the positions) - sh_spiral_conv_bwd_wgt_thin_ok() tells.  Writes
 * partial slabs into `workspace` in the layout and count of sh_spiral_conv_bwd_wgt (path_dtype SH_DTYPE_F32) or
 * sh_spiral_conv_bwd_wgt_bf16 (SH_DTYPE_BF16): the matching ..._reduce_multi launch finishes dW and dbias.
 * dx != NULL: the same launch also writes the layer's backward-data (what sh_spiral_conv_bwd_data[_bf16] computes from the
 * same table and buffer): dx[u,b,:] = act_prev'(x[u,b,:]) * sum_s dpre_ext[table_t[u,s],b,:] . W[:, s, :], row zero_prev
 * forced to zero; dx [n_in (+ extra)][B][16] vertex-major of the path's dtype, weight = the fp32 master [3][S*16] (rounded to
 * bf16 in the kernel on the bf16 path, like the fragment copies); x is both the layer input and the activation output
 * whose derivative multiplies (act_prev = SH_ACT_IDENTITY: no factor).  dx_planes != NULL (fp32 path, with dx): the three-plane
 * image of dx's n_in rows (see the ..._p3 entry points) is written with them. */
SH_API int sh_spiral_conv_bwd_wgt_thin_ok(int B, int n_in, int S, int Cin, int Cout, int path_dtype);
SH_API int sh_spiral_conv_bwd_wgt_thin(const float* dpre_ext, int64_t dp_sv, int64_t dp_sb, const void* x, int x_dtype, int64_t x_sv,
                                int64_t x_sb, const int32_t* table_t, void* workspace, size_t workspace_bytes,
                                const float* weight, void* dx, int64_t dx_sv, int64_t dx_sb, void* dx_planes, int act_prev,
                                int zero_prev, int B, int R, int n_in, int S, int Cin, int Cout, int path_dtype, sh_stream_t stream);

/* sh_spiral_conv_bwd_wgt in bf16: x / dpre bf16 (channels % 8 == 0) or fp32 with exactly 3 channels; writes fp32 partial
 * slabs into `workspace` (>= sh_spiral_conv_bwd_wgt_workspace_bf16 bytes); sh_spiral_conv_bwd_wgt_reduce_multi_bf16 sums
 * the slabs of up to 16 layers into fp32 dW / dbias (same contract as the fp32 pair above). */
SH_API size_t sh_spiral_conv_bwd_wgt_workspace_bf16(int B, int R, int S, int Cin, int Cout);
SH_API int sh_spiral_conv_bwd_wgt_bf16(const void* dpre, int dp_dtype, int64_t dp_sv, int64_t dp_sb, const void* x, int x_dtype,
                                       int64_t x_sv, int64_t x_sb, const int32_t* table, void* workspace,
                                       size_t workspace_bytes, int B, int R, int S, int Cin, int Cout, sh_stream_t stream);
SH_API int sh_spiral_conv_bwd_wgt_reduce_multi_bf16(int n_layers, const void* const* workspaces, float* const* dW,
                                                    float* const* dbias, const int* B, const int* R, const int* S,
                                                    const int* Cin, const int* Cout, sh_stream_t stream);

/* The latent dense layers (models.py:85-86,130,144) in bf16: `weight_bf16` is a bf16 working copy [N][K] of the fp32
 * master weight (sh_cast_f32_to_bf16, or written by sh_adam_step_bf16 as it updates the master); x / dy / y / dx are bf16
 * or fp32 (the latent code and its gradient stay fp32); bias, dW, dbias fp32.  N and K must be multiples of 8, M is free.
 * workspace >= sh_linear_workspace_bf16(M, N, K) bytes, 16-byte aligned (split-reduction partials). */
SH_API size_t sh_linear_workspace_bf16(int M, int N, int K);
SH_API int sh_cast_f32_to_bf16(const float* src, void* dst, int64_t n, sh_stream_t stream);
SH_API int sh_linear_fwd_bf16(const void* x, int x_dtype, const void* weight_bf16, const float* bias, void* y, int y_dtype,
                              int M, int N, int K, void* workspace, size_t workspace_bytes, sh_stream_t stream);
SH_API int sh_linear_bwd_data_bf16(const void* dy, int dy_dtype, const void* weight_bf16, void* dx, int dx_dtype, int M, int N,
                                   int K, void* workspace, size_t workspace_bytes, sh_stream_t stream);
SH_API int sh_linear_bwd_wgt_bf16(const void* dy, int dy_dtype, const void* x, int x_dtype, float* dW, float* dbias, int M,
                                  int N, int K, sh_stream_t stream);

/* sh_spmm / sh_act_backward on bf16 tensors (channels % 8 == 0, 16-byte aligned rows; strides in bf16 elements; CSR values
 * fp32; arithmetic fp32, one rounding of the result). */
SH_API int sh_spmm_bf16(const int32_t* rowptr, const int32_t* col, const float* val, const void* x, int64_t x_sv, int64_t x_sb,
                        void* y, int64_t y_sv, int64_t y_sb, const void* yprev, int64_t yp_sv, int64_t yp_sb, int act_prev,
                        int zero_row, int B, int rows, int C, sh_stream_t stream);
SH_API int sh_act_backward_bf16(const void* dy, int64_t dy_sv, int64_t dy_sb, const void* y, int64_t y_sv, int64_t y_sb,
                                void* dpre, int64_t dp_sv, int64_t dp_sb, int B, int R, int C, int act, int zero_row,
                                sh_stream_t stream);

/* sh_adam_step that also rewrites, for every tensor with shadow_bf16[i] != NULL, a bf16 working copy of the updated fp32
 * parameter (same element order): the next forward pass reads the working copy without a separate conversion pass. */
SH_API int sh_adam_step_bf16(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, float* const* steps, void* const* shadow_bf16, const int64_t* numel,
                             const float* lr, double beta1, double beta2, double eps, double weight_decay, sh_stream_t stream);

/* sh_stack_forward / sh_stack_backward for the bf16 path.  Tensors between steps are bf16 vertex-major; x is bf16, or fp32
 * with 3 channels; the output of the last step has type out_dtype (fp32 only for <= 16 channels), g the same type; gin[0]
 * has type gx_dtype; dpre_last has the type of the last step's output.  wfrag[i] / wfrag_t[i]: per conv step, buffers of
 * sh_conv_wfrag_bytes(S, cin, cout) / (S, cout, cin) bytes which the call fills from the fp32 master `weights` (one
 * conversion launch per pass) - unless wfrag_ready != 0: then they already hold the converted CURRENT weights (the caller
 * ran sh_conv_wfrag_prep_multi itself, e.g. once for both stacks and both orientations of a training step) and no
 * conversion is launched.  workspace[i] >= sh_spiral_conv_bwd_wgt_workspace_bf16.  Everything else as above. */
SH_API int sh_stack_forward_bf16(int n_steps, const sh_stack_step* steps, const void* x, int x_dtype, int x_layout, int rows0,
                                 int c0, int B, const float* const* weights, const float* const* biases, void* const* wfrag,
                                 int wfrag_ready, void* const* outs, int out_dtype, int out_layout, sh_stream_t stream);
SH_API int sh_stack_backward_bf16(int n_steps, const sh_stack_step* steps, const void* x, int x_dtype, int x_layout, int rows0,
                                  int c0, int B, const void* const* acts, const void* g, int out_dtype, int out_layout,
                                  const float* const* weights, void* const* gin, int gx_dtype, void* dpre_last,
                                  void* const* wfrag_t, int wfrag_ready, void* const* workspace,
                                  const size_t* workspace_bytes, float* const* dW, float* const* dbias, int need_x_grad,
                                  sh_stream_t stream);

/* =============================================================================================
 * Three-plane form of the fp32 path's matrix products (round 4).  Same operators, same fp32 tensors at every interface
 * as the fp32 entry points above (models.py:34-53 and its autograd); what changes is where the exact bf16x3 operand
 * split of SH_MMA_SPLIT3 happens: the PRODUCER of an activation / gradient writes, beside the fp32 tensor, its three
 * bf16 planes v = h + m + l (exact) once, and the conv kernels gather the planes and only multiply (six - or with
 * SH_P3_NP=9 all nine - partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
 *
 * Plane image of a tensor [rows][B][C], B % 16 == 0, C == 16 or C % 32 == 0 (sh_p3_bytes(rows, B, C) bytes, 0 =
 * unsupported shape; 16-byte aligned): fragment-major - the B-operand order of the matrix instruction, so a wave's gather
 * is one contiguous 1-KiB access (layout in csrc/p3_conv.hip).  sh_to_p3 converts rows of an fp32 tensor (strides as
 * everywhere: element (r, b, c) at x + r*sv + b*sb + c); the conv entry points can also write the image of their output
 * (yp / dxp != NULL; then Cout resp. Cin must be 16 or a multiple of 32) next to - or instead of (y / dx == NULL) - the
 * fp32 tensor.  Weights: sh_conv_wfrag3_prep_multi = sh_conv_wfrag_prep_multi with three planes per fragment
 * (sh_conv_wfrag3_bytes).  sh_spiral_conv_p3_ok(B, S, Cg, Nout): 1 when the kernels take a layer with Cg gathered and
 * Nout produced channels (its three-plane weight must fit LDS); otherwise the caller keeps the exact kernels. */
SH_API size_t sh_p3_bytes(int rows, int B, int C);
SH_API int sh_to_p3(const float* x, int64_t x_sv, int64_t x_sb, void* planes, int B, int rows, int C, sh_stream_t stream);
SH_API size_t sh_conv_wfrag3_bytes(int S, int Cg, int Nout);
SH_API int sh_conv_wfrag3_prep_multi(int n_layers, const float* const* weight, void* const* wfrag3, const int* S, const int* Cin,
                                     const int* Cout, const int* transpose, sh_stream_t stream);
SH_API int sh_spiral_conv_p3_ok(int B, int S, int Cg, int Nout);
/* 0: not taken; 1: LDS-resident weight (backward-data also takes rows without an image: dpre_f32 below); 2: streamed weight */
SH_API int sh_spiral_conv_p3_kind(int B, int S, int Cg, int Nout);
/* Diagnostics (like sh_profile_*; selects nothing): the number of plane-conv kernel launches (conv_p3 / conv_p3s, forward and
 * backward-data) this process has issued so far.  A caller that asked for SH_MMA_PLANES3 can tell whether the plane kernels
 * really ran or the SPLIT3 kernels served the call (batch not a multiple of 16, shapes sh_spiral_conv_p3_ok() refuses):
 * tests/conftest.py reports the [planes3] instance of a test that never moved this counter as skipped, not passed. */
SH_API int64_t sh_p3_launch_count(void);
/* sh_spmm that also writes the plane image of the rows it produces (y_planes: image of row 0 of y; NULL = plain sh_spmm);
 * y == NULL with y_planes: the image alone (y_sv / y_sb still describe the vertex-major tensor the image belongs to) */
SH_API int sh_spmm_p3(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t x_sv, int64_t x_sb,
                      float* y, int64_t y_sv, int64_t y_sb, void* y_planes, const float* yprev, int64_t yp_sv, int64_t yp_sb,
                      int act_prev, int zero_row, int B, int rows, int C, sh_stream_t stream);
/* sh_spiral_conv_fwd / sh_act_backward_tr that also write the plane image of what they store (y_planes / dpre_planes != NULL;
 * vertex-major result).  The conv writes it in its own epilogue where the kernel the dispatch picks has one, and issues
 * sh_to_p3 itself otherwise: the image is complete when the call returns either way. */
SH_API int sh_spiral_conv_fwd_img(const float* x, int64_t x_sv, int64_t x_sb, const int32_t* table, const float* weight,
                                  const float* bias, float* y, int64_t y_sv, int64_t y_sb, void* y_planes, int B, int R, int S, int Cin,
                                  int Cout, int act, int zero_row, int mma_mode, sh_stream_t stream);
SH_API int sh_act_backward_tr_img(const float* dy, int64_t dy_sv, int64_t dy_sb, const float* y, int64_t y_sv, int64_t y_sb, float* dpre,
                                  int64_t dp_sv, int64_t dp_sb, void* dpre_planes, int B, int R, int C, int act, int zero_row, int n_layers,
                                  const float* const* weight, float* const* weight_t, const int* S, const int* Cin, const int* Cout,
                                  sh_stream_t stream);
/* sh_spiral_conv_fwd with x given as its plane image xp ([n_in] rows).
 * Size limit of the gathered image (here and in sh_spiral_conv_bwd_data_p3): the kernels keep table entries pre-multiplied by
 * the image's row stride in 16-byte units in 32 bits, so the image must be smaller than 64 GiB (rows x B x C x 6 bytes); the
 * same holds for the fp32 tensor x of the streaming weight gradients (sh_spiral_conv_bwd_wgt*: rows x B x C x 4 bytes).  The
 * entry points do not know the gathered tensor's row count and cannot check it; sh_stack_forward / sh_stack_backward refuse
 * any tensor of 2^32 elements or more (SH_ERR_UNSUPPORTED), which is inside both limits. */
SH_API int sh_spiral_conv_fwd_p3(const void* xp, const int32_t* table, const void* wfrag3, const float* bias, float* y, int64_t y_sv,
                                 int64_t y_sb, void* yp, int B, int R, int S, int Cin, int Cout, int act, int zero_row,
                                 sh_stream_t stream);
/* sh_spiral_conv_bwd_data_z with dpre given as its plane image dprep (dpre_zero_row >= 0: the all-zero row the "no source"
 * entries point at - their products are skipped, bitwise the same result).  dpre_f32 != NULL: only rows < n_image_rows of dprep
 * are valid; the rows behind them (the pre-summed rows the transposed table refers to) are read from the fp32 tensor
 * (element strides dp_sv, dp_sb) and split by the kernel - their producers then need not write images.  yprev_planes (round 6;
 * may be NULL): the plane image of yprev ([n_in] rows x Cin channels) - the activation derivative is then evaluated from it
 * (h + m + l is the fp32 value, bit for bit) instead of from the fp32 tensor, which the caller need not keep cache-warm. */
SH_API int sh_spiral_conv_bwd_data_p3(const void* dprep, int dpre_zero_row, const float* dpre_f32, int64_t dp_sv, int64_t dp_sb,
                                      int n_image_rows, const int32_t* table_t, const void* wfrag3_t, float* dx, int64_t dx_sv,
                                      int64_t dx_sb, void* dxp, const float* yprev, int64_t yp_sv, int64_t yp_sb,
                                      const void* yprev_planes, int act_prev, int zero_row, int B, int n_in, int S, int Cin, int Cout,
                                      sh_stream_t stream);

/* sh_spiral_conv_bwd_data_p3 over RAGGED source lists (round 6; csrc/p3_conv.hip conv_p3r_kernel): dx[u] = act'(yprev[u]) x
 * sum_j dpre[rag_rows[u][j]] . W_{rag_pos[u][j]} over the entries with rag_pos >= 0 - no transposed table, no "no source" slots,
 * no pre-summed rows: the sums over several sources of one (row, position) are formed by the matrix pipe (linearity).  Every
 * source is a row of the image dprep.  sh_spiral_conv_p3_rag_ok: resident three-plane weight with at most four channel tiles per
 * workgroup, gathered channels (Cg = the layer's Cout) a multiple of 32, lists of at most 64 entries. */
SH_API int sh_spiral_conv_p3_rag_ok(int B, int S, int Cg, int Nout, int rag_L);
SH_API int sh_spiral_conv_bwd_data_p3_rag(const void* dprep, const int32_t* rag_rows, const int32_t* rag_pos, int rag_L, const void* wfrag3_t,
                                          float* dx, int64_t dx_sv, int64_t dx_sb, void* dxp, const float* yprev, int64_t yp_sv,
                                          int64_t yp_sb, const void* yprev_planes, int act_prev, int zero_row, int B, int n_in, int S,
                                          int Cin, int Cout, sh_stream_t stream);

/* Plane conv over GROUPED lists (csrc/p3_conv.hip conv_p3g_kernel, round 6): up to four output rows whose source lists overlap form a
 * group and share ONE list of the union of their sources - y[g_out[k][m]] = sum over the entries j of group k that member m reads
 * (byte m of g_pos[k][j] != 0xFF) of x[g_rows[k][j]] . W_{that byte} - so every row of the union is gathered once where the
 * one-row kernels (sh_spiral_conv_fwd_p3 through the table, sh_spiral_conv_bwd_data_p3_rag through ragged lists) gather it once
 * per member; matrix work and results' arithmetic are theirs (the order of a row's partial sums follows the group's list).
 * backward == 0: the forward pass (x = image of the layer input, wfrag3 = forward fragments, bias, activation `act`, masked
 * zero_row; Cg = the layer's Cin, Nout = its Cout, R = its output rows); backward == 1: backward-data (x = image of the
 * pre-activation gradient, wfrag3 = transposed fragments, the derivative of activation `act` at yprev / yprev_planes - the fp32
 * tensor or the image of the layer input - or neither; Cg = the layer's Cout, Nout = its Cin, R = its input rows).  y and / or yp
 * (the image of the result) are written.  .._grp_ok: resident three-plane weight with at most four channel tiles per workgroup,
 * Cg a multiple of 32, or 16 with at most two channel tiles (a k-step then spans two list entries), lists of at most 64 entries; .._grp_members: how many members per group the kernel of this shape takes (4 or 2;
 * 0 = shape not taken) - g_out always has four slots per group. */
SH_API int sh_spiral_conv_p3_grp_ok(int B, int S, int Cg, int Nout, int g_L);
SH_API int sh_spiral_conv_p3_grp_members(int B, int S, int Cg, int Nout);
/* 1 when the launch has enough groups x batch groups to fill the chip with its coarser work items (the sequencers' rule for taking
 * the grouped form; SH_P3_GRP_MIN_ITEMS_PER_CU, default 12). */
SH_API int sh_spiral_conv_p3_grp_pays(int B, int n_groups);
SH_API int sh_spiral_conv_p3_grp(const void* xp, const int32_t* g_rows, const uint32_t* g_pos, const int32_t* g_out, int n_groups, int g_L,
                                 const void* wfrag3, const float* bias, float* y, int64_t y_sv, int64_t y_sb, void* yp, const float* yprev,
                                 int64_t yp_sv, int64_t yp_sb, const void* yprev_planes, int act, int zero_row, int backward, int B, int R,
                                 int S, int Cg, int Nout, sh_stream_t stream);

/* The same over bf16 tensors (csrc/bf16_conv.hip conv_bf16r_kernel; the ragged sibling of sh_spiral_conv_bwd_data_bf16): dpre and dx
 * bf16 (element strides), wfrag_t the transposed bf16 fragments of sh_conv_wfrag_prep_multi, yprev bf16 or NULL.  The sums over
 * several sources of one (row, position) are formed in fp32 by the matrix pipe, where the dense form reads pre-summed rows that
 * were rounded to bf16 once more.  .._rag_ok: gathered channels (the layer's Cout) a multiple of 32, output channels a multiple
 * of 4, lists of at most 64 entries, a weight slice that fits LDS. */
SH_API int sh_spiral_conv_bf16_rag_ok(int B, int S, int Cg, int Nout, int rag_L);
SH_API int sh_spiral_conv_bwd_data_bf16_rag(const void* dpre, int64_t dp_sv, int64_t dp_sb, const int32_t* rag_rows, const int32_t* rag_pos,
                                            int rag_L, const void* wfrag_t, void* dx, int64_t dx_sv, int64_t dx_sb, const void* yprev,
                                            int64_t yp_sv, int64_t yp_sb, int act_prev, int zero_row, int B, int n_in, int S, int Cin,
                                            int Cout, sh_stream_t stream);

/* Weight gradient of a spiral conv in the three-plane form (csrc/wgrad_p3.hip, round 6; autograd of reference models.py:45,
 * dW = dpre^T . gather(x)): both operands given as their plane images - x_planes = image of the layer's input ([n_in] rows,
 * what its forward plane conv gathered), dpre_planes = image of the pre-activation gradient (rows [0, R) are read) - six
 * bf16 partial products per fp32 product, fp32 accumulation, the arithmetic of sh_spiral_conv_fwd_p3.  Writes partial slabs
 * (dW and dbias) into workspace, in the layout and for the deferred reduction of sh_spiral_conv_bwd_wgt(dW == NULL):
 * sh_spiral_conv_bwd_wgt_reduce_multi_kinds (kind 2).  .._p3_ok: 1 when the kernel takes the shape (batch % 16 == 0, Cin 16 or a
 * multiple of 32, Cout a multiple of 32); otherwise the caller keeps sh_spiral_conv_bwd_wgt.  dpre_zero_row: a row of dpre that
 * is all zero (the layer's dummy row), or -1 - needed only when R * (B / 16) is odd: the kernel sums pairs of 16-row units and
 * completes an odd count with that row's (SH_ERR_UNSUPPORTED without one). */
SH_API int sh_spiral_conv_bwd_wgt_p3_ok(int B, int R, int S, int Cin, int Cout);
SH_API size_t sh_spiral_conv_bwd_wgt_p3_workspace(int B, int R, int S, int Cin, int Cout);
SH_API int sh_spiral_conv_bwd_wgt_p3(const void* dpre_planes, int dpre_zero_row, const void* x_planes, const int32_t* table, void* workspace,
                                     size_t workspace_bytes, int B, int R, int S, int Cin, int Cout, sh_stream_t stream);
/* The same launch also carrying a pre-sum job, as sh_spiral_conv_bwd_wgt_presum does for the fp32 kernels: sum_out[r] = sum_e
 * sum_val[e] * dpre[sum_col[e]] for r < sum_rows over the FP32 gradient rows dpre (element strides dp_sv, dp_sb; sum_out has the
 * same strides; sum_out_planes != NULL: also the image of those rows) - sh_spmm's sums bit for bit, run by tail workgroups of
 * the launch (or, for rows that do not take 16-byte accesses, by sh_spmm_p3 in front of it).  sum_rows == 0: no job. */
SH_API int sh_spiral_conv_bwd_wgt_p3_presum(const void* dpre_planes, int dpre_zero_row, const void* x_planes, const int32_t* table,
                                            void* workspace, size_t workspace_bytes, const float* dpre, int64_t dp_sv, int64_t dp_sb,
                                            const int32_t* sum_rowptr, const int32_t* sum_col, const float* sum_val, float* sum_out,
                                            void* sum_out_planes, int sum_rows, int B, int R, int S, int Cin, int Cout,
                                            sh_stream_t stream);
/* sh_spiral_conv_bwd_wgt_reduce_multi over layers whose slabs were written under different plans: kinds[i] = 0 the fp32 plan
 * (sh_spiral_conv_bwd_wgt*, the thin-layer kernel), 1 the bf16 plan, 2 the three-plane plan (sh_spiral_conv_bwd_wgt_p3*). */
SH_API int sh_spiral_conv_bwd_wgt_reduce_multi_kinds(int n_layers, const void* const* workspaces, float* const* dW, float* const* dbias,
                                                     const int* B, const int* R, const int* S, const int* Cin, const int* Cout,
                                                     const int* kinds, sh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SH_KERNELS_H */
