/*
 * sh_preprocess.h - C ABI of libsh_preprocess.so: native (host, C++) mesh preprocessing for the spiral autoencoder -
 * SURVEY row f3.  Replaces the reference's one-off Python generators:
 *
 *   shp_qslim                  mesh_sampling.qslim_decimator_transformer   (mesh_sampling.py:98-211)  + _get_sparse_transform (:214-227)
 *   shp_barycentric_upsample   mesh_sampling.setup_deformation_transfer    (mesh_sampling.py:47-95; its closest-point search is
 *                              psbody-mesh's C++ AABB tree, which is not part of the reference repository)
 *   shp_spirals                utils_spiral.get_spirals                    (utils_spiral.py:130-417) incl. the shortest-path start
 *                              (single_source_shortest_path :101-125) and the adjacency of get_adj_trigs (:9-41)
 *
 * Plain host pointers and sizes, no allocation handed to the caller, return 0 on success / negative on error
 * (shp_last_error() gives the message).  semantichuman_amd/preprocess.py is the ctypes binding and also reads / writes
 * the reference's `downsampling_matrices*.pkl` layout (main.py:93-113).
 *
 * What "same result as the reference" means here: the decimation follows the reference's algorithm
 * step for step - heapq's sift order, stale-cost re-push, in-place rewriting of queued edges without re-heapifying, the
 * (cost, (r, c)) tuple order - so it selects the same vertices wherever the float64 costs of competing edges differ by
 * more than rounding; exact cost ties (symmetric synthetic meshes) are broken in the reference by the rounding noise of
 * numpy's SVD / BLAS and cannot be reproduced bit for bit.  The spiral traversal reproduces the reference's output
 * wherever it does not depend on CPython's set iteration order (checked against every committed hierarchy).
 */
#ifndef SH_PREPROCESS_H
#define SH_PREPROCESS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SHP_API __attribute__((visibility("default")))
#else
#define SHP_API
#endif

SHP_API const char* shp_last_error(void);

/* QSlim-style edge-collapse decimation down to n_target vertices (the reference passes ceil(nv * factor)).
 * verts [nv][3] float64, faces [nf][3].  Outputs: keep [<= nv] = indices of the surviving vertices in increasing order
 * (row i of the reference's D selects column keep[i]), *n_keep; faces_out [<= nf][3] = the surviving faces re-indexed to
 * the coarse numbering (the reference's new_faces), *nf_out. */
SHP_API int shp_qslim(const double* verts, int nv, const int32_t* faces, int nf, int n_target, int32_t* faces_out,
                      int32_t* nf_out, int32_t* keep, int32_t* n_keep);

/* Up-sampling coefficients: for every target (fine) vertex the closest point on the source (coarse) surface, expressed
 * as the reference expresses it (mesh_sampling.py:58-86): cols [n_tgt][3] = vertices of the closest triangle,
 * coeffs [n_tgt][3] = their coefficients (interior: barycentric; on an edge: least-squares fit of the target point by the
 * edge's two vertices, third coefficient 0; at a vertex: 1 for that vertex), part [n_tgt] = 0 face / 1-3 edge / 4-6 vertex
 * (psbody's convention).  U = csc_matrix((coeffs, (row, cols))). */
SHP_API int shp_barycentric_upsample(const double* src_v, int n_src, const int32_t* src_f, int nf, const double* tgt_v,
                                     int n_tgt, int32_t* cols, double* coeffs, int32_t* part);

/* Spiral orderings of every vertex (utils_spiral.get_spirals, counter-clockwise, padding 'zero', not random), n_steps
 * rings.  ref_points [n_ref]: the reference vertices the first neighbour is directed towards (main.py:50,166-171).
 * Output in CSR form: the spiral of vertex i is out[rowptr[i] .. rowptr[i+1]) (first entry i itself, -1 = the padding
 * vertex); rowptr [nv + 1]; out has room for out_cap entries, *out_len receives the total (call again with a larger
 * buffer if it exceeds out_cap: returns -3 then). */
SHP_API int shp_spirals(const double* verts, int nv, const int32_t* faces, int nf, const int32_t* ref_points, int n_ref,
                        int n_steps, int32_t* rowptr, int32_t* out, int64_t out_cap, int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* SH_PREPROCESS_H */
