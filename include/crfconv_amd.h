/* libcrfconv_amd.so -- C ABI of the MI355X (gfx950) CRFConv hot path.
 *
 * Plain pointers and sizes only (no torch / C++ types).  Two groups of entry points:
 *
 *  (A) Replacements for the reference's native extensions -- exactly what its FFI binds:
 *        utils/nearest_neighbors/knn_.h:2-19          (cpp_knn / cpp_knn_omp / cpp_knn_batch / cpp_knn_batch_omp)
 *        utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.h:84-91 (grid_subsampling)
 *      Host-pointer forms keep the reference signatures (same argument order and meaning,
 *      caller-owned buffers); `_dev` forms take device pointers + a hipStream_t and are what the
 *      Python host (crfconv_amd.utils.nearest_neighbors / cpp_subsampling) drives.
 *
 *  (B) The fused device kernels behind the nn.Module mirrors (crfconv_amd.models):
 *        CRF mean field   <- models/continuous_crf_conv_big.py:49-54, 63-72 (+ sparse :56-67)
 *        PointConv        <- models/point_conv_big.py:37-58
 *        neighbour max-pool / nearest up-sampling <- models/point_conv_big.py:74-77, 97-101
 *      All tensors are dense row-major fp32; neighbour tables are int32 GLOBAL row ids
 *      (cloud b, local id j  ->  b * n_src + j) produced once per batch by crfconv_index_narrow.
 *
 * Every function returns CRF_OK (0) or a negative error code; crfconv_last_error() gives the
 * message (thread-local).  Device entry points only enqueue work on `stream`; they never
 * allocate or synchronise unless their comment says so (workspace is caller-provided), so they
 * can be captured into a hipGraph.
 */
#ifndef CRFCONV_AMD_H
#define CRFCONV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* crf_stream_t; /* hipStream_t */

enum {
    CRF_OK = 0,
    CRF_ERR_ARG = -1,         /* bad shape / null pointer / unsupported size */
    CRF_ERR_HIP = -2,         /* a HIP runtime call failed */
    CRF_ERR_UNSUPPORTED = -3, /* valid request outside what the kernels implement */
    CRF_ERR_WORKSPACE = -4    /* workspace too small (size query tells how much) */
};

#define CRFCONV_ABI_VERSION 1
int crfconv_abi_version(void);
const char* crfconv_last_error(void);

/* ===================================================================== (A) kNN
 * Exact K nearest neighbours, float32 squared L2 accumulated x->y->z with one rounding per
 * operation (nanoflann.hpp:323-347), ascending distance, ties to the lower point index.
 * dim must be 3 (every reference call site: datasets/semantic3d_dataset.py:498, s3dis_dataset.py:413).
 * K <= 64, K <= npts. */

/* Host buffers; reference signatures knn_.h:2-19 (void return there; int status here). */
int crfconv_knn(const float* points, size_t npts, size_t dim, const float* queries, size_t nqueries,
                size_t K, long* indices);
int crfconv_knn_omp(const float* points, size_t npts, size_t dim, const float* queries,
                    size_t nqueries, size_t K, long* indices);
int crfconv_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                      const float* queries, size_t nqueries, size_t K, long* batch_indices);
int crfconv_knn_batch_omp(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                          const float* queries, size_t nqueries, size_t K, long* batch_indices);

/* Device buffers.  out_i64 / out_i32: either may be NULL; [B, nq, K] per-cloud LOCAL indices. */
size_t crfconv_knn_batch_dev_workspace(size_t batch_size, size_t npts, size_t nqueries, size_t K);
int crfconv_knn_batch_dev(const float* pts, size_t batch_size, size_t npts, size_t dim,
                          const float* queries, size_t nqueries, size_t K, int64_t* out_i64,
                          int32_t* out_i32, void* workspace, size_t workspace_bytes,
                          crf_stream_t stream);

/* Farthest point sampling per cloud (torch_cluster.fps as called at models/point_conv.py:381).  All descriptor arrays
 * are device int64 [n_clouds]: seg_start / seg_count = the cloud's rows in pos [N, 3]; n_sample = picks per cloud
 * (<= seg_count); first = local index of the first pick; out_start = offset of the cloud's picks in out.  Picks are
 * GLOBAL row ids in selection order; ties in the arg-max go to the lower index.  dist_ws: float [N] scratch. */
int crfconv_fps(const float* pos, int n_clouds, const int64_t* seg_start, const int64_t* seg_count,
                const int64_t* out_start, const int64_t* n_sample, const int64_t* first, float* dist_ws,
                int64_t* out, crf_stream_t stream);

/* ===================================================================== (A) grid subsampling
 * Barycentre / mean feature / majority label per voxel of edge `sampleDl`
 * (grid_subsampling.cpp:5-106).  Rows are emitted in ascending voxel key (the reference emits
 * hash-map order).  Returns M >= 0 (rows written) or a negative error; if M > cap nothing past
 * `cap` rows is written and -(M) ... is NOT used: the function returns CRF_ERR_WORKSPACE and
 * crfconv_last_error() states the required capacity.  feats/classes may be NULL. */
int64_t crfconv_grid_subsample(const float* points, int64_t N, const float* feats, int fdim,
                               const int32_t* classes, int ldim, float sampleDl, float* out_points,
                               float* out_feats, int32_t* out_classes, int64_t cap);
size_t crfconv_grid_subsample_dev_workspace(int64_t N, int fdim, int ldim);
/* Device buffers; synchronises `stream` once (the row count comes back to the host). */
int64_t crfconv_grid_subsample_dev(const float* points, int64_t N, const float* feats, int fdim,
                                   const int32_t* classes, int ldim, float sampleDl,
                                   float* out_points, float* out_feats, int32_t* out_classes,
                                   int64_t cap, void* workspace, size_t workspace_bytes,
                                   crf_stream_t stream);

/* ===================================================================== (B) neighbour tables
 * idx64 [B, n_tgt, K] per-cloud local ids into n_src points  ->  idx32 [B*n_tgt, K] global rows.
 * Out-of-range entries are clamped and counted into *bad_count (device int32, caller zeroes it);
 * the host must check it before any kernel consumes the table.  idx16 (may be NULL; needs n_src <= 65536)
 * additionally receives the per-cloud LOCAL ids as uint16 -- half the index bytes for the streaming kernels. */
int crfconv_index_narrow(const int64_t* idx64, int64_t B, int64_t n_tgt, int K, int64_t n_src,
                         int32_t* idx32, uint16_t* idx16, int32_t* bad_count, crf_stream_t stream);

/* Same, with columns sort_from .. K-1 of every row re-ordered by ascending source id (sort_from = 1 keeps the
 * self column of a self-query kNN table in place).  Consumers reduce over the columns of a row, so the order is
 * free; ascending ids make the k-th gathers of adjacent target rows hit adjacent source rows. */
int crfconv_index_narrow_sorted(const int64_t* idx64, int64_t B, int64_t n_tgt, int K, int64_t n_src,
                                int sort_from, int32_t* idx32, uint16_t* idx16, int32_t* bad_count,
                                crf_stream_t stream);

/* 30-bit Morton (Z-order) codes of B clouds of npts points (pts [B, npts, 3]): ten bits per axis of
 * clamp((p - lo) / ext * 1023, 0, 1023), lo = the cloud's per-axis minimum, ext = its largest extent.  The device collate sorts
 * every cloud along this curve (crfconv_amd.data.morton_order) so that neighbours sit in nearby rows.  box_ws: 4 B floats. */
int crfconv_morton_codes(const float* pts, int64_t B, int64_t npts, float* box_ws, int64_t* codes, crf_stream_t stream);

/* Any number of device-to-device copies (dst[j] <- src[j], nbytes[j] bytes, non-overlapping) in ONE launch: the ~25 tensors of a
 * freshly collated batch into the static buffers of a captured training step (crfconv_amd.data.MultiScaleData.load_). */
typedef struct { const void* src; void* dst; int64_t nbytes; } crf_copy_job;
int crfconv_copy_jobs(const crf_copy_job* jobs, int njobs, crf_stream_t stream);

/* Reverse (source-major) CSR of a table idx32 [E] with values in [0, m_src):
 * rev_ptr [m_src + 1], rev_eid [E] = edge ids e (= row * K + k) grouped by source row, ascending
 * e inside a group (deterministic summation order for every backward scatter). */
size_t crfconv_reverse_csr_workspace(int64_t E, int64_t m_src);
int crfconv_reverse_csr(const int32_t* idx32, int64_t E, int64_t m_src, int32_t* rev_ptr,
                        int32_t* rev_eid, void* workspace, size_t workspace_bytes,
                        crf_stream_t stream);
/* crfconv_index_narrow_sorted for up to 32 tables in ONE launch (a batch refresh narrows every table of the batch); same idx32 /
 * idx16 / bad counts as the one-table entry point, K <= 64.  jobs is a host array. */
typedef struct { const int64_t* idx64; int64_t B; int64_t n_tgt; int K; int64_t n_src; int sort_from; int32_t* idx32;
                 uint16_t* idx16; int32_t* bad_count; } crf_narrow_job;
int crfconv_index_narrow_batched(const crf_narrow_job* jobs, int njobs, crf_stream_t stream);
/* The same for up to 32 tables in ONE set of five launches (a batch refresh rebuilds every table's reverse CSR: 14 tables =
 * ~100 launches one by one): identical rev_ptr / rev_eid contents.  jobs is a host array. */
typedef struct { const int32_t* idx32; int64_t E; int64_t m_src; int32_t* rev_ptr; int32_t* rev_eid; } crf_rev_job;
size_t crfconv_reverse_csr_batched_workspace(const crf_rev_job* jobs, int njobs);
int crfconv_reverse_csr_batched(const crf_rev_job* jobs, int njobs, void* workspace, size_t workspace_bytes,
                                crf_stream_t stream);

/* ===================================================================== (B) CRF mean field
 * Rows m = B*N (flattened clouds), H hidden channels (4, 8, 16, 32 or 64).
 * Neighbour columns k0 .. K-1 of idx32 are used (k0 = 1 drops the self column,
 * continuous_crf_conv_big.py:45-47); Kn = K - k0 <= 63.
 *   s[i,k]  = softmax_k( -|y_i - y_j(i,k)|^2 )                       (:49-54)
 *   x_0 = z ;  x_t = z Q + (sum_k s[i,k] x_{t-1}[j(i,k)]) P          (:68-72 with Q = (I+C)^-1, P = C Q)
 * Outputs: s [m, K] (edge-id addressed: s[i*K + k], zero on columns < k0);  xs [T, m, H] = x_1 .. x_T
 * (z Q is recomputed per step from z: same read bytes as a stored copy, nothing extra to write).  K in {16, 32} with k0 = 1 takes the fused fast path (similarity + first
 * step in one launch, index / weight rows as aligned dwordx4 loads).  s may be NULL on that path when T == 1
 * (single-step inference: the weights are consumed inside the fused kernel and never re-read). */
int crfconv_meanfield_forward(const float* z, const float* y, const int32_t* idx32, int K, int k0,
                              int64_t m, int H, const float* Q, const float* P, int T, float* s,
                              float* xs, crf_stream_t stream);

/* Same with the index rows read from the uint16 local-id table (idx16 [m, K]; row i belongs to cloud i / n_tgt,
 * whose source rows start at cloud * n_src).  idx16 == NULL falls back to idx32.  Fast path only (K = 16 / 32,
 * k0 = 1); other shapes use idx32. */
int crfconv_meanfield_forward_u16(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                  int n_tgt, int n_src, int K, int k0, int64_t m, int H, const float* Q,
                                  const float* P, int T, float* s, float* xs, crf_stream_t stream);

/* The same forward as ONE launch with block-resident rows (csrc/crf_block.hip; round 6): a workgroup per CU owns consecutive rows for all
 * T steps -- index rows, soft-max weights and z Q stay in registers, the block's own rows of y | z | x_t sit in LDS and serve the
 * neighbours that lie inside the block (four fifths of them in Morton order), a grid barrier separates the steps.  The weights s and x_1 equal
 * crfconv_meanfield_forward_u16's bit for bit, the later iterates to rounding (their messages are added in-block columns first).  ws: crfconv_gridsync_workspace() bytes of zero words (left zero); a barrier that cannot
 * complete sets the sticky failure word (crfconv_gridsync_fail_word).
 * crfconv_meanfield_forward_block_rows: rows per workgroup for m rows on the current device, 0 = shape not covered (H = 8, K = 16,
 * k0 = 1, T >= 1, m <= 768 x CUs).  crfconv_block_locality: count[0] = number of table entries (columns k0 .. K-1) whose source row lies
 * in the SAME block of `rows` consecutive rows as their target -- the fraction count / (m (K - k0)) tells whether the form pays
 * (the host asks once per table). */
int crfconv_meanfield_forward_block_rows(int64_t m, int H, int K, int k0, int T);
int crfconv_meanfield_forward_block(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16,
                                    int n_tgt, int n_src, int K, int k0, int64_t m, int H, const float* Q,
                                    const float* P, int T, float* s, float* xs, void* ws, crf_stream_t stream);
/* Diagnostic twin: 100 MHz phase stamps per workgroup in dbg [blocks][64] (u64), 640 rows per workgroup, uint16 tables
 * (scratch/mf_block_stamps.py; profiles/r6_block_stamps.md). */
int crfconv_meanfield_forward_block_stamps(const float* z, const float* y, const int32_t* idx32, const uint16_t* idx16, int n_tgt, int n_src,
                                           int64_t m, const float* Q, const float* P, int T, float* s, float* xs, void* ws, int shape,
                                           unsigned long long* dbg, crf_stream_t stream);
int crfconv_block_locality(const int32_t* idx32, int64_t m, int K, int k0, int rows, unsigned long long* count, crf_stream_t stream);

/* One backward step, edge half:  given G = dL/dx_t and x_{t-1}:
 *   gm  = G P^T                              [m, H]
 *   ds (+)= <gm_i, x_{t-1}[j(i,k)]>          [m, K]    (accumulate != 0 adds to ds)
 *   mt  = sum_k s[i,k] x_{t-1}[j(i,k)]       [m, H]    (for dP = mt^T G; may be NULL) */
int crfconv_meanfield_bwd_edge(const float* G, const float* xprev, const float* s,
                               const int32_t* idx32, int K, int k0, int64_t m, int H,
                               const float* P, float* gm, float* ds, float* mt, int accumulate,
                               crf_stream_t stream);
/* Scatter half:  Gprev[j] = (add ? add[j] : 0) + sum_{e=(i,k) in rev(j)} s[e] gm[i]. */
int crfconv_meanfield_bwd_scatter(const float* gm, const float* s, const int32_t* rev_ptr,
                                  const int32_t* rev_eid, int K, int k0, int64_t m_src, int H,
                                  const float* add, float* Gprev, crf_stream_t stream);
/* The whole backward of the mean-field loop + similarity for the fast shapes (K in {16, 32}, k0 = 1) as T + 1 launches
 * (csrc/crf_bwd.hip) -- autograd of models/continuous_crf_conv_big.py:49-54, 63-72.  With G_T = gout:
 *   reverse walk x (T-1): Gs[i+1] = (A^T Gs[i]) P^T for steps t = T .. 2 (G_{t-1} = A^T (G_t P^T) = (A^T G_t) P^T);
 *            the first one reads rev_eid -> s[e] and leaves {e, s[e]} records in CSR order in ws for the later walks
 *   edge-all: per point, over all T steps with the index / weight rows and ds in registers:
 *            gm_i = Gs[i] P^T, m_i = sum_k s_ik x_{T-i-1}[j] (x_0 = z), ds += <gm_i, x[j]>, dzq = (sum_i Gs[i]) Q^T,
 *            then w = -2 s (ds - <s, ds>), dy_self = sum_k w_k (y_i - y_j)
 *   last reverse walk: dz = (A^T G_1) P^T + dzq and dy[j] = dy_self[j] + sum_{e in rev(j)} w[e] (y_j - y_i) from ONE
 *            pass over the reverse edge list; extra workgroups of the same launch finish dP, dQ.
 * The walks are load-balanced (a wavefront takes a contiguous range of the reverse edge list edge by edge and its rows sum
 * their segments from an LDS tile in edge order): bitwise reproducible, any in-degree.
 * Entry i of Gs / mts [T, m, H] belongs to step t = T - i; dzq [m, H] is scratch.
 * crfconv_meanfield_backward_param_grads_inside(H) == 1 (H in {8, 16}): dP = sum_i m_i^T G_i and dQ = z^T sum_i G_i
 * are accumulated inside edge-all on the matrix pipe (16x16x4 f32 MFMA outer products through a per-wave LDS tile) and
 * finished inside the last launch into dP, dQ [H, H]; mts and sumG are not touched and may be NULL; ticket = one device
 * word that is ZERO on entry (left zero).  Otherwise mts [T, m, H] and sumG [m, H] are written -- and Gs[0] = G_T = gout,
 * so that Gs is the whole stacked [T, m, H] operand -- and the caller finishes dP = sum_i mts[i]^T Gs[i], dQ = z^T sumG
 * (crfconv_linear_wgrad); dP, dQ, ticket may be NULL.  ws: crfconv_meanfield_backward_workspace(m, H, K) bytes, 16-byte
 * aligned.  m H 4 < 2^31. */
int crfconv_meanfield_backward_supported(int H, int K, int k0);
int crfconv_meanfield_backward_param_grads_inside(int H);
size_t crfconv_meanfield_backward_workspace(int64_t m, int H, int K);
int crfconv_meanfield_backward(const float* gout, const float* z, const float* y, const float* s, const float* xs,
                               const int32_t* idx32, const uint16_t* idx16, int n_tgt, int n_src,
                               const int32_t* rev_ptr, const int32_t* rev_eid, int K, int k0, int64_t m, int H,
                               const float* Q, const float* P, int T, float* Gs, float* dzq, float* mts, float* sumG,
                               float* dz, float* w, float* dy_self, float* dy, float* dP, float* dQ, void* ws,
                               size_t ws_bytes, unsigned* ticket, crf_stream_t stream);


/* Wide rows, H in {128, 256} (the GCRFConv(512, 256) / (256, 128) stages of the sparse networks,
 * models/point_conv.py:318-339): the graph part of the mean-field loop and its backward, one point per wavefront, any
 * K <= 64 / k0, entries of idx32 < 0 = no neighbour.  The H x H products of a step (x = z Q + m P) are plain dense GEMMs
 * left to the caller.
 *   similarity:      s[i,k] = softmax_k(-|y_i - y_j|^2)              aggregate:  out_i = sum_k s[i,k] x[j(i,k)]
 *   bwd_edge:        ds[i,k] (+)= <gm_i, xprev_j>
 *   scatter:         out[j] = (add ? add[j] : 0) + sum_{e in rev(j)} coef[e] src[e / K]        (similarity == 0)
 *                    out[j] = add[j] + sum_{e in rev(j)} coef[e] (src[j] - src[e / K])         (similarity != 0)
 *   similarity_bwd:  w = -2 s (ds - <s, ds>),  dy_self_i = sum_k w_ik (y_i - y_j) */
int crfconv_wide_similarity(const float* y, const int32_t* idx32, int K, int k0, int64_t m, int H, float* s,
                            crf_stream_t stream);
int crfconv_wide_aggregate(const float* x, const float* s, const int32_t* idx32, int K, int k0, int64_t m, int H,
                           float* out, crf_stream_t stream);
int crfconv_wide_bwd_edge(const float* gm, const float* xprev, const int32_t* idx32, int K, int k0, int64_t m, int H,
                          float* ds, int accumulate, crf_stream_t stream);
int crfconv_wide_scatter(const float* src, const float* coef, const int32_t* rev_ptr, const int32_t* rev_eid, int K,
                         int64_t m_src, int H, const float* add, int similarity, float* out, crf_stream_t stream);
int crfconv_wide_similarity_bwd(const float* ds, const float* s, const float* y, const int32_t* idx32, int K, int k0,
                                int64_t m, int H, float* w, float* dy_self, crf_stream_t stream);

/* Softmax + distance backward.  In: ds, s.  Out: w [m, K] = 2 * d(loss)/d(dist_ik) (must NOT
 * alias ds), dy_self[i] = sum_k w_ik (y_i - y_j). */
int crfconv_similarity_bwd(const float* ds, const float* s, const float* y, const int32_t* idx32,
                           int K, int k0, int64_t m, int H, float* w, float* dy_self,
                           crf_stream_t stream);
/* dy[j] = dy_self[j] + sum_{e=(i,k) in rev(j)} w[e] (y_j - y_i). */
int crfconv_similarity_bwd_scatter(const float* w, const float* y, const float* dy_self,
                                   const int32_t* rev_ptr, const int32_t* rev_eid, int K, int k0,
                                   int64_t m_src, int H, float* dy, crf_stream_t stream);

/* ===================================================================== (B) PointConv
 * Depth-wise point convolution with its per-edge weight MLP recomputed on the fly
 * (point_conv_big.py:37-58):  for target row i, neighbour j = idx32[i,k]:
 *   rel = p_tgt[i] - p_src[j];  h1 = lrelu_0.1(A1 rel + b1);  h2 = W2 h1;  w = a2*h2 + b2
 *   out[i,c] = sum_k w[c] * x[j,c]
 * A1 [d,3], b1 [d] fold Linear(3->d) with its BatchNorm; W2 [d,d] row-major [out,in];
 * a2,b2 [d] fold the second BatchNorm.  d in {4,8,16,32,64,128}; K <= 64.  `slope` is the LeakyReLU slope of
 * layer 1 (0.1 in point_conv_big.py:21, nn.LeakyReLU's default 0.01 in the sparse twin point_conv.py:24).
 * Table entries < 0 mean "no neighbour" (padded variable-degree tables): they contribute nothing. */

/* First and second moments of rel over all edges: out9 = {sum x,y,z, sum xx,xy,xz,yy,yz,zz}
 * as float64 (BatchNorm-1 batch statistics are analytic in these). */
size_t crfconv_pointconv_workspace(int64_t m_tgt, int K, int d);
int crfconv_pointconv_moments(const float* pos_src, const float* pos_tgt, const int32_t* idx32,
                              int K, int64_t m_tgt, double* out9, void* workspace,
                              size_t workspace_bytes, crf_stream_t stream);
/* The same sums finished on the device: mean [3] and covariance [3, 3] of rel over the n_edges edges, packed = {mean, cov} [12]
 * (the `mom` argument of fold1 / fold1_bwd) in float64 and the mean in float32, written into the CALLER's tensors -- a refresh
 * of a static batch recomputes them in place (two launches instead of ~17 and four copies). */
int crfconv_pointconv_moments_packed(const float* pos_src, const float* pos_tgt, const int32_t* idx32, int K, int64_t m_tgt,
                                     double n_edges, double* mean, double* cov, double* packed, float* mean32,
                                     void* workspace, size_t workspace_bytes, crf_stream_t stream);
/* The same for up to 16 tables in two launches (a batch refresh recomputes the moments of every PointConv layer's table):
 * identical outputs.  jobs is a host array. */
typedef struct { const float* pos_src; const float* pos_tgt; const int32_t* idx32; int K; int64_t m_tgt; double n_edges;
                 double* mean; double* cov; double* packed; float* mean32; } crf_moments_job;
size_t crfconv_pointconv_moments_batched_workspace(const crf_moments_job* jobs, int njobs);
int crfconv_pointconv_moments_batched(const crf_moments_job* jobs, int njobs, void* workspace, size_t workspace_bytes,
                                      crf_stream_t stream);
/* Batch statistics of h2 over all edges: stats [2, d] float64 = {sum(h2 - shift), sum (h2 - shift)^2},
 * shift [d] float32 out (= h2 at the mean rel; variance is shift-invariant). */
int crfconv_pointconv_stats(const float* pos_src, const float* pos_tgt, const int32_t* idx32, int K,
                            int64_t m_tgt, int d, const float* A1, const float* b1,
                            const float* W2, float slope, const float* mean_rel3, float* shift, double* stats,
                            void* workspace, size_t workspace_bytes, crf_stream_t stream);
/* Training forward in ONE edge pass.  BatchNorm-2 is affine per channel, so
 *   out_i = a2 * U_i + (a2 * shift + b2) * V_i,   U_i = sum_k (h2_k - shift) * x_j,   V_i = sum_k x_j
 * (U, V [m_tgt, d] float32 out) and the batch statistics of h2 (stats, shift: as crfconv_pointconv_stats) come
 * out of the same pass; crfconv_pointconv_combine folds BatchNorm-2 (as crfconv_pointconv_fold2 in training mode:
 * a2, b2, aux2 out, running statistics advanced) and finishes the convolution in one elementwise launch.
 * The backward reductions of BatchNorm-2 need no edge pass either:
 *   {sum_e g_w, sum_e g_w (h2 - shift)} = {sum_i gout_i * V_i, sum_i gout_i * U_i};
 * crfconv_pointconv_bwd_reduce_uv forms them and applies crfconv_pointconv_fold2_bwd (ca, cb, cc, dgamma2, dbeta2). */
int crfconv_pointconv_forward_uv(const float* x, const float* pos_src, const float* pos_tgt,
                                 const int32_t* idx32, int K, int64_t m_tgt, int d, const float* A1,
                                 const float* b1, const float* W2, float slope, const float* mean_rel3,
                                 float* shift, double* stats, float* U, float* V, void* workspace,
                                 size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);
/* The same launch CARRYING crfconv_crf_matrices_batched(c, H, n, Q, P) as n extra workgroups (round 5): the CRF layers' matrices depend
 * on parameters only, yet as a launch of their own (a 27 us chain of 64 dependent pivots on n workgroups) they sat in the forward's
 * launch chain; in the 31 us statistics pass of the network's first PointConv nothing waits for them.  For the widths
 * crfconv_pointconv_forward_uv_hosts(K, d) names (d = 8); results of both parts are those of the two separate calls. */
int crfconv_pointconv_forward_uv_hosts(int K, int d);
int crfconv_pointconv_forward_uv_hosting(const float* x, const float* pos_src, const float* pos_tgt, const int32_t* idx32, int K,
                                         int64_t m_tgt, int d, const float* A1, const float* b1, const float* W2, float slope,
                                         const float* mean_rel3, float* shift, double* stats, float* U, float* V, void* workspace,
                                         size_t workspace_bytes, unsigned* ticket, const float* const* c, const int* H, int n,
                                         float* const* Q, float* const* P, crf_stream_t stream);
int crfconv_pointconv_combine(const float* U, const float* V, const double* stats, const float* shift,
                              const float* gamma2, const float* beta2, double n_edges, float* run_mean,
                              float* run_var, float momentum, float eps, int64_t m_tgt, int d, float* a2, float* b2,
                              double* aux2, float* out, crf_stream_t stream);
int crfconv_pointconv_bwd_reduce_uv(const float* gout, const float* U, const float* V, int64_t m_tgt, int d,
                                    const float* shift, const double* aux2, const float* gamma2, double n_edges,
                                    int use_batch, float* ca, float* cb, float* cc, float* dgamma2, float* dbeta2,
                                    void* workspace, size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);
int crfconv_pointconv_forward(const float* x, const float* pos_src, const float* pos_tgt,
                              const int32_t* idx32, int K, int64_t m_tgt, int d, const float* A1,
                              const float* b1, const float* W2, float slope, const float* a2, const float* b2,
                              float* out, crf_stream_t stream);
/* Backward reductions, pass 1:  red1 [2, d] float64 = {sum_e g_w, sum_e g_w * (h2 - shift)} with
 * g_w[e,c] = gout[i,c] * x[j,c]  (BatchNorm-2 backward needs both before pass 2). */
int crfconv_pointconv_bwd_reduce(const float* x, const float* gout, const float* pos_src,
                                 const float* pos_tgt, const int32_t* idx32, int K, int64_t m_tgt,
                                 int d, const float* A1, const float* b1, const float* W2, float slope,
                                 const float* shift, double* red1, void* workspace, size_t workspace_bytes,
                                 crf_stream_t stream);
/* Pass 2:  g_h2[e,c] = ca[c] * g_w[e,c] + cb[c] * h2[e,c] + cc[c]  (the host folds BatchNorm-2's
 * backward into ca/cb/cc; eval mode: ca = a2, cb = cc = 0), then back through W2, lrelu, A1:
 *   dW2 [d,d] and dA1b1 [d,4] = {dA1[c][0..2], db1[c]} as float64 sums (the latter accumulated in
 * float64 throughout: the host's analytic BatchNorm-1 backward cancels their large common parts).  d <= 16. */
int crfconv_pointconv_bwd_params(const float* x, const float* gout, const float* pos_src,
                                 const float* pos_tgt, const int32_t* idx32, int K, int64_t m_tgt,
                                 int d, const float* A1, const float* b1, const float* W2, float slope,
                                 const float* ca, const float* cb, const float* cc, double* dW2,
                                 double* dA1b1, void* workspace, size_t workspace_bytes,
                                 crf_stream_t stream);
/* Pass 2 for wide layers (d >= 32, few edges): instead of reducing in-kernel, write per edge e = i*K + k
 *   h1 [E, d],  g_h2 [E, d] (same definition as above),  rel [E, 3]
 * so the host can form dW2 = g_h2^T h1, g_h1 = g_h2 W2, dA1 = (g_h1 * lrelu')^T rel, db1 with dense GEMMs. */
int crfconv_pointconv_bwd_dump(const float* x, const float* gout, const float* pos_src,
                               const float* pos_tgt, const int32_t* idx32, int K, int64_t m_tgt, int d,
                               const float* A1, const float* b1, const float* W2, float slope, const float* ca,
                               const float* cb, const float* cc, float* h1, float* gh2, float* rel,
                               crf_stream_t stream);
/* Wide-layer dA1 | db1 from the dumped tensors: with gw = g_h2 W2 [E, d] (a dense GEMM of the caller),
 *   dA1b1[c] = sum_e gw[e,c] * lrelu'(h1[e,c]) * {rel_x, rel_y, rel_z, 1}     float64 [d, 4],  d in {32, 64, 128}. */
size_t crfconv_pointconv_bwd_a1_workspace(int64_t n_edges, int d);
int crfconv_pointconv_bwd_a1(const float* gw, const float* h1, const float* rel, int64_t n_edges, int d,
                             float slope, double* dA1b1, void* workspace, size_t workspace_bytes,
                             crf_stream_t stream);
/* The wide layers' parameter pass for SEVERAL layers at once (nothing on the backward chain waits for it: queued, issued at the end):
 * crfconv_pointconv_bwd_dump_jobs = crfconv_pointconv_bwd_dump of every job, one launch per width present (the two ResNet blocks of a
 * level share a launch); crfconv_pointconv_bwd_a1_jobs = the slab pass of crfconv_pointconv_bwd_a1 of every job likewise (sums:
 * crfconv_reduce_jobs_f64).  Identical outputs / slabs.  jobs are host arrays. */
typedef struct { const float* x; const float* gout; const float* pos_src; const float* pos_tgt; const int32_t* idx32; int K; int64_t m_tgt;
                 int d; const float* A1; const float* b1; const float* W2; float slope; const float* ca; const float* cb; const float* cc;
                 float* h1; float* gh2; float* rel; } crf_pc_dump_job;
typedef struct { const float* gw; const float* h1; const float* rel; int64_t n_edges; int d; float slope; void* workspace;
                 size_t workspace_bytes; } crf_pc_a1_job;
int crfconv_pointconv_bwd_dump_jobs(const crf_pc_dump_job* jobs, int njobs, crf_stream_t stream);
/* The same parameter pass for K = 16 and d in {32, 64} WITHOUT per-edge tensors (csrc/pointconv_wide.hip): one wavefront per
 * target point, the three d x d products per edge (h2 = W2 h1, W2^T gh2, dW2 += gh2 (x) h1) on v_mfma_f32_16x16x4_f32.  Per
 * workgroup b < crfconv_pointconv_wide_params_nblk(m_tgt, d): dw2_partial[b] [d, d] float (row = output channel of W2) and
 * a1_partial[b] [d, 4] float64 ({dA1[c][0..2], db1[c]}); the caller sums the slabs (crfconv_reduce_jobs / crfconv_reduce_jobs_f64).
 * ca, cb, cc: the per-channel coefficients of crfconv_pointconv_bwd_reduce_uv (as crf_pc_dump_job).  One launch per width. */
typedef struct { const float* x; const float* gout; const float* pos_src; const float* pos_tgt; const int32_t* idx32; int K; int64_t m_tgt;
                 int d; const float* A1; const float* b1; const float* W2; float slope; const float* ca; const float* cb; const float* cc;
                 float* dw2_partial; double* a1_partial; } crf_pc_wide_job;
int crfconv_pointconv_wide_params_supported(int64_t m_tgt, int K, int d);
int64_t crfconv_pointconv_wide_params_nblk(int64_t m_tgt, int d);
int crfconv_pointconv_wide_params_jobs(const crf_pc_wide_job* jobs, int njobs, crf_stream_t stream);
int crfconv_pointconv_bwd_a1_jobs(const crf_pc_a1_job* jobs, int njobs, crf_stream_t stream);
/* Deferred sums.  crfconv_pointconv_bwd_params with dW2 = dA1b1 = NULL and crfconv_pointconv_bwd_a1 with dA1b1 = NULL leave their
 * partial slabs only (crfconv_pointconv_bwd_params_slabs / the aligned start of the a1 workspace, crfconv_pointconv_bwd_a1_nblk slabs);
 * crfconv_reduce_jobs_f64 then finishes ANY number of such sums in one launch per 32 jobs -- out[slot] = sum_b partial[b][slot] in
 * float64, partial float (is_float != 0) or double, same lane order and shuffle tree as the in-call reductions (identical results).
 * The parameter gradients of all PointConv layers of a backward pass are summed this way, once, in front of the batched fold. */
typedef struct { const void* partial; int is_float; int64_t nblk; int nslots; double* out; } crf_reduce64_job;
int crfconv_pointconv_bwd_params_slabs(void* workspace, int64_t m_tgt, int d, const float** slab_w2, const double** slab_a1,
                                       int64_t* nblk_out);
int64_t crfconv_pointconv_bwd_a1_nblk(int64_t n_edges, int d);
int crfconv_reduce_jobs_f64(const crf_reduce64_job* jobs, int njobs, crf_stream_t stream);
/* dx[j,c] = sum_{e=(i,k) in rev(j)} w_e[c] * gout[i,c]   (weight MLP recomputed per incoming edge). */
int crfconv_pointconv_bwd_input(const float* gout, const float* pos_src, const float* pos_tgt,
                                const int32_t* rev_ptr, const int32_t* rev_eid, int K,
                                int64_t m_src, int d, const float* A1, const float* b1,
                                const float* W2, float slope, const float* a2, const float* b2, float* dx,
                                crf_stream_t stream);
/* crfconv_pointconv_bwd_input and the ticketed crfconv_pointconv_bwd_reduce_uv of one layer in ONE launch (round 5): the input gradient
 * needs the forward coefficients only and nothing reads the reduction's outputs before the next launch.  Same results as the two calls. */
int crfconv_pointconv_bwd_input_reduce(const float* gout, const float* pos_src, const float* pos_tgt, const int32_t* rev_ptr,
                                       const int32_t* rev_eid, int K, int64_t m_src, int64_t m_tgt, int d, const float* A1, const float* b1,
                                       const float* W2, float slope, const float* a2, const float* b2, float* dx, const float* U,
                                       const float* V, const float* shift, const double* aux2, const float* gamma2, double n_edges,
                                       int use_batch, float* ca, float* cb, float* cc, float* dgamma2, float* dbeta2, void* workspace,
                                       size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);

/* BatchNorm folding of the weight MLP (one tiny workgroup each; d <= 128):
 *  fold1      W1 [d,3], gamma1/beta1, mom = {mean[3], cov[9]} of rel (float64) -> A1 [d,3], b1 [d];
 *             batch statistics of BN-1 are analytic in mom (mean1 = w.mu, var1 = w^T Sigma w); running
 *             statistics updated in place when use_batch != 0 and non-NULL; aux1 [3,d] float64 saved for backward.
 *  fold1_bwd  dA1b1 [d,4] float64 -> dW1 [d,3], dgamma1, dbeta1 (through the analytic statistics); when dW2_f64 [d,d] (the
 *             float64 accumulator of crfconv_pointconv_bwd_params) is given, also dW2_f32 = (float) dW2_f64 (both or neither NULL).
 *  fold2      stats [2,d] + shift (crfconv_pointconv_stats) -> a2, b2; aux2 [2,d] = {mean2, rstd2}.
 *  fold2_bwd  red [2,d] (crfconv_pointconv_bwd_reduce) -> ca, cb, cc for pass 2, dgamma2, dbeta2. */
int crfconv_pointconv_fold1(const float* W1, const float* gamma1, const float* beta1, const double* mom,
                            double n_edges, float* run_mean, float* run_var, float momentum, float eps,
                            int use_batch, int d, float* A1, float* b1, double* aux1, crf_stream_t stream);
int crfconv_pointconv_fold1_bwd(const float* W1, const float* gamma1, const double* mom, const double* aux1,
                                const double* dA1b1, float eps, int use_batch, int d, float* dW1,
                                float* dgamma1, float* dbeta1, const double* dW2_f64, float* dW2_f32, crf_stream_t stream);
/* fold1 of ALL PointConv layers of a network in one launch (a workgroup per job; fields as the arguments of
 * crfconv_pointconv_fold1): every input is known before the forward pass starts (models/point_conv_big.py:113-131 has ten). */
typedef struct {
    const float* W1; const float* gamma1; const float* beta1; const double* mom; double n_edges;
    float* run_mean; float* run_var; float momentum; float eps; int32_t use_batch; int32_t d;
    float* A1; float* b1; double* aux1;
} crf_fold1_job;
int crfconv_pointconv_fold1_batched(const crf_fold1_job* jobs, int njobs, crf_stream_t stream);
/* fold1_bwd of ALL PointConv layers of a backward pass in one launch (a workgroup per job; fields as the arguments of
 * crfconv_pointconv_fold1_bwd).  Nothing inside the pass reads dW1 / dgamma1 / dbeta1, so ops.deferred_weight_grads queues
 * the jobs and runs them once at the end. */
typedef struct {
    const float* W1; const float* gamma1; const double* mom; const double* aux1; const double* dA1b1;
    float eps; int32_t use_batch; int32_t d; int32_t pad_;
    float* dW1; float* dgamma1; float* dbeta1; const double* dW2_f64; float* dW2_f32;
} crf_fold1_bwd_job;
int crfconv_pointconv_fold1_bwd_batched(const crf_fold1_bwd_job* jobs, int njobs, crf_stream_t stream);
int crfconv_pointconv_fold2(const double* stats, const float* shift, const float* gamma2, const float* beta2,
                            double n_edges, float* run_mean, float* run_var, float momentum, float eps,
                            int use_batch, int d, float* a2, float* b2, double* aux2, crf_stream_t stream);
int crfconv_pointconv_fold2_bwd(const double* red, const float* shift, const double* aux2, const float* gamma2,
                                double n_edges, int use_batch, int d, float* ca, float* cb, float* cc,
                                float* dgamma2, float* dbeta2, crf_stream_t stream);

/* ===================================================================== (B) per-point Linear layers
 * Weight gradient of y = x W^T (+ b), the one contraction of models/common.py:30,35 vendor GEMMs handle
 * badly (reduction over m = 10^4..10^5 rows into a tiny [Co, Ci]):
 *   dW[co, ci] = sum_m G[m, co] X[m, ci];   db[co] = sum_m G[m, co]  (db may be NULL)
 * fp32 MFMA (v_mfma_f32_16x16x4_f32, exact f32), operands streamed once, fixed-order reduction. */
size_t crfconv_linear_wgrad_workspace(int64_t M, int Co, int Ci);
int crfconv_linear_wgrad(const float* G, const float* X, int64_t M, int Co, int Ci, float* dW, float* db,
                         void* workspace, size_t workspace_bytes, crf_stream_t stream);
/* The same contraction WITHOUT its final reduction: writes the row-slice partials to `workspace` (256-byte aligned,
 * crfconv_linear_wgrad_workspace bytes): float [nblk][Co][Ci] followed, when want_bias, by float [nblk][Co].
 * crfconv_reduce_jobs then finishes any number of such products (all weight gradients of a backward pass) in
 * ONE launch: out[j][s] = sum_b partial[j][b * nslots + s]. */
typedef struct { const float* partial; float* out; int32_t nblk; int32_t nslots; } crf_reduce_job;
int crfconv_linear_wgrad_partial(const float* G, const float* X, int64_t M, int Co, int Ci, int want_bias,
                                 void* workspace, size_t workspace_bytes, int* nblk_out, crf_stream_t stream);
/* The partial passes of several layers in one launch per tile class (the coarse levels' weight gradients are ~9 us launches nothing
 * on the backward chain waits for: queued, then issued together): identical partial slabs.  jobs is a host array; each workspace as
 * for crfconv_linear_wgrad_partial; crfconv_linear_wgrad_nblk = the slab count of a shape (the reduction job needs it). */
typedef struct { const float* G; const float* X; int64_t M; int Co; int Ci; int want_bias; void* workspace; size_t workspace_bytes; } crf_wgrad_job;
int crfconv_linear_wgrad_nblk(int64_t M, int Co, int Ci);
int crfconv_linear_wgrad_partial_jobs(const crf_wgrad_job* jobs, int njobs, crf_stream_t stream);
int crfconv_reduce_jobs(const crf_reduce_job* jobs, int njobs, crf_stream_t stream);
/* crfconv_reduce_jobs and crfconv_reduce_jobs_f64 (below / above: independent inputs, both at the end of a backward pass) in ONE launch
 * when each batch fits one table (96 / 32 jobs), else the two calls; identical results. */
int crfconv_reduce_jobs_both(const crf_reduce_job* jobs, int njobs, const crf_reduce64_job* jobs64, int njobs64, crf_stream_t stream);

/* Linear -> BatchNorm(train) -> LeakyReLU of the coarse levels as ONE launch (csrc/mlp_small.hip): replaces the
 * nn.Linear + FastBatchNorm1d + activation chain of models/common.py:34-40 where m <= 4096 rows (encoder / decoder
 * levels 3-5 of models/point_conv_big.py:113-131).  A 64 x 64 tile of Y = X W^T per workgroup on the fp32 matrix cores,
 * per-tile statistic records, a grid barrier (every workgroup resident: _supported checks the device), float64
 * coefficients, BatchNorm + LeakyReLU applied to the tile still in registers.  Y [M, Co] (kept for the backward),
 * A [M, Co], coef [4, Co] = a | b | mean | rstd as crfconv_bn_forward writes it; running statistics updated when given.
 * workspace: crfconv_mlp_small_workspace bytes.  sync_ws: crfconv_gridsync_workspace() bytes that are ZERO before the
 * first launch using them (the kernel leaves them zero; word 17 * 32 is non-zero only after a barrier that gave up). */
size_t crfconv_gridsync_workspace(void);
/* word index of the sticky failure flag inside that workspace (non-zero once a grid barrier timed out: the launch's
 * outputs are NaN-poisoned; the host must raise -- crfconv_amd.ops.check_gridsync) */
int crfconv_gridsync_fail_word(void);
int crfconv_mlp_small_supported(int64_t M, int Ci, int Co);
size_t crfconv_mlp_small_workspace(int64_t M, int Co);
int crfconv_mlp_small_forward(const float* X, const float* W, int64_t M, int Ci, int Co, const float* gamma,
                              const float* beta, float* run_mean, float* run_var, float momentum, float eps, float slope,
                              float* Y, float* A, float* coef, void* workspace, size_t workspace_bytes, void* sync_ws,
                              size_t sync_bytes, crf_stream_t stream);

/* crfconv_mlp_small_forward with the ResNet join folded into the same launch:
 * A = lrelu(lrelu(BN_train(X W^T), slope) + skip, join_slope)  (models/point_conv_big.py:84-88: slope = 1, join_slope = 0.01). */
int crfconv_mlp_small_forward_join(const float* X, const float* W, int64_t M, int Ci, int Co, const float* gamma,
                                   const float* beta, float* run_mean, float* run_var, float momentum, float eps, float slope,
                                   const float* skip, float join_slope, float* Y, float* A, float* coef, void* workspace,
                                   size_t workspace_bytes, void* sync_ws, size_t sync_bytes, crf_stream_t stream);

/* Backward of one MLP block  A = lrelu(BN_train(X W^T), slope)  (models/common.py:34-40, batch statistics) in two passes
 * over the activations and three launches: pass 1 streams (gA, Y, X) once and leaves the partials of sum g1, sum g1 yh,
 * G1^T X, Yh^T X and 1^T X (g1 = gA lrelu'(a y + b), yh = (y - mean) rstd; two MFMA accumulator sets share the X
 * fragment); the finalize launch turns them into dgamma, dbeta, dW = diag(a) [G1^T X - (dbeta / M) 1^T X -
 * diag(dgamma / M) Yh^T X] and the per-channel coefficients of gY; pass 2 forms gY in registers while loading (gA, Y)
 * and writes dX = gY W (skipped when dX is NULL).  gA, Y [M, Co]; X [M, Ci]; W [Co, Ci]; coef = the [4, Co] block of
 * crfconv_bn_forward / crfconv_bn_coef_from_records for Y.  Same results as crfconv_bn_backward ->
 * crfconv_linear_forward(transpose) + crfconv_linear_wgrad up to summation order.
 * ticket (crfconv_mlp_backward / _add / _cat, crfconv_pointconv_forward_uv / _bwd_reduce_uv): crfconv_ticket_bytes() ZERO device
 * bytes owned by the stream (the kernels leave them zero; large launches draw their tickets in two levels, one 128-byte line per
 * group), or NULL.  With it the LAST workgroup of the first pass to finish adds the partial rows and
 * does the finalize's channel part itself (write-through stores, one atomic ticket per workgroup, fixed summation order): one
 * launch less on the chain (here, for Co in {4, 8, .., 128}, when dW is NULL: two launches).  NULL: the launch-separated form. */
size_t crfconv_ticket_bytes(void);
int crfconv_mlp_backward_supported(int64_t M, int Ci, int Co);
size_t crfconv_mlp_backward_workspace(int64_t M, int Ci, int Co);
int crfconv_mlp_backward(const float* gA, const float* Y, const float* X, const float* W, const float* coef, float slope,
                         int64_t M, int Ci, int Co, float* dX, float* dW, float* dgamma, float* dbeta, void* workspace,
                         size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);
/* dW may be NULL in crfconv_mlp_backward / _add / _cat: the finalize launch then does its channel part only (dgamma, dbeta, the
 * coefficients dX needs) and the weight gradient is finished later, for any number of blocks in ONE launch, from the
 * workspaces those calls left behind (which must be untouched since): nothing inside a backward pass reads a weight gradient. */
typedef struct { const void* workspace; const float* coef; float* dW; int64_t M; int32_t Ci; int32_t Co; } crf_mlp_dw_job;
int crfconv_mlp_dw_jobs(const crf_mlp_dw_job* jobs, int njobs, crf_stream_t stream);
/* The same launch CARRYING crfconv_crf_matrices_backward_batched(c, Q, gQ, gP, H, n, dc) (below) as its first workgroups (round 6: both are
 * end-of-pass parameter work; the matrices' backward alone is a 13 us dependent chain on 15 workgroups).  njobs >= 1.  Results of both
 * are those of the two separate calls. */
int crfconv_mlp_dw_jobs_hosting(const crf_mlp_dw_job* jobs, int njobs, const float* const* c, const float* const* Q,
                                const float* const* gQ, const float* const* gP, const int* H, int n, float* const* dc,
                                crf_stream_t stream);
/* The same backward for a block whose input x has a second consumer -- the shortcut of a ResNet block
 * (models/point_conv_big.py:83-88: lin_in(x) and shortcut(x)): dX = gY W + dX_add, dX_add [M, Ci] the gradient that other
 * consumer already sent back (NULL: as crfconv_mlp_backward).  Replaces the accumulation pass autograd would run. */
int crfconv_mlp_backward_add(const float* gA, const float* Y, const float* X, const float* W, const float* coef, float slope,
                             int64_t M, int Ci, int Co, const float* dX_add, float* dX, float* dW, float* dgamma, float* dbeta,
                             void* workspace, size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);
/* The block whose input is the column concatenation [Xa | Xb] (the CRF layers' fusion_nn(cat[x, pairwise]),
 * models/continuous_crf_conv_big.py:76) without materialising it: Xa [M, split], Xb [M, Ci - split], split % 4 == 0;
 * the forward product is crfconv_linear_forward_cat, the backward writes dXa / dXb separately. */
int crfconv_linear_forward_cat(const float* Xa, const float* Xb, int split, const float* W, const float* bias, int64_t M,
                               int Ci, int Co, float* Y, float* stat_rec, crf_stream_t stream);
int crfconv_mlp_backward_cat(const float* gA, const float* Y, const float* Xa, const float* Xb, int split, const float* W,
                             const float* coef, float slope, int64_t M, int Ci, int Co, float* dXa, float* dXb, float* dW,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, unsigned* ticket,
                             crf_stream_t stream);

/* Fused BatchNorm (+ LeakyReLU) over rows x [M, C] (models/common.py:31,36-37; C % 4 == 0, C <= 1024).
 * forward:  use_batch_stats != 0 -> statistics of x (biased variance), running stats updated in place when non-NULL
 *           (momentum, unbiased variance -- torch.nn.BatchNorm1d semantics); else coefficients from running stats.
 *           coef [4, C] out = {a = gamma*rstd, b = beta - a*mean, mean, rstd};  y = lrelu(a x + b, slope)
 *           (slope = 1: no activation).
 * backward: gx = d/dx, dgamma, dbeta from gy = d/dy, the saved x and coef (training != 0: batch-statistics
 *           backward; else the affine-only backward). */
size_t crfconv_bn_workspace(int64_t M, int C);
int crfconv_bn_forward(const float* x, int64_t M, int C, const float* gamma, const float* beta, float* run_mean,
                       float* run_var, float momentum, float eps, int use_batch_stats, float slope, float* coef,
                       float* y, void* workspace, size_t workspace_bytes, crf_stream_t stream);
int crfconv_bn_backward(const float* gy, const float* x, const float* coef, int64_t M, int C, int training,
                        float slope, float* gx, float* dgamma, float* dbeta, void* workspace,
                        size_t workspace_bytes, crf_stream_t stream);

int crfconv_bn_apply(const float* x, int64_t M, int C, const float* coef, float slope, float* y, crf_stream_t stream);

/* Y [M, Co] = X [M, Ci] W^T (+ bias) on fp32 MFMA, X streamed once, W resident in LDS (needs
 * crfconv_linear_forward_supported(Ci, Co): a 16-channel weight slab of Ci inputs fits 64 KB; the slab is 64 channels
 * wide where that fits, else 32 / 16).  W is [Co, Ci] row-major, or
 * [Ci, Co] when transpose_w != 0 (the input-gradient product dX = G W).  stat_rec (may be NULL): float
 * [crfconv_linear_forward_stat_records(M)][Co][4] receives per-workgroup {shift, n, sum(y - shift), sum (y - shift)^2},
 * from which crfconv_bn_coef_from_records forms the BatchNorm coefficients (coef [4, Co], as crfconv_bn_forward) --
 * the statistics pass over Y disappears. */
int crfconv_linear_forward_supported(int Ci, int Co);
size_t crfconv_linear_forward_stat_records(int64_t M);
int crfconv_linear_forward(const float* X, const float* W, const float* bias, int64_t M, int Ci, int Co,
                           int transpose_w, float* Y, float* stat_rec, crf_stream_t stream);
/* Row plumbing of the training step (csrc/rows.hip), so that a captured step launches no framework kernel:
 *  crfconv_cat2: out [m, ca + cb] = [xa | xb] -- the torch.cat in front of the fusion layers
 *    (models/continuous_crf_conv_big.py:71, models/point_conv_big.py:107) on the levels where the two-pointer Linear does not
 *    apply; crfconv_split2 is its backward (two contiguous gradients in one pass).  ca, cb multiples of 4.
 *  crfconv_add_i64: x[0 .. n) += delta (the num_batches_tracked counters of all BatchNorm layers, one launch). */
int crfconv_cat2(const float* xa, const float* xb, int64_t m, int ca, int cb, float* out, crf_stream_t stream);
int crfconv_split2(const float* g, int64_t m, int ca, int cb, float* ga, float* gb, crf_stream_t stream);
int crfconv_add_i64(int64_t* x, int64_t n, int64_t delta, crf_stream_t stream);
/* A bounded device-side gate between two streams (csrc/rows.hip): gate = 4 zeroed uint64 device words (mark | consumed | enabled |
 * timeouts).  crfconv_gate_mark (inside the training graph, where its coarse levels begin) adds one mark; crfconv_gate_wait (first
 * launch of the collate graph on the side stream) holds its stream until an unconsumed mark exists -- for at most max_wait_us
 * (<= 100 000) microseconds, then goes ahead and counts a timeout (three in a row switch the gate off: gate[2] = 0); it returns at
 * once while gate[2] == 0.  No host involvement, no event: a wait on an event recorded INSIDE a replayed graph sees the previous
 * replay's record. */
int crfconv_gate_mark(uint64_t* gate, crf_stream_t stream);
int crfconv_gate_wait(uint64_t* gate, int max_wait_us, crf_stream_t stream);

/* C [M, N] = A [M, K] B (+ bias [N]) (+ addend [M, N]) on fp32 MFMA for the shapes crfconv_linear_forward does not take:
 * the coarse-level Linear forward (models/common.py:30,35 at 640 .. 10 240 rows, up to 512 channels), every dX = gY W of
 * the MLP / ResNet-block backward and g_h1 = g_h2 W2 of the wide PointConv layers (models/point_conv_big.py:45-47) --
 * replaces the rocBLAS / hipBLASLt call behind torch.nn.functional.linear / torch.mm / torch.addmm at those sites.
 * b_is_nk != 0: B is [N, K] row-major (C = A B^T, the F.linear weight layout), else [K, N] row-major.  bias and addend
 * may be NULL; addend may alias C.  Any N, K >= 1 (16-byte accesses when both are multiples of 4, element-wise otherwise --
 * the 13-class logits).  32 x 32 ... 64 x 64 outputs per workgroup, both operands through double-buffered LDS tiles, the tile
 * shape picked so the grid covers the chip; fixed summation order. */
/* Backward of one coarse-level MLP block A = lrelu(BN_train(X W^T), slope) (models/common.py:34-40) behind its one-launch forward, in
 * two launches: row-tile partials of the two BatchNorm channel sums, then dX [M, Ci] = gY W (+ addend [M, Ci], may be NULL) on the
 * tiled product with gY [M, Co] -- the gradient in front of the BatchNorm -- formed in its operand load from (gA, Y, coef) and stored
 * for the weight gradient; dgamma / dbeta on the way.  Same results as crfconv_bn_backward + crfconv_gemm up to summation order.
 * Ci, Co multiples of 4, Co <= 512; workspace: crfconv_mlp_small_backward_workspace(M, Co) bytes. */
int crfconv_mlp_small_backward_supported(int64_t M, int Ci, int Co);
size_t crfconv_mlp_small_backward_workspace(int64_t M, int Co);
int crfconv_mlp_small_backward(const float* gA, const float* Y, const float* coef, const float* W, const float* addend, int64_t M,
                               int Ci, int Co, int training, float slope, float* gY, float* dX, float* dgamma, float* dbeta,
                               void* workspace, size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);
/* ticket: crfconv_ticket_bytes() of zero device words per stream (left zero): the last workgroup of each column slab of the first
 * launch finishes the two channel sums (float64, row-tile order), so the product launch loads finished means instead of every one
 * of its workgroups re-summing all row-tile partials (round 5). */
/* C [M, N] = A [M, K] B^T (B [N, K], the F.linear weight; N, K multiples of 4) on the same tiled kernel, with the BatchNorm statistic
 * records of C written by the epilogue: stat_rec float [crfconv_gemm_stat_records(M)][N][4] = {shift, rows, sum (v - shift),
 * sum (v - shift)^2} per 16-row group and channel -- the input of crfconv_bn_coef_from_records, so no statistics pass over C runs
 * (the Linear + BatchNorm blocks between the one-launch kernel's row limit and the row-streaming forms). */
size_t crfconv_gemm_stat_records(int64_t M);
int crfconv_gemm_stats(const float* A, const float* B, int64_t M, int N, int K, float* C, float* stat_rec, crf_stream_t stream);
/* ---- several INDEPENDENT coarse-level MLP blocks per launch (round 5).  A coarse launch is a chain of dependent memory round trips
 * on a grid that covers a fraction of the chip, so two blocks whose inputs are both ready -- unary_nn[i] and pairwise_nn[i] of a CRF
 * layer (models/continuous_crf_conv_big.py:56-60), shortcut and lin_in of a strided ResNet block (models/point_conv_big.py:79-88) --
 * run side by side for the price of the longer one.  jobs: host arrays of 1 .. 4 entries; per job the same tiles, summation order
 * and results as the one-block calls above / below.
 *   crfconv_gemm_stats_jobs              = crfconv_gemm_stats per job, one launch
 *   crfconv_bn_apply_from_records_jobs   = crfconv_bn_apply_from_records per job, one launch
 *   crfconv_mlp_small_backward_jobs      = crfconv_mlp_small_backward per job, two launches in all */
typedef struct { const float* A; const float* B; int64_t M; int N; int K; float* C; float* stat_rec; } crf_gemm_stats_job;
int crfconv_gemm_stats_jobs(const crf_gemm_stats_job* jobs, int njobs, crf_stream_t stream);
typedef struct { const float* stat_rec; int64_t nrec; const float* x; int64_t M; int C; const float* gamma; const float* beta;
                 float* run_mean; float* run_var; float momentum; float eps; const float* skip; float slope; float* coef; float* y;
} crf_bn_apply_job;
int crfconv_bn_apply_from_records_jobs(const crf_bn_apply_job* jobs, int njobs, crf_stream_t stream);
typedef struct { const float* gA; const float* Y; const float* coef; const float* W; const float* addend; int64_t M; int Ci; int Co;
                 int training; float slope; float* gY; float* dX; float* dgamma; float* dbeta; void* workspace; size_t workspace_bytes;
} crf_mlp_bwd_job;
int crfconv_mlp_small_backward_jobs(const crf_mlp_bwd_job* jobs, int njobs, unsigned* ticket, crf_stream_t stream);
/* The same in ONE launch (round 6): tile-sum workgroups first, the product's workgroups wait inside the launch for their job's channel
 * means (bounded; a wait that gives up sets the sticky failure word of sync_ws and leaves NaN in that tile).  sync_ws: the zeroed words of
 * crfconv_gridsync_workspace(), left zero.  Bit-identical to the two launches. */
int crfconv_mlp_small_backward_jobs_one_launch(const crf_mlp_bwd_job* jobs, int njobs, unsigned* ticket, unsigned* sync_ws,
                                               crf_stream_t stream);
/* Up to 8 independent products C_j = A_j B_j (B_j [K_j, N_j]; N, K multiples of 4) per launch -- the g_h1 = g_h2 W2 products of all wide
 * PointConv layers of a backward pass; same tiles and summation order as crfconv_gemm on each.  jobs is a host array. */
typedef struct { const float* A; const float* B; float* C; int64_t M; int N; int K; } crf_gemm_job;
int crfconv_gemm_jobs(const crf_gemm_job* jobs, int njobs, crf_stream_t stream);
int crfconv_gemm_supported(int64_t M, int N, int K);
int crfconv_gemm(const float* A, const float* B, const float* bias, const float* addend, int64_t M, int N, int K,
                 int b_is_nk, float* C, crf_stream_t stream);
int crfconv_bn_coef_from_records(const float* stat_rec, int64_t M, int C, const float* gamma, const float* beta,
                                 float* run_mean, float* run_var, float momentum, float eps, float* coef,
                                 crf_stream_t stream);
/* crfconv_bn_coef_from_nrecords + crfconv_bn_apply in one launch (every workgroup combines the records of its own 16 channels, then
 * applies y = lrelu(a x + b (+ skip), slope) to its row tile; skip != NULL = crfconv_bn_apply_add, the ResNet join): identical coef,
 * running statistics and y. */
int crfconv_bn_apply_from_records(const float* stat_rec, int64_t nrec, const float* x, int64_t M, int C, const float* gamma,
                                  const float* beta, float* run_mean, float* run_var, float momentum, float eps, const float* skip,
                                  float slope, float* coef, float* y, crf_stream_t stream);
/* The same with an explicit record count (the records of crfconv_gemm_stats: one per 16-row group). */
int crfconv_bn_coef_from_nrecords(const float* stat_rec, int64_t nrec, int64_t M, int C, const float* gamma, const float* beta,
                                  float* run_mean, float* run_var, float momentum, float eps, float* coef, crf_stream_t stream);

/* Q = M^-1 for the symmetric positive definite M = I + c^T c of a CRF layer (H <= 64; Gauss-Jordan in
 * float64, one workgroup, no host sync -- capturable into a hipGraph, unlike a LAPACK-style inverse). */
int crfconv_spd_inverse(const float* M, int H, float* Q, crf_stream_t stream);
/* The same for 64 < H <= 512 (the wide CRF stages of the sparse networks, models/point_conv.py:318-339): Gauss-Jordan without
 * pivoting on one workgroup, matrix in global memory; M symmetric positive definite (M = I + c^T c), M != Q. */
int crfconv_spd_inverse_wide(const float* M, int H, float* Q, crf_stream_t stream);
/* Both loop-invariant matrices of a CRF layer from its compatibility factor c [H, H] (C = c^T c) in one launch:
 *   Q = (I + C)^-1,  P = C Q = I - Q          (continuous_crf_conv_big.py:67-72: (z + m C)(I + C)^-1 = z Q + m P)
 * and the matching backward: dc from dQ / dP (either may be NULL). */
int crfconv_crf_matrices(const float* c, int H, float* Q, float* P, crf_stream_t stream);
int crfconv_crf_matrices_backward(const float* c, const float* Q, const float* dQ, const float* dP, int H,
                                  float* dc, crf_stream_t stream);
/* The same for n <= 8 layers in ONE launch each way (one workgroup per layer: the Gauss-Jordan sweep is a ~20 us latency
 * chain whatever H is).  Host arrays of n device pointers / sizes; gQ[i] / gP[i] may be NULL (= zero). */
int crfconv_crf_matrices_batched(const float* const* c, const int* H, int n, float* const* Q, float* const* P,
                                 crf_stream_t stream);
int crfconv_crf_matrices_backward_batched(const float* const* c, const float* const* Q, const float* const* gQ,
                                          const float* const* gP, const int* H, int n, float* const* dc,
                                          crf_stream_t stream);

/* ===================================================================== (B) pooling / up-sampling
 * out[i,c] = max_k x[idx32[i,k], c];  arg [m_tgt, C] int32 = winning k (first maximum). */
int crfconv_neighbor_maxpool_forward(const float* x, const int32_t* idx32, int K, int64_t m_tgt,
                                     int C, float* out, int32_t* arg, crf_stream_t stream);
/* The strided shortcut of a ResNet block (models/point_conv_big.py:74-83: shortcut MLP, then max over the sub_idx
 * neighbours) without the normalised fine-level tensor: out[i,c] = max_k (a[c] x[idx32[i,k], c] + b[c]) with coef = the
 * [4, C] block of crfconv_bn_coef_from_records for x.  arg as above; the backward is crfconv_neighbor_maxpool_backward. */
int crfconv_neighbor_maxpool_affine_forward(const float* x, const float* coef, const int32_t* idx32, int K, int64_t m_tgt,
                                            int C, float* out, int32_t* arg, crf_stream_t stream);
/* dx[j,c] = sum over incoming edges (i,k) with arg[i,c] == k of gout[i,c]. */
int crfconv_neighbor_maxpool_backward(const float* gout, const int32_t* arg,
                                      const int32_t* rev_ptr, const int32_t* rev_eid, int K,
                                      int64_t m_src, int C, float* dx, crf_stream_t stream);
/* out[i] = x[idx32[i]]  (K = 1 table: nearest up-sampling, point_conv_big.py:97-101). */
int crfconv_gather_rows(const float* x, const int32_t* idx32, int64_t m_tgt, int C, float* out,
                        crf_stream_t stream);
/* dx[j] = sum_{i in rev(j)} gout[i]. */
int crfconv_gather_rows_backward(const float* gout, const int32_t* rev_ptr,
                                 const int32_t* rev_eid, int64_t m_src, int C, float* dx,
                                 crf_stream_t stream);

/* out = lrelu(a + b, slope) over n floats (n % 4 == 0): the residual join of the ResNet block
 * (point_conv_big.py:86-88); backward gin = gout * (out > 0 ? 1 : slope), shared by both addends. */
int crfconv_add_lrelu(const float* a, const float* b, int64_t n, float slope, float* out, crf_stream_t stream);
/* The classifier's  MLP -> nn.Dropout(p)  (models/point_conv_big.py:131-134) without the intermediate tensor:
 * out = dropout(lrelu(a x + b, slope), p).  The mask is counter-based -- element e is kept iff a hash of (seed, *counter, e)
 * reaches p 2^32 -- with `counter` one int64 DEVICE word the caller advances between training steps (so a captured graph
 * draws a new mask at every replay); crfconv_dropout_backward(g, n = M C, p, seed, counter) applies the same mask to the
 * gradient, nothing is stored.  Kept elements are scaled by 1 / (1 - p).  counter_used (may be NULL): one int64 device
 * word that receives the counter value this call masked with -- hand THAT to the backward, so that a second forward before
 * the first backward (which advances the live counter) cannot change the first one's mask. */
int crfconv_bn_apply_dropout(const float* x, int64_t M, int C, const float* coef, float slope, float p, uint64_t seed,
                             const int64_t* counter, float* out, int64_t* counter_used, crf_stream_t stream);
int crfconv_dropout_backward(const float* g, int64_t n, float p, uint64_t seed, const int64_t* counter, float* gin,
                             crf_stream_t stream);
/* The same mask applied by the kernel that PRODUCES g: Y = mask .* (X W^T, or X W when transpose_w != 0) / (1 - p), element
 * e = row * Co + column -- the input gradient of the Linear behind the dropout (the classifier's last layer,
 * models/point_conv_big.py:131-134), so that no separate pass over [M, Co] runs.  Shapes as crfconv_linear_forward. */
int crfconv_linear_forward_dropout(const float* X, const float* W, int64_t M, int Ci, int Co, int transpose_w, float p,
                                   uint64_t seed, const int64_t* counter, float* Y, crf_stream_t stream);
/* The per-point classifier, models/point_conv_big.py:131-134: MLP(Ci -> Co = 4 Ci: Linear, BatchNorm, LeakyReLU) -> nn.Dropout(p)
 * -> nn.Linear(Co -> C2), training mode, without ever storing a [M, Co] tensor: every pass recomputes X W1^T on the matrix pipe
 * (bit-identical each time).  Supported: Co = 128, Ci in {16, 32}, C2 <= 16.
 *   stats:    stat_rec [crfconv_head_stat_records(M)][Co][4] = the BatchNorm statistic records of X W1^T (the tuples of
 *             crfconv_linear_forward) WITHOUT the product being stored; crfconv_bn_coef_from_nrecords turns them into
 *   forward:  coef [4, Co];
 *             logits [M, C2] = dropout(lrelu(a (X W1^T) + b, slope)) W2^T + b2 (b2 may be NULL) with the counter-based mask of
 *             crfconv_bn_apply_dropout (same seed / counter / element numbering e = row * Co + channel: identical logits);
 *             mask_bits [crfconv_head_mask_words(M)] uint32 receives the mask (one bit per element) for the backward;
 *             counter_used as in crfconv_bn_apply_dropout.
 *   backward: g [M, C2] = d loss / d logits  ->  dX [M, Ci] (may be NULL), dW1 [Co, Ci], dgamma / dbeta [Co], dW2 [C2, Co],
 *             db2 [C2] (may be NULL); workspace of crfconv_head_backward_workspace bytes.  Four launches: partial pass, float64
 *             totals, dX, parameters. */
int crfconv_head_supported(int64_t M, int Ci, int Co, int C2);
size_t crfconv_head_mask_words(int64_t M);
size_t crfconv_head_stat_records(int64_t M);
int crfconv_head_stats(const float* X, const float* W1, int64_t M, int Ci, int Co, float* stat_rec, crf_stream_t stream);
size_t crfconv_head_backward_workspace(int64_t M, int Ci, int Co, int C2);
int crfconv_head_forward(const float* X, const float* W1, const float* coef, float slope, float p, uint64_t seed,
                         const int64_t* counter, const float* W2, const float* b2, int64_t M, int Ci, int Co, int C2,
                         float* logits, uint32_t* mask_bits, int64_t* counter_used, crf_stream_t stream);
int crfconv_head_backward(const float* g, const float* X, const float* W1, const float* coef, float slope, float p,
                          const float* W2, const uint32_t* mask_bits, int64_t M, int Ci, int Co, int C2, float* dX, float* dW1,
                          float* dgamma, float* dbeta, float* dW2, float* db2, void* workspace, size_t workspace_bytes,
                          crf_stream_t stream);
/* The ResNet join of models/point_conv_big.py:84-88 in one pass: out = lrelu(a x + b + skip, slope), coef = the [4, C]
 * block of crfconv_bn_forward / crfconv_bn_coef_from_records for x (a BatchNorm without activation), skip / out [M, C].
 * Same arithmetic as crfconv_bn_apply(slope 1) followed by crfconv_add_lrelu, without the intermediate tensor. */
int crfconv_bn_apply_add(const float* x, int64_t M, int C, const float* coef, const float* skip, float slope, float* out,
                         crf_stream_t stream);
int crfconv_add_lrelu_backward(const float* gout, const float* out, int64_t n, float slope, float* gin,
                               crf_stream_t stream);

/* ===================================================================== (B) training loss
 * Replaces trainval.py:101-104: F.cross_entropy(y_pred, data.y.reshape(-1) - 1, weight=class_weights,
 * ignore_index=ignore).  logits [m, C] float32; target [m] int64, the class of row r is target[r] - label_shift
 * (label_shift = 1 folds the reference's "- 1" into the kernel); weight [C] float32 or NULL.
 *   loss = sum_r w[t_r] (lse_r - logits[r, t_r]) / sum_r w[t_r]   over rows with t_r != ignore_index
 * Outputs: lse [m] (kept for the backward), sums [3] float64 = {sum w*nll, sum w, rows whose class is outside
 * [0, C) and not ignore_index -- counted, contribute nothing}, loss [1] float32 (device).
 * Backward: dlogits[r,c] = grad_loss[0] * w[t_r] * (softmax(logits[r])[c] - [c == t_r]) / sums[1]. */
size_t crfconv_softmax_ce_workspace(int64_t m);
int crfconv_softmax_ce_forward(const float* logits, const int64_t* target, const float* weight, int64_t m, int C,
                               int64_t ignore_index, int64_t label_shift, float* lse, double* sums, float* loss,
                               void* workspace, size_t workspace_bytes, unsigned* ticket, crf_stream_t stream);
/* ticket: crfconv_ticket_bytes() of zero device words per stream (left zero) -- the last workgroup of the forward folds the per-block
 * sums (one launch); NULL: the fold as a second launch.  Same sums, same order, either way. */
int crfconv_softmax_ce_backward(const float* logits, const int64_t* target, const float* weight, const float* lse,
                                const double* sums, const float* grad_loss, int64_t m, int C, int64_t ignore_index,
                                int64_t label_shift, float* dlogits, crf_stream_t stream);

/* torch.optim.SGD step (trainval.py:69-72, 105) over one flat float32 parameter vector of n elements:
 *   g = grad + weight_decay * p;  buf = first_step ? g : momentum * buf + (1 - dampening) * g;
 *   g = nesterov ? g + momentum * buf : buf;  p -= lr * g          (momentum == 0: buf unused, may be NULL). */
int crfconv_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                     float dampening, float weight_decay, int nesterov, int first_step, crf_stream_t stream);
/* The same with hyper = {lr, momentum, dampening, weight_decay, grad_scale} (5 floats) in DEVICE memory, so that a
 * captured graph of the step honours a learning-rate scheduler; grad is multiplied by grad_scale first (1 / world size
 * when the bucket holds the all-reduce SUM; 1 otherwise); momentum_buf is required (zero-filled when momentum is 0). */
int crfconv_sgd_step_hyper(float* param, const float* grad, float* momentum_buf, int64_t n, const float* hyper,
                           int nesterov, int first_step, crf_stream_t stream);
/* The same, guarded by the grid-barrier failure word (crfconv_gridsync_fail_word() of the caller's barrier workspace, device
 * memory; NULL = unguarded): while *fail_word != 0 the launch changes NOTHING -- a one-launch kernel whose barrier timed out has
 * NaN-poisoned its outputs and therefore this step's gradient, and parameters / momentum must survive until the host reads the
 * word (it is sticky; captured replays are protected too). */
int crfconv_sgd_step_guarded(float* param, const float* grad, float* momentum_buf, int64_t n, const float* hyper,
                             int nesterov, int first_step, const unsigned* fail_word, crf_stream_t stream);
/* The guard over SEVERAL sticky words (fail_words: HOST array of n_fail_words <= 8 device pointers -- every barrier workspace of
 * the device: eager streams and the captured-graph buffer each have their own) and over reduced_flag (device float, may be NULL):
 * the slot behind the flat gradient bucket that every rank fills with crfconv_sgd_guard_publish BEFORE the gradient all-reduce and
 * that the all-reduce sums with the bucket -- non-zero on every rank when any rank's step failed, so all replicas skip together
 * (the failed rank's NaN gradient is in everybody's bucket by then). */
int crfconv_sgd_step_guarded_all(float* param, const float* grad, float* momentum_buf, int64_t n, const float* hyper,
                                 int nesterov, int first_step, const unsigned* const* fail_words, int n_fail_words,
                                 const float* reduced_flag, crf_stream_t stream);
/* *slot = 1.0f when any of the words is set, else 0.0f (one tiny launch; capture-safe). */
int crfconv_sgd_guard_publish(const unsigned* const* fail_words, int n_fail_words, float* slot, crf_stream_t stream);

/* ---- device-side pieces of the collate (datasets/semantic3d_dataset.py:512-528), csrc/collate.hip
 * crfconv_random_subsets: for each level l < nlevels, out[l][0 .. s[l]) (device int64) = a uniformly random subset of
 *   {0 .. n[l]-1} of size s[l] in ASCENDING order -- `torch.randperm(n)[: n // ratio]` of :517 followed by the sort the
 *   device collate applies.  Counter-based: the subset is a function of (seed, *counter, l) -- *counter is a DEVICE word the
 *   caller advances per batch, so a captured graph draws new subsets at every replay.  The draws are NOT torch.randperm's.
 *   n, s, out are host arrays; n[l] <= 2^20; nlevels <= 8.
 * crfconv_argsort_codes: order [B, N] int64 = per-cloud stable argsort of code [B, N] int64 (30-bit Morton codes):
 *   bit-identical to torch.argsort(code, dim=1, stable=True), without scratch memory (capturable on ROCm 7.2). */
int crfconv_random_subsets(const int* n, const int* s, int64_t* const* out, int32_t* const* rank, int nlevels, uint64_t seed,
                           const int64_t* counter, crf_stream_t stream);
/* rank (host array of device int32 [n[l]] pointers, or NULL; entries may be NULL): rank[l][i] = position of point i in subset l, -1
 * outside -- the membership table of crfconv_upindex_from_table.
 * crfconv_upindex_from_table: up_idx [B, N] int64 = for every point of a level the position (in the subset shared by the B clouds, whose
 *   positions are sub_pos) of the nearest subset member -- knn_batch(sub_pos, pos, 1) of datasets/semantic3d_dataset.py:524, bit-identical incl.
 *   the (distance, position) order on ties --, answered from the level's own K-nearest table neighbor_idx [B, N, K] int64 (distance
 *   order, as crfconv_knn_batch_dev writes it) where a member is in it; the ~1 % of points without one go onto a list and a second launch
 *   gives each a wavefront that scans sub_pos [B, S, 3] (round 5: two launches per level instead of a grid build + search).
 *   rank [N] int32: the subset's membership table (position in the subset, -1 outside; crfconv_random_subsets writes it).
 *   workspace: crfconv_upindex_workspace(B, N) bytes whose first 8 are ZERO before the first use (the launches leave them zero). */
size_t crfconv_upindex_workspace(int64_t B, int64_t N);
int crfconv_upindex_from_table(const float* pos, const float* sub_pos, const int64_t* neighbor_idx, const int32_t* rank, int64_t B, int64_t N,
                               int K, int64_t S, int64_t* up_idx, void* workspace, size_t workspace_bytes, crf_stream_t stream);
/* Rows of up to 8 [B, N, row_bytes[j]] tensors picked by one index list in one launch: dst[j][b][s] = src[j][b][index[s]]
 * (index [S], shared by all clouds: datasets/semantic3d_dataset.py:524-526 pos[:, choice], neighbor_idx[:, choice]) or
 * src[j][b][index[b][s]] (per_cloud != 0: the Morton permutation, farthest-point picks).  row_bytes multiples of 4; src, dst,
 * row_bytes are host arrays; out-of-range picks are clamped. */
int crfconv_gather_rows_batched(const void* const* src, void* const* dst, const int* row_bytes, int njobs, const int64_t* index,
                                int per_cloud, int64_t B, int64_t N, int64_t S, crf_stream_t stream);
size_t crfconv_argsort_codes_workspace(int64_t B, int64_t N);
int crfconv_argsort_codes(const int64_t* code, int64_t B, int64_t N, int64_t* order, void* workspace,
                          size_t workspace_bytes, crf_stream_t stream);

/* ===================================================================== (B) discrete (label-space) CRF layer
 * models/discrete_crf_conv.py:40-63.  One mean-field step with GIVEN edge weights s [m, K] (edge-id addressed like
 * every per-edge array; entries of idx32 < 0 = no neighbour):  xout = z Q + (sum_k s_ik xin[idx32[i,k]]) P.
 * The layer's update  q <- softmax(-u - (sum_e w_e q_j) C)  (:58-60) is this with z = -u, Q = I, P = -C and a row
 * soft-max on top.  Backward through crfconv_meanfield_bwd_edge / _bwd_scatter. */
int crfconv_meanfield_step(const float* xin, const float* z, const float* s, const int32_t* idx32, int K, int k0,
                           int64_t m, int H, const float* Q, const float* P, float* xout, crf_stream_t stream);

/* Edge weights of :49-54:  w[i,k] = sum_g Wg[g] exp(-| fk[j, g, :] - fk[i, g, :] |^2),  j = idx32[i,k]  (0 where
 * j < 0);  fk [m, G * H] = f F_g for the G <= 8 Gaussian kernels, H <= 256 hidden channels.
 * Backward: gw [m, K] -> dfk [m, G * H] (dfk_self is scratch of the same size; the source side walks the reverse
 * CSR, no atomics) and dW_partial [crfconv_kernel_weights_partials(m)] float64 block partials laid out [block][8]
 * whose column sums over blocks are dWg[0..G). */
int crfconv_kernel_weights_forward(const float* fk, const int32_t* idx32, int K, const float* Wg, int G, int H,
                                   int64_t m, float* w, crf_stream_t stream);
size_t crfconv_kernel_weights_partials(int64_t m);
int crfconv_kernel_weights_backward(const float* gw, const float* fk, const int32_t* idx32, const int32_t* rev_ptr,
                                    const int32_t* rev_eid, int K, const float* Wg, int G, int H, int64_t m,
                                    float* dfk_self, float* dfk, double* dW_partial, crf_stream_t stream);

/* ===================================================================== (C) callers either side of the network
 * SURVEY 8(f) rows 2-3.  All device pointers; nothing here synchronises.
 *
 * Confusion matrix, utils/metrics.py:13-27 (runningScore._fast_hist + update): for every row r with class
 * t = y_true[r] - label_shift such that 0 <= t < n_class and t != ignore_index,  hist[t * n_class + p] += 1, where
 * p = y_pred[r], or the FIRST arg-max of logits[r, 0..n_class) when logits != NULL (trainval.py:108).  hist is
 * int64 [n_class, n_class], accumulated.  Rows whose p falls outside [0, n_class) are skipped and counted in
 * *bad_count (np.bincount would widen and the reference's reshape raise). */
int crfconv_confusion_accumulate(const int64_t* y_true, const int64_t* y_pred, const float* logits, int64_t n_rows,
                                 int n_class, int64_t ignore_index, int64_t label_shift, int64_t* hist,
                                 int32_t* bad_count, crf_stream_t stream);

/* Vote accumulator, trainval.py:186-189: test_probs[point_idx[r]] = smooth * test_probs[point_idx[r]] +
 * (1 - smooth) * prob[r] in float32 exactly as numpy evaluates it (two rounded products, one rounded sum);
 * prob = probs[r] (float32 [n_rows, C]) or, when logits != NULL, soft-max of logits[r] (trainval.py:178).
 * point_idx int64 [n_rows], distinct within one call; entries outside [0, n_cloud) are skipped and counted. */
int crfconv_vote_accumulate(const float* probs, const float* logits, const int64_t* point_idx, int64_t n_rows, int C,
                            double smooth, float* test_probs, int64_t n_cloud, int32_t* bad_count, crf_stream_t stream);
/* The same, counting the updates of every point in visits [n_cloud] (int32, the caller's zeros at the start), and the merge of two tables
 * accumulated apart:  acc <- what ONE accumulator holds that applied acc's updates first and `later`'s after them (a running mean
 * v <- s v + (1 - s) p applied n times scales what was there by s^n:  acc[p] = acc[p] s^later_visits[p] + later[p];  acc_visits +=
 * later_visits).  crfconv_amd.sampling.VoteAccumulator.merge() folds the ranks' tables in rank order after an all-gather: crops of one
 * scene sharded over GPUs give the votes of a single GPU that saw rank 0's crops first, then rank 1's (trainval.py:188-189 is
 * order-dependent; this fixes the order). */
int crfconv_vote_accumulate_counted(const float* probs, const float* logits, const int64_t* point_idx, int64_t n_rows, int C, double smooth,
                                    float* test_probs, int64_t n_cloud, int32_t* bad_count, int32_t* visits, crf_stream_t stream);
int crfconv_vote_fold(float* acc, int32_t* acc_visits, const float* later, const int32_t* later_visits, int64_t n, int C, double smooth,
                      crf_stream_t stream);

/* Re-projection, trainval.py:200-203: preds[i] = uint8(first arg-max of test_probs[proj_idx[i]]) + label_offset. */
int crfconv_vote_project(const float* test_probs, const int64_t* proj_idx, int64_t n_proj, int C, int64_t n_cloud,
                         int label_offset, uint8_t* preds, int32_t* bad_count, crf_stream_t stream);

/* np.argmin / np.min of a float64 array (first index on ties): semantic3d_dataset.py:424-425, 451. */
size_t crfconv_argmin_workspace(void);
int crfconv_argmin_f64(const double* values, int64_t n, double* out_value, int64_t* out_index, void* workspace,
                       size_t workspace_bytes, crf_stream_t stream);

/* One draw of the possibility sampler, semantic3d_dataset.py:426-450, for a cloud of n float32 points:
 *   centre = float64(points[*pick_index]) + noise[0..3)   (noise may be NULL)
 *   crop   = the k points nearest to centre by float64 squared distance (ties: lower point id first)
 *   d      = float32 distances formed as the reference does (float64 squares rounded to float32, summed in float32)
 *   possibility[q] += float64((1 - d / max d)^2) * point_weight[q]   (point_weight NULL = test split, weight 1)
 * Outputs, row t describing crop element perm[t] (perm = a permutation of 0..k-1, NULL = nearest first; the
 * reference shuffles): out_idx int64 [k]; out_xyz float32 [k, 3] with x, y centred on the seed (:436-437);
 * out_center float64 [3] (may be NULL). */
size_t crfconv_possibility_crop_workspace(int64_t n, int64_t k);
int crfconv_possibility_crop(const float* points, int64_t n, int64_t k, const int64_t* pick_index, const double* noise,
                             const int64_t* perm, const double* point_weight, double* possibility, int64_t* out_idx,
                             float* out_xyz, double* out_center, void* workspace, size_t workspace_bytes,
                             crf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CRFCONV_AMD_H */
