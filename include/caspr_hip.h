/*
 * caspr_hip.h -- C ABI of libcaspr_hip.so: the MI355X (gfx950) kernels behind the CaSPR
 * encode -> advect -> sample path.
 *
 * The reference (davrempe/caspr) has no FFI layer of its own: its operator boundary is the set of
 * Python symbols it imports from third-party CUDA extensions.  Each entry point below names the
 * reference call site (file:line under /root/reference/caspr) whose operator it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch tensors); the library
 *     never allocates, frees or retains pointers; no global state apart from the last-error string;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - functions are re-entrant and never call hipSetDevice (one process per GPU);
 *   - return 0 on success, a negative CASPR_E* code otherwise (caspr_last_error_string() explains);
 *   - activations are POINT-MAJOR f32: (batch, points, channels) with an explicit row stride `ld*`
 *     (multiple of 4 floats).  The reference's channels-first (B,C,P) tensors are the transposed
 *     view of the same data;
 *   - index tensors are int32, row-major, contiguous;
 *   - "packed" weights are produced by caspr_pack_weight_f32 (MFMA A-fragment order).
 */
#ifndef CASPR_HIP_H
#define CASPR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASPR_OK 0
#define CASPR_EINVAL (-1)   /* bad argument (shape / alignment / null pointer)         */
#define CASPR_ELAUNCH (-2)  /* hipLaunchKernel / hipGetLastError reported a failure    */
#define CASPR_EUNSUP (-3)   /* shape outside what the kernels are instantiated for     */

const char *caspr_last_error_string(void);
int caspr_abi_version(void);

/* ---------------- input preparation: models/tpointnet2.py:75,79-90 ------------------------------
 * x (B,T,N,4) [x,y,z,t] -> xyz (B*T,N,3) ; feat (B*T,N,8) = [x^2,y^2,z^2,xz,xy,yz,0,0] (or fewer
 * terms: quad/pairs flags as TPointNet2.augment_quad / augment_pairs; row stride 8, zero padded).
 * The global PointNet reads x itself (point-major (B, T*N, 4)).                                   */
int caspr_prep_input_f32(const float *x, int BT, int N, int quad, int pairs, float *xyz, float *feat,
                         void *stream);

/* ---------------- Kaolin furthest_point_sampling + fps_gather_by_index: models/pointnet2.py:384-387
 * xyz (B,n,3) -> idx (B,M) int32 and (optionally, may be NULL) new_xyz (B,M,3) = xyz[idx].
 * Contract = oracle/point_ops.c:oracle_fps (start index 0, temp=1e10, padding guard, tie rule).
 * n <= 36,864 (Kaolin's kernel takes any n; every configuration of the reference has n <= 4096): clouds up to 4096 points
 * stay in registers, larger ones keep their running minimum in LDS (one cloud per workgroup either way); the same
 * selections bit for bit.  EINVAL above the limit.                                                   */
int caspr_fps_f32(const float *xyz, int B, int n, int M, int guard, int32_t *idx, float *new_xyz,
                  void *stream);

/* Kaolin fps_gather_by_index on point-major features: out[b,j,:] = feat[b,idx[b,j],:]  (pointnet2.py:385) */
int caspr_gather_points_f32(const float *feat, int ldf, const int32_t *idx, int B, int n, int M, int C,
                            float *out, int ldo, void *stream);

/* ---------------- Kaolin ball_query inside PointNet2GroupingLayer: models/pointnet2.py:340-342,391
 * xyz (B,n,3), new_xyz (B,M,3) -> idx (B,M,ns) int32.  Contract = oracle_ball_query.              */
int caspr_ball_query_f32(const float *xyz, const float *new_xyz, int B, int n, int M, float radius,
                         int ns, int32_t *idx, void *stream);
/* The two queries of a set-abstraction level (PointNet2SetAbstraction builds one grouper per radius over the same xyz / new_xyz:
 * models/pointnet2.py:338-342, both called at :391) in one pass over the cloud: idx_a (B,M,ns_a), idx_b (B,M,ns_b), each row what
 * caspr_ball_query_f32 writes for its radius, bit for bit.                                                                      */
int caspr_ball_query2_f32(const float *xyz, const float *new_xyz, int B, int n, int M, float radius_a, int ns_a, int32_t *idx_a,
                          float radius_b, int ns_b, int32_t *idx_b, void *stream);

/* Kaolin group_gather_by_index + centre subtraction + xyz||feat concat (pointnet2.py:391-398):
 * out (B,M,3+C,ns) exactly as the reference's grouper returns it.  feat point-major (B,n,C), may be NULL. */
int caspr_group_points_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                           const int32_t *idx, int B, int n, int M, int C, int ns, float *out,
                           void *stream);

/* ---------------- fused grouper + PointNetFeatureExtractor: models/pointnet2.py:391-409,649-703
 * For every centre: gather the ns neighbours (xyz - centre || feat), run 3 x (conv1d k=1 ->
 * GroupNorm(16) per neighbourhood -> ReLU [not after the last]) and max over the ns samples.
 * Weights are packed with caspr_pack_weight_f32 from the K-permuted matrix [feat (C, padded to 4) |
 * xyz (3) | 0...].  out[b, m, out_off : out_off+C3] (row stride ldo).
 * feat_kind: 0 = generic features; CASPR_FEAT_QUAD | CASPR_FEAT_PAIRS = `feat` is the quadratic augmentation of `xyz`
 *   written by caspr_prep_input_f32 ([x2 y2 z2] then [xz xy yz], tpointnet2.py:79-90).  The first layer runs on inputs
 *   centred on a sample of the neighbourhood (an exact reformulation in front of the per-neighbourhood GroupNorm); with
 *   feat_kind set the centred features come straight from the coordinates, x^2 - x0^2 = (x - x0)(x + x0), instead of
 *   from the difference of two rounded squares -- same function, closer to its exact value.                       */
#define CASPR_FEAT_QUAD 1
#define CASPR_FEAT_PAIRS 2
/* Low parts (round 5; the register kernel's shapes only: all three widths <= 64).  A ball that holds ONE point yields the GroupNorm
 * chain of that point's features alone, and a group of two channels that differ by ~1e-4 multiplies the f32 ROUNDING OF THE INPUT
 * by up to 1 / (2 sqrt(eps)) = 158; so a level may hand its output to the next one as an unevaluated sum hi + lo:
 *   CASPR_FEAT_LO_OUT: out rows are [channels (ldo / 2) | their low parts (ldo / 2)]; out[.., ldo / 2 + out_off + c] receives what
 *     the f32 value lacks of the kernel's internal f64 result (meaningful where that result is f64: balls of <= 4 distinct points);
 *   CASPR_FEAT_LO_IN: feat rows are [C channels .. | low parts at column ldf / 2 ..]; the f64 reference column and the f64
 *     re-evaluation of small balls read hi + lo (the MFMA's deviation columns read hi).  Not with QUAD / PAIRS (those take the exact
 *     f64 products of the coordinates for the same purpose).                                                                        */
#define CASPR_FEAT_LO_IN 4
#define CASPR_FEAT_LO_OUT 8
/* The call in two halves, for a caller that wants them on two streams (caspr_sa_mlp_max_ws_f32 with a workspace only: the list in it is what
 * makes the halves' output rows disjoint): CASPR_SA_ONLY_MFMA = the list + the MFMA kernel (every neighbourhood the f64 re-evaluation does
 * not take; on the LDS kernel's shapes: everything), CASPR_SA_ONLY_F64 = the f64 re-evaluation of the small balls alone (a no-op on shapes
 * that have none).  The two calls together write exactly what the plain call writes, bit for bit, in any order.                        */
#define CASPR_SA_ONLY_MFMA 16
#define CASPR_SA_ONLY_F64 32
int caspr_sa_mlp_max_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                         const int32_t *idx, int B, int n, int M, int C, int ns, int feat_kind,
                         const float *w1p, const float *b1, const float *g1, const float *be1, int C1,
                         const float *w2p, const float *b2, const float *g2, const float *be2, int C2,
                         const float *w3p, const float *b3, const float *g3, const float *be3, int C3,
                         float *out, int ldo, int out_off, void *stream);
/* The same call with a scratch of caspr_sa_mlp_max_workspace_ints(B, M) 32-bit integers (device memory, contents irrelevant on entry,
 * undefined on return; must stay valid until the call's kernels have run on `stream`).  On the register kernel's shapes (all widths
 * <= 64) the neighbourhoods whose result the f64 re-evaluation of small balls replaces anyway are listed there first and the MFMA
 * kernel computes only the others -- same outputs, bit for bit (a NULL workspace = caspr_sa_mlp_max_f32).                          */
long caspr_sa_mlp_max_workspace_ints(int B, int M);
int caspr_sa_mlp_max_ws_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                            const int32_t *idx, int B, int n, int M, int C, int ns, int feat_kind,
                            const float *w1p, const float *b1, const float *g1, const float *be1, int C1,
                            const float *w2p, const float *b2, const float *g2, const float *be2, int C2,
                            const float *w3p, const float *b3, const float *g3, const float *be3, int C3,
                            float *out, int ldo, int out_off, int32_t *workspace, void *stream);
/* The same level with its first layer PRE-AGGREGATED (the LDS kernel's shapes: C1 >= 64).  Layer 1 is linear in front of its GroupNorm and
 * its feature part does not depend on the centre -- W [p - c ; f] = W_x (p - c) + W_f f -- so W_f f is computed once per SOURCE point by a
 * plain conv over the level's n points (pre (B, n, ldp), no bias) instead of once per (centre, sample) pair; the kernel gathers pre rows
 * (C1 floats per sample instead of C + 3), adds wx1 (C1, 3) . (p - c) + b1 and continues with GroupNorm 1, layers 2 and 3 and the max.
 * An exact reformulation of pointnet2.py:391-409,677-698 (the two parts of the dot product are rounded separately: ~1e-7).           */
int caspr_sa_mlp_max_pre_f32(const float *xyz, const float *new_xyz, const float *pre, int ldp, const int32_t *idx, int B, int n,
                             int M, int ns, const float *wx1, const float *b1, const float *g1, const float *be1, int C1,
                             const float *w2p, const float *b2, const float *g2, const float *be2, int C2,
                             const float *w3p, const float *b3, const float *g3, const float *be3, int C3,
                             float *out, int ldo, int out_off, void *stream);
/* The pre-aggregated first layer as ROWS (the row-materialised form of the coarsest level, models/pointnet2.py: _run_rows):
 * Y[(b*M+j)*ns+s, 0:C1] = pre[b, idx[b,j,s], 0:C1] + wx (C1,3) . (xyz[b, idx[b,j,s]] - new_xyz[b,j]) + bias -- the raw output of layer 1. */
int caspr_group_rows_pre_f32(const float *xyz, const float *new_xyz, const float *pre, int ldp, const int32_t *idx, int B, int n,
                             int M, int C1, int ns, const float *wx, const float *bias, float *Y, int ldy, void *stream);



/* ---------------- Kaolin three_nn + inverse-distance weights: models/pointnet2.py:514-518
 * unknown (B,n,3), known (B,m,3) -> dist (B,n,3) [sqrt], idx (B,n,3), weight (B,n,3) (may be NULL) */
int caspr_three_nn_f32(const float *unknown, const float *known, int B, int n, int m, float *dist,
                       int32_t *idx, float *weight, void *stream);

/* Kaolin three_interpolate + the concat with the skip features (pointnet2.py:519-523), point-major:
 * out[b,i,0:C] = sum_k w[b,i,k] * feat[b,idx[b,i,k],0:C] ; out[b,i,C:C+C2] = skip[b,i,0:C2] ;
 * columns up to ldo are zero-filled.  skip may be NULL (C2 = 0).  in_scale/in_shift (B,C) (may be
 * NULL): feat is read as max(feat*scale+shift, 0) (in_relu) -- the producer's GroupNorm+ReLU.       */
int caspr_three_interp_f32(const float *feat, int ldf, const int32_t *idx, const float *weight,
                           const float *in_scale, const float *in_shift, int in_relu, const float *skip,
                           int lds, int B, int m, int n, int C, int C2, float *out, int ldo, void *stream);
/* Feature propagation with its first conv on the COARSE level (pointnet2.py:514-525: three_interpolate -> concat skip -> conv -> GroupNorm;
 * interpolation and a pointwise conv commute).  u (B,m,ldu) = W_p h over the coarse rows (caspr_conv1x1_*, no bias; W_p = the conv's first
 * C_prev input columns); y[b,i,0:C] = sum_k weight[b,i,k] * u[b,idx[b,i,k],0:C] + wskip (C x C2, row-major: the conv's last C2 input columns)
 * . skip[b,i,0:C2] + bias, C2 <= 8; scale / shift (B,C) = the GroupNorm(G) of y folded with gamma / beta, as caspr_conv1x1_gn_* return them.
 * ws: caspr_three_interp_add_gn_ws_bytes(B, n, G) bytes of device scratch.                                                           */
long caspr_three_interp_add_gn_ws_bytes(int B, int n, int G);
int caspr_three_interp_add_gn_f32(const float *u, int ldu, const int32_t *idx, const float *weight, const float *skip, int lds,
                                  int C2, const float *wskip, const float *bias, int B, int m, int n, int C, float *y, int ldy,
                                  int G, const float *gamma, const float *beta, float eps, float *scale, float *shift, void *ws,
                                  long ws_bytes, void *stream);


/* ---------------- pointwise conv (nn.Conv1d k=1 / nn.Linear) on MFMA f32 --------------------------
 * Replaces the cuDNN/cuBLAS calls behind pointnet.py:37-41, pointnet2.py:525,247,
 * tpointnet2.py:99-105.   Y[b,p,co] = act( sum_k W[co,k] * in(X[b,p,k]) + bias[co] + bbias[b,co] )
 *   in(x) = x                                   if in_scale == NULL
 *         = max(x*in_scale[b,k]+in_shift[b,k],0) (in_relu, for k >= in_relu_from) or without the max -- the previous
 *           layer's GroupNorm(+ReLU) folded into the operand load (scale/shift from caspr_gn_stats_f32)
 *   act  = identity (0) or sigmoid (1), optionally | CASPR_CONV_ROW_INVARIANT: the result of a row then depends on that
 *          row's data only -- not on P or on the row's position in the batch entry (same kernel, same K order for every
 *          row tile).  The hyper-network conv of the CNF runs over frames-as-rows with this flag so that a frame's gates do
 *          not change with the batch it is part of (sharding invariance).
 * Kernel choice inside (by shape only, never by B, so a batch ENTRY's result does not depend on the batch around it): P <= 16
 * rows per entry without a fused input transform -> one workgroup per (16 outputs, entry); Cout <= 16 -> streaming kernel with
 * one row tile per wave; P >= 128 and Cin >= 192 -> streaming kernel; otherwise the LDS-tiled kernel.  The guarantee is per
 * batch entry, NOT per row: without CASPR_CONV_ROW_INVARIANT a row's bits may depend on P (which kernel) and on its position
 * among the P rows (K order rotated by row tile).  Callers that put independent items along P and need them bitwise
 * independent of their neighbours (the hyper conv) set the flag; the training path's row products (train/flow_grad.py: all
 * frames or all points of the local shard as ONE entry) do not, so training gradients are shard-invariant to rounding
 * (tests: 1e-5 of the gradient norm), not bitwise.
 * caspr_pack_weight_f32: W (Cout,Cin) row-major [+ column offset/count to pack a slice] -> packed
 * buffer of caspr_packed_size(Cout, ncols) floats.                                                 */
long caspr_packed_size(int Cout, int Cin);
int caspr_pack_weight_f32(const float *w, int ldw, int Cout, int col0, int ncols, float *packed,
                          void *stream);
#define CASPR_CONV_ROW_INVARIANT 0x100
int caspr_conv1x1_f32(const float *wp, const float *bias, const float *bbias, const float *X, int ldx,
                      const float *in_scale, const float *in_shift, int in_relu, int in_relu_from,
                      float *Y, int ldy, int B, int P, int Cin, int Cout, int act, void *stream);

/* The same conv with the products on the bf16 matrix pipe ("bf16x6", csrc/gemm_bf16x6.hip): every f32 operand is split
 * EXACTLY into three bf16 numbers and six of the nine partial products (all but the three below 2^-23 |a||b|) are
 * accumulated in f32 -- error vs f64 no larger than a sequential f32 FMA chain's (tools/micro/bf16x6_gemm.hip), at
 * 1.5-1.8x the rate of the f32 MFMA kernels.  Same argument meaning as caspr_conv1x1_f32; restricted to Cin % 32 == 0,
 * Cout % 4 == 0, P % 128 == 0 and ldx >= Cin (CASPR_EINVAL otherwise: the caller falls back to caspr_conv1x1_f32).
 * wpk = the weight split and packed once by caspr_pack_weight_bf16x3 into caspr_bf16x3_packed_bytes(Cout, Cin)
 * bytes (0 if the shape is not supported).  The Python host's default where the shape allows (ops.set_matmul_mode). */
long caspr_bf16x3_packed_bytes(int Cout, int Cin);
int caspr_pack_weight_bf16x3(const float *w, int ldw, int Cout, int col0, int ncols, void *packed, void *stream);
int caspr_conv1x1_bf16x6_f32(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx,
                             const float *in_scale, const float *in_shift, int in_relu, int in_relu_from,
                             float *Y, int ldy, int B, int P, int Cin, int Cout, int act, void *stream);

/* conv -> GroupNorm statistics in one pass (the model's conv -> GroupNorm -> ReLU blocks: pointnet.py:37-42,
 * pointnet2.py:575-590 / 247, tpointnet2.py:96-111): caspr_conv1x1_bf16x6_f32 (act = 0) whose epilogue also leaves, per
 * (batch entry, 128-point tile, output channel), the f32 mean / sum of squared deviations / max / min of the tile's outputs in
 * ws; a second small kernel combines them pairwise in f64 in a fixed order into what caspr_gn_stats_f32 returns for Y: scale, shift (B,Cout),
 * optionally pmax (B,Cout) and the moments mean / rstd (B,G) (NULL: not wanted).  The 2 x |Y| read pass of caspr_gn_stats_f32
 * disappears; Y itself may be NULL when only the statistics are needed (pointnet.py:41-42: the output is max-pooled).
 * ws: caspr_conv_gn_ws_bytes(B, P, Cout) bytes, 16-byte aligned.  Same shape restrictions as caspr_conv1x1_bf16x6_f32.  */
long caspr_conv_gn_ws_bytes(int B, int P, int Cout);
int caspr_conv1x1_gn_bf16x6_f32(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx,
                                const float *in_scale, const float *in_shift, int in_relu, int in_relu_from,
                                float *Y, int ldy, int B, int P, int Cin, int Cout, int G, const float *gamma,
                                const float *beta, float eps, float *scale, float *shift, float *pmax, float *mean,
                                float *rstd, void *ws, long ws_bytes, void *stream);
/* ... with the statistics POOLED over `pool` consecutive batch entries (scale / shift / pmax (B / pool, Cout), mean / rstd
 * (B / pool, G)) while in_scale / in_shift / bbias stay per entry: the head's first layer (tpointnet2.py:96-99) normalises its
 * output per SEQUENCE but reads PointNet++ features whose GroupNorm (pointnet2.py:247) is per FRAME -- with the last, purely
 * linear PointNet++ layer folded into this layer's weight (caspr_amd/models/tpointnet2.py), its input is that GroupNorm's raw
 * operand, so the conv runs over frames (B = frames, P = points per frame) and pools T frames per statistic.            */
int caspr_conv1x1_gn_pooled_bf16x6_f32(const void *wpk, const float *bias, const float *bbias, const float *X, int ldx,
                                       const float *in_scale, const float *in_shift, int in_relu, int in_relu_from,
                                       float *Y, int ldy, int B, int P, int Cin, int Cout, int G, int pool,
                                       const float *gamma, const float *beta, float eps, float *scale, float *shift,
                                       float *pmax, float *mean, float *rstd, void *ws, long ws_bytes, void *stream);

/* The same two contracts for the LARGE layers (>= 512 output channels: the 1600-wide head convs of tpointnet2.py:96-105, the
 * 512-wide feature-propagation / final layers of pointnet2.py:525,247), csrc/gemm_bf16x6w.hip: workgroup tile 128 points x
 * 512 channels on v_mfma_f32_32x32x16_bf16, 256 accumulators per lane in the accumulator file, weight fragments straight from
 * global memory (no LDS for that operand), the activation read and split once per 512 channels.  The first Cout - Cout % 512
 * channels run there (wpk_main: caspr_pack_weight_x6w over those rows), a remainder (1600 = 3 x 512 + 64) on
 * conv1x1_bf16x6_kernel (wpk_tail: caspr_pack_weight_bf16x3 over the remaining rows; NULL when Cout % 512 == 0); both write one
 * output and one statistics array.  G > 0: also the GroupNorm statistics of the output, arguments as
 * caspr_conv1x1_gn_bf16x6_f32 (Y may then be NULL); G == 0: plain conv (gamma ... ws ignored).  act is always the identity.
 * Needs Cin % 32 == 0, Cin >= 64, P % 128 == 0, Cout >= 512, Cout % 4 == 0.                                             */
long caspr_x6w_packed_bytes(int Cout, int Cin);
int caspr_pack_weight_x6w(const float *w, int ldw, int Cout, int col0, int ncols, void *packed, void *stream);
int caspr_conv1x1_x6w_f32(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                          const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                          int P, int Cin, int Cout, int G, const float *gamma, const float *beta, float eps, float *scale,
                          float *shift, float *pmax, float *mean, float *rstd, void *ws, long ws_bytes, void *stream);
/* ... with the statistics pooled over `pool` consecutive batch entries (G > 0), as caspr_conv1x1_gn_pooled_bf16x6_f32: the head's first
 * layer reads per-FRAME normalised features but normalises its own output per SEQUENCE (tpointnet2.py:96-99).                       */
int caspr_conv1x1_x6w_pooled_f32(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                                 const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                                 int P, int Cin, int Cout, int G, int pool, const float *gamma, const float *beta, float eps, float *scale,
                                 float *shift, float *pmax, float *mean, float *rstd, void *ws, long ws_bytes, void *stream);

/* The same layer in PIECES, for a caller that needs part of the output's GroupNorm statistics before the whole layer is done: the
 * latent ODE starts from the max over points of the first 64 normalised channels of the head's 1600 -> 1600 layer (tpointnet2.py:100,111,
 * caspr.py:169) -- inside group 0 of 16, complete after the first 512-channel tile -- so the solve can run BESIDE the rest of the layer.
 *   _part_     : channel tiles mt_begin .. mt_end-1 of the 512-channel kernel (+ the remainder below 512 channels when with_tail), output
 *                rows and per-tile statistics partials into ws; reserve_cus compute units stay free for the kernel that runs beside it;
 *   _finalize_ : scale / shift / pmax / mean / rstd of groups g_begin .. g_end-1 from ws (pool as in the _pooled_ entry).
 * All pieces + a finalize over all groups == caspr_conv1x1_x6w_f32 bit for bit.                                                   */
int caspr_conv1x1_x6w_part_f32(const void *wpk_main, const void *wpk_tail, const float *bias, const float *bbias, const float *X, int ldx,
                               const float *in_scale, const float *in_shift, int in_relu, int in_relu_from, float *Y, int ldy, int B,
                               int P, int Cin, int Cout, int mt_begin, int mt_end, int with_tail, int reserve_cus, void *ws, long ws_bytes,
                               void *stream);
int caspr_conv_gn_finalize_f32(const void *ws, long ws_bytes, int B, int P, int Cout, int G, int g_begin, int g_end, int pool,
                               const float *gamma, const float *beta, float eps, float *scale, float *shift, float *pmax, float *mean,
                               float *rstd, void *stream);

/* GroupNorm statistics of Y (B,P,C) (nn.GroupNorm(G,C), biased variance, eps):
 *   scale[b,c] = gamma[c]*rstd[b,g(c)] ; shift[b,c] = beta[c] - mean[b,g(c)]*scale[b,c]
 * and, when pmax != NULL, pmax[b,c] = max_p (Y[b,p,c]*scale+shift) (tpointnet2.py:111, pointnet.py:42).
 * Two deterministic passes (f64 partial sums per 1024-point split, then a fixed-order combine);
 * ws = caller-provided scratch of at least caspr_gn_ws_bytes(B,P,C,G) bytes.  C/G %% 4 == 0.       */
long caspr_gn_ws_bytes(int B, int P, int C, int G);
int caspr_gn_stats_f32(const float *Y, int ldy, int B, int P, int C, int G, const float *gamma,
                       const float *beta, float eps, float *scale, float *shift, float *pmax,
                       void *ws, long ws_bytes, void *stream);

/* ---------------- latent ODE: models/latent_ode_model.py:45-70,139-147 (fixed-step RK4) -----------
 * z0 (B,D) rows at stride ldz ; times (Tu) ascending (made relative to times[0] as :58) ;
 * w*p = the four Linear weights packed with caspr_pack_weight_f32 ; out (B,Tu,D).
 * D <= 64, H <= 512, H % 64 == 0.  Up to 16 sequences share one workgroup (batch = MFMA columns). */
int caspr_latent_rk4_f32(const float *z0, int ldz, const float *times, int B, int Tu, int D, int H,
                         int steps, const float *w0p, const float *b0, const float *w1p, const float *b1,
                         const float *w2p, const float *b2, const float *w3p, const float *b3, float *out,
                         void *stream);

/* Same solve spread over a team of 32 workgroups per 16 sequences that keep the weights resident in LDS and meet at
 * three team barriers per evaluation (see csrc/ode.hip): ~3x lower latency of this serial stage.  H must be 512 and
 * B <= 64 (the 32*ceil(B/16) workgroups have to be co-resident); results equal caspr_latent_rk4_f32 up to the
 * re-association of the layer sums.  ws >= caspr_latent_team_ws_bytes(B), 256-byte aligned.                    */
long caspr_latent_team_ws_bytes(int B);
int caspr_latent_rk4_team_f32(const float *z0, int ldz, const float *times, int B, int Tu, int D, int H,
                              int steps, const float *w0p, const float *b0, const float *w1p, const float *b1,
                              const float *w2p, const float *b2, const float *w3p, const float *b3, float *out,
                              void *ws, long ws_bytes, void *stream);

/* ---------------- point CNF: models/cnf.py:70-128 + odefunc.py:119-142 + diffeq_layers.py:83-90
 * + normalization.py:59-108 (fixed-step RK4 of the gated 3-512-512-512-3 ODE function).
 * hyper (BT, 2*(3*H+3)) = per-frame context terms, columns [gate l0 | gate l1 | gate l2 | gate l3 |
 *   bias l0 | .. | bias l3] = W_hyper[:,1:] . c (+ gate bias)  (from caspr_conv1x1_f32);
 * tcol (2*(3*H+3)) = column 0 of the hyper weights (multiplies t), same order.
 * w0 (H,3), b0 (H) ; w1p, w2p packed (H,H) ; w3 (3,H), b3 (3).
 * mbn_in / mbn_out: 12 floats each [weight(3), bias(3), running_mean(3), running_var(3)] of the
 *   MovingBatchNorm applied before / after the block in the direction of travel (NULL = none).
 * reverse: integrate t: T_end -> 0 and use MBN._reverse (sampling); else 0 -> T_end and MBN._forward.
 * e / logp_in / logp_out (BT,n,*) may be NULL together: no divergence is integrated (sampling).     */
int caspr_cnf_rk4_f32(const float *y_in, const float *hyper, int ldh, const float *tcol,
                      const float *w0, const float *b0, const float *w1p, const float *b1,
                      const float *w2p, const float *b2, const float *w3, const float *b3, int H,
                      float t_end, int steps, int reverse, const float *mbn_in, const float *mbn_out,
                      const float *e, const float *logp_in, float *logp_out, float *y_out, int BT,
                      int n, void *stream);

/* The same solve with the two hidden layers on the bf16 matrix pipe in the exact three-way split of
 * caspr_conv1x1_bf16x6_f32 (the Python host's default, ops.set_matmul_mode).  Two kernels behind one entry, chosen by the
 * presence of e ONLY (never by BT or n: a frame's result does not depend on the batch around it):
 *   e == NULL (sampling, cnf.py:71-74 with logpx = None): csrc/ode_bf16x6w.hip -- a workgroup owns 128 points, a wave all
 *     512 hidden units of its 32 points on v_mfma_f32_32x32x16_bf16; layer 1's accumulators / activations live in the
 *     accumulator file, layer 2 runs in four 128-row passes over them; weights stream through a four-deep LDS ring;
 *   e given (forward()/NLL with the Hutchinson divergence, odefunc.py:119-142): csrc/ode_bf16x6.hip -- a workgroup owns 32
 *     points + their 32 tangent columns, a wave all 512 hidden units of its 16 columns on v_mfma_f32_16x16x32_bf16.
 * In both the hidden activation never leaves the registers of its lane (accumulator fragments are the next layer's operand
 * fragments) and LDS only stages the shared weight pieces.  w1x / w2x = the (512,512) hidden weights packed by
 * caspr_pack_weight_cnf_x6 (caspr_cnf_x6_packed_bytes() bytes each: the images of both kernels); every other argument as
 * caspr_cnf_rk4_f32, including e / logp_in / logp_out (NULL together, or given together).
 * reverse | CASPR_CNF_NARROW (sampling only): the 64-point kernel of csrc/ode_bf16x6.hip (a wave owns 16 points) instead of the
 *   128-point one -- for launches that do not fill the chip, where a solve lasts as long as ONE workgroup does: the accuracy
 *   guard's check solve (64 samples per frame at half the steps, models/caspr.py) takes half the time on it.  The caller's
 *   choice, per call: never made from BT or n inside the library.                                                       */
#define CASPR_CNF_NARROW 2
long caspr_cnf_x6_packed_bytes(void);
int caspr_pack_weight_cnf_x6(const float *w, int ldw, void *packed, void *stream);
int caspr_cnf_rk4_x6_f32(const float *y_in, const float *hyper, int ldh, const float *tcol,
                         const float *w0, const float *b0, const void *w1x, const float *b1,
                         const void *w2x, const float *b2, const float *w3, const float *b3, int H,
                         float t_end, int steps, int reverse, const float *mbn_in, const float *mbn_out,
                         const float *e, const float *logp_in, float *logp_out, float *y_out, int BT,
                         int n, void *stream);

/* ---------------- Chamfer (tk3dv.extern.chamfer.ChamferDistance): utils/evaluations.py:40 --------
 * p (B,n,3), q (B,m,3) -> dist1 (B,n) = min_j |p_i-q_j|^2 , dist2 (B,m).                            */
int caspr_chamfer_f32(const float *p, const float *q, int B, int n, int m, float *dist1, float *dist2,
                      void *stream);

/* ---------------- approximate EMD (utils/emd.py: emd_cuda approxmatch + matchcost; call site
 * utils/evaluations.py:45) -- p (B,n,3), q (B,m,3) -> cost (B) = sum_{k,l} match[k,l] |p_k - q_l| with the
 * 10-level annealed soft assignment of approxmatch (the caller divides by n, evaluations.py:46).
 * ws >= caspr_emd_ws_bytes(B,n,m).                                                                   */
long caspr_emd_ws_bytes(int B, int n, int m);
int caspr_emd_f32(const float *xyz1, const float *xyz2, int B, int n, int m, float *cost, void *ws,
                  long ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CASPR_HIP_H */
