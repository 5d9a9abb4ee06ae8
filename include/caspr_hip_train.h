/* caspr_hip_train.h -- C-ABI of the training tier of libcaspr_hip.so (gradient kernels).
 *
 * The reference gets every one of these from torch.autograd: `loss.backward()` at
 * train_utils.py:173 walks the graph recorded by models/caspr.py:76-115 (CaSPR.forward) and
 * caspr_losses.py:31-70.  Each entry below names the forward call site whose autograd node it
 * replaces.  Same conventions as caspr_hip.h: device pointers, f32, point-major rows, the
 * caller's stream, int return code (0 = ok, text from caspr_last_error_string()).
 * Dense reductions (weight / GroupNorm parameter gradients) combine partial sums in a fixed order; the two
 * scatter-adds exist as float-atomic kernels (as in Kaolin) and as a deterministic segment gather
 * (caspr_segment_sum_f32), which is what the training path calls: gradients are reproducible run to run. */
#ifndef CASPR_HIP_TRAIN_H
#define CASPR_HIP_TRAIN_H
#include "caspr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* GroupNorm statistics as caspr_gn_stats_f32 plus the moments the backward pass needs:
 * mean (B,G), rstd (B,G).  (forward node: nn.GroupNorm at pointnet.py:38-41, pointnet2.py:247,525) */
int caspr_gn_stats_train_f32(const float *Y, int ldy, int B, int P, int C, int G, const float *gamma,
                             const float *beta, float eps, float *scale, float *shift, float *pmax,
                             float *mean, float *rstd, void *ws, long ws_bytes, void *stream);

/* Weight / bias gradient of the pointwise conv (nn.Conv1d(k=1) / nn.Conv2d(k=1) / nn.Linear nodes at
 * pointnet.py:37-41, pointnet2.py:525,247, tpointnet2.py:99-105):
 *   dW[co,k] (+)= sum_{b,p} dY[b,p,co] * in(X[b,p,k])     dbias[co] (+)= sum_{b,p} dY[b,p,co]
 * in() = the same folded GroupNorm(+ReLU) operand transform as caspr_conv1x1_f32.  dW is (Cout,Cin)
 * row-major, i.e. the layout of the reference's parameter.  dbias may be NULL.  accumulate != 0 adds
 * to the existing contents (weight used at several sites).  ws >= caspr_wgrad_ws_bytes(B*P,Cin,Cout).
 * The data gradient dX = dY . W needs no entry of its own: it is caspr_conv1x1_f32 with the
 * transposed weight packed by caspr_pack_weight_f32.                                                */
long caspr_wgrad_ws_bytes(long rows, int Cin, int Cout);
int caspr_conv1x1_wgrad_f32(const float *dY, int lddy, const float *X, int ldx, const float *in_scale,
                            const float *in_shift, int in_relu, int in_relu_from, int B, int P, int Cin,
                            int Cout, float *dW, float *dbias, int accumulate, void *ws, long ws_bytes,
                            void *stream);

/* The same contract with the products on the bf16 matrix pipe in the exact three-way split of caspr_conv1x1_bf16x6_f32
 * (csrc/backward.hip: conv1x1_wgrad_bf16x6_kernel; both operands are split on the way to LDS, fragments come out of the
 * row-major LDS image through ds_read_b64_tr_b16).  Same workspace, same fixed-order slab reduction (bit-reproducible).  */
int caspr_conv1x1_wgrad_bf16x6_f32(const float *dY, int lddy, const float *X, int ldx, const float *in_scale,
                            const float *in_shift, int in_relu, int in_relu_from, int B, int P, int Cin,
                            int Cout, float *dW, float *dbias, int accumulate, void *ws, long ws_bytes,
                            void *stream);

/* GroupNorm(+ReLU) backward.  Y = raw conv output (B,P,ldy); dA (B,P,ldd) = gradient w.r.t. the
 * normalised (and, if relu, rectified) activation, or NULL for zero; dMax (B,C) + aMax (B,C) = gradient
 * of the max over points of the normalised, NOT rectified feature and its arg-max point (torch.max at
 * tpointnet2.py:111 / pointnet.py:42), or NULL.  dY (B,P,lddy) = gradient w.r.t. Y (may alias dA).
 * dgamma / dbeta (C) (+)= parameter gradients.  ws >= caspr_gn_bwd_ws_bytes(B,P,C,G).              */
long caspr_gn_bwd_ws_bytes(long B, int P, int C, int G);
int caspr_gn_bwd_f32(const float *Y, int ldy, const float *dA, int ldd, const float *dMax,
                     const int32_t *aMax, float *dY, int lddy, long B, int P, int C, int G,
                     const float *mean, const float *rstd, const float *gamma, const float *beta,
                     int relu, float *dgamma, float *dbeta, int accumulate, void *ws, long ws_bytes,
                     void *stream);

/* arg-max over points of y*scale+shift, first index on ties (the index torch.max keeps for backward). */
long caspr_argmax_ws_bytes(long B, int P, int C);
int caspr_argmax_points_f32(const float *Y, int ldy, int B, int P, int C, const float *scale,
                            const float *shift, int32_t *out, void *ws, long ws_bytes, void *stream);

/* out[b,c] = sum_p A[b,p,c]: gradient of the per-sequence bias that carries the tiled global feature
 * through the head's first conv (tpointnet2.py:96-99, pointnet.py:44-46).                           */
long caspr_colsum_ws_bytes(long B, int P, int C);
int caspr_colsum_batched_f32(const float *A, int ld, int B, int P, int C, float *out, void *ws,
                             long ws_bytes, void *stream);

/* three_interpolate backward (Kaolin three_interpolate grad, call site pointnet2.py:519):
 * dFeat[b, idx[b,i,k], c] += weight[b,i,k] * dOut[b,i,c], c < C.  dFeat must be initialised by the
 * caller (float atomics: the summation order, hence the last bit, can vary run to run).             */
int caspr_three_interp_bwd_f32(const float *dOut, int ldo, const int32_t *idx, const float *weight,
                               int B, int m, int n, int C, float *dFeat, int ldf, void *stream);

/* Training layout of the grouper (Kaolin PointNet2GroupingLayer, call site pointnet2.py:391): one row
 * per (b, centre j, sample s): G[(b*M+j)*ns+s] = [xyz[b,i]-new_xyz[b,j] | feat[b,i,0:C] | 0 pad],
 * i = idx[b,j,s]; channel order = the reference's (xyz first).  Backward scatters the feature columns
 * back: dFeat[b,i,c] += dG[row,3+c] (float atomics).
 * centred != 0: every row has the row of its neighbourhood's sample 0 subtracted (exact in front of a conv whose
 *   GroupNorm groups are single channels: the per-neighbourhood constant W x0 + b cancels in the normalisation, and
 *   the weight gradient sum_s dy_s x_s is unchanged because sum_s dy_s = 0 there); with feat_kind = CASPR_FEAT_QUAD |
 *   CASPR_FEAT_PAIRS (feat = caspr_prep_input_f32's augmentation of xyz) the centred features are formed from the
 *   coordinate differences as in caspr_sa_mlp_max_f32.  Used for the first level's 16-channel scale.           */
int caspr_group_rows_f32(const float *xyz, const float *new_xyz, const float *feat, int ldf,
                         const int32_t *idx, int B, int n, int M, int C, int ns, int centred, int feat_kind,
                         float *G, int ldg, void *stream);
int caspr_group_rows_bwd_f32(const float *dG, int ldg, const int32_t *idx, int B, int n, int M, int C,
                             int ns, float *dFeat, int ldf, void *stream);

/* Deterministic form of the two scatter-adds: a gather over precomputed segments (CSR over the target rows, the entries
 * of each segment in ascending source-row order):  dst[t, c] (+)= sum_{e in [seg_start[t], seg_start[t+1])}
 * w[e] * src[seg_row[e], col0 + c],  c < C  (w == NULL: 1).  The training path builds the segments once per level from
 * the three-NN / ball-query indices (a stable sort of the target ids) and calls this instead of the atomic kernels:
 * gradients are then bit-reproducible run to run.                                                                   */
int caspr_segment_sum_f32(const float *src, int lds, int col0, const int32_t *seg_start, const int32_t *seg_row,
                          const float *seg_w, long targets, int C, float *dst, int ldd, int accumulate, void *stream);

/* GroupNorm(16) over one neighbourhood (ns rows) at a time (PointNetFeatureExtractor, pointnet2.py:
 * 649-703).  Y (NB*ns, ldy).  Forward writes mean/rstd (NB,16) and either the dense activation
 * A = relu?(gn(Y)) or, for the last layer, maxout (NB, ldm) = max over the ns rows and arg (NB,C) =
 * first row attaining it (torch.max at pointnet2.py:701).  Backward takes dA (dense) or dMax + arg.  */
int caspr_gn_rows_f32(const float *Y, int ldy, long NB, int ns, int C, const float *gamma,
                      const float *beta, float eps, int relu, float *A, int lda, float *mean,
                      float *rstd, float *maxout, int ldm, int32_t *arg, void *stream);
long caspr_gn_rows_bwd_ws_bytes(int C);
int caspr_gn_rows_bwd_f32(const float *Y, int ldy, long NB, int ns, int C, const float *gamma,
                          const float *beta, int relu, const float *mean, const float *rstd,
                          const float *dA, int lda, const float *dMax, int ldm, const int32_t *arg,
                          float *dY, int lddy, float *dgamma, float *dbeta, int accumulate, void *ws,
                          long ws_bytes, void *stream);

/* The latent ODE's training path in one launch each way (latent_ode_model.py:45-70,139-147 through torch.autograd + torchdiffeq's
 * adjoint in the reference, train_utils.py:173; here the gradient of the discrete RK4 map).
 * caspr_latent_rk4_team_tape_f32: caspr_latent_rk4_team_f32 (include/caspr_hip.h: same arguments, same values) that also leaves the tape
 *   of its E = 4 steps (Tu - 1) evaluations, in the order they are made: tape_x (E, B, xw) the evaluation's input (xw >= D, a multiple
 *   of 4), tape_h1 / h2 / h3 (E, B, 512) the three tanh outputs.  Evaluations a zero-length interval skips leave their rows untouched
 *   (zero-fill the tape: they then contribute nothing to the weight gradients).
 * caspr_latent_rk4_team_adjoint_f32: the reverse sweep.  gout (B, Tu, D) = dL/d(out) -> gz (B, D) = dL/dz0 and the per-layer deltas
 *   d0 / d1 / d2 (E, B, 512), d3 (E, B, xw) (rows as the tape), from which dW_l = d_l^T tape_l and db_l = colsum(d_l) follow as ONE
 *   caspr_conv1x1_wgrad_f32 per layer over the E B rows.  w3tp .. w0tp: caspr_pack_weight of the TRANSPOSED weights (W3^T (512, D) ..
 *   W0^T (D, 512)).  Both: D <= 64, H == 512, B <= 64; ws: caspr_latent_team_ws_bytes(B), 256-byte aligned.  Deterministic.          */
int caspr_latent_rk4_team_tape_f32(const float *z0, int ldz, const float *times, int B, int Tu, int D, int H, int steps,
                                   const float *w0p, const float *b0, const float *w1p, const float *b1,
                                   const float *w2p, const float *b2, const float *w3p, const float *b3, float *out,
                                   float *tape_x, int xw, float *tape_h1, float *tape_h2, float *tape_h3, void *ws,
                                   long ws_bytes, void *stream);
int caspr_latent_rk4_team_adjoint_f32(const float *gout, const float *times, int B, int Tu, int D, int H, int steps,
                                      const float *w3tp, const float *w2tp, const float *w1tp, const float *w0tp,
                                      const float *tape_h1, const float *tape_h2, const float *tape_h3, float *d0,
                                      float *d1, float *d2, float *d3, int xw, float *gz, void *ws, long ws_bytes,
                                      void *stream);

/* Gated softplus layer of the CNF's ODE function (ConcatSquashLinear + Softplus, diffeq_layers.py:83-90,
 * odefunc.py:98-105) on value and tangent rows of Z (2R, ldz) = the layer's matrix product; frame f = p / n has its
 * own gate / beta rows (hyper networks of the context).  Row layout: blk = R puts the value of point p in row p and its
 * tangent (Hutchinson, odefunc.py:13-31) in row R + p; blk = a power of two dividing R (32: what the fused conv epilogue of
 * caspr_conv1x1_cnf_act_bf16x6_f32 needs) interleaves blocks of blk value rows and the blk tangent rows of the same points:
 * value row = (p / blk) 2 blk + p % blk, tangent row = value row + blk.
 *   H[v(p)] = softplus((Z[v(p)]+b)*gate[f]+beta[f])     H[t(p)] = sigmoid(same) * Z[t(p)]*gate[f]
 * Backward: dZ (2R, lddz), dgate / dbeta (R/n, C) summed over each frame's points in a fixed order.   */
int caspr_cnf_act_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta,
                      long R, int n, int C, long blk, float *H, int ldh, void *stream);
int caspr_cnf_act_bwd_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta,
                          const float *dH, int ldd, long R, int n, int C, long blk, float *dZ, int lddz,
                          float *dgate, float *dbeta, void *stream);
/* ... for the hidden layer in front of the 3-channel output layer (odefunc.py:103: no activation behind it): dH is that layer's
 * data gradient dZo Wo (dZo (2R, ldo >= 3) in the same row layout, Wo (3, ldw >= C)), formed on the fly.                  */
int caspr_cnf_act_bwd_out_f32(const float *Z, int ldz, const float *b, const float *gate, const float *beta,
                              const float *dZo, int ldo, const float *Wo, int ldw, long R, int n, int C, long blk,
                              float *dZ, int lddz, float *dgate, float *dbeta, void *stream);

/* Epilogue of the ODE function's 3-channel output layer (odefunc.py:103-105 and the Hutchinson contraction of odefunc.py:13-31) on
 * the rows of Zo (2R, ldo >= 3) in the row layout blk: a (R,3) = (Zo_value + b) gate[f] + beta[f], nd (R) = - sum_j Zo_tangent_j
 * gate[f]_j e_j.  gate / beta: rows of a (frames, ldg) tensor.  Backward: dZo (2R, 4) (column 3 zero), dgate / dbeta (frames, 3).  */
int caspr_cnf_out_f32(const float *Zo, int ldo, const float *b, const float *gate, const float *beta, int ldg,
                      const float *E, long R, int n, long blk, float *A, float *ND, void *stream);
int caspr_cnf_out_bwd_f32(const float *dA, const float *dND, const float *Zo, int ldo, const float *b, const float *gate,
                          int ldg, const float *E, long R, int n, long blk, float *dZo, float *dgate, float *dbeta,
                          void *stream);

/* A hidden layer of the ODE function in one launch (diffeq_layers.py:83-90 + odefunc.py:98-105 on value and tangent rows):
 * Z = X W^T on the bf16x6 conv kernel (wpk: caspr_pack_weight_bf16x3 of W (Cout, Cin)), H = the gated softplus of
 * caspr_cnf_act_f32 applied in the conv's epilogue.  X (2 frames n, ldx), Z, H in the row layout blk = 32; n % 64 == 0,
 * Cin % 32 == 0, Cout % 4 == 0.  Same values as caspr_conv1x1_bf16x6_f32 followed by caspr_cnf_act_f32 (one pass less
 * over Z).  Backward: caspr_cnf_act_bwd_f32 on (Z, dH), then the data / weight gradient convs as for any layer.          */
int caspr_conv1x1_cnf_act_bf16x6_f32(const void *wpk, const float *b, const float *gate, const float *beta,
                                     const float *X, int ldx, float *Z, int ldz, float *H, int ldh, int frames,
                                     int n, int Cin, int Cout, void *stream);

/* ... and its BACKWARD in the epilogue of the data-gradient conv that produces its dH: X = dZ of the NEXT layer (2 frames n, ldx),
 * wpk = caspr_pack_weight_bf16x3 of that layer's TRANSPOSED weight (Cout = this layer's width, Cin = the next one's), Z = this
 * layer's raw product, (b, gate, beta) its parameters -> dZ (2 frames n, lddz) = what caspr_cnf_act_bwd_f32 returns for
 * dH = X W, and dgate / dbeta (frames, Cout) from per-tile partial sums in a fixed order (ws: caspr_conv1x1_cnf_act_bwd_ws_bytes).
 * dH itself is never written.  Row layout blk = 32.                                                                       */
long caspr_conv1x1_cnf_act_bwd_ws_bytes(int frames, int n, int Cout);
int caspr_conv1x1_cnf_act_bwd_bf16x6_f32(const void *wpk, const float *X, int ldx, const float *Z, int ldz, const float *b,
                                         const float *gate, const float *beta, float *dZ, int lddz, float *dgate,
                                         float *dbeta, void *ws, long ws_bytes, int frames, int n, int Cin, int Cout,
                                         void *stream);

/* First layer of the ODE function (3 -> C) fused with its gate + softplus on value (y) and tangent (e) rows:
 * H (2R, C) as caspr_cnf_act_f32 with Z = [W0 y ; W0 e] in the row layout `blk`.  Y, E (R,3) point-indexed; W0 (C,3).
 * Backward, with ch = caspr_cnf_in_bwd_chunk(C) and ns = caspr_cnf_in_bwd_splits(C, n): dgate / dbeta (R/n, ns, C) and
 * dW0_part (R/n * ns, C, 3) = partial sums per frame and point split (sum over frames and splits = dW0; db0 = sum_f
 * gate*dbeta); dY_part (ceil(C/ch), R, 3) = per-chunk partial sums of dL/dy.  All sums in a fixed order.      */
int caspr_cnf_in_bwd_chunk(int C);
int caspr_cnf_in_bwd_splits(int C, int n);
int caspr_cnf_in_f32(const float *Y, const float *E, const float *W0, const float *b, const float *gate,
                     const float *beta, long R, int n, int C, long blk, float *H, void *stream);
int caspr_cnf_in_bwd_f32(const float *Y, const float *E, const float *W0, const float *b, const float *gate,
                         const float *beta, const float *dH, long R, int n, int C, long blk, float *dgate,
                         float *dbeta, float *dW0_part, float *dY_part, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CASPR_HIP_TRAIN_H */
