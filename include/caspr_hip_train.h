/* caspr_hip_train.h -- C-ABI of the training tier of libcaspr_hip.so (gradient kernels).
 *
 * The reference gets every one of these from torch.autograd: `loss.backward()` at
 * train_utils.py:173 walks the graph recorded by models/caspr.py:76-115 (CaSPR.forward) and
 * caspr_losses.py:31-70.  Each entry below names the forward call site whose autograd node it
 * replaces.  Same conventions as caspr_hip.h: device pointers, f32, point-major rows, the
 * caller's stream, int return code (0 = ok, text from caspr_last_error_string()).
 * All reductions combine partial sums in a fixed order: gradients are reproducible run to run.  */
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

/* GroupNorm(+ReLU) backward.  Y = raw conv output (B,P,ldy), dA (B,P,ldd) = gradient w.r.t. the
 * normalised (and, if relu, rectified) activation; on return dA holds the gradient w.r.t. Y.
 * dgamma / dbeta (C) (+)= parameter gradients.  ws >= caspr_gn_bwd_ws_bytes(B,P,C,G).              */
long caspr_gn_bwd_ws_bytes(long B, int P, int C, int G);
int caspr_gn_bwd_f32(const float *Y, int ldy, float *dA, int ldd, long B, int P, int C, int G,
                     const float *mean, const float *rstd, const float *gamma, const float *beta,
                     int relu, float *dgamma, float *dbeta, int accumulate, void *ws, long ws_bytes,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CASPR_HIP_TRAIN_H */
