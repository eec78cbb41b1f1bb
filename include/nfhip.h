/*
 * nfhip.h -- C ABI of libnfhip.so, the MI355X (gfx950 / CDNA4) flow-transform engine.
 *
 * Drop-in boundary for the forward / inverse + log-det-Jacobian hot path of
 * tatsy/normalizing-flows-pytorch (pure Python upstream: there is no upstream FFI, so every entry
 * point cites the reference *Python* function it replaces).  Plain pointers and sizes only, no torch
 * types: the host side (normalizing-flows-pytorch_amd/_native.py) binds these with ctypes and wraps
 * them in torch.autograd.Function objects behind the reference's nn.Module surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated; tensors are NCHW
 *     (B, C, H, W), 2-D data is (B, D) == (B, C=D, H=1, W=1);
 *   - `ld` is the running log-det-Jacobian vector (B,), updated IN PLACE (the reference does the same:
 *     coupling.py:110, modules.py:249,305,480);
 *   - no allocation, no synchronisation, no host reads inside: every call only enqueues kernels on
 *     `stream` (a hipStream_t), so a whole step can be captured into a hipGraph;
 *   - gradient accumulators (g_* of parameters) are ACCUMULATED INTO (+=): the caller zero-fills them;
 *   - return value: 0 (hipSuccess) or the hipError_t of the failed launch; NF_E_* for argument errors.
 *
 * `mode` / `odd` select the split map of AbstractCoupling (coupling.py:16-30), see SURVEY.md appendix A:
 *   z0 = the TRANSFORMED half, z1 = the CONDITIONING half, both of shape (B, Ch, h, w).
 */
#ifndef NFHIP_H
#define NFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nf_stream_t; /* hipStream_t */

enum nf_split_mode {
    NF_SPLIT_1D = 0,      /* squeeze1d / unsqueeze1d        flows/squeeze.py:64-83  : Ch=D/2, h=w=1      */
    NF_SPLIT_CHECKER = 1, /* checker_split / checker_merge  flows/squeeze.py:32-61  : Ch=2C, h=H/2, w=W/2 */
    NF_SPLIT_CHANNEL = 2, /* channel_split / channel_merge  flows/squeeze.py:5-17   : Ch=C/2, h=H, w=W    */
    NF_SPLIT_NONE = 3     /* no split: every element is transformed (MAF, flows/maf.py:101-107)           */
};

enum nf_chan_op {
    NF_ACTNORM = 0, /* p0 = log_scale, p1 = bias                    flows/modules.py:225-256 */
    NF_FLOWBN = 1   /* p0 = mean, p1 = var, p2 = log_gamma, p3 = beta  flows/modules.py:259-322 */
};

#define NF_E_BADARG 10001
#define NF_E_UNSUPPORTED 10002

/* library / device probes (host side) */
int nf_version(void);
int nf_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len);

/* ---- index maps (bit-exact) -------------------------------------------------------------------------------- */
/* half[b,m,i,j] = z[b, src(m,i,j)]  for which = 0 (z0) or 1 (z1).       squeeze.py:5-10, :32-44, :64-72      */
int nf_half_gather(const float* z, float* half, int which, int mode, int odd, int64_t B, int C, int H, int W,
                   nf_stream_t stream);
/* full[b, src(m,i,j)] = half[b,m,i,j]; every other position of `full` is set to 0 (autograd of the gather).  */
int nf_half_scatter(const float* half, float* full, int which, int mode, int odd, int64_t B, int C, int H, int W,
                    nf_stream_t stream);
/* Squeeze2d.forward / Unsqueeze2d.backward: (B,C,H,W) -> (B,4C,H/2,W/2)  squeeze.py:86-96, :162-165, :186-189 */
int nf_squeeze2d(const float* z, float* out, int64_t B, int C, int H, int W, nf_stream_t stream);
/* Unsqueeze2d.forward / Squeeze2d.backward: (B,4C,h,w) -> (B,C,2h,2w)    squeeze.py:99-111, :167-170, :181-184 */
int nf_unsqueeze2d(const float* z, float* out, int64_t B, int C, int H, int W, nf_stream_t stream);

/* ---- affine coupling: AffineCoupling._transform/_inverse_transform + split + merge  coupling.py:32-43,:104-122
 * t_ptr / s_ptr: conditioner output, element (b,m,i,j) at  ptr[b*param_bstride + (m*h+i)*w+j];
 * for a coupling layer t_ptr = params, s_ptr = params + Ch*h*w, param_bstride = 2*Ch*h*w (coupling.py:106-107);
 * for MAF (NF_SPLIT_NONE) the two MADE outputs (maf.py:103-104).
 *   s = tanh(s_raw) * s_log_scale[0] + s_bias[0]
 *   forward : y0 = z0*exp(s) + t,  ld[b] += sum s ;  inverse: y0 = exp(-s)*(z0 - t),  ld[b] -= sum s
 *   y = merge(y0, z1) written in full (pass-through half copied).                                              */
int nf_affine_coupling_fwd(const float* z, const float* t_ptr, const float* s_ptr, int64_t param_bstride,
                           const float* s_log_scale, const float* s_bias, float* y, float* ld, int mode, int odd,
                           int inverse, int64_t B, int C, int H, int W, nf_stream_t stream);
/* autograd of the forward direction.  g_z is written in full (pass-through half = g_y);
 * g_t / g_s have the layout of t_ptr / s_ptr; g_scale[0] += , g_bias[0] += (caller zero-fills).
 * g_ld is read only (it passes through unchanged).                                                             */
int nf_affine_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* t_ptr,
                           const float* s_ptr, int64_t param_bstride, const float* s_log_scale, const float* s_bias,
                           float* g_z, float* g_t, float* g_s, float* g_scale, float* g_bias, int mode, int odd,
                           int64_t B, int C, int H, int W, nf_stream_t stream);

/* ---- per-channel affine bijectors: ActNorm and flow BatchNorm ---------------------------------------------- */
/* y = f_c(x), ld[b] += (forward) / -= ... the layer's scalar log-det, P = H*W pixels.  modules.py:246-256,
 * :300-305, :315-320.   NF_ACTNORM: p0=log_scale p1=bias.  NF_FLOWBN: p0=mean p1=var p2=log_gamma p3=beta.      */
int nf_chan_affine_fwd(int op, const float* x, const float* p0, const float* p1, const float* p2, const float* p3,
                       float* y, float* ld, int inverse, int64_t B, int C, int P, nf_stream_t stream);
/* autograd of the forward direction (appendix B2/B4): g_x written; g_p0/g_p1 (ActNorm: g_log_scale, g_bias;
 * flow-BN with affine=True: g_log_gamma, g_beta; pass NULL to skip) accumulated.                               */
int nf_chan_affine_bwd(int op, const float* g_y, const float* g_ld, const float* x, const float* p0,
                       const float* p1, const float* p2, const float* p3, float* g_x, float* g_pa, float* g_pb,
                       int64_t B, int C, int P, nf_stream_t stream);
/* per-channel statistics over (B, P): sum[c] += sum x   (pass 1)  /  sqdev[c] += sum (x - sum[c]/n)^2 (pass 2)
 * ActNorm data-dependent init (modules.py:238-244) and flow-BN batch stats (modules.py:284-287).               */
int nf_chan_sum(const float* x, float* sum, int64_t B, int C, int P, nf_stream_t stream);
int nf_chan_sqdev(const float* x, const float* sum, float* sqdev, int64_t B, int C, int P, nf_stream_t stream);
/* flow-BN train-mode bookkeeping in one launch (modules.py:285-294): batch_mean = sum/n,
 * batch_var = sqdev/n + eps, running = (1-momentum)*running + momentum*batch.                                  */
int nf_flowbn_finalize(const float* sum, const float* sqdev, float* batch_mean, float* batch_var,
                       float* running_mean, float* running_var, float eps, float momentum, int64_t n, int C,
                       nf_stream_t stream);
/* ActNorm init finalize (modules.py:240-243): log_scale = log(sqrt(sqdev/(n-1)) + eps), bias = sum/n.          */
int nf_actnorm_init_finalize(const float* sum, const float* sqdev, float* log_scale, float* bias, float eps,
                             int64_t n, int C, nf_stream_t stream);

/* ---- invertible 1x1 convolution: per-pixel C x C mat-vec  modules.py:470-497 ----------------------------------
 * y[b,:,p] = M z[b,:,p]  (M row-major C x C; transpose != 0 applies M^T: the autograd of z).
 * if ld != NULL: ld[b] += ld_sign * P * sum(log_s)   (modules.py:479-480, :494-495).
 * Forward uses M = W = P L' U' (modules.py:471-473), the inverse M = W^-1 obtained from the SAME LU factors and
 * init-time pivots the reference hands to torch.lu_solve (modules.py:485-492).                                  */
int nf_invconv_apply(const float* z, const float* M, int transpose, float* y, float* ld, const float* log_s,
                     float ld_sign, int64_t B, int C, int P, nf_stream_t stream);
/* g_M[r,c] += sum_{b,p} g_y[b,r,p] * z[b,c,p]   (appendix B3)                                                   */
int nf_invconv_wgrad(const float* g_y, const float* z, float* g_M, int64_t B, int C, int P, nf_stream_t stream);

/* ---- Logit  modules.py:141-156 --------------------------------------------------------------------------------
 * forward: xc = clamp(x, eps, 1-eps); y = log(xc/(1-xc)); ld[b] += sum -(y - 2 softplus(y))
 * inverse: y = sigmoid(x); ld[b] += sum (x - 2 softplus(x)).      n = elements per sample.                       */
int nf_logit_fwd(const float* x, float* y, float* ld, float eps, int inverse, int64_t B, int64_t n,
                 nf_stream_t stream);
int nf_logit_bwd(const float* g_y, const float* g_ld, const float* x, float* g_x, float eps, int64_t B, int64_t n,
                 nf_stream_t stream);

/* ---- Flow++ mixture-of-logistics coupling  coupling.py:172-210, modules.py:64-97, :186-212 ---------------------
 * params: conditioner output (B, (2+3K)*Ch, h, w) with channel sections [a | b | logit(pi) | mu | s]
 * (coupling.py:140,177); mixture k of transformed channel m lives at section channel k*Ch + m (coupling.py:180-182).
 * forward : z0 -> MixLogCDF -> Logit(eps) -> * exp(a) + b,  a = tanh(a_raw)*a_log_scale + a_bias; ld accumulates
 *           log pdf, the logit log-det and sum a.   inverse: the exact reverse with the bisection of
 *           modules.py:196-212 (bracket +-1e3, <= 100 iterations, batch-global exit rule reproduced with the
 *           device flag `stuck_flag` (int32[1], zeroed by the call): 25 iterations, then 75 more iff any element
 *           still has |hi-lo| >= 1e-4, i.e. iff the reference would not have left the loop at iteration 25.
 *           `scratch`: 3*B*Ch*h*w floats (bracket + target carried between the two phases).                       */
int nf_mixlog_coupling_fwd(const float* z, const float* params, const float* a_log_scale, const float* a_bias,
                           float* y, float* ld, int K, float logit_eps, int mode, int odd, int64_t B, int C, int H,
                           int W, nf_stream_t stream);
int nf_mixlog_coupling_inv(const float* z, const float* params, const float* a_log_scale, const float* a_bias,
                           float* y, float* ld, float* scratch, int* stuck_flag, int K, int mode, int odd, int64_t B,
                           int C, int H, int W, nf_stream_t stream);
int nf_mixlog_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* params,
                           const float* a_log_scale, const float* a_bias, float* g_z, float* g_params,
                           float* g_scale, float* g_bias, int K, float logit_eps, int mode, int odd, int64_t B,
                           int C, int H, int W, nf_stream_t stream);

/* ---- NLL of the training harness  main.py:49-51, :85 -------------------------------------------------------------
 * loss[0] += -(1/B) * sum_b ( -0.5*|z_b|^2 - 0.5*D*log(2 pi) + ld[b] );  g_z = z / B,  (g_ld = -1/B is constant) */
int nf_nll_loss(const float* z, const float* ld, float* loss, int64_t B, int64_t D, nf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFHIP_H */
