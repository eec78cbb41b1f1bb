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
/* out = base + scatter(half): the gradient of a tensor consumed whole and through the gather of one half (image splits only).        */
int nf_half_scatter_add(const float* half, const float* base, float* out, int which, int mode, int odd, int64_t B, int C, int H, int W,
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

/* ---- fused training-mode flow BatchNorm for the head of a RealNVP / MAF flow step (modules.py:283-307) -------------
 * nf_flowbn_stats   : ws[0:C] += sum (x - center), ws[C:2C] += sum (x - center)^2, ws[2C:3C] = center  (ONE pass of
 *                     shifted sums; center = the running mean; caller zero-fills ws[0:2C]).
 * nf_flowbn_head_fwd: mean / biased variance (+eps inside) from ws, batch_* and running_* buffers written, y = BN(x),
 *                     ld[b] += pixels * sum(log_gamma - 0.5 log var); z1c != NULL additionally gathers the conditioning
 *                     half of the following coupling (mode / odd as above).
 * nf_flowbn_head_bwd: g_x = (g_h + scatter(g_z1c)) * exp(log_gamma) / sqrt(var)   (affine=False: statistics are
 *                     constants for autograd); g_z1c nullable.                                                       */
int nf_flowbn_stats(const float* x, const float* center, float* ws, int64_t B, int C, int P, nf_stream_t stream);
int nf_flowbn_head_fwd(const float* x, const float* ws, const float* log_gamma, const float* beta, float* batch_mean,
                       float* batch_var, float* running_mean, float* running_var, float eps, float momentum, float* y,
                       float* z1c, float* ld, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream);
/* The two launches above in ONE persistent launch (csrc/flowbn_head.hip: k_flowbn_head_fused): the workgroups exchange their
 * per-plane shifted sums through ws_zero -- nf_flowbn_head_fused_ws_floats(B, C, H, W) floats of ZEROS; 0 = the shape is not
 * taken (more workgroups of 1 024 elements than compute units, C > 64, B C > 3 072, H W not a power of two in 16 .. 1 024): use the
 * two launches -- and add them in sample order (bit-reproducible without the ordered mode).  Replaces the same reference lines (flows/modules.py:283-307).  */
int nf_flowbn_head_fused_ws_floats(int64_t B, int C, int H, int W);
int nf_flowbn_head_fused(const float* x, const float* log_gamma, const float* beta, float* batch_mean, float* batch_var,
                         float* running_mean, float* running_var, float eps, float momentum, float* y, float* z1c, float* ld,
                         float* ws_zero, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream);
int nf_flowbn_head_bwd(const float* g_h, const float* g_z1c, const float* var, const float* log_gamma, float* g_x,
                       int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream);

/* ---- invertible 1x1 convolution: per-pixel C x C mat-vec  modules.py:470-497 ----------------------------------
 * y[b,:,p] = M z[b,:,p]  (M row-major C x C; transpose != 0 applies M^T: the autograd of z).
 * if ld != NULL: ld[b] += ld_sign * P * sum(log_s)   (modules.py:479-480, :494-495).
 * Forward uses M = W = P L' U' (modules.py:471-473), the inverse M = W^-1 obtained from the SAME LU factors and
 * init-time pivots the reference hands to torch.lu_solve (modules.py:485-492).                                  */
int nf_invconv_apply(const float* z, const float* M, int transpose, float* y, float* ld, const float* log_s,
                     float ld_sign, int64_t B, int C, int P, nf_stream_t stream);
/* g_M[r,c] += sum_{b,p} g_y[b,r,p] * z[b,c,p]   (appendix B3)                                                   */
int nf_invconv_wgrad(const float* g_y, const float* z, float* g_M, int64_t B, int C, int P, nf_stream_t stream);

/* PLU weight assembly W = P (L o L_mask + I) (U o U_mask + diag(sign_s exp(log_s)))  (modules.py:471-473) and its
 * autograd: g_L, g_U (masked), g_log_s[i] = diag term + pixels * sum_b g_ld[b] (the log-det path, modules.py:480);
 * g_ld may be NULL; accumulate != 0: g_L/g_U/g_log_s are +=.  One workgroup, C <= 64.                                                                       */
int nf_invconv_weight_fwd(const float* P, const float* L, const float* U, const float* L_mask, const float* U_mask,
                          const float* sign_s, const float* log_s, float* W, int C, nf_stream_t stream);
int nf_invconv_weight_bwd(const float* g_W, const float* P, const float* L, const float* U, const float* L_mask,
                          const float* U_mask, const float* sign_s, const float* log_s, const float* g_ld, float* g_L,
                          float* g_U, float* g_log_s, int accumulate, int C, int64_t B, int pixels, nf_stream_t stream);

/* the same for up to NF_PLU_MAX_LAYERS layers per launch, one workgroup per layer (an image Glow has 129 of these single-
 * workgroup, latency-bound launches per direction).  forward uses P .. log_s, W, C; backward additionally g_W, g_ld (nullable),
 * g_L, g_U, g_log_s, accumulate, B, pixels.                                                                               */
#define NF_PLU_MAX_LAYERS 192     /* (x 128 bytes of descriptor = 24 KB of kernel arguments; see NF_SLAB_SUM_MAX) */
typedef struct nf_plu_desc {
    const float* P; const float* L; const float* U; const float* L_mask; const float* U_mask; const float* sign_s;
    const float* log_s;
    float* W;
    const float* g_W;
    const float* g_ld;
    float* g_L; float* g_U; float* g_log_s;
    int64_t B;
    int C;
    int accumulate;
    float pixels;
    int reserved;
} nf_plu_desc;
int nf_invconv_weight_fwd_multi(const nf_plu_desc* descs, int n_layers, nf_stream_t stream);
int nf_invconv_weight_bwd_multi(const nf_plu_desc* descs, int n_layers, nf_stream_t stream);

/* ---- fused head of a Glow flow step for C <= 4 (2-D data, 1..4 channel images) -------------------------------------
 * forward : h = W ((z - bias) / exp(log_scale)) per pixel with W = P L' U' assembled in-kernel; z1c = the contiguous
 *           conditioning half of h (what the coupling's conditioner reads); ld[b] += pixels*(sum log_s - sum
 *           log_scale); W_out (C x C, nullable) receives W for the backward pass.
 *           = ActNorm.forward + InvertibleConv1x1.forward + the split gather  (modules.py:246-250, :470-482,
 *           coupling.py:33) in one launch.
 * backward: G = g_h + scatter(g_z1c) (g_z1c nullable); g_z = (W^T G) / exp(log_scale); g_log_scale, g_bias, g_W are
 *           ACCUMULATED (+=, caller zero-fills or passes .grad buffers); g_W then feeds nf_invconv_weight_bwd
 *           (which also adds the pixels * sum g_ld term of log_s; sum_g_ld (nullable, +=) receives sum_b g_ld so
 *           that launch can be given the scalar with B = 1).                                                   */
int nf_glow_head_fwd(const float* z, const float* log_scale, const float* bias, const float* P, const float* L,
                     const float* U, const float* L_mask, const float* U_mask, const float* sign_s, const float* log_s,
                     float* h, float* z1c, float* W_out, float* ld, int mode, int odd, int64_t B, int C, int H, int W,
                     nf_stream_t stream);
int nf_glow_head_bwd(const float* g_h, const float* g_z1c, const float* g_ld, const float* z, const float* log_scale,
                     const float* bias, const float* W_saved, float* g_z, float* g_log_scale, float* g_bias, float* g_W,
                     float* sum_g_ld, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream);
/* nf_glow_head_bwd in two parts (same arithmetic): the data gradient on the pass, the parameter sums of n <= NF_GLOW_HEAD_MULTI_MAX heads
 * of one shape and split mode in one launch where the pass ends (in front of nf_invconv_weight_bwd_multi, which reads g_W, sum_g_ld). */
typedef struct nf_glow_head_small_params_desc {
    const float *g_h, *g_z1c /* nullable */, *g_ld, *z, *log_scale, *bias, *W_saved;
    float *g_log_scale, *g_bias, *g_W, *sum_g_ld /* nullable */;
    int odd, reserved;
} nf_glow_head_small_params_desc;
int nf_glow_head_bwd_data(const float* g_h, const float* g_z1c, const float* log_scale, const float* W_saved, float* g_z, int mode, int odd,
                          int64_t B, int C, int H, int W, nf_stream_t stream);
int nf_glow_head_bwd_params_multi(const nf_glow_head_small_params_desc* descs, int n, int mode, int64_t B, int C, int H, int W,
                                  nf_stream_t stream);
/* The same head for 9 <= C <= 64 channels of image data with the 1x1 weight W given assembled (nf_invconv_weight_fwd_multi), on
 * the fp32 matrix cores; (H * W) % 16 == 0, mode NF_SPLIT_CHANNEL | NF_SPLIT_CHECKER.  (flows/modules.py:246-249, :470-482,
 * flows/coupling.py:33)
 *   fwd: h = W ((x - bias) / exp(log_scale)),  z1c = untouched half of h under the split map,  ld += H W (sum log_s - sum log_scale)
 *   bwd: g_x, and  g_log_scale / g_bias / g_W += their gradients (atomic: zero or accumulate-into at launch); the gradient of log_s
 *        and the PLU factors follows from g_W and sum g_ld as for nf_invconv_apply (nf_invconv_weight_bwd_multi).              */
int nf_glow_head_w_usable(int64_t B, int C, int H, int W, int mode);
int nf_glow_head_w_fwd(const float* x, const float* act_log_scale, const float* act_bias, const float* Wm, const float* log_s,
                       float* h, float* z1c, float* ld, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream);
int nf_glow_head_w_bwd(const float* g_h, const float* g_ld, const float* x, const float* act_log_scale, const float* act_bias,
                       const float* Wm, float* g_x, float* g_log_scale, float* g_bias, float* g_W, int64_t B, int C, int H, int W,
                       nf_stream_t stream);
/* The same backward in two parts (flows/modules.py:246-256, 478-487 differentiated; bit for bit what nf_glow_head_w_bwd computes):
 *   nf_glow_head_w_bwd_data          g_x alone -- the only result the rest of the backward pass waits for;
 *   nf_glow_head_w_bwd_params_multi  g_W, g_log_scale, g_bias (+=) of n <= NF_GLOW_HEAD_MULTI_MAX heads of ONE shape in one launch,
 *                                    from the g_h / g_ld / x the caller kept -- where the pass ends, in front of
 *                                    nf_invconv_weight_bwd_multi, which reads g_W.                                              */
#define NF_GLOW_HEAD_MULTI_MAX 128
typedef struct nf_glow_head_params_desc {
    const float *g_h, *g_ld, *x, *act_log_scale, *act_bias, *W;
    float *g_log_scale, *g_bias, *g_W;
} nf_glow_head_params_desc;
int nf_glow_head_w_bwd_data(const float* g_h, const float* act_log_scale, const float* Wm, float* g_x, int64_t B, int C, int H, int W,
                            nf_stream_t stream);
int nf_glow_head_w_bwd_params_multi(const nf_glow_head_params_desc* descs, int n, int64_t B, int C, int H, int W, nf_stream_t stream);

/* ---- Logit  modules.py:141-156 --------------------------------------------------------------------------------
 * forward: xc = clamp(x, eps, 1-eps); y = log(xc/(1-xc)); ld[b] += sum -(y - 2 softplus(y))
 * inverse: y = sigmoid(x); ld[b] += sum (x - 2 softplus(x)).      n = elements per sample.                       */
int nf_logit_fwd(const float* x, float* y, float* ld, float eps, int inverse, int64_t B, int64_t n,
                 nf_stream_t stream);
int nf_logit_bwd(const float* g_y, const float* g_ld, const float* x, float* g_x, float eps, int64_t B, int64_t n,
                 nf_stream_t stream);

/* The other elementwise bijector modules (flows/modules.py:125-183: Sigmoid, Tanh, Arctanh; no reference model builds them), one pass per
 * direction with ld[b] += the per-sample log-det.  kind 0: sigmoid (Sigmoid.forward), 1: logit of clamp(x, 1e-8, 1 - 1e-8)
 * (Sigmoid.backward), 2: tanh (Tanh.forward = Arctanh.backward), 3: arctanh (Tanh.backward = Arctanh.forward).  nf_bijector_bwd: the
 * autograd of kinds 0, 2, 3 as forward directions (g_x written).                                                                 */
#define NF_BIJ_SIGMOID 0
#define NF_BIJ_SIGMOID_INV 1
#define NF_BIJ_TANH 2
#define NF_BIJ_ARCTANH 3
int nf_bijector_fwd(const float* x, float* y, float* ld, int kind, int64_t B, int64_t n, nf_stream_t stream);
int nf_bijector_bwd(const float* g_y, const float* g_ld, const float* x, float* g_x, int kind, int64_t B, int64_t n, nf_stream_t stream);
/* Squeeze1d / Unsqueeze1d as flow layers (flows/squeeze.py:114-151): out (B, D) = cat(z[:, odd::2], z[:, 1 - odd::2]); inverse != 0: the
 * inverse map (in = the concatenated halves, out = the interleaved row).  D even.                                                 */
int nf_squeeze1d(const float* in, float* out, int64_t B, int D, int odd, int inverse, nf_stream_t stream);

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
/* Density data (two features, K <= 8): batches of at least `min_rows` rows take the one-row-per-thread kernels of the three
 * calls around this comment (shared transcendentals, staged 16-byte traffic: the bandwidth form), smaller ones the
 * one-component-per-lane kernels (the latency form).  Default 262 144 rows; min_rows < 0 restores it; returns 0.               */
int nf_mixlog_rows_config(int64_t min_rows);
int nf_mixlog_coupling_inv(const float* z, const float* params, const float* a_log_scale, const float* a_bias,
                           float* y, float* ld, float* scratch, int* stuck_flag, int K, int mode, int odd, int64_t B,
                           int C, int H, int W, nf_stream_t stream);
int nf_mixlog_coupling_bwd(const float* g_y, const float* g_ld, const float* z, const float* params,
                           const float* a_log_scale, const float* a_bias, float* g_z, float* g_params,
                           float* g_scale, float* g_bias, int K, float logit_eps, int mode, int odd, int64_t B,
                           int C, int H, int W, nf_stream_t stream);
/* image data: the same with g_scale / g_bias left as per-workgroup partial sums partials[0 .. n) | partials[n .. 2 n), n =
 * nf_mixlog_bwd_blocks(...) (0: the shape is served by a kernel without this form), for the caller's nf_slab_sum.              */
int nf_mixlog_bwd_blocks(int K, int mode, int64_t B, int C, int H, int W);
int nf_mixlog_coupling_bwd_partials(const float* g_y, const float* g_ld, const float* z, const float* params,
                                    const float* a_log_scale, const float* a_bias, float* g_z, float* g_params, float* partials,
                                    int K, float logit_eps, int mode, int odd, int64_t B, int C, int H, int W, nf_stream_t stream);

/* ---- standalone MixLogCDF  modules.py:186-212 (module surface forward / backward(x, log_pi, mu, s, log_df_dz)) -----
 * x, out (B, n); log_pi, mu, s (B, K, n) with n = the non-batch extent of x and log_pi already normalised over K
 * (coupling.py:180).  fwd: out = exp(logsumexp_k(log_pi + logsigmoid(u_k))), ld[b] += sum_e logsumexp_k(log_pi + logpdf_k).
 * inv: the bisection of modules.py:196-212 (bracket +-1e3, 25 iterations, 75 more iff some |hi-lo| >= 1e-4 after 25 -- the
 * reference's batch-global rule, device flag `stuck_flag` int32[1]), x = mid, ld[b] -= sum_e log pdf(x).  scratch: 2*B*n
 * floats.  bwd: analytic gradients with respect to x, log_pi, mu and s.  K <= 32.               */
int nf_mixlogcdf_fwd(const float* x, const float* log_pi, const float* mu, const float* s, float* out, float* ld, int K,
                     int64_t B, int64_t n, nf_stream_t stream);
int nf_mixlogcdf_bwd(const float* g_out, const float* g_ld, const float* x, const float* log_pi, const float* mu,
                     const float* s, float* g_x, float* g_log_pi, float* g_mu, float* g_s, int K, int64_t B, int64_t n,
                     nf_stream_t stream);
int nf_mixlogcdf_inv(const float* target, const float* log_pi, const float* mu, const float* s, float* x, float* ld,
                     float* scratch, int* stuck_flag, int K, int64_t B, int64_t n, nf_stream_t stream);

/* ---- Flow++ density step pair: the coupling above on two features (NF_SPLIT_1D, D = 2, K <= 8), followed by the NEXT flow
 * step's ActNorm (flows/flowpp.py:60-66 alternates ActNorm and coupling; flows/modules.py:246-249) in the same pass:
 *     h = (y - next_bias) / exp(next_log_scale),   ld += coupling log-det - sum_c next_log_scale[c]
 * backward takes g_h / g_ld for that output, ADDS the ActNorm's parameter gradients into g_next_log_scale / g_next_bias
 * (g_log_scale_c = -sum g_h h - sum g_ld, g_bias_c = -sum g_h / exp(log_scale_c)) next to g_scale / g_bias, and writes the
 * coupling's g_z, g_params as nf_mixlog_coupling_bwd does.  next_* are [2] device vectors and must not be NULL.          */
int nf_flowpp_vec_couple_fwd(const float* z, const float* params, const float* a_log_scale, const float* a_bias,
                             const float* next_log_scale, const float* next_bias, float* y, float* ld, int K,
                             float logit_eps, int odd, int64_t B, nf_stream_t stream);
int nf_flowpp_vec_couple_bwd(const float* g_h, const float* g_ld, const float* z, const float* params,
                             const float* a_log_scale, const float* a_bias, const float* next_log_scale,
                             const float* next_bias, float* g_z, float* g_params, float* g_scale, float* g_bias,
                             float* g_next_log_scale, float* g_next_bias, int K, float logit_eps, int odd, int64_t B,
                             nf_stream_t stream);

/* ---- fused masked / weight-normed linear + BatchNorm1d + ReLU chain on fp32 MFMA --------------------------------------
 * The building block of MADE (flows/maf.py:49-64: F.linear(z, W*M, b) -> BatchNorm1d -> relu) and of the MLP
 * conditioner (flows/modules.py:342-413: BatchNorm1d -> ReLU -> WeightNorm(Linear), residual adds).  One launch =
 * one linear layer of up to NF_MAX_NETS independent nets (MAF's s-net and t-net share a launch):
 *
 *     act[n,i] = in[n,i]                                      (no input BatchNorm)
 *              = relu( (in[n,i] - mean_i) * invstd_i * gamma_i + beta_i )   ("normalise on load")
 *     out[n,o] = sum_i act[n,i] * Weff[o,i] + bias[o] (+ residual[n,o])
 *     Weff     = weight * mask                                (MADE: "mask baked into the tile loader")
 *              = weight * weight_g / (||weight||_dim0 + wn_eps)   (flows/weight_norm.py:35-41)
 *     stat_sum[o] += sum_n (out - bias[o]),  stat_sqsum[o] += sum_n (out - bias[o])^2     ("statistics on store")
 *
 * training != 0: mean/var of `in` come from bn_sum/bn_sqsum/bn_center (what the producing launch stored; biased
 * variance, eps added under the root like nn.BatchNorm1d), block 0 updates running_mean / running_var (unbiased,
 * momentum) / num_batches_tracked and writes save_mean / save_invstd for the backward pass.
 * training == 0: running statistics are used.   Limits: I, O <= 32 (every reference conditioner is 32 wide).
 * The GEMM runs on v_mfma_f32_32x32x2_f32 (exact fp32, 32 rows x 32 outputs per wave per 16 issues).            */
#define NF_MAX_NETS 2
/* Atomically accumulated per-feature vectors (stat_sum/stat_sqsum, hence bn_sum/bn_sqsum; sum_g/sum_gx, hence
 * cbn_sum_g/cbn_sum_gx; g_bias) are REPLICATED: NF_STAT_REPL replicas of 32 floats (replica r at offset 32*r), workgroup
 * b adds into replica b % NF_STAT_REPL and every consumer sums the replicas -- same-address atomics cost ~23 ns each
 * serialised, replication cuts that chain by NF_STAT_REPL.                                                          */
#define NF_STAT_REPL 8
typedef struct nf_linear_desc {
    const float* in;          /* (N, I) */
    const float* weight;      /* (O, I) */
    const float* weight_g;    /* (I,)  weight-norm gain, or NULL */
    const float* mask;        /* (O, I) MADE mask, or NULL */
    const float* bias;        /* (O,) */
    const float* residual;    /* (N, O) or NULL */
    float* out;               /* (N, O) */
    const float* bn_gamma;    /* (I,) input BatchNorm affine; NULL = no input BatchNorm/ReLU */
    const float* bn_beta;
    const float* bn_sum;      /* (I,) training: sum_n (in - bn_center) */
    const float* bn_sqsum;    /* (I,) training: sum_n (in - bn_center)^2 */
    const float* bn_center;   /* (I,) */
    float* bn_running_mean;   /* (I,) */
    float* bn_running_var;    /* (I,) */
    int64_t* bn_num_batches;  /* scalar or NULL */
    float* bn_save_mean;      /* (I,) training: written */
    float* bn_save_invstd;    /* (I,) training: written */
    float* stat_sum;          /* (O,) or NULL */
    float* stat_sqsum;        /* (O,) or NULL */
} nf_linear_desc;
int nf_linear_bn_fwd(const nf_linear_desc* descs, int n_nets, int64_t N, int I, int O, int training, float bn_eps,
                     float bn_momentum, float wn_eps, nf_stream_t stream);

/* autograd of nf_linear_bn_fwd in training mode.  The gradient G of `out` is assembled on load:
 *     G[n,o] = g_direct[n,o] + g_skip[n,o] + BNbwd(gn_src)[n,o]
 *     BNbwd(gn)[n,o] = cbn_gamma_o * cbn_invstd_o * ( gn - cbn_sum_g_o / N - xhat * cbn_sum_gx_o / N ),
 *     xhat = (out[n,o] - cbn_mean_o) * cbn_invstd_o         (the BatchNorm that CONSUMED `out`; batch statistics
 *                                                            carry gradient)
 * (each term optional: NULL pointer = absent).  Results:
 *     g_store[n,o] = G                    (optional; needed where `out` also feeds a residual connection)
 *     g_bias[o]   += sum_n G ;  g_weff[slab][o,i] = this workgroup's part of sum_n G[n,o] act[n,i]  (gradient wrt
 *                    the EFFECTIVE weight; nf_linear_bwd_slabs(N) slabs, no zero-fill needed)
 *     gn_out[n,i]  = (sum_o G[n,o] Weff[o,i]) * [act[n,i] > 0]   and  sum_g[i] += sum_n gn_out,
 *     sum_gx[i]   += sum_n gn_out * xhat_in      (== g_beta, g_gamma of the input BatchNorm; the producer of `in`
 *                                                 finishes the BatchNorm backward on load)
 *     without input BatchNorm: gn_out = G Weff is the gradient of `in` itself.                                    */
typedef struct nf_linear_bwd_desc {
    const float* in;            /* (N, I) forward input */
    const float* weight;        /* (O, I) */
    const float* weight_g;      /* (I,) or NULL */
    const float* mask;          /* (O, I) or NULL */
    const float* bn_gamma;      /* (I,) input BatchNorm (NULL = none) */
    const float* bn_beta;
    const float* bn_save_mean;
    const float* bn_save_invstd;
    const float* g_direct;      /* (N, O) or NULL */
    const float* g_skip;        /* (N, O) or NULL */
    const float* gn_src;        /* (N, O) or NULL */
    const float* out;           /* (N, O) forward output (needed with gn_src) */
    const float* cbn_gamma;     /* (O,) consumer BatchNorm */
    const float* cbn_save_mean;
    const float* cbn_save_invstd;
    const float* cbn_sum_g;     /* (O,) */
    const float* cbn_sum_gx;    /* (O,) */
    float* g_store;             /* (N, O) or NULL */
    float* g_bias;              /* (O,) += */
    float* g_weff;              /* (nf_linear_bwd_slabs(N), O, I) written */
    float* gn_out;              /* (N, I) or NULL */
    float* sum_g;               /* (I,) += (with input BatchNorm) */
    float* sum_gx;              /* (I,) += */
} nf_linear_bwd_desc;
int nf_linear_bn_bwd(const nf_linear_bwd_desc* descs, int n_nets, int64_t N, int I, int O, float wn_eps,
                     nf_stream_t stream);

/* number of partial-sum slabs nf_linear_bn_bwd writes into g_weff for N rows (one per workgroup: the weight gradient
 * is reduced without atomics, deterministically, by nf_weight_grad_finalize).                                      */
int nf_linear_bwd_slabs(int64_t N);

/* gradient of the effective weight -> gradients of the stored parameters, n_layers jobs per launch (one workgroup
 * each).  g_weff holds n_slabs partial (O, I) sums from nf_linear_bn_bwd.
 *   mask != NULL    : g_weight = g_weff * mask                                         (maf.py:54)
 *   weight_g != NULL: weight-norm backward, g_weight (= g_v) and g_weight_g             (weight_norm.py:35-41)
 * plus up to two plain vector gradients finished in the same launch (a bias, or a BatchNorm's g_gamma / g_beta):
 * vec_dst = vec_src.  accumulate != 0 turns every store into += (direct accumulation into a .grad buffer).
 * g_weff == NULL: vector jobs only.                                                                               */
typedef struct nf_weight_grad_desc {
    const float* g_weff;    /* (n_slabs, O, I) or NULL */
    const float* weight;    /* (O, I) */
    const float* weight_g;  /* (I,) or NULL */
    const float* mask;      /* (O, I) or NULL */
    float* g_weight;        /* (O, I) */
    float* g_weight_g;      /* (I,) or NULL */
    const float* vec_src0;
    float* vec_dst0;
    const float* vec_src1;
    float* vec_dst1;
    int vec_n0;
    int vec_n1;
    int I;
    int O;
    int n_slabs;
    int accumulate;
    int vec_repl;           /* replicas of vec_src0 / vec_src1 to sum (1 or NF_STAT_REPL, stride 32 floats) */
    int reserved;
} nf_weight_grad_desc;
int nf_weight_grad_finalize(const nf_weight_grad_desc* descs, int n_layers, float wn_eps, nf_stream_t stream);

/* ---- fused 3x3 / 1x1 convolution + BatchNorm2d + ReLU chain on fp32 MFMA (NCHW) ---------------------------------------
 * The building block of the image conditioner ConvNet (flows/modules.py:416-438: WN(conv3x3) -> [BN -> ReLU -> WN(conv3x3)]
 * x 4 with residual adds -> BN -> ReLU -> WN(conv1x1)); the convolutional twin of nf_linear_bn_fwd with the same contract:
 *
 *     act[b,i,y,x] = in[b,i,y,x]                                                    (no input BatchNorm)
 *                  = relu( (in - mean_i) * invstd_i * gamma_i + beta_i )            ("normalise on load")
 *     out[b,o,y,x] = sum_{i,dy,dx} act[b,i,y+dy,x+dx] * weight[o,i,dy,dx] + bias[o] (+ residual[b,o,y,x])   zero padding
 *     stat_sum[o] += sum (out - bias[o]),  stat_sqsum[o] += sum (out - bias[o])^2   over (b, y, x)   ("statistics on store")
 *
 * weight is the EFFECTIVE weight (the weight-norm arithmetic of all convolutions of a model is one nf_weight_norm_fwd launch).
 * training / running statistics / save_mean / save_invstd exactly as nf_linear_bn_fwd, the batch count is B*H*W.
 * Limits: ksize 3 (pad 1): I <= 96, O <= 32;  ksize 1: I <= 32, O <= 192;  I <= 32 with an input BatchNorm, O <= 32 with
 * statistics; W a power of two and the spatial size tiling into 128-pixel groups of whole rows or whole samples
 * (nf_conv_bn_usable != 0).                              */
typedef struct nf_conv_desc {
    const float* in;          /* (B, I, H, W) */
    const float* weight;      /* (O, I, k, k) effective weight */
    const float* bias;        /* (O,) */
    const float* residual;    /* (B, O, H, W) or NULL */
    float* out;               /* (B, O, H, W) */
    const float* bn_gamma;    /* (I,) input BatchNorm affine; NULL = no input BatchNorm/ReLU */
    const float* bn_beta;
    const float* bn_sum;      /* NF_STAT_REPL x 32, training: sum (in - bn_center) */
    const float* bn_sqsum;
    const float* bn_center;   /* (I,) */
    float* bn_running_mean;   /* (I,) */
    float* bn_running_var;    /* (I,) */
    int64_t* bn_num_batches;  /* scalar or NULL */
    float* bn_save_mean;      /* (I,) training: written */
    float* bn_save_invstd;    /* (I,) training: written */
    float* stat_sum;          /* NF_STAT_REPL x 32 or NULL */
    float* stat_sqsum;
    const float* wpk;         /* optional: the weight's LDS images (nf_conv_weight_pack of this very weight); read by the large-batch
                               * 3x3 kernels (csrc/conv_bulk.hip), which split the weight themselves when it is NULL */
    int valid_h, valid_w;     /* 0 = the whole map.  Else the VALID extent of a map kept in power-of-two storage (H, W of the call):
                               * pixels with y >= valid_h or x >= valid_w are dead -- read as zero like the outside of the image,
                               * absent from every batch statistic (whose count is B * valid_h * valid_w), never written.  This is
                               * how maps whose sides are no powers of two (MNIST: 14 x 14, 7 x 7) run on these kernels.             */
} nf_conv_desc;
int nf_conv_bn_usable(int64_t B, int I, int O, int H, int W, int ksize);
/* Large batches (B*H*W > min_pixels, default 16384: beyond the persistent chain's 128 tiles): the 3x3 layers with <= 32 input and 32
 * output channels run on csrc/conv_bulk.hip -- independent waves, three-way bf16 split on the matrix pipe -- behind nf_conv_bn_fwd /
 * nf_conv_bn_bwd (data pass); same results to fp32 rounding.  Switches for tests and A/B runs (-1 = keep): on, min_pixels, nblk
 * (pixel blocks per wave: 0 automatic, 1, 2).  Environment: NF_CONV_BULK=0 (off).                                                  */
int nf_conv_bulk_config(int on, int64_t min_pixels, int nblk);
int nf_conv_bn_fwd(const nf_conv_desc* desc, int64_t B, int I, int O, int H, int W, int ksize, int training, float bn_eps,
                   float bn_momentum, nf_stream_t stream);

/* ---- the whole ConvNet conditioner in ONE persistent launch (csrc/conv_chain.hip)  modules.py:416-438 ---------------------
 * x (B, I0, H, W) -> conv3x3 -> 2 x [BN, ReLU, conv3x3, BN, ReLU, conv3x3, + skip] -> BN, ReLU, conv1x1 -> out (B, O, H, W), with
 * the effective (weight-normed) weights w[0..5] and biases b[0..5] of the six convolutions and the five BatchNorm2d layers
 * (training: batch statistics exchanged across the grid in-kernel, running statistics / num_batches_tracked updated; evaluation:
 * running statistics, no exchange).  Same results and the same by-products as six nf_conv_bn_fwd launches: acts[l] = output of
 * convolution l (bias and residual included), save_mean[l] / save_invstd[l] = what BatchNorm l normalised with -- what
 * nf_conv_bn_bwd reads.  A workgroup owns whole samples: H * W <= 256 and a power of two, ceil(B * H * W / tile) <=
 * NF_CONVNET_MAX_BLOCKS co-resident workgroups (nf_convnet_chain_usable != 0), tile = 256 / 128 pixels for 16 x 16 maps, 64 below
 * (128 when 64-pixel tiles would exceed the co-residency limit).
 * ws_zero: nf_convnet_chain_ws_floats(...) floats that are ZERO at launch (exchange slots: NF_CONVNET_WS_FLOATS for the BatchNorm
 * statistics; 16 x 16 maps at 2 B <= 128 split every sample over two workgroups, which hand their boundary rows to each other through
 * further slots -- then ws_zero is needed in evaluation mode too).  */
#define NF_CONVNET_MAX_BLOCKS 128
#define NF_CONVNET_WS_FLOATS (5 * (128 + 8) * 64 * 2)     /* a row of 64 eight-byte slots per workgroup and per group of 16, per BatchNorm */
int nf_convnet_chain_ws_floats(int64_t B, int I0, int O_out, int H, int W);
int nf_convnet_chain_blocks(int64_t B, int I0, int O_out, int H, int W);   /* workgroups of a launch (0: shape not taken) */
typedef struct nf_convnet_desc {
    const float* x;
    const float* w[6];
    const float* b[6];
    const float* gamma[5];
    const float* beta[5];
    float* rmean[5];
    float* rvar[5];
    int64_t* nbt[5];          /* each scalar or NULL */
    float* acts[5];           /* (B, 32, H, W), written */
    float* out;               /* (B, O, H, W), written */
    float* save_mean[5];      /* (32,), training: written */
    float* save_invstd[5];
    float* ws_zero;
    /* Optional (cp_z != NULL): the affine coupling the conditioner belongs to (flows/coupling.py:104-122), in the epilogue of the
     * output convolution.  The conditioner's input x is half 1 of the split map (cp_mode, cp_odd) of cp_z; with [shift | raw] the
     * convolution's O = 2 I0 outputs, s = cp_a tanh(raw) + cp_c and `out` receives [exp(s) | tanh(raw)] (exp(-s) when cp_inverse) --
     * what nf_convnet_chain_bwd reads as cp_out:
     *     cp_y = merge(z0 exp(s) + shift, z1),  cp_ld[b] += sum s      (cp_inverse: (z0 - shift) exp(-s), cp_ld[b] -= sum s).    */
    const float* cp_z;        /* (B, cp_C, Hf, Wf): Hf, Wf = H, W (NF_SPLIT_CHANNEL) or 2 H, 2 W (NF_SPLIT_CHECKER) */
    float* cp_y;              /* (B, cp_C, Hf, Wf) written */
    float* cp_ld;             /* (B,) += */
    const float* cp_a;        /* scalars */
    const float* cp_c;
    int cp_mode, cp_odd, cp_C, cp_inverse;
    /* Optional: the weights of layer l as LDS images (nf_conv_weight_pack below), NULL = the kernel splits w[l] itself.  With the
     * images a layer's weights reach LDS as direct global -> LDS loads issued under the previous layer's exchanges.            */
    const float* wpk[6];
    /* Optional (hd_x != NULL; needs the coupling, cp_z != NULL): the HEAD of an image Glow step -- ActNorm and the invertible 1 x 1
     * convolution in front of the coupling (flows/glow.py:24-46, flows/modules.py:246-249, 471-480) -- in the prologue of the launch, the
     * work of nf_glow_head_w_fwd without its launch:  h = W ((hd_x - hd_bias) / exp(hd_ls)) per pixel.  cp_z is then an OUTPUT (the
     * head's result h, (B, cp_C, Hf, Wf), kept for the backward), hd_x1 receives the conditioning half of h -- the conditioner's input:
     * `x` must point to the same buffer -- and cp_ld[b] += Hf Wf sum_c (hd_log_s[c] - hd_ls[c]).  9 <= cp_C <= 64 (hd_W is the
     * assembled cp_C x cp_C weight of the 1 x 1 convolution, nf_invconv_weight_fwd_multi).                                          */
    const float* hd_x;        /* (B, cp_C, Hf, Wf) the step's input */
    const float* hd_ls;       /* (cp_C,) ActNorm log_scale */
    const float* hd_bias;     /* (cp_C,) ActNorm bias */
    const float* hd_W;        /* (cp_C, cp_C) */
    const float* hd_log_s;    /* (cp_C,) log |diagonal of U| of the PLU factors */
    float* hd_x1;             /* (B, I0, H, W) written */
    /* ... of 2 <= cp_C <= 4 channels (hd_W == NULL): the weight is assembled in the launch from its PLU factors, as nf_glow_head_fwd
     * does, and saved to hs_Wout for the backward pass                                                                          */
    const float* hs_P;        /* (cp_C, cp_C) each: P, L, U, L_mask, U_mask; (cp_C,) sign_s */
    const float* hs_L;
    const float* hs_U;
    const float* hs_Lm;
    const float* hs_Um;
    const float* hs_sign;
    float* hs_Wout;           /* (cp_C, cp_C) written, or NULL */
    /* ws_gen = 0: ws_zero holds zeros.  ws_gen = k > 0: ws_zero was zero when the FIRST launch that shares it began and every launch
     * since (either direction, serialised on one stream) came with its own k: the launch tags its slots 8 k + 1 .. 8 k + 7 and takes
     * nothing else for an arrival (one buffer per train step instead of one per launch: fused_conv._chain_slots).                 */
    int ws_gen, ws_reserved;
} nf_convnet_desc;
int nf_convnet_chain_usable(int64_t B, int I0, int O_out, int H, int W);
int nf_convnet_chain_fwd(const nf_convnet_desc* desc, int64_t B, int I0, int O_out, int H, int W, int training, float bn_eps,
                         float bn_momentum, nf_stream_t stream);
/* The data gradient of the same conditioner in one persistent launch (the six nf_conv_bn_bwd data passes of the split form below):
 * reads the forward's by-products, writes what the deferred weight-gradient launches (nf_conv_bn_wgrad_multi) read afterwards.
 * Deterministic (the batch sums are exchanged in a fixed order, no atomics).  training == 0: the statistics are constants (no mean
 * terms in the BatchNorm backward); the sums are still produced -- they are the gradients of beta and gamma.                    */
typedef struct nf_convnet_bwd_desc {
    const float* w[6];          /* effective weights */
    const float* gamma[5];
    const float* beta[5];
    const float* save_mean[5];
    const float* save_invstd[5];
    const float* acts[5];       /* (B, 32, H, W) pre-BatchNorm outputs of the forward pass */
    const float* g_out;         /* (B, O, H, W) */
    float* gn[5];               /* (B, 32, H, W) written: gradient at BatchNorm l's output, ReLU mask applied */
    float* sum_g[5];            /* NF_STAT_REPL x 32, ZERO at launch; replica 0 receives the batch sum of gn[l] */
    float* sum_gx[5];           /* ... of gn[l] * xhat_l */
    float* g_store[2];          /* (B, 32, H, W) written: gradients of acts[4] and acts[2] (the g_skip operands of layers 2 and 0) */
    float* g_x;                 /* (B, I0, H, W) written, or NULL */
    float* ws_zero;             /* NF_CONVNET_WS_FLOATS zeros (exchange slots) */
    /* Optional (cp_g_y != NULL): the backward of the fused coupling.  g_out is then not read: the gradient of the conditioner's
     * output is computed from (cp_g_y, cp_g_ld, cp_z, cp_out) on the way into the first transposed convolution and WRITTEN to
     * cp_g_out (the deferred weight-gradient pass of the 1 x 1 convolution reads it); cp_g_z receives the full gradient of cp_z:
     * g_y0 exp(s) on the transformed half, g_y1 + (gradient of the conditioner's input) on the other (g_x is not written).    */
    const float* cp_g_y;        /* (B, cp_C, Hf, Wf) */
    const float* cp_g_ld;       /* (B,) */
    const float* cp_z;
    const float* cp_out;        /* (B, O, H, W) the forward's out: [exp(s) | tanh(raw)] */
    const float* cp_a;
    const float* cp_c;          /* (not read: the forward left exp(s)) */
    float* cp_g_z;              /* (B, cp_C, Hf, Wf) written */
    float* cp_g_out;            /* (B, O, H, W) written */
    float* cp_g_a;              /* scalars, += (atomic) */
    float* cp_g_c;
    int cp_mode, cp_odd, cp_C;
    int ws_gen;                 /* as in nf_convnet_desc */
    /* Optional: gradient accumulators of the BatchNorm parameters, (32,) each, += (one workgroup adds: no atomics). */
    float* g_gamma[5];
    float* g_beta[5];
    const float* wpk[6];        /* optional, as in nf_convnet_desc (the backward reads the transposed images of the same buffers) */
    /* Optional (hd_g_h != NULL, with the fused coupling, 2 <= cp_C <= 4 or 9 <= cp_C <= 64): the data gradient of the head that FOLLOWS this coupling
     * (ActNorm + 1 x 1 convolution of the next step, nf_glow_head_w_bwd_data) in the prologue -- cp_g_y is then WRITTEN first,
     * cp_g_y[c] = (sum_r hd_W[r][c] hd_g_h[r]) / exp(hd_ls[c]) per pixel, and read afterwards as usual.                        */
    const float* hd_g_h;        /* (B, cp_C, Hf, Wf) gradient at that head's output */
    const float* hd_W;          /* (cp_C, cp_C) its assembled weight */
    const float* hd_ls;         /* (cp_C,) its ActNorm log_scale */
} nf_convnet_bwd_desc;
int nf_convnet_chain_bwd(const nf_convnet_bwd_desc* desc, int64_t B, int I0, int O_out, int H, int W, int training,
                         nf_stream_t stream);

/* The chain kernels run their fp32 convolutions on the bf16 matrix pipe: every operand is split into three bf16 values (x = h + m + l
 * exactly), six of the nine partial products are accumulated in fp32 -- fp32 accuracy (measured error against float64 below that of
 * v_mfma_f32_32x32x2_f32, tools/probes/bf16x3_probe.hip) at 0.375 of the matrix-pipe time.  nf_conv_weight_pack writes, for n
 * effective weights, the split weights in the exact LDS layout the kernels read (an "image": 3 planes x 36 slots x 32 rows x 8 bf16 =
 * NF_CONV_PACK_IMAGE_FLOATS floats), once per pass instead of once per workgroup and launch:
 *   3 x 3 (O = 32, I <= 96):  ceil(I / 32) forward images (K = input channel chunk), then as many transposed ones (K = output channel)
 *   1 x 1 (I = 32, O <= 192): one forward image in the row order of the fused coupling (row 2 p = shift channel, 2 p + 1 = its
 *                             scale channel), one transposed image.
 * dst: nf_conv_weight_pack_images(O, I, ksize) * NF_CONV_PACK_IMAGE_FLOATS floats per weight (0 images = shape not packable).       */
#define NF_CONV_PACK_IMAGE_FLOATS (3 * 36 * 128)
#define NF_CONV_PACK_MAX_LAYERS 1024
typedef struct nf_conv_pack_desc {
    const float* w;             /* (O, I, k, k) effective weight */
    float* dst;
    int O, I, ksize, reserved;
} nf_conv_pack_desc;
int nf_conv_weight_pack_images(int O, int I, int ksize);
int nf_conv_weight_pack(const nf_conv_pack_desc* descs, int n, nf_stream_t stream);
/* Self-test of that arithmetic (no reference counterpart: flows/modules.py:416-438's convolutions are plain fp32): D (32, 32) =
 * A (32, K) * B (K, 32) on one wave, K a multiple of 16; mode 0 = v_mfma_f32_32x32x2_f32, mode 1 = the chain kernels' own split
 * and six-product accumulation.  tests/test_gpu_ops.py asserts error(mode 1) <= error(mode 0) against float64.                        */
int nf_selftest_gemm32(const float* A, const float* B, float* D, int K, int mode, nf_stream_t stream);

/* autograd of nf_conv_bn_fwd in training mode; the gradient G of `out` is assembled on load exactly as in
 * nf_linear_bn_bwd (G = g_direct + g_skip + BNbwd(gn_src), each term optional).  Results:
 *     g_store = G (optional);  g_bias[replica][o] += sum G  (NF_STAT_REPL replicas, stride 256 floats);
 *     g_weff[slab] = this workgroup's part of the gradient of the effective weight, stored (k*k, O, I) so that the stores
 *                    coalesce (nf_conv_bwd_slabs slabs, written; nf_slab_sum with taps = k*k sums them into (O, I, k, k));
 *     gn_out = (transposed convolution of G) * [act > 0], sum_g / sum_gx += its batch sums (replicas, stride 32) -- or the
 *     gradient of `in` itself without an input BatchNorm.                                                              */
typedef struct nf_conv_bwd_desc {
    const float* in;            /* (B, I, H, W) forward input */
    const float* weight;        /* (O, I, k, k) effective weight */
    const float* bn_gamma;      /* (I,) input BatchNorm (NULL = none) */
    const float* bn_beta;
    const float* bn_save_mean;
    const float* bn_save_invstd;
    const float* g_direct;      /* (B, O, H, W) or NULL */
    const float* g_skip;
    const float* gn_src;
    const float* out;           /* forward output (needed with gn_src) */
    const float* cbn_gamma;     /* (O,) consumer BatchNorm */
    const float* cbn_save_mean;
    const float* cbn_save_invstd;
    const float* cbn_sum_g;     /* NF_STAT_REPL x 32, or NULL (evaluation-mode statistics are constants) */
    const float* cbn_sum_gx;
    float* g_store;             /* (B, O, H, W) or NULL */
    float* g_bias;              /* NF_STAT_REPL x 256, += ; or NULL */
    float* g_weff;              /* (nf_conv_bwd_slabs, k*k, O, I) written */
    float* gn_out;              /* (B, I, H, W) or NULL */
    float* sum_g;               /* NF_STAT_REPL x 32, += (with input BatchNorm) */
    float* sum_gx;
    const float* wpk;           /* optional, as in nf_conv_desc (the data gradient reads the transposed image of the same buffer) */
    int valid_h, valid_w;       /* as in nf_conv_desc (0 = the whole map) */
} nf_conv_bwd_desc;
int nf_conv_bwd_slabs(int64_t B, int H, int W);
int nf_conv_bn_bwd(const nf_conv_bwd_desc* desc, int64_t B, int I, int O, int H, int W, int ksize, nf_stream_t stream);
/* Split form for a model's backward pass: nf_conv_bn_bwd with g_weff == NULL (then g_bias must be NULL too) computes only what
 * the next layer waits for -- g_store, gn_out, sum_g / sum_gx.  The weight-gradient slabs and bias sums of up to
 * NF_CONV_WGRAD_MAX layers of one shape (B, I, O, H, W, ksize) are then produced by ONE launch of nf_conv_bn_wgrad_multi, any
 * time after the layers' data passes (same descriptors, with g_weff / g_bias set; g_store, gn_out, sum_g, sum_gx are ignored).
 * Nothing but the optimizer waits for them: deferred to the end of the backward pass they fill the machine instead of sitting
 * on every layer's latency chain.                                                                                          */
#define NF_CONV_WGRAD_MAX 16
int nf_conv_bn_wgrad_multi(const nf_conv_bwd_desc* descs, int n, int64_t B, int I, int O, int H, int W, int ksize,
                           nf_stream_t stream);
/* Slabs per layer of an nf_conv_bn_wgrad_multi launch of n_layers layers -- g_weff of each descriptor holds THIS many slabs (not
 * nf_conv_bwd_slabs): one workgroup per compute unit over the whole launch, each keeping its tap tiles in registers over
 * tiles / slabs tiles.                                                                                                     */
int nf_conv_wgrad_slabs(int64_t B, int H, int W, int n_layers);
/* The same pass for ANY number of layers of one shape in ONE launch (round 6: config 4 at 64 samples per GPU queues 320 hidden layers per
 * resolution; sixteen per launch were 2 .. 8 tiles per workgroup -- prologue, slab write and launch gap outweighed the tiles).  The
 * descriptors do not fit the kernel arguments: they are written to the device table `table_dev` (n descriptors, caller's scratch) by tiny
 * launches of up to 320 in front of the pass (by-value arguments: a captured hipGraph replays them without touching host memory) and read
 * there by the workgroups.  `slabs` workgroups per layer (1 .. 128, <= tiles of 128 pixels); g_weff of each descriptor holds that many.  */
#define NF_CONV_WGRAD_TABLE_MAX 4096
int nf_conv_bn_wgrad_table(const nf_conv_bwd_desc* descs, nf_conv_bwd_desc* table_dev, int n, int slabs, int64_t B, int I, int O, int H,
                           int W, int ksize, nf_stream_t stream);

/* dst[e] (+)= sum_{s < n_slabs} src[s * stride + e], e < n: every slab / replica sum of one conditioner backward in ONE
 * launch (weight-gradient slabs, bias and BatchNorm-parameter replicas).                                                */
#define NF_SLAB_SUM_MAX 1024     /* (1024 x 48 bytes of descriptors + the per-job workgroup ranges = 52 KB of kernel arguments.  Rounds 1 - 5 kept every
                                  * descriptor array of a multi-launch under 4 KB; gfx950 / ROCm 7.2 takes 64 KB -- measured -- and a launch costs ~4.5 us
                                  * inside a hipGraph whatever it does: the C4 step went from 693 to ~560 launches by raising these caps alone)            */
typedef struct nf_slab_sum_desc {
    const float* src;
    float* dst;
    int64_t n;
    int64_t stride;
    int n_slabs;
    int accumulate;
    int taps;       /* > 1: the slabs are nf_conv_bn_bwd's (taps, O, I) order and dst is the (O, I, taps) weight gradient */
    int reserved;
} nf_slab_sum_desc;
int nf_slab_sum(const nf_slab_sum_desc* descs, int n_jobs, nf_stream_t stream);

/* ---- invertible residual block (Residual Flow), D <= 4 features, hidden width 32  iresblock.py:17-109, :229-278 ------
 * g(x) = W3 lipswish(W2 lipswish(W1 x + b1) + b2) + b3 with the EFFECTIVE (spectrally normalised) weights.
 * nf_resmlp_fwd: y = x + g(x) (y nullable) and, by `mode`: 0 nothing; 1 ld[b] += ld_sign * log|det(I + J_b)| (exact,
 *   iresblock.py:17-32); 2 ld[b] += ld_sign * mean_s sum_{k<=n_terms[s]} coef[s*64 + k-1] * v_s^T (J_b^T)^k v_s with host-
 *   drawn noise (B, S, D) -- the power-series / Russian-roulette estimators (iresblock.py:35-81) on the exact per-sample
 *   Jacobian instead of nested autograd sweeps.
 * nf_resmlp_fixed_point_step: one iteration x <- z - g(x) of the inverse (iresblock.py:243-249); flags (int32[>=it+1],
 *   zeroed by the caller) carry the reference's batch-global exit: the step is a no-op when flags[it-1] == 0, and it sets
 *   flags[it] when some |x_new - x| >= ftol.
 * nf_spectral_weights: flows/spectral_norm.py:26-43 for up to 3 matrices in one launch (one power iteration, u / v
 *   updated in place, W_eff = W_bar * min(coeff/(sigma+eps), 1)); gated by the same flags when flags != NULL.          */
int nf_resmlp_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                  const float* b3, const float* beta1, const float* beta2, float* y, float* ld, float ld_sign, int mode,
                  const float* noise, const float* coef, const int* n_terms, int S, int64_t B, int D, nf_stream_t stream);
int nf_resmlp_fixed_point_step(const float* z, float* x, const float* W1, const float* b1, const float* W2, const float* b2,
                               const float* W3, const float* b3, const float* beta1, const float* beta2, int* flags,
                               int iteration, float ftol, int64_t B, int D, nf_stream_t stream);
int nf_spectral_weights(const float* const* W_bar, float* const* u, float* const* v, float* const* W_eff, const int* rows,
                        const int* cols, int n_mats, float coeff, float eps, const int* flags, int iteration,
                        nf_stream_t stream);

/* Training backward of the block (iresblock.py:84-109, :112-185): gradients of  <d_g, g(x)> + d_ld[0] * s^T J v  with
 * s = v + sum_{k <= n_terms} coef[k-1] (J^T)^k v held constant (the Neumann-series gradient estimator; v = noise (B, D), coef
 * from the host's Russian-roulette draw), second derivatives of the LipSwish network in closed form.  d_x (B, D) = gradient of
 * x through g and the surrogate (the residual connection's identity term is the caller's); g_params, ZERO at launch, receives
 * the gradients of the EFFECTIVE parameters: W1 (32 x D) | b1 (32) | W2 (32 x 32) | b2 (32) | W3 (D x 32) | b3 (D) | beta1 | beta2.
 * nf_spectral_weights_bwd maps gradients of effective weights to the stored ones (spectral_norm.py:36-43, u / v constant),
 * ACCUMULATING into g_W_bar.                                                                                                */
int nf_resmlp_train_bwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                        const float* b3, const float* beta1, const float* beta2, const float* noise, const float* coef,
                        int n_terms, const float* d_g, const float* d_ld, float* d_x, float* g_params, int64_t B, int D,
                        nf_stream_t stream);
int nf_spectral_weights_bwd(const float* const* W_bar, const float* const* u, const float* const* v,
                            const float* const* g_W_eff, float* const* g_W_bar, const int* rows, const int* cols, int n_mats,
                            float coeff, float eps, nf_stream_t stream);

/* ---- weight normalisation of many layers per launch  flows/weight_norm.py:35-41 ----------------------------------------
 * w[o, m] = v[o, m] * g[m] / (||v[:, m]|| + eps), m = (input channel, ky, kx) flattened, O = output channels.
 * forward writes w; backward writes (accumulate = 0) or adds to (1) g_v, g_g from g_w.  <= NF_WN_MAX_LAYERS layers per call. */
#define NF_WN_MAX_LAYERS 896
typedef struct nf_wn_desc {
    const float* v;        /* (O, M) weight_v */
    const float* g;        /* (M,) weight_g */
    float* w;              /* (O, M) effective weight (forward) */
    const float* g_w;      /* (O, M) gradient of w (backward) */
    float* g_v;            /* (O, M) */
    float* g_g;            /* (M,) */
    int O;
    int M;
    int accumulate;
    int reserved;
} nf_wn_desc;
int nf_weight_norm_fwd(const nf_wn_desc* descs, int n_layers, float eps, nf_stream_t stream);
int nf_weight_norm_bwd(const nf_wn_desc* descs, int n_layers, float eps, nf_stream_t stream);

/* ---- the whole MLP conditioner in one persistent launch (small / medium batches) ------------------------------------
 * flows/modules.py:393-413 with n_blocks = 2: six weight-normed linears (width 32), five BatchNorm1d.  Same mathematics
 * as the nf_linear_bn_* chain; used when N <= NF_MLP_MAX_ROWS, where those launches are latency-bound: one 16-wave
 * workgroup per NF_MLP_ROWS_PER_BLOCK rows keeps every activation in registers, training-mode batch statistics cross
 * the grid through NF_STAT_REPL-replicated atomics + a software grid barrier (all workgroups co-resident by construction).
 * params: NF_MLP_N_PARAM_PTRS device pointers on the HOST, linear l = 0..5: weight_v (O_l, I_l), weight_g (I_l), bias (O_l);
 *         then BatchNorm j = 0..4: gamma, beta, running_mean, running_var, num_batches_tracked (int64, may be NULL).
 * save_stats (5, 2, 32): batch mean and 1/sqrt(var + eps) per BatchNorm, written in training mode (backward input).
 * ws_zero: NF_MLP_WS_FLOATS floats that are ZERO at launch (per-workgroup exchange slots + barrier counter).               */
#define NF_MLP_LINEARS 6
#define NF_MLP_BNS 5
#define NF_MLP_N_PARAM_PTRS 43
#define NF_MLP_ROWS_PER_BLOCK 128
#define NF_MLP_MAX_BLOCKS 128
#define NF_MLP_MAX_ROWS 16384
#define NF_MLP_WS_FLOATS (6 * 128 * 64 * 2 + 64)
int nf_mlp_chain_fwd(const float* x, const void* const* params, float* out, float* save_stats, float* ws_zero, int64_t N,
                     int I0, int O_out, int training, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
/* autograd of nf_mlp_chain_fwd, one launch (the forward is recomputed from x and save_stats; evaluation mode takes the
 * running statistics as constants).  grads: NF_MLP_N_GRAD_PTRS device pointers on the HOST, linear l: g_weight_v, g_weight_g,
 * g_bias; then BatchNorm j: g_gamma, g_beta -- written (accumulate = 0) or += (accumulate = 1, e.g. .grad buffers).
 * g_x (N, I0) written, nullable.  ws_zero as above (a fresh zero region per call); slabs: NF_MLP_BWD_SLAB_FLOATS floats of
 * scratch (contents irrelevant, re-usable by the next call on the stream).                                            */
#define NF_MLP_N_GRAD_PTRS 28
#define NF_MLP_BWD_SLAB_FLOATS (128 * 7 * 2 * 1056)
int nf_mlp_chain_bwd(const float* x, const void* const* params, const float* save_stats, const float* g_out, float* g_x,
                     void* const* grads, int accumulate, float* ws_zero, float* slabs, int64_t N, int I0, int O_out,
                     int training, float bn_eps, float wn_eps, nf_stream_t stream);

/* ---- one whole Glow flow step on vector data in one persistent launch per direction --------------------------------
 * dims = (D,), D = 2 or 4: ActNorm (modules.py:246-250) -> invertible 1x1 with the PLU weight assembled in-kernel
 * (modules.py:470-482) -> affine coupling (coupling.py:104-113, 1-D split squeeze.py:68-69) whose conditioner is the MLP
 * of nf_mlp_chain_*.  Replaces glow head + conditioner + coupling launches (3 forward, 4 backward) for N <= NF_MLP_MAX_ROWS.
 * head: NF_GLOW_HEAD_PTRS device pointers on the HOST: actnorm log_scale (D), bias (D); P, L, U, L_mask, U_mask (D, D);
 *       sign_s, log_s (D); the coupling's s_log_scale, s_bias (1).   y (N, D) written; ld (N) += log-det of the step.
 * backward: g_z (N, D) written; g_ld (N) nullable (the step adds to ld, so d/d ld passes through unchanged and also
 * feeds log_scale, log_s and the coupling scale); head_grads: NF_GLOW_HEAD_GRAD_PTRS pointers: g_log_scale, g_bias, g_L,
 * g_U, g_log_s, g_s_log_scale, g_s_bias; mlp_grads / accumulate / ws_zero / slabs as in nf_mlp_chain_bwd.          */
#define NF_GLOW_HEAD_PTRS 11
#define NF_GLOW_HEAD_GRAD_PTRS 7
int nf_glow_step_vec_fwd(const float* z, float* y, float* ld, const void* const* head, const void* const* mlp_params,
                         float* save_stats, float* ws_zero, int64_t N, int D, int odd, int training, float bn_eps,
                         float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_glow_step_vec_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                         const void* const* mlp_params, const float* save_stats, void* const* head_grads,
                         void* const* mlp_grads, int accumulate, float* ws_zero, float* slabs, int64_t N, int D, int odd,
                         int training, float bn_eps, float wn_eps, nf_stream_t stream);
/* the INVERSE of the same step in one launch (ActNorm.backward o InvertibleConv1x1.backward o AffineCoupling.backward,
 * modules.py:250-256, :484-497, coupling.py:115-122): y (N, D) -> z, ld -= the step's log-det.  head as above EXCEPT slot 2:
 * the row-swap matrix of the LAPACK pivots (what torch.lu_solve applies to its right-hand side; = P^T for consistent factors);
 * W^-1 = U'^-1 L'^-1 Pp is formed inside.  The conditioner runs in the mode `training` says (batch statistics + running-
 * statistics update, or running statistics); save_stats: NF_GLOW_FLOW_SAVE_FLOATS floats of scratch.                          */
int nf_glow_step_vec_inv(const float* y, float* z, float* ld, const void* const* head, const void* const* mlp_params,
                         float* save_stats, float* ws_zero, int64_t N, int D, int odd, int training, float bn_eps,
                         float bn_momentum, float wn_eps, nf_stream_t stream);

/* ---- one whole RealNVP flow step on vector data in one persistent launch per direction (training mode) -----------------
 * dims = (D,), D = 2 or 4: flow BatchNorm with batch statistics (modules.py:283-307, affine=False) -> affine coupling whose
 * conditioner is the MLP of nf_mlp_chain_* (coupling.py:104-113).  head: NF_REALNVP_HEAD_PTRS pointers on the HOST: flow-BN
 * log_gamma, beta, batch_mean, batch_var, running_mean, running_var (D each; the four buffers are updated), s_log_scale,
 * s_bias (1).  save_stats: NF_REALNVP_SAVE_FLOATS floats.  Everything else as in nf_glow_step_vec_*.                       */
#define NF_REALNVP_HEAD_PTRS 8
#define NF_REALNVP_SAVE_FLOATS 328
int nf_realnvp_step_vec_fwd(const float* z, float* y, float* ld, const void* const* head, const void* const* mlp_params,
                            float* save_stats, float* ws_zero, int64_t N, int D, int odd, float flow_bn_eps,
                            float flow_bn_momentum, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_realnvp_step_vec_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                            const void* const* mlp_params, const float* save_stats, float* g_s_log_scale, float* g_s_bias,
                            void* const* mlp_grads, int accumulate, float* ws_zero, float* slabs, int64_t N, int D, int odd,
                            float bn_eps, float wn_eps, nf_stream_t stream);
/* Evaluation mode and the inverse.  flow_bn_momentum = NF_FBN_RUNNING in nf_realnvp_step_vec_fwd / nf_realnvp_flow_pack selects
 * evaluation mode for the whole step (flow BatchNorm and conditioner on their running statistics, modules.py:296-298; no
 * buffer is touched, no grid exchange).  nf_realnvp_step_vec_inv is the step's INVERSE in one launch (AffineCoupling.backward
 * + BatchNorm.backward, coupling.py:115-122, modules.py:309-322): y -> z, ld -= the step's log-det; training != 0: the flow
 * BatchNorm inverts with its stored BATCH buffers and the conditioner runs in training mode (what the reference does), else
 * running statistics.  Records of an inverse run are packed with NF_FBN_BATCH_BUFFERS or NF_FBN_RUNNING accordingly.          */
#define NF_FBN_RUNNING (-1.0f)
#define NF_FBN_BATCH_BUFFERS (-2.0f)
int nf_realnvp_step_vec_inv(const float* y, float* z, float* ld, const void* const* head, const void* const* mlp_params,
                            float* save_stats, float* ws_zero, int64_t N, int D, int odd, int training, float bn_eps,
                            float bn_momentum, float wn_eps, nf_stream_t stream);

/* ---- one whole MAF flow step on vector data in one persistent launch per direction (training mode) ------------------
 * dims = (D,), D <= 4, N <= NF_MAF_MAX_ROWS: flow BatchNorm with batch statistics (modules.py:283-307, affine=False) ->
 * z @ perm (maf.py:100) -> the MADE pair (maf.py:49-64; masks applied while the weights are staged) -> affine transform of
 * all features + log-det (maf.py:103-106).  Replaces 7 launches forward / 13 backward of the per-layer path.
 * head: NF_MAF_HEAD_PTRS device pointers on the HOST: flow-BN log_gamma, beta, batch_mean, batch_var, running_mean,
 *       running_var (D each; the four buffers are updated), perm (D, D), s_log_scale, s_bias (1).
 * made_params: NF_MAF_PARAM_PTRS pointers: net s then net t, each: layer l = 0..3: weight (O_l, I_l), mask (O_l, I_l),
 *       bias (O_l); then BatchNorm1d j = 0..2: gamma, beta, running_mean, running_var, num_batches_tracked (int64 / NULL).
 * save_stats: NF_MAF_SAVE_FLOATS floats written by the forward, read by the backward.  ws_zero: NF_MAF_WS_FLOATS floats that
 * are ZERO at launch.  Backward: g_z (N, D) written; g_ld nullable (passes through unchanged to the caller's graph);
 * made_grads: NF_MAF_GRAD_PTRS pointers (net s, net t: per layer g_weight, g_bias; per BatchNorm g_gamma, g_beta) and the two
 * scalars g_s_log_scale, g_s_bias are ACCUMULATED by atomics: pass .grad buffers or zero-filled temporaries.
 * slabs: NF_MAF_BWD_SLAB_FLOATS floats of scratch.                                                                        */
#define NF_MAF_HEAD_PTRS 9
#define NF_MAF_PARAM_PTRS 54
#define NF_MAF_GRAD_PTRS 28
#define NF_MAF_ROWS_PER_BLOCK 128
#define NF_MAF_MAX_BLOCKS 128
#define NF_MAF_MAX_ROWS 16384
#define NF_MAF_SAVE_FLOATS 392
#define NF_MAF_WS_FLOATS ((4 * 128 * 128 + 4 * 8 * 128) * 2 + 64)
#define NF_MAF_BWD_SLAB_FLOATS (128 * 2 * 4 * 2 * 1056)
int nf_maf_step_fwd(const float* z, float* y, float* ld, const void* const* head, const void* const* made_params,
                    float* save_stats, float* ws_zero, int64_t N, int D, float flow_bn_eps, float flow_bn_momentum,
                    float bn_eps, nf_stream_t stream);
/* nf_maf_step_fwd with flow_bn_momentum = NF_FBN_RUNNING: the evaluation-mode step (every BatchNorm on its running statistics,
 * no buffer touched, no grid exchange; save_stats unused).  nf_maf_step_inv: the INVERSE step in one launch
 * (AutoregressiveTransfrom.backward, maf.py:108-119: D sequential passes of both MADEs, pass i fixes feature i; then perm^T and
 * BatchNorm.backward, modules.py:309-322): y -> z, ld -= the step's log-det.  training != 0: the MADE BatchNorms normalise with
 * batch statistics and update their running statistics once per pass, the flow BatchNorm inverts with its batch buffers;
 * ws_zero: D x NF_MAF_WS_FLOATS zero floats (one exchange workspace per pass).  Unlike the reference the input is not mutated. */
int nf_maf_step_inv(const float* y, float* z, float* ld, const void* const* head, const void* const* made_params, float* ws_zero,
                    int64_t N, int D, int training, float bn_eps, nf_stream_t stream);
int nf_maf_step_bwd(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                    const void* const* made_params, const float* save_stats, void* const* made_grads, float* g_s_log_scale,
                    float* g_s_bias, float* ws_zero, float* slabs, int64_t N, int D, nf_stream_t stream);
/* Deferred fold for a flow of S such steps (flows/maf.py stacks them): nf_maf_step_bwd_partial is nf_maf_step_bwd without its
 * last two phases (the fenced grid exchange and the fold of the weight-gradient slabs) -- g_z and the BatchNorm1d gamma / beta
 * gradients are complete on return, the step's slabs (slabs_step: blocks x NF_MAF_SLAB_WG_FLOATS, blocks = ceil(N /
 * NF_MAF_ROWS_PER_BLOCK), a region of its own per step) and scalar sums (head_rec_step: blocks x NF_MAF_HEAD_REC_WG floats) stay
 * behind.  nf_maf_fold_all then ACCUMULATES the weight / bias / s_log_scale / s_bias gradients of all S steps in one launch
 * (per 16 steps): made_params_all = S x NF_MAF_PARAM_PTRS and made_grads_all = S x NF_MAF_GRAD_PTRS pointers (host arrays, the
 * steps' tables back to back), g_s_log_scale_all / g_s_bias_all = S pointers each; slabs_all / head_rec_all = the S regions
 * back to back, in the same step order.                                                                                    */
#define NF_MAF_SLAB_WG_FLOATS (2 * 4 * 2 * 1056)
#define NF_MAF_HEAD_REC_WG 16
int nf_maf_step_bwd_partial(const float* z, const float* g_y, const float* g_ld, float* g_z, const void* const* head,
                            const void* const* made_params, const float* save_stats, void* const* made_grads, float* ws_zero,
                            float* slabs_step, float* head_rec_step, int64_t N, int D, nf_stream_t stream);
int nf_maf_fold_all(const void* const* made_params_all, void* const* made_grads_all, float* const* g_s_log_scale_all,
                    float* const* g_s_bias_all, int S, const float* slabs_all, const float* head_rec_all, int blocks, int D,
                    nf_stream_t stream);

/* ---- the whole backward of one Flow++ density flow step on (N, 2) data, K <= 8 -------------------------------------------
 * = nf_flowpp_vec_couple_bwd (or nf_mixlog_coupling_bwd when next_* are NULL) followed by nf_flowpp_cond_bwd on the
 * conditioning feature, as two launches: the coupling's backward runs inside the conditioner's backward kernel, which computes
 * the gradient of the (N, 2 + 3K) conditioner output from the SAVED output `params` instead of reading it.  g_z (N, 2) is
 * written; every parameter gradient is ACCUMULATED.  workspace: NF_FLOWPP_BWD_WS_FLOATS floats.  phase: 0 = both launches on
 * `stream`; 1 = the backward kernel only, 2 = the slab finalize only (same arguments) -- a caller may put the finalize on a second
 * stream so that it overlaps the next step's backward kernel; the workspace is then busy until that finalize has run.        */
/* forward of the same step in ONE launch: params (N, 2 + 3K) is written for the backward, y (N, 2) written, ld (N,) += .      */
int nf_flowpp_vec_step_fwd(const float* z, const float* W0, const float* b0, const float* Wg, const float* bg, const float* ln1_g,
                           const float* ln1_b, const float* pos, const float* Wq, const float* bq, const float* W2,
                           const float* b2, const float* ln2_g, const float* ln2_b, const float* W5, const float* b5,
                           const float* a_log_scale, const float* a_bias, const float* next_log_scale, const float* next_bias,
                           float* params, float* y, float* ld, int K, float logit_eps, int odd, int64_t N, nf_stream_t stream);
int nf_flowpp_vec_step_bwd(const float* g_h, const float* g_ld, const float* z, const float* params, const float* W0,
                           const float* b0, const float* Wg, const float* bg, const float* ln1_g, const float* ln1_b,
                           const float* pos, const float* Wq, const float* bq, const float* W2, const float* b2,
                           const float* ln2_g, const float* ln2_b, const float* W5, const float* b5, const float* a_log_scale,
                           const float* a_bias, const float* next_log_scale, const float* next_bias, float* g_z, float* g_W0,
                           float* g_b0, float* g_Wg, float* g_bg, float* g_ln1_g, float* g_ln1_b, float* g_pos, float* g_Wq,
                           float* g_bq, float* g_W2, float* g_b2, float* g_ln2_g, float* g_ln2_b, float* g_W5, float* g_b5,
                           float* g_scale, float* g_bias, float* g_next_log_scale, float* g_next_bias, float* workspace, int K,
                           float logit_eps, int odd, int64_t N, int phase, nf_stream_t stream);
/* Deferred finalize: a flow of such steps calls nf_flowpp_vec_step_bwd with phase = 1 (the backward kernel only: g_z complete,
 * the parameter-gradient partials stay in `workspace` -- one NF_FLOWPP_BWD_WS_FLOATS region PER STEP, untouched until folded)
 * and, after the last step, nf_flowpp_vec_step_finalize ONCE with every step's descriptor: the same accumulation into the
 * gradient buffers as phase 2, eight steps per launch.  All steps share K and N.                                            */
typedef struct nf_flowpp_fin_desc {
    const float* workspace;
    float *g_W0, *g_b0, *g_Wg, *g_bg, *g_ln1_g, *g_ln1_b, *g_pos, *g_Wq, *g_bq, *g_W2, *g_b2, *g_ln2_g, *g_ln2_b, *g_W5, *g_b5;
    float *g_scale, *g_bias;
    const float* next_log_scale;                          /* NULL: no ActNorm rides the step                                  */
    float *g_next_log_scale, *g_next_bias;
    int odd, reserved;
} nf_flowpp_fin_desc;
int nf_flowpp_vec_step_finalize(const nf_flowpp_fin_desc* descs, int n, int K, int64_t N, nf_stream_t stream);

/* ---- a whole flow of S fused vector Glow steps (nf_glow_step_vec_*) in ONE launch per direction ---------------------------
 * flows/glow.py: the (N, D in {2, 4}) model IS a sequence of such steps; rows stay in their workgroup from step to step, the
 * batch statistics are exchanged grid-wide inside the launch exactly as in the single-step kernels.
 *   steps_dev : DEVICE array of S step records (nf_glow_flow_step_bytes() bytes each), packed on the host one by one with
 *               nf_glow_flow_pack from the same pointer tables nf_glow_step_vec_fwd / _bwd take (grad tables may be NULL for
 *               a forward-only flow) and copied to the device by the caller; pointers only, valid while the parameters live;
 *   ys        : (S, N, D) every step's output (the last one is the flow's output; all of them are the backward's inputs);
 *   saves     : S x NF_GLOW_FLOW_SAVE_FLOATS; ws_zero: S x NF_MLP_WS_FLOATS zero floats per direction;
 *   backward  : g_y is the gradient of ys[S-1], gzs (S, N, D) scratch whose FIRST slice is the gradient of z0 on return;
 *               slabs2 = 2 x NF_MLP_BWD_SLAB_FLOATS; gradients are accumulated (accumulate != 0) or stored into the sinks.   */
#define NF_GLOW_FLOW_MAX_STEPS 1024
#define NF_GLOW_FLOW_SAVE_FLOATS 320
int nf_glow_flow_step_bytes(void);
int nf_glow_flow_pack(void* dst_host, const void* const* head, const void* const* mlp_params, void* const* head_grads,
                      void* const* mlp_grads, int D, int odd);
int nf_glow_flow_vec_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, float* ws_zero,
                         int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
/* inverse of the whole run, last step first (records packed with the pivot matrix in head slot 2, see nf_glow_step_vec_inv);
 * zs2: (2, N, D) scratch whose FIRST slice holds the flow's input on return; ld -= the run's log-det.                        */
int nf_glow_flow_vec_inv(const void* steps_dev, int S, const float* y, float* zs2, float* ld, float* saves, float* ws_zero,
                         int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_glow_flow_vec_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld,
                         float* gzs, const float* saves, int accumulate, float* ws_zero, float* slabs2, int64_t N, int D,
                         int training, float bn_eps, float wn_eps, nf_stream_t stream);
/* The same run of Glow steps as S single-step launches per direction (records as kernel arguments: steps_host = the HOST copy
 * of the packed table) with a DEFERRED fold in the backward: each step's launch ends right after its data gradient, leaving
 * its weight-gradient slabs (slabs_all: S x blocks x NF_MLP_BWD_SLAB_WG_FLOATS, blocks = ceil(N / NF_MLP_ROWS_PER_BLOCK)) and
 * its workgroups' head sums (head_rec: S x blocks x 64) behind; ONE more launch folds every step into the parameter
 * gradients (steps_dev = the device copy of the same table).  For batches where the whole-flow launch loses;
 * buffers and results as nf_glow_flow_vec_*.                                                                               */
#define NF_MLP_BWD_SLAB_WG_FLOATS (7 * 2 * 1056)
int nf_glow_flow_steps_fwd(const void* steps_host, int S, const float* z0, float* ys, float* ld, float* saves, float* ws_zero,
                           int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_glow_flow_steps_bwd(const void* steps_host, const void* steps_dev, int S, const float* z0, const float* ys,
                           const float* g_y, const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero,
                           float* slabs_all, float* head_rec, int64_t N, int D, int training, float bn_eps, float wn_eps,
                           nf_stream_t stream);
/* ... and for a run of RealNVP steps (training mode; records of nf_realnvp_flow_pack, saves = S x NF_REALNVP_SAVE_FLOATS)          */
int nf_realnvp_flow_steps_fwd(const void* steps_host, int S, const float* z0, float* ys, float* ld, float* saves, float* ws_zero,
                              int64_t N, int D, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_realnvp_flow_steps_bwd(const void* steps_host, const void* steps_dev, int S, const float* z0, const float* ys,
                              const float* g_y, const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero,
                              float* slabs_all, float* head_rec, int64_t N, int D, float bn_eps, float wn_eps, nf_stream_t stream);
/* the same for a run of RealNVP steps (nf_realnvp_step_vec_*: training-mode flow BatchNorm + AffineCoupling, flows/realnvp.py);
 * records by nf_realnvp_flow_pack (head = the 8 pointers of nf_realnvp_step_vec_fwd), saves = S x NF_REALNVP_SAVE_FLOATS.     */
int nf_realnvp_flow_pack(void* dst_host, const void* const* head, const void* const* mlp_params, float* g_s_log_scale,
                         float* g_s_bias, void* const* mlp_grads, int D, int odd, float flow_bn_eps, float flow_bn_momentum);
int nf_realnvp_flow_vec_fwd_eval(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves,
                                 float* ws_zero, int64_t N, int D, float bn_eps, float wn_eps, nf_stream_t stream);
int nf_realnvp_flow_vec_inv(const void* steps_dev, int S, const float* y, float* zs2, float* ld, float* saves, float* ws_zero,
                            int64_t N, int D, int training, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_realnvp_flow_vec_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, float* ws_zero,
                            int64_t N, int D, float bn_eps, float bn_momentum, float wn_eps, nf_stream_t stream);
int nf_realnvp_flow_vec_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld,
                            float* gzs, const float* saves, int accumulate, float* ws_zero, float* slabs2, int64_t N, int D,
                            float bn_eps, float wn_eps, nf_stream_t stream);
/* The whole-flow backward with the gradient fold DEFERRED: the launch ends every step after its data gradient and leaves the step's
 * weight-gradient slabs (slabs_all: S x blocks x NF_MLP_BWD_SLAB_WG_FLOATS, blocks = ceil(N / NF_MLP_ROWS_PER_BLOCK)) and head sums
 * (head_rec: S x blocks x 64) behind; one fold launch for all S steps follows it on the same stream.  In-kernel the fold is 8.4 of a
 * step's 34 us at two workgroups (C1: 1.89 -> 1.6 ms per train step).  Same results up to the summation order of the fold.           */
int nf_glow_flow_vec_bwd_deferred(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld,
                                  float* gzs, const float* saves, int accumulate, float* ws_zero, float* slabs_all, float* head_rec,
                                  int64_t N, int D, int training, float bn_eps, float wn_eps, nf_stream_t stream);
int nf_realnvp_flow_vec_bwd_deferred(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y,
                                     const float* g_ld, float* gzs, const float* saves, int accumulate, float* ws_zero, float* slabs_all,
                                     float* head_rec, int64_t N, int D, float bn_eps, float wn_eps, nf_stream_t stream);

/* RealNVP runs with D = 2 and N <= 256 (training mode: nf_realnvp_flow_vec_fwd / nf_realnvp_flow_vec_bwd_deferred) can be served by
 * ONE workgroup that holds the whole batch (csrc/flow_solo.hip): no meeting in global memory.  mode bit 0 = the forward run, bit 1 =
 * the backward run (it reads the activations the one-workgroup forward stashed, so it needs bit 0 as well; the environment variable
 * NF_FLOW_SOLO sets the mode at load; mode < 0 changes nothing); returns 0.  The backward run is a PAIR of workgroups that must share an
 * XCD: it is taken only on devices whose XCC count divides 8 (hipDeviceAttributeNumberOfXccs), the grid kernels serve the others.
 * Buffer sizes of such a run do NOT follow the grid kernels' rule; ask, whichever kernel ends up serving the call:
 *   nf_realnvp_flow_save_floats(N, D)  floats PER STEP of `saves`: the S statistics records (S x NF_REALNVP_SAVE_FLOATS, stride
 *                                      NF_REALNVP_SAVE_FLOATS) are followed -- at the next multiple of 32 floats, so that the regions
 *                                      start on a 128-byte line -- by S x NF_FLOW_SOLO_STASH_FLOATS floats of stashed BatchNorm
 *                                      inputs, S x NF_FLOW_SOLO_GBUF_FLOATS floats of backward scratch (the backward WRITES them) and
 *                                      S x NF_FLOW_SOLO_TAB_FLOATS floats of table images for the shapes the one-workgroup kernels
 *                                      take (`saves` itself on a 128-byte line: the kernels return NF_E_BADARG otherwise);
 *   nf_realnvp_flow_bwd_regions(N, D)  regions per step of slabs_all / head_rec of nf_realnvp_flow_vec_bwd_deferred
 *                                      (NF_FLOW_SOLO_REGIONS for those shapes, ceil(N / NF_MLP_ROWS_PER_BLOCK) otherwise).
 * Both return the count (> 0), not an error code.                                                                                  */
#define NF_FLOW_SOLO_MAX_ROWS 256
#define NF_FLOW_SOLO_REGIONS 4
#define NF_FLOW_SOLO_STASH_FLOATS (5 * 32 * 256)
#define NF_FLOW_SOLO_GBUF_FLOATS (6 * 4 * 512 * 4)   /* per step: what the backward's data-path workgroup hands to its weight-gradient workgroup */
#define NF_FLOW_SOLO_TAB_FLOATS 5888                 /* per step: the forward's parameter / constant tables as an LDS image for the backward */
int nf_flow_solo_config(int mode);
int nf_realnvp_flow_save_floats(int64_t N, int D);
int nf_realnvp_flow_bwd_regions(int64_t N, int D);

/* The persistent kernels above wait on each other with BOUNDED spin loops (a grid of <= NF_MLP_MAX_BLOCKS workgroups is
 * co-resident on an otherwise idle MI355X by construction).  A loop that gives up is counted; a non-zero count means some
 * launch produced garbage (device shared with another job?).  Synchronises the device.                                    */
int nf_persistent_timeouts(int* count);

/* Makes that failure LOUD.  Sets the poll budget of every spin loop (default 2^22; tests lower it to provoke a failure),
 * optionally clears the counters (reset != 0; synchronises), and returns the address of ONE pinned host word that a loop
 * which gives up sets to 1 with system scope: the host can look at it after every launch without synchronising
 * (_native.call and FlowTrainer.train_on_batch raise when it is non-zero).                                                */
int nf_persistent_config(int64_t spin_limit, int reset, void** host_error_word);

/* Deterministic mode (csrc/nf_det.h; main.py:308-311 has the reference's switch).  on != 0: every batch sum that meets at one address
 * through float atomics -- per-sample log-dets of slab kernels, per-channel statistics and parameter gradients of the layerwise kernels,
 * the folds of the persistent kernels -- is added in a FIXED order (workgroups in linear block order through a turnstile, waves in wave
 * order inside one), so that two runs of a launch from identical inputs are bit-identical.  A verification mode: the ordered tails
 * serialise (~1 us per workgroup).  Synchronises the device.  The Python binding calls it with 1 when NF_DETERMINISTIC=1.
 * nf_deterministic_timeouts: turnstile waits that gave up (must be 0; a non-zero count means a launch finished unordered).          */
int nf_deterministic(int on);
int nf_deterministic_enabled(void);
int nf_deterministic_timeouts(int* count);

/* Launch-time residency check: how many workgroups of the largest persistent kernel of the MLP-chain family / the MAF-step
 * family the current device holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units).  The host side
 * keeps grids within it (min with NF_MLP_MAX_BLOCKS / NF_MAF_MAX_BLOCKS), else it takes the multi-launch path.           */
int nf_persistent_capacity(int* mlp_blocks, int* maf_blocks);

/* ---- Flow++ conditioner for density data, whole network in one launch  coupling.py:142-149, modules.py:500-578 ---------
 * out = Linear5(LN2(GatedAttn1(LN1(GatedLinear(Linear0(x))))))  for x (N, I0 <= 4), hidden width 32, O <= 64 outputs;
 * GatedAttn with ONE position: q = Wq (h + pos) + bq (rows 64..95 of conv1), v = W2 q + b2, h + v[:32]*sigmoid(v[32:]).
 * x may be a strided view: element (n, i) at x[n * x_row_stride + i * x_col_stride] -- e.g. the conditioning half of the
 * coupling input itself (squeeze.py:68-69), which removes the gather launch.                                             */
int nf_flowpp_cond_fwd(const float* x, const float* W0, const float* b0, const float* Wg, const float* bg,
                       const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                       const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                       const float* b5, float* out, int64_t x_row_stride, int x_col_stride, int64_t N, int I0, int O,
                       nf_stream_t stream);
/* autograd of nf_flowpp_cond_fwd: the forward is recomputed per 16-row tile from x, every parameter gradient is
 * ACCUMULATED (+=; zero-filled temporaries or .grad buffers); g_x element (n, i) at g_x[n * gx_row_stride + i * gx_col_stride]
 * is written, or += when gx_accumulate (straight into the coupling's input gradient: no scatter, no add launch); nullable.
 * Two launches: the tile
 * kernel leaves one partial-sum slab per block in `workspace` (>= NF_FLOWPP_BWD_WS_FLOATS floats, contents irrelevant,
 * re-usable by the next call on the same stream), a small kernel folds the slabs into the destinations.
 * g_Wq / g_bq address rows 64..95 of conv1's gradient (the V/K rows receive exact zeros, like the reference).          */
#define NF_FLOWPP_BWD_WS_FLOATS (256 * 7744)
int nf_flowpp_cond_bwd(const float* x, const float* W0, const float* b0, const float* Wg, const float* bg,
                       const float* ln1_g, const float* ln1_b, const float* pos, const float* Wq, const float* bq,
                       const float* W2, const float* b2, const float* ln2_g, const float* ln2_b, const float* W5,
                       const float* b5, const float* g_out, float* g_x, float* g_W0, float* g_b0, float* g_Wg,
                       float* g_bg, float* g_ln1_g, float* g_ln1_b, float* g_pos, float* g_Wq, float* g_bq, float* g_W2,
                       float* g_b2, float* g_ln2_g, float* g_ln2_b, float* g_W5, float* g_b5, float* workspace,
                       int64_t x_row_stride, int x_col_stride, int64_t gx_row_stride, int gx_col_stride, int gx_accumulate,
                       int64_t N, int I0, int O, nf_stream_t stream);

/* ---- Flow++ conditioner for IMAGE data (csrc/flowpp_img.hip)  coupling.py:159-166, modules.py:519-578 ------------------------
 * net = Conv2d(I0, 32, 3) -> GatedConv2d(32) -> LayerNorm(32,H,W) -> GatedAttn(32, 4 heads) -> LayerNorm -> Conv2d(32, O, 3) on
 * square maps of side 1 .. 16 (nf_flowpp_img_usable != 0; the mid shapes of flows/flowpp.py:22-57 on images up to 32 x 32).
 * Per-sample work throughout (no batch statistics): four launches forward, nine backward, all NCHW fp32.
 * STORAGE: (H, W) in every call below is the IMAGE; the activation / gradient tensors of these calls are (B, C, S, S) with
 * S = nf_flowpp_img_storage(H, W) = the next power of two >= H, at least 4 (S = H for the CIFAR / MNIST pyramids: 16, 8, 4), the image
 * in the top-left corner.  Pixels outside the image are DEAD: the convolutions read zero there (as outside the map), LayerNorm and
 * the attention run over the H * W live positions only, what the kernels leave there is meaningless (the caller pads its input
 * with anything and crops the output).  The (32, H, W) parameters (LayerNorm affines, position embedding) and their gradients keep
 * the reference's layout.  nf_flowpp_img_celu_bwd is elementwise: pass it the storage extent.
 *
 * nf_flowpp_img_conv: out (B, Co, H, W) = conv3x3(in, pad 1) + bias (bias nullable), exact-fp32 matrix-core GEMM.
 *   in_mode 0: in is (B, Ci, H, W);   in_mode 1: in is (B, Ci / 2, H, W) and the convolution sees concat_elu(in) = elu([in, -in])
 *   (GatedConv2d's inner convolution, modules.py:500-517);   transposed 0: weight (Co, Ci, 3, 3);   transposed 1: weight (Ci, Co, 3, 3)
 *   read transposed with flipped taps -- the DATA GRADIENT of the convolution that owns `weight`, `in` being the gradient of its
 *   output (in_mode 0, bias NULL).
 *   ksplit >= 1 cuts the input-channel axis into that many ranges of 32-channel chunks, one workgroup row each: `out` then holds
 *   ksplit partial-sum slabs (ksplit, B, Co, H, W) whose sum is the result (bias in slab 0); nf_flowpp_img_conv_ksplit proposes
 *   a count that fills the chip (the 32 -> O convolution's data gradient at 4 x 4 has K = 9 * 1344 and four pixel tiles).
 * nf_flowpp_img_conv_wgrad: partial sums of the weight gradient sum_b g_out (x) in  and of the bias gradient sum g_out (slab_b nullable)
 *   in n_slabs slabs, TAP-MAJOR (n_slabs, 9, Co, Ci) and (n_slabs, Co), every element WRITTEN; nf_slab_sum (stride Co*Ci*9, taps 9 /
 *   stride Co, taps 1) folds
 *   them into the destinations; nf_flowpp_img_wgrad_slabs proposes the count (<= NF_FLOWPP_IMG_MAX_SLABS); in / in_mode as above.
 * nf_flowpp_img_celu_bwd: g_x += elu'(x) * g_cat[:, :C] - elu'(-x) * g_cat[:, C:]  (x (B, C, H, W), g_cat (B, 2 C, H, W)).
 * nf_flowpp_img_mid_fwd: x = conv0 output, a = the gated convolution's output (both (B, 32, H, W)) ->
 *   LN2( A(LN1(x + elu(a) * sigmoid(elu(-a)))) ),  A(t) = t + y * sigmoid(gate), [y, gate] = conv2(attention(conv1(t + pos))).
 * nf_flowpp_img_mid_bwd: its autograd, recomputing the forward from x and a; g_x / g_a written, parameter gradients ACCUMULATED (with
 *   per_sample != 0 the (32, H, W) ones -- LayerNorm affines, position embedding -- are instead WRITTEN per sample, (B, 32, H, W));
 *   g_out may be g_out_slabs partial-sum slabs (g_out_slabs, B, 32, H, W) of a K-split nf_flowpp_img_conv, summed on load.          */
#define NF_FLOWPP_IMG_MAX_KSPLIT 64
#define NF_FLOWPP_IMG_MAX_SLABS 64
int nf_flowpp_img_usable(int64_t B, int Ci, int Co, int H, int W);
int nf_flowpp_img_storage(int H, int W);                     /* side S of the storage map, 0 if the map is not served */
int nf_flowpp_img_conv_ksplit(int64_t B, int Ci, int Co, int H, int W);
int nf_flowpp_img_conv(const float* in, const float* weight, const float* bias, float* out, int64_t B, int Ci, int Co, int H, int W,
                       int in_mode, int transposed, int ksplit, nf_stream_t stream);
int nf_flowpp_img_wgrad_slabs(int64_t B, int Ci, int Co, int H, int W);
int nf_flowpp_img_conv_wgrad(const float* in, const float* g_out, float* slab_w, float* slab_b, int n_slabs, int64_t B, int Ci, int Co,
                             int H, int W, int in_mode, nf_stream_t stream);
/* ... of n <= NF_FLOWPP_IMG_WGRAD_MAX convolutions of ONE shape in one launch (each with its own slabs; slab_b nullable) */
#define NF_FLOWPP_IMG_WGRAD_MAX 16
typedef struct nf_flowpp_img_wgrad_desc {
    const float *in, *g_out;
    float *slab_w, *slab_b;
} nf_flowpp_img_wgrad_desc;
int nf_flowpp_img_conv_wgrad_multi(const nf_flowpp_img_wgrad_desc* descs, int n, int n_slabs, int64_t B, int Ci, int Co, int H, int W,
                                   int in_mode, nf_stream_t stream);
int nf_flowpp_img_celu_bwd(const float* x, const float* g_cat, float* g_x, int64_t B, int C, int H, int W, nf_stream_t stream);
int nf_flowpp_img_mid_fwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                          const float* conv1_w, const float* conv1_b, const float* conv2_w, const float* conv2_b,
                          const float* ln2_g, const float* ln2_b, float* out, int64_t B, int H, int W, nf_stream_t stream);
int nf_flowpp_img_mid_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                          const float* conv1_w, const float* conv1_b, const float* conv2_w, const float* conv2_b,
                          const float* ln2_g, const float* ln2_b, const float* g_out, float* g_x, float* g_a, float* g_ln1_g,
                          float* g_ln1_b, float* g_pos, float* g_conv1_w, float* g_conv1_b, float* g_conv2_w, float* g_conv2_b,
                          float* g_ln2_g, float* g_ln2_b, int per_sample, int64_t B, int H, int W, int g_out_slabs, nf_stream_t stream);

/* The same middle cut BY ATTENTION HEAD (csrc/flowpp_img_att.hip), for batches that leave most of the chip idle with one workgroup per
 * sample: H = W = 16 (nf_flowpp_img_att_usable != 0).  Forward = att_fwd (B x 4 workgroups: gate, LayerNorm 1, a head's rows of
 * conv1, the softmax sweeps -> mixed (B, 32, H, W) and cj (B, 4, H*W), cj = max + log sum of a column's scores) + post_fwd (conv2, gate,
 * LayerNorm 2).  Backward = post_bwd (LayerNorm 2 / conv2 backward -> g3, g_mixed (B, 32, H, W)) + att_bwd (B x 4: softmax backward, the
 * head's conv1 rows, gt_part (B, 4, 32, H*W) = its part of the tokens' gradient) + pre_bwd (sum of the parts, position embedding,
 * LayerNorm 1, gate -> g_x, g_a).  Parameter gradients ACCUMULATED; with per_sample != 0 the (32, H, W) ones (LayerNorm affines, position
 * embedding) are instead WRITTEN per sample into (B, 32, H, W) buffers that nf_slab_sum folds (n_slabs = B).  Every kernel recomputes
 * what it needs from x and a.                                                                                                       */
int nf_flowpp_img_att_usable(int64_t B, int H, int W);
int nf_flowpp_img_att_fwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                          const float* conv1_w, const float* conv1_b, float* mixed, float* cj, int64_t B, int H, int W,
                          nf_stream_t stream);
int nf_flowpp_img_post_fwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* mixed,
                           const float* conv2_w, const float* conv2_b, const float* ln2_g, const float* ln2_b, float* out, int64_t B,
                           int H, int W, nf_stream_t stream);
int nf_flowpp_img_post_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* mixed,
                           const float* conv2_w, const float* conv2_b, const float* ln2_g, const float* ln2_b, const float* g_out,
                           int g_out_slabs, float* g3, float* g_mixed, float* g_conv2_w, float* g_conv2_b, float* g_ln2_g,
                           float* g_ln2_b, int per_sample, int64_t B, int H, int W, nf_stream_t stream);
int nf_flowpp_img_att_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* pos,
                          const float* conv1_w, const float* conv1_b, const float* mixed, const float* cj, const float* g_mixed,
                          float* gt_part, float* g_conv1_w, float* g_conv1_b, int64_t B, int H, int W, nf_stream_t stream);
int nf_flowpp_img_pre_bwd(const float* x, const float* a, const float* ln1_g, const float* ln1_b, const float* g3,
                          const float* gt_part, float* g_x, float* g_a, float* g_ln1_g, float* g_ln1_b, float* g_pos, int per_sample,
                          int64_t B, int H, int W, nf_stream_t stream);

/* ---- on-device synthetic batches (csrc/datagen.hip)  flows/dataset.py:13-34, :120; replaces the per-step H2D copy main.py:79 --
 * kind 0 moons, 1 circles, 2 normals: out (n, 2), per_sample = 2;  3 cifar-like uniform uint8 / 255: out (n, per_sample).
 * Counter-based Philox4x32-10 keyed by (seed, *step, sample): stateless and reproducible; `step` (device int64, NULL = 0) is read
 * by the kernel, so a captured graph draws a fresh batch per replay once nf_sample_advance (step += 1) follows it.             */
int nf_sample_data(int kind, float* out, int64_t n, int per_sample, int64_t seed, const int64_t* step, nf_stream_t stream);
int nf_sample_advance(int64_t* step, nf_stream_t stream);

/* ---- NLL of the training harness  main.py:49-51, :85 -------------------------------------------------------------
 * loss[0] += -(1/B) * sum_b ( -0.5*|z_b|^2 - 0.5*D*log(2 pi) + ld[b] )   (caller zero-fills loss)            */
int nf_nll_loss(const float* z, const float* ld, float* loss, int64_t B, int64_t D, nf_stream_t stream);
/* autograd: g_z = g_loss[0] * z / B,  g_ld[b] = -g_loss[0] / B   (g_loss: device scalar) */
int nf_nll_loss_bwd(const float* z, const float* g_loss, float* g_z, float* g_ld, int64_t B, int64_t D,
                    nf_stream_t stream);

/* ---- fused Adam over flat buffers (torch.optim.Adam semantics, main.py:56-64) --------------------------------------
 * step[0] += 1 (device int32), then for every i < n:  g = grad[i]*grad_scale + wd*p ; m,v moments ; p update with the
 * bias corrections of step[0] and the learning rate lr[0] (device float, so hipGraph replays see new values).     */
int nf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int* step, const float* lr,
                 float beta1, float beta2, float eps, float weight_decay, float grad_scale, int64_t n,
                 nf_stream_t stream);

/* dst_k[0:n_k] = src_k[0:n_k] for up to NF_COPY_MAX tensors in one launch (gradient gather into the flat bucket)   */
#define NF_COPY_MAX 128
typedef struct nf_copy_desc {
    const float* src;
    float* dst;
    int64_t n;
} nf_copy_desc;
int nf_multi_copy(const nf_copy_desc* descs, int n_tensors, nf_stream_t stream);
/* dst[0 .. n) = 0 (the train step's memsets: the flat gradient bucket of main.py:84's optimizer.zero_grad(), the scratch arena)          */
int nf_zero_fill(float* dst, int64_t n, nf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFHIP_H */
