// Records shared by the persistent vector-flow kernels (mlp_chain.hip: 16-row tiles, grids of workgroups; flow_solo.hip: the whole
// batch in one workgroup): parameter / gradient-sink pointer tables of one flow step and the weight-gradient slab geometry.
#pragma once
#include "nf_common.h"

#define NF_MC_WAVES (NF_MLP_ROWS_PER_BLOCK / 16)         // 8 or 16 (multiple of 4: the weight-gradient jobs come in fours)
#define NF_MC_NKQ (NF_MC_WAVES / 4)                       // row groups of 64 per workgroup
#define NF_MC_THREADS (NF_MC_WAVES * NF_WAVE)
#define NF_MC_NL NF_MLP_LINEARS
#define NF_MC_NB NF_MLP_BNS

// Parameter pointers carry the GLOBAL address space on the device side: a whole-flow kernel reads them from a record in
// memory, and a pointer loaded from memory is otherwise generic -> flat loads, whose completion counts on BOTH the vector-
// memory and the LDS counters (43 flat instructions in k_glow_flow_fwd before this; none now).
#if defined(__HIP_DEVICE_COMPILE__)
#define NF_G __attribute__((address_space(1)))
#else
#define NF_G
#endif
#define NF_GSET(field, value) field = (decltype(field))(value)
struct NfMlpP {
    const NF_G float* v[NF_MC_NL]; const NF_G float* g[NF_MC_NL]; const NF_G float* b[NF_MC_NL];
    const NF_G float* gamma[NF_MC_NB]; const NF_G float* beta[NF_MC_NB];
    NF_G float* rmean[NF_MC_NB]; NF_G float* rvar[NF_MC_NB]; NF_G int64_t* nbt[NF_MC_NB];
};

// ---- the fused vector Glow step (ActNorm -> invertible 1x1 -> affine coupling around this MLP), dims = (D,), D = 2 or 4 ----
struct NfGlowV {
    const float* z; float* y; float* ld;                                     // forward: input, output, log-det (in place +=)
    const float* g_y; const float* g_ld; float* g_z;                         // backward
    const NF_G float *ls, *bs, *P, *L, *U, *Lm, *Um, *sign_s, *log_s, *a, *c;     // ActNorm, PLU factors, coupling scale / shift
    NF_G float *g_ls, *g_bs, *g_L, *g_U, *g_log_s, *g_a, *g_c;
    NF_G float *bmean, *bvar, *rmean, *rvar;                                 // flow-BatchNorm head (HEAD == 2): ls = log_gamma, bs = beta
    float fbn_eps, fbn_mom;
    int fbn_mode;                                                            // 0: batch statistics computed here; 1: running statistics; 2: the stored batch buffers
    int D, odd;
};
struct NfMlpG { NF_G float* v[NF_MC_NL]; NF_G float* g[NF_MC_NL]; NF_G float* b[NF_MC_NL]; NF_G float* gamma[NF_MC_NB]; NF_G float* beta[NF_MC_NB]; };
struct NfGlowFlowStep { NfMlpP p; NfMlpG g; NfGlowV h; };     // the static pointers of one step: parameters, gradient sinks

#define NF_MC_SLAB_Q 1056                                // 32 x 32 weight-gradient partial + 32 bias partial
#define NF_MC_SLAB_L (NF_MC_NKQ * NF_MC_SLAB_Q)          // one partial per 64-row group
#define NF_MC_NLS (NF_MC_NL + 1)                         // + one product for the fused Glow step's scalar gradients
#define NF_MC_SLAB (NF_MC_NLS * NF_MC_SLAB_L)

// the whole-batch kernels of flow_solo.hip (RealNVP steps, D = 2, N <= 256, training mode); 0 from the plan = not taken
int nf_solo_plan(int64_t N, int D, int backward);
int nf_solo_fwd(const void* steps_dev, int S, const float* z0, float* ys, float* ld, float* saves, int save_stride, int64_t N, float bn_eps,
                float bn_momentum, float wn_eps, hipStream_t stream);
int nf_solo_bwd_steps_ok(int S);
int nf_solo_bwd(const void* steps_dev, int S, const float* z0, const float* ys, const float* g_y, const float* g_ld, float* gzs,
                const float* saves, int save_stride, int accumulate, float* ws_zero, float* slabs_all, float* head_rec, int64_t N,
                float wn_eps, hipStream_t stream);
