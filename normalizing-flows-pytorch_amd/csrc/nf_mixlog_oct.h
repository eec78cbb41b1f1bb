// One mixture component per lane ("octet" = eight consecutive lanes share an element): helpers shared by the stand-alone
// mixture-of-logistics kernels (mixlog.hip) and the Flow++ coupling step fused into the conditioner's backward (flowpp_cond.hip).
// The log-sum-exp / softmax reductions over the components are three DPP steps each (quad_perm xor 1, xor 2,
// row_half_mirror) -- VALU speed, no LDS.
#pragma once
#include "nf_common.h"

template <int CTRL>
__device__ __forceinline__ float nf_dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float nf_oct_sum(float v) {
    v += nf_dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += nf_dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    v += nf_dpp_mov<0x141>(v);         // row_half_mirror: lane i <-> 7 - i of each group of eight
    return v;
}
__device__ __forceinline__ float nf_oct_max(float v) {
    v = fmaxf(v, nf_dpp_mov<0xB1>(v));
    v = fmaxf(v, nf_dpp_mov<0x4E>(v));
    v = fmaxf(v, nf_dpp_mov<0x141>(v));
    return v;
}
// hardware transcendentals for the octet kernels (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp): the libm forms are 30-60
// instructions each, and with eight lanes per element the per-element scalar math is issued eight times as often
#ifndef NF_FEXP_DEFINED
#define NF_FEXP_DEFINED
__device__ __forceinline__ float nf_fexp(float x) { return __expf(x); }
#endif
__device__ __forceinline__ float nf_flog(float x) { return __logf(x); }
__device__ __forceinline__ float nf_ftanh(float x) {              // 1 - 2 / (1 + e^{2x}), saturates cleanly
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}
struct NfOct { float lp, mu, s, es, a_raw, b; };     // this lane's component + the element's affine parameters

__device__ __forceinline__ void nf_oct_load(const float* __restrict__ P, int64_t nh, int K, int kk, NfOct& m) {
    const bool on = kk < K;
    const int k = on ? kk : 0;
    const float lp = P[(2 + k) * nh], mu = P[(2 + K + k) * nh], sv = P[(2 + 2 * K + k) * nh];
    m.a_raw = P[0];
    m.b = P[nh];
    m.lp = on ? lp : -INFINITY;
    m.mu = on ? mu : 0.f;
    m.s = on ? sv : 0.f;
    const float mx = nf_oct_max(m.lp);
    const float lse = mx + nf_flog(nf_oct_sum(nf_fexp(m.lp - mx)));         // F.log_softmax over the mixture axis (coupling.py:180)
    m.lp -= lse;
    m.es = nf_fexp(-m.s);
}
// log CDF and log PDF of the mixture at x (modules.py:64-97), identical on the eight lanes; u, l of this lane's component
__device__ __forceinline__ void nf_oct_eval(const NfOct& m, float x, float& lcdf, float& lpdf, float& u, float& l) {
    u = (x - m.mu) * m.es;
    l = nf_flog(1.f + nf_fexp(-fabsf(u)));
    const float c = m.lp + (fminf(u, 0.f) - l);
    const float d = m.lp + (u - m.s - 2.f * (fmaxf(u, 0.f) + l));
    const float cm = nf_oct_max(c), dm = nf_oct_max(d);
    lcdf = cm + nf_flog(nf_oct_sum(nf_fexp(c - cm)));
    lpdf = dm + nf_flog(nf_oct_sum(nf_fexp(d - dm)));
}

