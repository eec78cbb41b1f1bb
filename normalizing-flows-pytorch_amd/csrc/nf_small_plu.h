// The invertible 1 x 1 convolution's weight from its PLU factors for C <= 4 (flows/modules.py:470-476), shared by the stand-alone head
// kernels (glow_head.hip) and the head in the prologue of the image conditioner's chain launch (conv_chain.hip): the same FMAs in the
// same order in both, so the two produce the same bits.
#pragma once

// W = P L' U' for C <= 4, computed redundantly by every thread that needs it (a few dozen FMAs)
template <int CT>
__device__ __forceinline__ void nf_small_plu(const float* __restrict__ Pm, const float* __restrict__ L,
                                             const float* __restrict__ U, const float* __restrict__ Lm,
                                             const float* __restrict__ Um, const float* __restrict__ sign_s,
                                             const float* __restrict__ log_s, float (&Wm)[CT][CT]) {
    float Lp[CT][CT], Up[CT][CT], T[CT][CT];
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            Lp[r][c] = L[r * CT + c] * Lm[r * CT + c] + (r == c ? 1.f : 0.f);
            Up[r][c] = U[r * CT + c] * Um[r * CT + c] + (r == c ? sign_s[r] * expf(log_s[r]) : 0.f);
        }
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < CT; ++k) a = fmaf(Lp[r][k], Up[k][c], a);
            T[r][c] = a;
        }
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < CT; ++k) a = fmaf(Pm[r * CT + k], T[k][c], a);
            Wm[r][c] = a;
        }
}

