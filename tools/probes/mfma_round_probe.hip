// How does v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 ROUND?  K = 32 dot products on the matrix pipe against the exact (double) result,
// next to a sequential fmaf chain and a mul + add chain in fp32 on the vector ALU.  Reported per variant: mean SIGNED error and RMS error in
// units of the result's ulp.  A pipe that truncates (or rounds its partial sums toward zero) shows as a negative mean on all-positive data.
//   hipcc --offload-arch=gfx950 -O3 -o normalizing-flows-pytorch_amd/build/mfma_round_probe tools/probes/mfma_round_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: (16, 32) row-major, B: (32, 16) row-major -> D (16, 16)
__global__ void k16(const float* A, const float* B, float* D, float* Dv, float* Dm) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < 32; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 32 + k0 + g], B[(k0 + g) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
    // vector ALU: lane -> 4 entries
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float f = 0.f, m = 0.f;
        for (int k = 0; k < 32; ++k) {
            f = __builtin_fmaf(A[row * 32 + k], B[k * 16 + i], f);
            m = __fadd_rn(m, __fmul_rn(A[row * 32 + k], B[k * 16 + i]));
        }
        Dv[row * 16 + i] = f;
        Dm[row * 16 + i] = m;
    }
}
// 32x32x2: A (32, 32) row-major, B (32, 32) row-major -> D (32, 32).  A operand: lane (i = l & 31, k = l >> 5); D: row 8 b + 4 ... standard layout
__global__ void k32(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < 32; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 32 + k0 + h], B[(k0 + h) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i] = acc[r];
}
static double ulp_of(double x) { int e; std::frexp(std::fabs(x) > 1e-300 ? x : 1e-300, &e); return std::ldexp(1.0, e - 24); }
struct Stat { double s = 0, s2 = 0; long n = 0; void add(double e) { s += e; s2 += e * e; ++n; } };
int main() {
    float *dA, *dB, *dD, *dV, *dM, *dD32;
    hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 4096); hipMalloc(&dV, 4096); hipMalloc(&dM, 4096); hipMalloc(&dD32, 4096);
    for (int mode = 0; mode < 3; ++mode) {
        Stat s16, sf, sm, s32, sh;
        srand(1 + mode);
        for (int rep = 0; rep < 400; ++rep) {
            std::vector<float> A(1024), B(1024), D(256), V(256), M(256), D32(1024);
            for (auto& v : A) { float u = (rand() + 0.5f) / (RAND_MAX + 1.0f); v = mode == 0 ? 2.f * u - 1.f : (mode == 1 ? 0.5f + u : (u < 0.5f ? -1.f : 1.f) * std::exp(8.f * u - 4.f)); }
            for (auto& v : B) { float u = (rand() + 0.5f) / (RAND_MAX + 1.0f); v = mode == 0 ? 2.f * u - 1.f : (mode == 1 ? 0.5f + u : 2.f * u - 1.f); }
            hipMemcpy(dA, A.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 4096, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, dV, dM);
            hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD32);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(V.data(), dV, 1024, hipMemcpyDeviceToHost);
            hipMemcpy(M.data(), dM, 1024, hipMemcpyDeviceToHost); hipMemcpy(D32.data(), dD32, 4096, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ex = 0; float hf = 0.f;
                for (int k = 0; k < 32; ++k) { ex += (double)A[i * 32 + k] * (double)B[k * 16 + j]; hf = std::fmaf(A[i * 32 + k], B[k * 16 + j], hf); }
                // scale: ulp of the largest partial magnitude (sum of |products|) -- the error unit of a length-32 accumulation
                double mag = 0; for (int k = 0; k < 32; ++k) mag += std::fabs((double)A[i * 32 + k] * (double)B[k * 16 + j]);
                const double u = ulp_of(mag);
                s16.add((D[i * 16 + j] - ex) / u); sf.add((V[i * 16 + j] - ex) / u); sm.add((M[i * 16 + j] - ex) / u); sh.add((hf - ex) / u);
            }
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double ex = 0, mag = 0;
                for (int k = 0; k < 32; ++k) { ex += (double)A[i * 32 + k] * (double)B[k * 32 + j]; mag += std::fabs((double)A[i * 32 + k] * (double)B[k * 32 + j]); }
                s32.add((D32[i * 32 + j] - ex) / ulp_of(mag));
            }
        }
        auto pr = [&](const char* n, Stat& s) { printf("  %-34s mean %+.4f ulp   rms %.4f ulp   (n %ld)\n", n, s.s / s.n, std::sqrt(s.s2 / s.n), s.n); };
        printf("%s  (error / ulp of sum |a_k b_k|, K = 32)\n", mode == 0 ? "a, b uniform in (-1, 1)" : mode == 1 ? "a, b uniform in (0.5, 1.5): all products positive" : "a log-uniform magnitude with sign, b uniform in (-1, 1)");
        pr("v_mfma_f32_16x16x4_f32 (8 steps)", s16); pr("v_mfma_f32_32x32x2_f32 (16 steps)", s32); pr("v_fma_f32 chain (device)", sf);
        pr("v_mul + v_add chain (device)", sm); pr("fmaf chain (host)", sh);
    }
    return 0;
}
