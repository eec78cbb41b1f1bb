// Library probes and the NLL reduction of the training harness (main.py:49-51, :85).
#include <string.h>

#include "nf_common.h"
#include "nf_det.h"

NF_DET_STATE(nf_ms)
NF_DET_HOST_API(nf_ms)

extern "C" int nf_version(void) { return 100; }

extern "C" int nf_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.gcnArchName, (size_t)arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return 0;
}

// loss[0] += -(1/B) * sum_b ( -0.5 |z_b|^2 - 0.5 D log(2 pi) + ld[b] )
__global__ void __launch_bounds__(NF_BLOCK) k_nll_loss(const float* __restrict__ z, const float* __restrict__ ld,
                                                       float* __restrict__ loss, int64_t B, int64_t D) {
    __shared__ float scratch[NF_BLOCK / NF_WAVE];
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    const int64_t n = B * D;
    // 16 bytes per load where the tensors allow it (one 4-byte load per thread and trip left this pass at 36 % of the HBM rate)
    if ((((uintptr_t)z | (uintptr_t)ld) & 15) == 0) {
        const int64_t n4 = n >> 2, b4 = B >> 2;
        const float4* z4 = reinterpret_cast<const float4*>(z);
        const float4* l4 = reinterpret_cast<const float4*>(ld);
        for (int64_t t = gtid; t < n4; t += gstride) {
            const float4 v = z4[t];
            acc -= 0.5f * ((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        }
        for (int64_t t = 4 * n4 + gtid; t < n; t += gstride) acc = fmaf(-0.5f * z[t], z[t], acc);
        for (int64_t b = gtid; b < b4; b += gstride) {
            const float4 v = l4[b];
            acc += (v.x + v.y) + (v.z + v.w);
        }
        for (int64_t b = 4 * b4 + gtid; b < B; b += gstride) acc += ld[b];
    } else {
        for (int64_t t = gtid; t < n; t += gstride) {
            const float v = z[t];
            acc = fmaf(-0.5f * v, v, acc);
        }
        for (int64_t b = gtid; b < B; b += gstride) acc += ld[b];
    }
    const float cst = -0.5f * (float)D * 1.8378770664093453f;   // log(2 pi)
    if (gtid == 0) acc += cst * (float)B;
    const float tot = nf_block_sum(acc, scratch);
    if (threadIdx.x == 0) NF_DET_ADD1(nf_ms, loss, -tot / (float)B);
}

// dst[0 .. n) = 0: the step's two memsets (gradient bucket, scratch arena).  16-byte stores from 2 048 workgroups: 33.5 MB (the CIFAR Glow's
// bucket) in ~10 us; the framework's fill kernel took 48 us for it (profiles/r06_c4_step_kernels.txt: three fills, 0.145 ms of a 23 ms step).
__global__ void __launch_bounds__(NF_BLOCK) k_zero_fill(float* __restrict__ dst, int64_t n, int64_t head, int64_t n4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (t < head) dst[t] = 0.f;                                    // (unaligned head, < 4 elements)
    float4* d4 = reinterpret_cast<float4*>(dst + head);
    for (int64_t i = t; i < n4; i += stride) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t tail0 = head + 4 * n4;
    if (t < n - tail0) dst[tail0 + t] = 0.f;
}
extern "C" int nf_zero_fill(float* dst, int64_t n, nf_stream_t stream) {
    if (n < 0 || (dst == nullptr && n > 0)) return NF_E_BADARG;
    if (n == 0) return 0;
    const int64_t mis = (int64_t)((reinterpret_cast<uintptr_t>(dst) & 15) / 4);
    int64_t head = mis ? 4 - mis : 0;
    if (head > n) head = n;
    const int64_t n4 = (n - head) / 4;
    int64_t g = (n4 + NF_BLOCK - 1) / NF_BLOCK;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)g), dim3(NF_BLOCK), 0, (hipStream_t)stream, dst, n, head, n4);
    NF_CHECK_LAUNCH();
    return 0;
}

extern "C" int nf_nll_loss(const float* z, const float* ld, float* loss, int64_t B, int64_t D, nf_stream_t stream) {
    if (B <= 0 || D <= 0) return NF_E_BADARG;
    unsigned g = nf_grid_for(B * D, NF_BLOCK * 8);
    if (g > 2048) g = 2048;                         // (one same-address atomic per workgroup: 24 k of them were the whole launch at 2^24 rows)
    hipLaunchKernelGGL(k_nll_loss, dim3(g), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, ld, loss, B, D);
    NF_CHECK_LAUNCH();
    return 0;
}

// g_z = g * z / B ; g_ld = -g / B      (g = upstream gradient of the scalar loss, read from device memory)
__global__ void __launch_bounds__(NF_BLOCK) k_nll_loss_bwd(const float* __restrict__ z, const float* __restrict__ g,
                                                           float* __restrict__ gz, float* __restrict__ gld, int64_t B,
                                                           int64_t D) {
    const float s = g[0] / (float)B;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = gtid; t < B * D; t += gstride) gz[t] = s * z[t];
    for (int64_t b = gtid; b < B; b += gstride) gld[b] = -s;
}

extern "C" int nf_nll_loss_bwd(const float* z, const float* g_loss, float* g_z, float* g_ld, int64_t B, int64_t D,
                               nf_stream_t stream) {
    if (B <= 0 || D <= 0) return NF_E_BADARG;
    hipLaunchKernelGGL(k_nll_loss_bwd, dim3(nf_grid_for(B * D)), dim3(NF_BLOCK), 0, (hipStream_t)stream, z, g_loss, g_z,
                       g_ld, B, D);
    NF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// deterministic mode (nf_det.h): the switch reaches every translation unit that orders batch sums by float atomics
__attribute__((visibility("hidden"))) int nf_ca_det_set(int on);
__attribute__((visibility("hidden"))) int nf_ca_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_cpl_det_set(int on);
__attribute__((visibility("hidden"))) int nf_cpl_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_lg_det_set(int on);
__attribute__((visibility("hidden"))) int nf_lg_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_fbh_det_set(int on);
__attribute__((visibility("hidden"))) int nf_fbh_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_ic_det_set(int on);
__attribute__((visibility("hidden"))) int nf_ic_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_icm_det_set(int on);
__attribute__((visibility("hidden"))) int nf_icm_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_gh_det_set(int on);
__attribute__((visibility("hidden"))) int nf_gh_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_ghm_det_set(int on);
__attribute__((visibility("hidden"))) int nf_ghm_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_ml_det_set(int on);
__attribute__((visibility("hidden"))) int nf_ml_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_lb_det_set(int on);
__attribute__((visibility("hidden"))) int nf_lb_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_cvb_det_set(int on);
__attribute__((visibility("hidden"))) int nf_cvb_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_cbk_det_set(int on);
__attribute__((visibility("hidden"))) int nf_cbk_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_ccd_det_set(int on);
__attribute__((visibility("hidden"))) int nf_ccd_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_fpc_det_set(int on);
__attribute__((visibility("hidden"))) int nf_fpc_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_fpi_det_set(int on);
__attribute__((visibility("hidden"))) int nf_fpi_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_fpa_det_set(int on);
__attribute__((visibility("hidden"))) int nf_fpa_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_mdc_det_set(int on);
__attribute__((visibility("hidden"))) int nf_mdc_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_mcd_det_set(int on);
__attribute__((visibility("hidden"))) int nf_mcd_det_timeouts(unsigned* out);
__attribute__((visibility("hidden"))) int nf_rsm_det_set(int on);
__attribute__((visibility("hidden"))) int nf_rsm_det_timeouts(unsigned* out);

static int g_nf_det_on = 0;

extern "C" int nf_deterministic(int on) {
    // (synchronises: the mode words live in device globals, and a launch in flight must not see the turn reset under it)
    hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) return (int)se;
    int e = nf_ms_det_set(on);
    if (e == 0) e = nf_ca_det_set(on);
    if (e == 0) e = nf_cpl_det_set(on);
    if (e == 0) e = nf_lg_det_set(on);
    if (e == 0) e = nf_fbh_det_set(on);
    if (e == 0) e = nf_ic_det_set(on);
    if (e == 0) e = nf_icm_det_set(on);
    if (e == 0) e = nf_gh_det_set(on);
    if (e == 0) e = nf_ghm_det_set(on);
    if (e == 0) e = nf_ml_det_set(on);
    if (e == 0) e = nf_lb_det_set(on);
    if (e == 0) e = nf_cvb_det_set(on);
    if (e == 0) e = nf_cbk_det_set(on);
    if (e == 0) e = nf_ccd_det_set(on);
    if (e == 0) e = nf_fpc_det_set(on);
    if (e == 0) e = nf_fpi_det_set(on);
    if (e == 0) e = nf_fpa_det_set(on);
    if (e == 0) e = nf_mdc_det_set(on);
    if (e == 0) e = nf_mcd_det_set(on);
    if (e == 0) e = nf_rsm_det_set(on);
    if (e == 0) g_nf_det_on = on ? 1 : 0;
    return e;
}

extern "C" int nf_deterministic_enabled(void) { return g_nf_det_on; }

extern "C" int nf_deterministic_timeouts(int* count) {
    if (count == nullptr) return NF_E_BADARG;
    hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) return (int)se;
    unsigned total = 0, v = 0;
    int e = nf_ms_det_timeouts(&v);
    total += v;
    if (e == 0) { e = nf_ca_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_cpl_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_lg_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_fbh_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_ic_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_icm_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_gh_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_ghm_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_ml_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_lb_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_cvb_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_cbk_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_ccd_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_fpc_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_fpi_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_fpa_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_mdc_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_mcd_det_timeouts(&v); total += v; }
    if (e == 0) { e = nf_rsm_det_timeouts(&v); total += v; }
    if (e != 0) return e;
    *count = (int)total;
    return 0;
}
