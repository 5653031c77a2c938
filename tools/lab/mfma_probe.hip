// Practical fp32-MFMA ceiling on this box: pure v_mfma_f32_32x32x2_f32 loop on random vs zero register
// data, 1 or 2 waves per SIMD; reports TFLOP/s (wall) and the effective shader clock (s_memtime / wall).
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe(const float* in, float* out, long long* cyc, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xFFFFF]; b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF]; }
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 3) & 7], b[(i + 2) & 7], acc[3], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    out[tid] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    float *din, *dzero, *dout;
    long long* dcyc;
    hipMalloc(&din, N * 4); hipMalloc(&dzero, N * 4); hipMalloc(&dout, 2048 * 256 * 4); hipMalloc(&dcyc, 2048 * 8);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dzero, 0, N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;   // 32 MFMAs per iteration
    for (int blocks : {256, 512}) for (int zero = 0; zero < 2; ++zero) {
        const float* src = zero ? dzero : din;
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, src, dout, dcyc, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, src, dout, dcyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
        printf("blocks=%d (%d wave/SIMD) data=%s: %.3f ms  %.1f TFLOP/s  memtime ticks/MFMA=%.1f  ticks/us=%.1f\n", blocks,
               blocks / 256, zero ? "zero" : "random", ms, flops / ms / 1e9, (double)c / (iters * 32.0), c / (ms * 1e3));
    }
    return 0;
}
