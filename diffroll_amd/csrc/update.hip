// gfx950 (MI355X, CDNA4): the small HBM-bound kernels around the network - posterior update of all nine samplers +
// classifier-free combine + Philox noise (task/diffusion.py:804-1055), q_sample / extract_x0 (:31-64), frame confusion
// counts (:381-383), the roll -> note-run scan (:1185-1233).
#include "update_quad.h"

namespace dr {

__global__ __launch_bounds__(256) void update_kernel(const UpdateArgs a) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= a.n) return;
    reinterpret_cast<float4*>(a.x)[i4] = update_quad(a, i4);
}

__global__ void set_dyn_kernel(DynParams* d, unsigned long long seed, int first_sample, float w, float onepw) {
    d->seed = seed; d->first_sample = first_sample; d->w = w; d->onepw = onepw;
}
hipError_t launch_set_dyn(DynParams* d, unsigned long long seed, int first_sample, float w, float onepw, hipStream_t s) {
    hipLaunchKernelGGL(set_dyn_kernel, dim3(1), dim3(1), 0, s, d, seed, first_sample, w, onepw);
    return hipGetLastError();
}

hipError_t launch_update(const UpdateArgs& a, hipStream_t s) {
    const long n4 = a.n / 4;
    hipLaunchKernelGGL(update_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t launch_fill(float* p, float v, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}

// q_sample (mode 0, task/diffusion.py:31-46): out = sac[t] * x + s1m[t] * y;  extract_x0 (mode 1, :49-64):
// out = (x - s1m[t] * y) / sac[t], t per sample.  HBM-bound (12 B per element); each operation rounds once, in
// the reference's order (contraction off, IEEE division), so results are bit-identical to the torch expression.
__global__ __launch_bounds__(256) void noise_mix_kernel(int mode, const float* __restrict__ x, const float* __restrict__ y,
                                                        const int64_t* __restrict__ t, const float* __restrict__ sac,
                                                        const float* __restrict__ s1m, int n_steps, long per_sample,
                                                        float* __restrict__ out) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    long ti = t[b];
    ti = ti < 0 ? 0 : (ti >= n_steps ? n_steps - 1 : ti);
    const float a = sac[ti], c = s1m[ti];
    const long base = (long)b * per_sample;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (long)gridDim.x * 256) {
        const float xv = x[base + i], yv = y[base + i];
        out[base + i] = mode == 0 ? (a * xv) + (c * yv) : (xv - c * yv) / a;
    }
}
hipError_t launch_noise_mix(int mode, const float* x, const float* y, const int64_t* t, const float* sac,
                            const float* s1m, int n_steps, int B, long per_sample, float* out, hipStream_t s) {
    if (B <= 0 || per_sample <= 0) return hipSuccess;
    const long bx = (per_sample + 255) / 256;
    hipLaunchKernelGGL(noise_mix_kernel, dim3((unsigned)(bx < 1024 ? bx : 1024), (unsigned)B), dim3(256), 0, s, mode, x, y, t,
                       sac, s1m, n_steps, per_sample, out);
    return hipGetLastError();
}

// Roll -> note runs (task/diffusion.py:1185-1233 with onsets == frames, rule1): one thread per
// (sample, pitch) column walks the T frames once, backwards, so every note start learns its offset in
// O(T) total; lanes of a wave cover 64 consecutive pitches of a frame (coalesced 256-B reads).  Index
// work: results are exact integers.
__global__ __launch_bounds__(128) void note_runs_kernel(const float* __restrict__ roll, int* __restrict__ note_end,
                                                        int T, float thr) {
    const int b = blockIdx.x, p = threadIdx.x;
    if (p >= 88) return;
    const float* col = roll + (long)b * T * 88 + p;
    int* out = note_end + (long)b * T * 88 + p;
    int end = 0;          // offset (exclusive) of the run containing frame t, 0 when frame t is off
    for (int t = T - 1; t >= 0; --t) {
        const bool on = col[(long)t * 88] > thr;
        if (on) { if (end == 0) end = t + 1; } else end = 0;
        const bool prev_on = (t > 0) && (col[(long)(t - 1) * 88] > thr);
        out[(long)t * 88] = (on && !prev_on) ? end : 0;
    }
}
hipError_t launch_note_runs(const float* roll, int* note_end, int B, int T, float thr, hipStream_t s) {
    hipLaunchKernelGGL(note_runs_kernel, dim3((unsigned)B), dim3(128), 0, s, roll, note_end, T, thr);
    return hipGetLastError();
}

// Frame-level confusion counts of task/diffusion.py:381-383 (sklearn precision_recall_fscore_support,
// average='binary', on label.flatten() vs pred.flatten() > threshold): HBM-bound, 8 B per element,
// integer-exact (per-lane counters -> wave shuffles -> one 64-bit atomic per wave), so the metric does
// not depend on launch geometry.
__global__ __launch_bounds__(256) void frame_counts_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ label, float thr, long n,
                                                           unsigned long long* counts) {
    unsigned tp = 0, fp = 0, fn = 0;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const bool p = pred[i] > thr, l = label[i] > 0.5f;
        tp += (p && l); fp += (p && !l); fn += (!p && l);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        tp += __shfl_xor(tp, off); fp += __shfl_xor(fp, off); fn += __shfl_xor(fn, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], (unsigned long long)tp);
        atomicAdd(&counts[1], (unsigned long long)fp);
        atomicAdd(&counts[2], (unsigned long long)fn);
    }
}
hipError_t launch_frame_counts(const float* pred, const float* label, float thr, long n,
                               unsigned long long* counts, hipStream_t s) {
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(frame_counts_kernel, dim3((unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048)), dim3(256),
                       0, s, pred, label, thr, n, counts);
    return hipGetLastError();
}

}  // namespace dr
