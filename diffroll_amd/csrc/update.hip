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
// V4: one float4 per lane and operand (per_sample % 4 == 0 and 16-byte aligned tensors: every roll).
template <int V4>
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
    if constexpr (V4) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* y4 = reinterpret_cast<const float4*>(y + base);
        float4* o4 = reinterpret_cast<float4*>(out + base);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (per_sample >> 2); i += (long)gridDim.x * 256) {
            const float4 xv = x4[i], yv = y4[i];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = mode == 0 ? (a * xs[e]) + (c * ys[e]) : (xs[e] - c * ys[e]) / a;
            o4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (long)gridDim.x * 256) {
            const float xv = x[base + i], yv = y[base + i];
            out[base + i] = mode == 0 ? (a * xv) + (c * yv) : (xv - c * yv) / a;
        }
    }
}
hipError_t launch_noise_mix(int mode, const float* x, const float* y, const int64_t* t, const float* sac,
                            const float* s1m, int n_steps, int B, long per_sample, float* out, hipStream_t s) {
    if (B <= 0 || per_sample <= 0) return hipSuccess;
    const bool v4 = (per_sample & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) == 0);
    const long items = v4 ? per_sample >> 2 : per_sample;
    const long bx = (items + 255) / 256;
    const dim3 grid((unsigned)(bx < 1024 ? bx : 1024), (unsigned)B);
    if (v4) hipLaunchKernelGGL(noise_mix_kernel<1>, grid, dim3(256), 0, s, mode, x, y, t, sac, s1m, n_steps, per_sample, out);
    else hipLaunchKernelGGL(noise_mix_kernel<0>, grid, dim3(256), 0, s, mode, x, y, t, sac, s1m, n_steps, per_sample, out);
    return hipGetLastError();
}

// Roll -> note runs (task/diffusion.py:1185-1233 with onsets == frames, rule1): note_end[b][t][p] = the frame at which
// the note that STARTS at (t, p) ends (exclusive), 0 elsewhere.  Index work: results are exact integers.
// 1 / 4 / 8 workgroups per clip (pitch ranges), three passes instead of a serial walk per pitch column (round 5:
// profiles/r05_membound_kernels*.txt):
//   1. the thresholded roll as one bit per (pitch, frame) in LDS - mask[p][t / 32], LDS atomics, rows read coalesced;
//   2. barrier;
//   3. every output element in memory order: (t, p) starts a note when its bit is set and bit t - 1 is not; its end is the
//      first clear bit behind t in that pitch's mask (a word scan with ctz; bits >= T are clear, so the scan ends at T at
//      the latest).  Starts are rare, the scan is short, the stores are coalesced.
// Rolls longer than NOTE_RUNS_MAX_T frames (mask > 150 KB of LDS) take the column walk below.
constexpr int NOTE_RUNS_MAX_T = 12000;
__global__ __launch_bounds__(512) void note_runs_kernel(const float* __restrict__ roll, int* __restrict__ note_end,
                                                        int T, float thr) {
    extern __shared__ unsigned note_mask[];            // [PG][W], W = T / 32 + 2 words (one spare clear word per pitch)
    const int W = T / 32 + 2;
    const int b = blockIdx.x;
    // gridDim.y workgroups share a clip by pitch ranges of PG = 88 / gridDim.y pitches (rows are read in PG-float segments)
    const int PG = 88 / (int)gridDim.y, p0 = (int)blockIdx.y * PG;
    const int n = T * PG;                              // (T <= NOTE_RUNS_MAX_T: 32-bit index math)
    const float* src = roll + (long)b * T * 88 + p0;
    int* dst = note_end + (long)b * T * 88 + p0;
    for (int i = threadIdx.x; i < PG * W; i += 512) note_mask[i] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 512) {
        const int t = i / PG, p = i - t * PG;
        if (src[t * 88 + p] > thr) atomicOr(&note_mask[p * W + (t >> 5)], 1u << (t & 31));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 512) {
        const int t = i / PG, p = i - t * PG;
        const unsigned* m = note_mask + p * W;
        int end = 0;
        const bool on = (m[t >> 5] >> (t & 31)) & 1u;
        const bool prev = t > 0 && ((m[(t - 1) >> 5] >> ((t - 1) & 31)) & 1u);
        if (on && !prev) {
            int w = t >> 5;
            unsigned clear = ~m[w] & (0xFFFFFFFFu << (t & 31));       // clear bits at or behind t in this word (bit t itself is set)
            while (clear == 0u) clear = ~m[++w];                      // (the spare word is all clear: terminates)
            end = min(w * 32 + __builtin_ctz(clear), T);
        }
        dst[t * 88 + p] = end;
    }
}
// the serial form (one thread per (sample, pitch) column, backwards): any T
__global__ __launch_bounds__(128) void note_runs_columns_kernel(const float* __restrict__ roll, int* __restrict__ note_end,
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
    if (T <= NOTE_RUNS_MAX_T) {
        const int groups = T >= 256 ? 8 : (T >= 64 ? 4 : 1);          // workgroups per clip (11 / 22 / 88 pitches each)
        const size_t lds = (size_t)(88 / groups) * (T / 32 + 2) * 4;
        hipLaunchKernelGGL(note_runs_kernel, dim3((unsigned)B, (unsigned)groups), dim3(512), lds, s, roll, note_end, T, thr);
    } else {
        hipLaunchKernelGGL(note_runs_columns_kernel, dim3((unsigned)B), dim3(128), 0, s, roll, note_end, T, thr);
    }
    return hipGetLastError();
}
hipError_t init_update_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&note_runs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Frame-level confusion counts of task/diffusion.py:381-383 (sklearn precision_recall_fscore_support,
// average='binary', on label.flatten() vs pred.flatten() > threshold): HBM-bound, 8 B per element, integer-exact.
// Every block counts its slice (per-lane counters -> wave shuffles -> LDS) and parks three partial sums; the block that
// draws the last ticket adds the partials up in block order and writes {TP, FP, FN}.  (Until round 5 every WAVE added its
// counts to the three result words with device atomics: ~2000 same-address atomics at 50 ns each - 101 us for 1.4 MB.)
// work = [0..2] result, [3] ticket (zero between launches: re-armed by the last block), then one 16-byte partial per block.
constexpr int FRAME_COUNTS_MAX_BLOCKS = 256;
__global__ __launch_bounds__(256) void frame_counts_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ label, float thr, long n,
                                                           unsigned long long* work) {
    unsigned tp = 0, fp = 0, fn = 0;
    const long stride = (long)gridDim.x * 256;
    const long n4 = ((((uintptr_t)pred | (uintptr_t)label) & 15) == 0) ? (n >> 2) : 0;      // float4 part
#pragma unroll 4
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 pv = reinterpret_cast<const float4*>(pred)[i], lv = reinterpret_cast<const float4*>(label)[i];
        const float ps[4] = {pv.x, pv.y, pv.z, pv.w}, ls[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool p = ps[e] > thr, l = ls[e] > 0.5f;
            tp += (p && l); fp += (p && !l); fn += (!p && l);
        }
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const bool p = pred[i] > thr, l = label[i] > 0.5f;
        tp += (p && l); fp += (p && !l); fn += (!p && l);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        tp += __shfl_xor(tp, off); fp += __shfl_xor(fp, off); fn += __shfl_xor(fn, off);
    }
    __shared__ unsigned sh[4][3];
    __shared__ unsigned last_s;
    __shared__ unsigned long long tot[4][3];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // the partials travel as ONE 16-byte write-through store per block (sc0 sc1) and are re-read with L2-bypassing loads:
    // agent-scope atomic stores to neighbouring words serialise at the memory channel like the atomics they replace
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)(work + 4), 0, (unsigned)FRAME_COUNTS_MAX_BLOCKS * 16u, 0x00020000);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = tp; sh[threadIdx.x >> 6][1] = fp; sh[threadIdx.x >> 6][2] = fn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32x4 mine = {sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0], sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1],
                            sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2], 0u};       // (a block counts < 2^32 elements: launcher)
        __builtin_amdgcn_raw_buffer_store_b128(mine, pr, (int)blockIdx.x * 16, 0, 17);
        // stores -> s_waitcnt vmcnt(0) -> relaxed ticket -> L2-bypassing loads: the hand-off form of the split-K reduction
        // (gemm_body.h); a release / acquire fence pair here costs an L2 write-back + invalidate per block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ticket = __hip_atomic_fetch_add(work + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = ticket == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;
    unsigned long long c[3] = {0, 0, 0};
    if (threadIdx.x < gridDim.x) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(pr, (int)threadIdx.x * 16, 0, 17);
        c[0] = v.x; c[1] = v.y; c[2] = v.z;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] += __shfl_xor(c[k], off);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) tot[threadIdx.x >> 6][k] = c[k];
    __syncthreads();
    if (threadIdx.x < 3) work[threadIdx.x] = tot[0][threadIdx.x] + tot[1][threadIdx.x] + tot[2][threadIdx.x] + tot[3][threadIdx.x];
    if (threadIdx.x == 0) __hip_atomic_store(work + 3, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
size_t frame_counts_work_words() { return 4 + 2 * (size_t)FRAME_COUNTS_MAX_BLOCKS; }
hipError_t launch_frame_counts(const float* pred, const float* label, float thr, long n,
                               unsigned long long* work, hipStream_t s) {
    // ~4096 elements (32 KB of the two operands: four float4 pairs per lane) per block, at most FRAME_COUNTS_MAX_BLOCKS
    // blocks (every block costs one same-address ticket atomic, ~50 ns each)
    long blocks = (n + 4095) / 4096;
    blocks = blocks < 1 ? 1 : (blocks > FRAME_COUNTS_MAX_BLOCKS ? FRAME_COUNTS_MAX_BLOCKS : blocks);
    if (n / blocks >= (1ll << 32)) return hipErrorInvalidValue;      // (a block's partial counts are 32-bit)
    hipLaunchKernelGGL(frame_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, label, thr, n, work);
    return hipGetLastError();
}

}  // namespace dr
