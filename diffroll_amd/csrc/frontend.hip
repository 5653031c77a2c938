// gfx950 (MI355X, CDNA4): the mel front-end's own kernels - reflect padding, the per-frame Stockham FFT power
// spectrum, per-sample / per-frame min-max, normalise + mask + trim (torchaudio MelSpectrogram, model/diffwave.py:635-662,
// model/utils.py:10-32).  The mel filterbank and the conditioner projections are GEMMs (gemm.hip).
#include "device_common.h"

namespace dr {

// ---------------------------------------------------------------------------------------------
// front-end helpers
// ---------------------------------------------------------------------------------------------
// center=True, pad_mode='reflect': out[b][i] = wav[b][reflect(i - pad)], row stride Lp (floats)
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ wav, float* __restrict__ out,
                                                          int L, int pad, int Lp) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Lp) return;
    float v = 0.f;
    if (i < L + 2 * pad) {
        int p = i - pad;
        if (p < 0) p = -p;
        if (p >= L) p = 2 * (L - 1) - p;
        v = wav[(long)b * L + p];
    }
    out[(long)b * Lp + i] = v;
}

// the same, four outputs per lane: quads that lie inside the clip (all but the 2 * pad / 4 at its ends) are one aligned
// float4 load + store (needs L % 4 == 0 and pad % 4 == 0: every configuration of config/spec/mel.yaml)
__global__ __launch_bounds__(256) void reflect_pad4_kernel(const float* __restrict__ wav, float* __restrict__ out,
                                                           int L, int pad, int Lp) {
    const int b = blockIdx.y;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= Lp) return;
    const float* w = wav + (long)b * L;
    float4 v;
    const int p = i - pad;
    if (p >= 0 && p + 3 < L) {
        v = *reinterpret_cast<const float4*>(w + p);
    } else {
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int q = p + k;
            if (q < 0) q = -q;
            if (q >= L) q = 2 * (L - 1) - q;
            e[k] = (i + k < L + 2 * pad) ? w[q] : 0.f;
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
    }
    *reinterpret_cast<float4*>(out + (long)b * Lp + i) = v;
}

hipError_t launch_reflect_pad(const float* wav, float* out, int B, int L, int pad, hipStream_t s) {
    const int Lp = (L + 2 * pad + 3) & ~3;
    if ((L & 3) == 0 && (pad & 3) == 0 && (((uintptr_t)wav | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(reflect_pad4_kernel, dim3((unsigned)((Lp / 4 + 255) / 256), (unsigned)B), dim3(256), 0, s,
                           wav, out, L, pad, Lp);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((Lp + 255) / 256), (unsigned)B), dim3(256), 0, s,
                       wav, out, L, pad, Lp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// STFT power spectrum by FFT (torchaudio Spectrogram: torch.stft(center, reflect, hann, onesided) -> / sqrt(sum w^2)
// -> |.|^2; model/diffwave.py:635,643).  One workgroup per (clip, frame): the N real samples of the frame are
// windowed and packed as N/2 complex points z[n] = x[2n] + i x[2n+1] in LDS, transformed by a Stockham
// autosort FFT (radix-4 passes, one radix-2 pass when log2(N/2) is odd; ping-pong buffers, natural order out, no
// bit reversal), and split into the N/2+1 bins of the real transform
//     X[k] = E[k] + W_N^k O[k],   E = (Z[k] + conj Z[N/2-k]) / 2,   O = -i (Z[k] - conj Z[N/2-k]) / 2.
// Same O(eps log N) rounding behaviour as the FFT the reference runs (the earlier windowed-DFT-as-GEMM summed
// 2048 terms per bin in fp32 and differed from it by up to 2e-5 in the normalised log-mel; this path: ~3e-6).
// HBM-bound: reads 4 N bytes, writes 4 bins_p bytes per frame; twiddles (N complex) and window stay L2-resident.
// Output: power (B, TF, bins_p) row-major = "frame stride bins_p, plane stride 4" for the mel GEMM's X operand;
// bins >= N/2+1 are written as 0.
// ---------------------------------------------------------------------------------------------
DR_DEVINL float2 cmul(const float2 a, const float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__global__ __launch_bounds__(256) void stft_power_kernel(const float* __restrict__ wav_pad, const float* __restrict__ win,
                                                         const float2* __restrict__ tw, float* __restrict__ power,
                                                         int Lp, int TF, int N, int hop, int bins_p, float norm) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf0 = reinterpret_cast<float2*>(smem);
    const int H = N >> 1;                              // complex points
    float2* buf1 = buf0 + H;
    const int t = blockIdx.x, b = blockIdx.y;
    const float* x = wav_pad + (long)b * Lp + (long)t * hop;
    for (int n = threadIdx.x; n < H; n += 256) {
        const float2 xv = *reinterpret_cast<const float2*>(x + 2 * n);      // hop and the pad are multiples of 4
        const float2 wv = *reinterpret_cast<const float2*>(win + 2 * n);
        buf0[n] = make_float2(xv.x * wv.x, xv.y * wv.y);
    }
    __syncthreads();
    float2* in = buf0;
    float2* out = buf1;
    // tw[k] = exp(-2 pi i k / N), k in [0, N): the H-point transform's roots are tw[2 m]
    int Ns = 1;
    for (; Ns * 4 <= H; Ns *= 4) {                     // radix-4 passes
        const int Q = H >> 2;
        for (int j = threadIdx.x; j < Q; j += 256) {
            const int k = j & (Ns - 1);
            const int step = (H / (4 * Ns)) * 2;       // index step in tw for angle -2 pi k / (4 Ns)
            float2 u0 = in[j], u1 = in[j + Q], u2 = in[j + 2 * Q], u3 = in[j + 3 * Q];
            if (k) {
                u1 = cmul(u1, tw[k * step]);
                u2 = cmul(u2, tw[2 * k * step]);
                u3 = cmul(u3, tw[3 * k * step]);
            }
            const float2 a0 = make_float2(u0.x + u2.x, u0.y + u2.y), a1 = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 a2 = make_float2(u1.x + u3.x, u1.y + u3.y), a3 = make_float2(u1.x - u3.x, u1.y - u3.y);
            const int j0 = ((j - k) << 2) + k;
            out[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
            out[j0 + Ns] = make_float2(a1.x + a3.y, a1.y - a3.x);        // a1 - i a3
            out[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
            out[j0 + 3 * Ns] = make_float2(a1.x - a3.y, a1.y + a3.x);    // a1 + i a3
        }
        __syncthreads();
        float2* tmp = in; in = out; out = tmp;
    }
    if (Ns < H) {                                      // one radix-2 pass (log2 H odd)
        const int Q = H >> 1;
        for (int j = threadIdx.x; j < Q; j += 256) {
            const int k = j & (Ns - 1);
            const int step = (H / (2 * Ns)) * 2;
            const float2 u0 = in[j];
            float2 u1 = in[j + Q];
            if (k) u1 = cmul(u1, tw[k * step]);
            const int j0 = ((j - k) << 1) + k;
            out[j0] = make_float2(u0.x + u1.x, u0.y + u1.y);
            out[j0 + Ns] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2* tmp = in; in = out; out = tmp;
    }
    // split + |.|^2 ; Z = in[], natural order
    float* prow = power + ((long)b * TF + t) * bins_p;
    for (int k = threadIdx.x; k < bins_p; k += 256) {
        float v = 0.f;
        if (k <= H) {
            const float2 A = in[k == H ? 0 : k];
            const float2 Bz = in[(H - k) & (H - 1)];
            const float2 Bc = make_float2(Bz.x, -Bz.y);
            const float2 E = make_float2(0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y));
            const float2 D = make_float2(0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y));
            const float2 O = make_float2(D.y, -D.x);                     // -i D
            const float2 w = (k == H) ? make_float2(-1.f, 0.f) : tw[k];
            const float2 WO = cmul(w, O);
            // the reference's order: spec_f / window.pow(2).sum().sqrt()  ->  .abs()  ->  .pow(2.0)
            const float re = (E.x + WO.x) / norm, im = (E.y + WO.y) / norm;
            const float mag = sqrtf(re * re + im * im);
            v = mag * mag;
        }
        prow[k] = v;
    }
}

hipError_t launch_stft_power(const float* wav_pad, const float* win, const float* tw, float* power, int B, int Lp, int TF,
                             int N, int hop, int bins_p, float norm, hipStream_t s) {
    if (N < 8 || (N & (N - 1)) || N > 16384) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stft_power_kernel, dim3((unsigned)TF, (unsigned)B), dim3(256), (size_t)N * 8, s, wav_pad, win,
                       reinterpret_cast<const float2*>(tw), power, Lp, TF, N, hop, bins_p, norm);
    return hipGetLastError();
}

// per-sample min / max (model/utils.py:25-26) over the n_rows x TF valid values.  A clip's log-mel (119 KB at 125 frames,
// 595 KB at 640) is cut over MINMAX_CHUNKS-at-most workgroups (~32 KB each; until round 5 ONE workgroup walked a whole
// clip: 16 CUs busy, 14 us at 16 x 125 frames, 61 us at 4 x 640): wavefront shuffles, one LDS hop across the 4 waves, the
// chunk's (min, max) parked in scratch, a ticket per clip; the workgroup that draws the last ticket folds the chunks.
// min / max are exact whatever the order: bit-identical to the single-workgroup form.
constexpr int MINMAX_CHUNKS = 32;
size_t minmax_scratch_floats(int B) { return (size_t)B * (2 * MINMAX_CHUNKS + 1); }
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, float* __restrict__ mm, float* __restrict__ scratch,
                                                     int planes, int TF, int n_rows) {
    const int b = blockIdx.y, ck = blockIdx.x, nck = gridDim.x;
    const float4* xb = reinterpret_cast<const float4*>(x) + (long)b * planes * TF;
    const int vplanes = (n_rows + 3) >> 2;
    const long total = (long)vplanes * TF;
    const long per = (total + nck - 1) / nck;
    const long i1 = min(total, (ck + 1) * per);
    float mn = INFINITY, mx = -INFINITY;
    // only the LAST valid plane can hold rows >= n_rows (229 mel rows = 57 planes + 1 row): no division in the loop
    const long last0 = (long)(vplanes - 1) * TF;
    const int tail = n_rows - (vplanes - 1) * 4;             // valid rows of the last plane (1..4)
#pragma unroll 4
    for (long i = ck * per + threadIdx.x; i < i1; i += 256) {
        const float4 v = xb[i];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const int valid = i >= last0 ? tail : 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < valid) { mn = fminf(mn, vv[e]); mx = fmaxf(mx, vv[e]); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    if (nck == 1) { mm[b * 2 + 0] = mn; mm[b * 2 + 1] = mx; return; }
    float* part = scratch + (long)b * (2 * MINMAX_CHUNKS + 1);
    unsigned* ticket = reinterpret_cast<unsigned*>(part + 2 * MINMAX_CHUNKS);
    __hip_atomic_store(part + 2 * ck, mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 2 * ck + 1, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (stores -> s_waitcnt vmcnt(0) -> relaxed ticket -> agent-scope loads: as the split-K reduction of gemm_body.h)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nck - 1) return;
    for (int k = 0; k < nck; ++k) {
        mn = fminf(mn, __hip_atomic_load(part + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        mx = fmaxf(mx, __hip_atomic_load(part + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    mm[b * 2 + 0] = mn;
    mm[b * 2 + 1] = mx;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 'framewise' normalisation (model/utils.py:11-19): min / max over the n_rows frequency bins of every frame;
// one thread per (sample, frame), lanes walk consecutive frames (coalesced float4 per plane)
__global__ __launch_bounds__(256) void minmax_frame_kernel(const float* __restrict__ x, float* __restrict__ mm,
                                                           int planes, int TF, int n_rows) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= TF) return;
    const float4* xb = reinterpret_cast<const float4*>(x) + (long)b * planes * TF + t;
    float mn = INFINITY, mx = -INFINITY;
    const int vplanes = (n_rows + 3) >> 2;
    for (int pl = 0; pl < vplanes; ++pl) {
        const float4 v = xb[(long)pl * TF];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (pl * 4 + e < n_rows) { mn = fminf(mn, vv[e]); mx = fmaxf(mx, vv[e]); }
    }
    mm[((long)b * TF + t) * 2 + 0] = mn;
    mm[((long)b * TF + t) * 2 + 1] = mx;
}
hipError_t launch_minmax_frame(const float* logmel, float* mm, int B, int planes, int TF, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(minmax_frame_kernel, dim3((unsigned)((TF + 255) / 256), (unsigned)B), dim3(256), 0, s, logmel, mm,
                       planes, TF, n_rows);
    return hipGetLastError();
}

hipError_t launch_minmax(const float* logmel, float* mm, float* scratch, int B, int planes, int TF, int n_rows, hipStream_t s) {
    const long bytes = (long)((n_rows + 3) >> 2) * TF * 16;
    long chunks = (bytes + 32767) / 32768;
    chunks = chunks < 1 ? 1 : (chunks > MINMAX_CHUNKS ? MINMAX_CHUNKS : chunks);
    if (!scratch) chunks = 1;
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(256), 0, s, logmel, mm, scratch, planes, TF, n_rows);
    return hipGetLastError();
}

// (x - min) / (max - min), NaN -> 0 (model/utils.py:27-31 with min=0,max=1); mask -> -1
// (model/diffwave.py:649-654); trim to T frames (:662); pad rows -> 0.
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                        float* __restrict__ specP4, float* __restrict__ plain,
                                                        int planes_in, int planes_out, int TF, int T, int n_rows,
                                                        int mt0, int mt1, int mf0, int mf1, int framewise) {
#pragma clang fp contract(off)
    const int b = blockIdx.z, pl = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    // imagewise: one (min, max) per sample; framewise: one per (sample, frame)
    const long mi = framewise ? ((long)b * TF + t) * 2 : (long)b * 2;
    const float mn = mm[mi], mx = mm[mi + 1];
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (pl * 4 < n_rows) {
        const float4 v = reinterpret_cast<const float4*>(x)[((long)b * planes_in + pl) * TF + t];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const bool en_t = mt0 >= 0, en_f = mf0 >= 0;
        const bool in_t = (t >= mt0 && t < mt1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = pl * 4 + e;
            if (f >= n_rows) continue;
            float s = (vv[e] - mn) / (mx - mn);
            s = s * (1.0f - 0.0f) + 0.0f;
            if (s != s) s = 0.f;
            const bool in_f = (f >= mf0 && f < mf1);
            const bool masked = (en_t || en_f) && (!en_t || in_t) && (!en_f || in_f);
            o[e] = masked ? -1.f : s;
            if (plain) plain[((long)b * n_rows + f) * T + t] = o[e];
        }
    }
    reinterpret_cast<float4*>(specP4)[((long)b * planes_out + pl) * T + t] = make_float4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_normalize(const float* logmel, const float* mm, float* specP4, float* spec_plain, int B,
                            int planes_in, int planes_out, int TF, int T, int n_rows, int mt0, int mt1, int mf0,
                            int mf1, hipStream_t s, int framewise) {
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)planes_out, (unsigned)B),
                       dim3(256), 0, s, logmel, mm, specP4, spec_plain, planes_in, planes_out, TF, T, n_rows, mt0,
                       mt1, mf0, mf1, framewise);
    return hipGetLastError();
}

hipError_t init_frontend_kernels() {
    // the FFT front-end holds two n_fft/2-point complex buffers: n_fft * 8 bytes (128 KiB at its largest size, 16384)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_power_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace dr
