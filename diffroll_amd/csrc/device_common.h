// Shared device-side helpers of the gfx950 kernels (included by every kernel translation unit; not an interface).
#pragma once
#include "kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace dr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DR_DEVINL __device__ __forceinline__

// DR_FAULT (compile-time, LITMUS builds only - results are WRONG on purpose; tests/test_gpu_fused.py, tools/xcd_stress.py):
//   1 = the producers' LDS-DMA hand-over barrier without its s_waitcnt vmcnt(0) (the round-3 race: consumers may read the
//       buffer's previous occupant),  2 = the tensors handed to other workgroups stored PLAIN whatever the placement
//       (no sc1 write-through: a group spread over several XCDs then reads stale lines / stale memory)
#ifndef DR_FAULT
#define DR_FAULT 0
#endif

// DR_BOUNDS (compile-time, checker builds only: tools/checked_build.sh): every hand-computed LDS address and every
// in-range buffer offset of the GEMM bodies and the fused kernel is compared with the region it must stay in; the
// first violation is recorded in g_bounds (code, two details) and counted - nothing traps, the run completes and
// dr_debug_bounds reports.  (The LDS-DMA X-tile loads and pw_body's activation loads go out of range ON PURPOSE -
// the hardware bounds check of the buffer descriptor is the conv's zero padding - and are not checked.)
#ifdef DR_BOUNDS
static __device__ unsigned long long g_bounds[4];     // per translation unit: {code of the first violation, detail, detail, violations}
DR_DEVINL void bounds_fail(unsigned code, long a, long b) {
    if (atomicCAS(&g_bounds[0], 0ull, (unsigned long long)code) == 0ull) { g_bounds[1] = (unsigned long long)a; g_bounds[2] = (unsigned long long)b; }
    atomicAdd(&g_bounds[3], 1ull);
}
DR_DEVINL unsigned lds_off(const void* p) {          // byte address inside the workgroup's LDS allocation
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
#define DR_CHECK(cond, code, a, b) do { if (!(cond)) bounds_fail((code), (long)(a), (long)(b)); } while (0)
// a 16-byte LDS access at p must lie inside [lo, hi)
#define DR_CHECK_LDS(p, lo, hi, code) do { const unsigned o_ = lds_off(p); if (o_ < (unsigned)(lo) || o_ + 16u > (unsigned)(hi)) bounds_fail((code), o_, (hi)); } while (0)
// host side of the checker: what a launch will touch of each tensor argument (base + extent of the buffer descriptors /
// flat accesses built from GemmArgs) against the device allocation the pointer lives in (hipMemGetAddressRange)
static unsigned long long g_host_violations = 0;
static void host_extent(const void* p, size_t bytes, const char* what, const char* kernel) {
    if (!p || !bytes) return;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return; }   // not a device allocation
    if ((const char*)p < (const char*)base || (const char*)p + bytes > (const char*)base + size) {
        ++g_host_violations;
        fprintf(stderr, "[DR_BOUNDS] %s: %s needs %zu bytes at +%zd of a %zu-byte allocation\n", kernel, what, bytes,
                (ssize_t)((const char*)p - (const char*)base), size);
    }
}
static void check_gemm_extents(const GemmArgs& a, int epi, int prec, const char* kernel) {
    const size_t slab = prec ? 24576 : 16384;
    const int mts = a.mt0 + a.MT;
    host_extent(a.Wp, (size_t)mts * a.kchunks * a.taps * slab, "packed weights", kernel);
    host_extent(a.bias, (size_t)mts * 128 * 4, "bias", kernel);
    if (epi == EPI_GATE) host_extent(a.bias2, (size_t)mts * 128 * 4, "bias2", kernel);
    const long nbx = a.x_bmod ? (a.NB < a.x_bmod ? a.NB : a.x_bmod) : a.NB;
    if (!prec) host_extent(a.X, (size_t)((nbx - 1) * a.x_bs + (long)(a.x_planes - 1) * a.x_ps + (long)(a.T - 1) * a.x_fs + 4) * 4, "X", kernel);
    const long nby = a.NB + (a.dual > 0 ? a.dual : 0);
    const long planes = (a.y_rows + 3) / 4;
    if (!(a.out_s3 & 1)) host_extent(a.Y, (size_t)((nby - 1) * a.y_bs + (planes - 1) * a.y_ps + (long)(a.T - 1) * a.y_fs + 4) * 4, "Y", kernel);
    if (a.Y2 && !(a.out_s3 & 2)) host_extent(a.Y2, (size_t)((nby - 1) * a.y2_bs + (planes - 1) * a.y_ps + (long)(a.T - 1) * a.y_fs + 4) * 4, "Y2", kernel);
    if (epi == EPI_GATE && a.cond) host_extent(a.cond, (size_t)((long)(a.n_cond > 1 ? a.n_cond - 1 : 0) * a.c_bs + (long)mts * 128 * a.T) * 4, "conditioner", kernel);
    if (epi == EPI_GATE && a.cond2) host_extent(a.cond2, (size_t)mts * 128 * a.T * 4, "conditioner (shared)", kernel);
    if (epi == EPI_RES_SKIP && a.skip && mts * 128 > a.y_rows)
        host_extent(a.skip, (size_t)((long)(a.NB - 1) * a.s_bs + (long)(mts * 128 - a.y_rows) * a.T) * 4, "skip", kernel);
    if (a.ws) host_extent(a.ws, a.ws_floats * 4, "split-K workspace", kernel);
    if (a.ws_cnt) host_extent(a.ws_cnt, a.ws_cnt_n * 4, "split-K counters", kernel);
}
// every kernel translation unit keeps its own g_bounds / g_host_violations; DR_BOUNDS_TU(tag) defines the pair of
// readers that read_bounds() / reset_bounds() (gemm.hip) aggregate over the units
#define DR_BOUNDS_TU(tag)                                                                                       \
    hipError_t read_bounds_##tag(unsigned long long* out4) {                                                    \
        hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_bounds), 4 * sizeof(unsigned long long));         \
        if (e == hipSuccess && g_host_violations) { if (!out4[0]) out4[0] = 999; out4[3] += g_host_violations; } \
        return e;                                                                                               \
    }                                                                                                           \
    hipError_t reset_bounds_##tag() {                                                                           \
        unsigned long long z[4] = {0, 0, 0, 0};                                                                 \
        g_host_violations = 0;                                                                                  \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), z, sizeof z);                                            \
    }
#define DR_CHECK_EXTENTS(a, epi, prec, kernel) check_gemm_extents((a), (epi), (prec), (kernel))
#else
#define DR_CHECK_EXTENTS(a, epi, prec, kernel) do {} while (0)
#define DR_CHECK(cond, code, a, b) do {} while (0)
#define DR_CHECK_LDS(p, lo, hi, code) do {} while (0)
#define DR_BOUNDS_TU(tag)                                                                      \
    hipError_t read_bounds_##tag(unsigned long long*) { return hipErrorNotSupported; }         \
    hipError_t reset_bounds_##tag() { return hipErrorNotSupported; }
#endif
// per-unit readers (DR_BOUNDS_TU) and one-time initialisers (dynamic-LDS attributes), called from gemm.hip
hipError_t read_bounds_gemm(unsigned long long*);  hipError_t reset_bounds_gemm();
hipError_t read_bounds_stack(unsigned long long*); hipError_t reset_bounds_stack();
hipError_t read_bounds_tail(unsigned long long*);  hipError_t reset_bounds_tail();
hipError_t init_stack_kernels();
hipError_t init_tail_kernels();
hipError_t init_frontend_kernels();

struct A8 { float4 v[8]; };                          // A fragments of one K step: [group g][row tile mi]
struct A12 { uint4 v[12]; };                         // split-bf16 A fragments of one K step: [(g*3 + piece)*2 + mi]
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// "S3" split precision: an fp32 value x is carried as three bf16 pieces x = p0 + p1 + p2 (each the
// round-to-nearest-even bf16 of the remaining residual; 3 x 8 significant bits reconstruct the 24-bit
// fp32 significand EXACTLY).  A product a*b is formed from the six piece products with i + j <= 2
// (dropping terms <= 2^-24 |ab|, i.e. one fp32 ulp) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// measured dot-product error is below that of a plain fp32 FMA chain, at 16/6 = 2.67x the matrix rate
// of v_mfma_f32_32x32x2_f32.  S3 tensor layout: [batch][piece 3][plane8 = channel/8][frame][8 bf16].
// ---------------------------------------------------------------------------------------------
DR_DEVINL uint32_t bf16_rne_bits(float x) {          // fp32 bits of bf16(x) (low 16 bits zero)
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}
DR_DEVINL void split3(float x, uint32_t (&pc)[3]) {
    pc[0] = bf16_rne_bits(x);
    const float r1 = x - __uint_as_float(pc[0]);     // exact
    pc[1] = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(pc[1]);    // exact
    pc[2] = bf16_rne_bits(r2);
}
// store 4 consecutive channels (one C/D register quad) of frame t as the three bf16 pieces:
// dst = S3 tensor base of this sample; the quad is the low (half = 0) or high 8 bytes of its plane8 unit
// COH / write_through: as store_f4 - 8-byte write-through (sc1) stores for tensors handed to other workgroups of the launch
template <int COH = 0>
DR_DEVINL void store_s3_quad(float* dst, const float (&v)[4], int row0, int t, int T, int P8, const int write_through = 0) {
    uint32_t pc[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(v[e], pc[e]);
    const long plane8 = row0 >> 3;
    const int half = (row0 >> 2) & 1;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint2 w;
        w.x = (pc[0][p] >> 16) | pc[1][p];
        w.y = (pc[2][p] >> 16) | pc[3][p];
        char* q = reinterpret_cast<char*>(dst) + (((long)p * P8 + plane8) * T + t) * 16 + half * 8;
        if constexpr (COH && DR_FAULT != 2) {
            if (write_through) {       // wave-uniform
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 d = {w.x, w.y};
                asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(q), "v"(d) : "memory");
                continue;
            }
        }
        *reinterpret_cast<uint2*>(q) = w;
    }
}
DR_DEVINL f32x16 mma_bf16(const uint4 a, const uint4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// sched_group_barrier helper (masks: 0x8 MFMA, 0x100 DS read): the next N instructions of that class
// are scheduled here, in program order - used to pin the fragment-read software pipeline.
template <int MASK, int N>
DR_DEVINL void sgb() {
    if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
}
// N x (2 MFMAs, 1 LDS read [, 1 vector-memory read for the first V]): fragment reads and prefetch loads issued
// in the shadow of the running MFMAs
template <int N, int V>
DR_DEVINL void sgb_mix() {
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (V > 0) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        sgb_mix<N - 1, (V > 0 ? V - 1 : 0)>();
    }
}
// (x + residual) / math.sqrt(2.0) (model/diffwave.py:151) is an IEEE fp32 division by fp32(sqrt 2) in ATen.
// For a constant divisor the correctly rounded quotient takes three instructions (Markstein): q = RN(x y),
// r = x - q d (exact, fma), q' = RN(q + r y) with y = RN(1/d) - verified bit-identical to x / d over every
// fp32 significand - instead of the ~10-instruction v_div_scale / v_div_fmas / v_div_fixup sequence.
DR_DEVINL float div_sqrt2(float x) {
    constexpr float d = 1.41421356237309504880f, y = 1.0f / d;
    const float q = x * y;
    const float r = fmaf(-q, d, x);
    return fmaf(r, y, q);
}
// N x (PER MFMAs, 1 vector-memory read)
template <int N, int PER>
DR_DEVINL void sgb_spread() {
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x8, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        sgb_spread<N - 1, PER>();
    }
}
DR_DEVINL float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// 16-byte store of one P4 quad.  COH = 1: write-through (sc1) - the line leaves this XCD's L2 for memory, where
// every other XCD's sc1 load finds it; the instruction sits in inline asm (no builtin takes a flat pointer with a
// cache policy), so the COMPILER does not count it: every wave drains with an explicit s_waitcnt vmcnt(0)
// before it signals (stack_kernel's group barrier).  The trailing s_nop covers the store-data hazard.
template <int COH>
DR_DEVINL void store_f4(float* dst, const float4 v, const int write_through) {
    if constexpr (COH && DR_FAULT != 2) {
        if (write_through) {           // wave-uniform (a kernel-wide mode)
            const f32x4 d = {v.x, v.y, v.z, v.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(d) : "memory");
            return;
        }
    }
    *reinterpret_cast<float4*>(dst) = v;
}
DR_DEVINL float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// Gate of the residual block (model/diffwave.py:146-147) on the hardware transcendentals: v_exp_f32 (2^x) and
// v_rcp_f32, ~1 ulp each.  sigmoid(u) = 1 / (1 + 2^(-u log2 e)); tanh(v) = 1 - 2 / (2^(2 v log2 e) + 1), which
// saturates correctly at +-1 (2^x -> inf / 0) and has absolute error <= ~1.2e-7 near 0.  ~10 instructions per
// output instead of ~60 for the libm-accurate expf / tanhf / IEEE divisions: the gate is 64 transcendental
// evaluations per lane and was 17k of the conv kernel's 640k cycles.
DR_DEVINL float gatef_(float u, float v) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * u));
    const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008177792681472f * v) + 1.0f), 1.0f);
    return sg * th;
}

}  // namespace dr
