// gfx950 (MI355X, CDNA4) kernels of the DiffRoll sampling engine.
//
// One implicit-GEMM design carries every contraction of the path, in three consumer flavours that share
// the LDS-DMA producers, the P4 / S3 activation layouts and the packed weights:
//   gemm_kernel<.., PREC=0>  exact fp32 on v_mfma_f32_32x32x2_f32      (default; 64/128-frame blocks)
//   gemm_kernel<.., PREC=1>  split-bf16 on v_mfma_f32_32x32x16_bf16    (opt-in "bf16x3", hot kernels)
//   gemm16_kernel            exact fp32 on v_mfma_f32_16x16x4_f32      (96/160-frame blocks, hot kernels)
// used for
//   * dilated Conv1d (k taps) + conditioner add + sigmoid*tanh gate   (model/diffwave.py:139-147)
//   * 1x1 output projection + residual/skip update (+ h + d_next)      (model/diffwave.py:149-151, :138)
//   * input / skip / output projections of the net                     (model/diffwave.py:667-668, 683-685)
//   * conditioner projections, step-embedding MLP                      (hoisted; :126,128,65-74)
//   * the STFT as a windowed-DFT GEMM and the mel filterbank GEMM      (torchaudio MelSpectrogram)
// plus small HBM-bound kernels: posterior update (all nine samplers) + classifier-free combine + Philox
// noise (task/diffusion.py:804-1055), reflect padding, per-sample min/max + normalise/mask/trim
// (model/utils.py:21-32, model/diffwave.py:644-662), frame confusion counts (:381-383) and the
// roll -> note-run scan (:1185-1233).
//
// Written for wave64, 512-thread workgroups (4 consumer + 4 producer waves); gfx950 only.
#include "kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace dr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DR_DEVINL __device__ __forceinline__

// DR_ABLATE (compile-time, measurement builds only; results are WRONG when non-zero):
//   1 = no A-fragment prefetch in the K loop, 2 = (S3 path) A fragments always from slab 0 (cache-hot),
//   9 = producers do no loads / LDS writes (barriers only)
#ifndef DR_ABLATE
#define DR_ABLATE 0
#endif

// DR_BOUNDS (compile-time, checker builds only: tools/checked_build.sh): every hand-computed LDS address and every
// in-range buffer offset of the GEMM bodies and the fused kernel is compared with the region it must stay in; the
// first violation is recorded in g_bounds (code, two details) and counted - nothing traps, the run completes and
// dr_debug_bounds reports.  (The LDS-DMA X-tile loads and pw_body's activation loads go out of range ON PURPOSE -
// the hardware bounds check of the buffer descriptor is the conv's zero padding - and are not checked.)
#ifdef DR_BOUNDS
__device__ unsigned long long g_bounds[4];            // {code of the first violation, detail, detail, violations}
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
hipError_t read_bounds(unsigned long long* out4) {
    hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_bounds), 4 * sizeof(unsigned long long));
    if (e == hipSuccess && g_host_violations) { if (!out4[0]) out4[0] = 999; out4[3] += g_host_violations; }
    return e;
}
hipError_t reset_bounds() {
    unsigned long long z[4] = {0, 0, 0, 0};
    g_host_violations = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), z, sizeof z);
}
#define DR_CHECK_EXTENTS(a, epi, prec, kernel) check_gemm_extents((a), (epi), (prec), (kernel))
#else
#define DR_CHECK_EXTENTS(a, epi, prec, kernel) do {} while (0)
#define DR_CHECK(cond, code, a, b) do {} while (0)
#define DR_CHECK_LDS(p, lo, hi, code) do {} while (0)
hipError_t read_bounds(unsigned long long*) { return hipErrorNotSupported; }
hipError_t reset_bounds() { return hipErrorNotSupported; }
#endif

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
DR_DEVINL void store_s3_quad(float* dst, const float (&v)[4], int row0, int t, int T, int P8) {
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
    if constexpr (COH) {
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

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM kernel.
//   block = 512 threads = 8 waves, SPECIALISED (measured with the s_memtime hook: with one wave per
//   SIMD every non-MFMA instruction cluster in the consumer's in-order stream is exposed - X loads,
//   select/add VALU, ds_write and the barrier cost 6.7 of 74.6 ticks per MFMA - while instructions of
//   ANOTHER wave on the same SIMD overlap the 64-cycle MFMAs):
//     waves 0-3  consumers, one per SIMD, 4 (M) x 1 (N): wave tile 32 rows x 64*NI frames (for the paired
//                epilogues the 32 rows are 16 gate/cos + 16 filter/sin rows of the SAME channels, so
//                pairing is register-local).  They only issue MFMAs, the A-fragment loads and the
//                B-fragment ds_reads.
//     waves 4-7  producers: stage the X tile of the NEXT chunk with LDS-DMA (hardware zero padding)
//                while the consumers compute the current one.
//   One s_barrier per chunk hands a staged buffer over (double buffered).
//
//   A operand (weights): NEVER staged through LDS.  The packed layout is fragment-shaped, so every
//   consumer wave loads the 4 float4 A-fragments of a K step (4 channel groups x its 32 rows) straight
//   from L2 into VGPRs - buffer loads off the M tile's weight panel (resource in SGPRs, fixed per-lane
//   offset, scalar per-step offset: no vector address arithmetic), two fully coalesced 512-B segments per
//   instruction, one step ahead of use, spread through the first group's MFMAs.  The 128-row weight panel
//   of an M tile is L2-resident: blockIdx % MT pins a panel to an XCD.
//   B operand (activations): X tile [KS*8 planes][FW = BN + 2*halo frames][float4] in LDS; all taps of
//   the dilated conv read it at shifted frame offsets with conflict-free ds_read_b128.
//   K loop: for chunk (32*KS input channels) for tap for sub-chunk: 64*NI MFMAs per consumer wave.
// ---------------------------------------------------------------------------------------------
//   PREC = 1 ("S3"): the X input and the weights are split-bf16 (see above): X tile rows are
//   [(sub*2 + g)*6 + piece*2 + kq] (16 channels per group g, 8 per kq half), the consumers run 6
//   v_mfma_f32_32x32x16_bf16 per (group, row tile, frame tile) instead of 8 fp32 MFMAs per 16 channels.
//   COH = 1 (the fused residual-stack kernel only): the tensors this body exchanges with OTHER workgroups of the
//   same launch - its X input (hd) and its EPI_GATE output (g) - are read with sc1 loads and written with sc1
//   (write-through) stores, the placement-independent hand-off form of MI355X_MICROARCH.md "inter-workgroup
//   visibility"; everything else (weights, biases, conditioner: written before the launch) stays plain.
template <int NI, int KS, int EPI, int PREC, int COH>
DR_DEVINL void gemm_body(const GemmArgs& a, char* smem, const int mt, const int nt, const int ks) {
    constexpr int BN = 64 * NI;
    constexpr int XP = (PREC ? 12 : 8) * KS;      // 16-byte rows per X tile
    // Consumer wave arrangement: 4 (M) x 1 (N) - every wave owns 32 distinct rows x all 64*NI frames of the
    // block, so no two waves issue the same A-fragment loads (a CU's vector-memory path is the stressed
    // resource: with 2 x 2 the two N-waves fetched identical fragments); the X tile is shared through LDS.
    // Paired epilogues (gate / |.|^2) find both members of a pair inside one 32-row MFMA tile: packed rows
    // of a wave are [16 gate (cos) channels, 16 filter (sin) channels], i.e. C/D register quads q and q+2
    // of the same lane.
    constexpr bool PAIRED = (EPI == EPI_GATE || EPI == EPI_POWER);
    constexpr int WNC = 1;                        // consumer waves along N
    constexpr int MI = 1;                         // 32-row MFMA tiles per wave
    constexpr int NW = 2 * NI;                    // 32-frame MFMA tiles per wave
    constexpr int WROWS = MI * 32;                // rows per wave
    constexpr int WFR = NW * 32;                  // frames per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long tick0 = a.dbg ? clock64() : 0;   // measurement hook (null in production launches)

    const int halo = ((a.taps - 1) >> 1) * a.dil;
    const int FW = BN + 2 * halo;
    float4* Xs = reinterpret_cast<float4*>(smem);   // [2][XP][FW]
    // EPI_RES_SKIP: the tile of h (residual rows) / skip (skip rows) this block read-modify-writes,
    // [32 planes][BN frames] float4, DMA'd by the producers at kernel start and read by the epilogue.
    // (Holding it in 64 prefetch VGPRs instead cost the compiler the B-fragment software pipelining.)
    float4* Rs = Xs + 2 * XP * FW;
#ifdef DR_BOUNDS
    const unsigned xs0 = lds_off(Xs), xs1 = xs0 + 2u * XP * FW * 16u, rs1 = xs1 + (EPI == EPI_RES_SKIP ? 32u * BN * 16u : 0u);
    if (tid == 0 && !COH) DR_CHECK(rs1 <= (unsigned)a.lds_bytes, 100, rs1, a.lds_bytes);      // the regions fit the launch's LDS
#endif

    // (mt, nt, ks) = this block's M tile, frame tile and K split: chosen by the caller (gemm_kernel below)
    const int tps = (a.T + BN - 1) / BN;
    const int b = nt / tps;
    const int t0 = (nt % tps) * BN;
    const int NS = a.kchunks * a.taps;              // K steps (32 channels x 1 tap each)
    const int cps = a.kchunks / KS / a.ksplit;      // chunks of this block: [c0, c1)
    const int c0 = ks * cps, c1 = c0 + cps;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers (LDS-DMA)
        // Each X-tile plane row (FW float4 = frames t0-halo .. t0+BN+halo-1 of 4 channels) is copied
        // global -> LDS by ceil(FW/64) `buffer_load_dwordx4 ... lds` instructions (64 lanes x 16 B, LDS
        // destination = wave-uniform base + lane*16).  The buffer descriptor covers exactly frames
        // [0, T) of that plane, so frames outside the clip - the conv's zero padding and the tail of the
        // last tile - come back as 0 from the hardware bounds check: no VALU, no ds_write, no VGPR
        // staging.  Producers therefore issue a handful of instructions per chunk and no longer steal
        // issue slots from the consumers' MFMA stream (measured: 70.7 -> 66 ticks per MFMA when idle).
        const int pw = wave - 4;
        const int bx = a.x_bmod ? (b % a.x_bmod) : b;
        const float* Xg = a.X + (long)bx * a.x_bs;
        const int last_plane = a.x_planes - 1;
        const unsigned recs = ((unsigned)(a.T - 1) * (unsigned)a.x_fs + 4u) * 4u;   // bytes of one plane row
        const int wl = (FW + 63) >> 6;                  // wave-loads per plane row
        const int total = XP * wl;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        auto issue = [&](int chunk) {
            for (int i = pw; i < total; i += 4) {
                const int pl = i / wl, seg = i - pl * wl;
                const int f = seg * 64 + lane;
                // planes beyond Cin (K padding) re-read the last valid plane: finite data x zero weights
                const float* src;
                if constexpr (PREC) {
                    const int sg = pl / 6, rem = pl - sg * 6, pce = rem >> 1, kq = rem & 1;
                    const int pc = min(chunk * (4 * KS) + sg * 2 + kq, last_plane);       // plane8
                    src = Xg + (long)pce * a.x_piece + (long)pc * a.x_ps;
                } else {
                    src = Xg + (long)min(chunk * XP + pl, last_plane) * a.x_ps;
                }
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, recs, 0x00020000);
                const int voff = (t0 - halo + f) * (int)a.x_fs * 4;   // negative / past the end => reads 0
                float4* dst = Xs + (((chunk - c0) & 1) * XP + pl) * FW + seg * 64;
                if (f < FW) DR_CHECK_LDS(dst + lane, xs0, xs1, 101);
                if (f < FW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)dst, 16, voff, 0, 0, COH ? 16 : 0);
            }
        };
        if constexpr (EPI == EPI_RES_SKIP) {
            const unsigned rrecs = (unsigned)a.T * 16u;
            constexpr int RWL = BN / 64;
            for (int i = pw; i < 32 * RWL; i += 4) {
                const int pl = i / RWL, seg = i - pl * RWL;
                const int row0 = mt * 128 + pl * 4;
                const float* src = (row0 < a.y_rows)
                    ? a.Y + (long)b * a.y_bs + (long)(row0 >> 2) * a.y_ps
                    : a.skip + (long)b * a.s_bs + (long)((row0 - a.y_rows) >> 2) * a.T * 4;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rrecs, 0x00020000);
                const int voff = (t0 + seg * 64 + lane) * 16;
                DR_CHECK_LDS(Rs + pl * BN + seg * 64 + lane, xs1, rs1, 102);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(Rs + pl * BN + seg * 64), 16, voff, 0, 0, 0);
            }
        }
#if DR_ABLATE != 9
        issue(c0);
#endif
        for (int chunk = c0; chunk < c1; ++chunk) {
            // hand-over #chunk: this wave's DMA of tile #chunk must have LANDED before the barrier releases the
            // consumers - barriers do not drain VMEM, and hipcc does not reliably insert the wait for a
            // __syncthreads() behind LDS-DMA builtins (it did in the stand-alone kernels and did NOT in the fused
            // ones: tools/isa_audit.py; the consumers then read the previous occupant of the buffer whenever the tile
            // was slower than their own first weight fragments - observed with cross-XCD hand-offs).  Hence explicit.
            // The consumers' matching barrier opens their chunk; only then may the OTHER buffer be refilled (the
            // consumers finished reading it before they arrived here).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#if DR_ABLATE != 9
            if (chunk + 1 < c1) issue(chunk + 1);
#endif
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wr = wave / WNC, wc = wave % WNC;
    const int r = lane & 31, hi = lane >> 5;
    // this lane's A fragments inside a slab: fp32 [g][hi][row][4] (16 KiB); S3 [g16][piece][kq][row][8 bf16] (24 KiB)

    f32x16 acc[MI][NW];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // Epilogue operands are NOT prefetched into VGPRs: bias / d2 / conditioner are L2-resident and are
    // loaded as unconditional batches at the start of the epilogue (a conditional load there compiles to
    // a branch + s_waitcnt vmcnt(0) per quad), the EPI_RES_SKIP read-modify-write tile waits in LDS (Rs).
    float4 eop[MI][NW][4];

    if constexpr (PREC == 1) {
        // A fragments through buffer loads with scalar per-step offsets (see the fp32 path): slab = 24 KiB,
        // [g16 2][piece 3][kq 2][row 128][8 bf16]
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.Wp + (long)mt * NS * 6144), 0, (unsigned)NS * 24576u, 0x00020000);
        const int wvo = (hi * 128 + wr * WROWS + r) * 16;
        static_assert(MI == 1, "one 32-row MFMA tile per consumer wave");
        auto load_a3 = [&](int slab) -> A12 {
            A12 o;
#pragma unroll
            for (int gp = 0; gp < 6; ++gp) {
#if DR_ABLATE == 2          // measurement build: always the same slab (L1-hot A loads)
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, gp * 4096, 0);
#else
                DR_CHECK(slab >= 0 && wvo + slab * 24576 + gp * 4096 + 16 <= NS * 24576, 103, slab, NS);
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, slab * 24576 + gp * 4096, 0);
#endif
                o.v[gp * 2] = make_uint4(u.x, u.y, u.z, u.w);
            }
            return o;
        };
        A12 wA = load_a3(c0 * KS * a.taps), wB;
#if DR_ABLATE == 1
        wB = wA;
#endif
        const int cen = (a.taps - 1) >> 1;
        const int per_chunk = a.taps * KS;
        const uint4* Xs3 = reinterpret_cast<const uint4*>(Xs);
        // B fragments of one 16-channel group: 3 pieces x NW 32-frame tiles
        struct BF3 { uint4 v[3][NW]; };
        BF3 b0, b1;
        auto xaddr = [&](int chunk, int q) -> const uint4* {
            const int j = q / KS, sub = q - j * KS;
            return Xs3 + (((chunk - c0) & 1) * XP + sub * 12 + hi) * FW + halo + (j - cen) * a.dil + wc * WFR + r;
        };
        auto rd3 = [&](const uint4* Xb, int g) -> BF3 {
            BF3 o;
#pragma unroll
            for (int pz = 0; pz < 3; ++pz)
#pragma unroll
                for (int ni = 0; ni < NW; ++ni) {
                    DR_CHECK_LDS(Xb + (g * 6 + pz * 2) * FW + ni * 32, xs0, xs1, 104);
                    o.v[pz][ni] = Xb[(g * 6 + pz * 2) * FW + ni * 32];
                }
            return o;
        };
        // six piece products per accumulator, smallest terms first; consecutive MFMAs go to different
        // accumulators (the pinned schedule keeps program order: no dependent back-to-back pairs)
        auto mma6 = [&](const uint4 a0, const uint4 a1, const uint4 a2, const BF3& bf) {
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a2, bf.v[0][ni], acc[0][ni]);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a0, bf.v[2][ni], acc[0][ni]);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a1, bf.v[1][ni], acc[0][ni]);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a1, bf.v[0][ni], acc[0][ni]);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a0, bf.v[1][ni], acc[0][ni]);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a0, bf.v[0][ni], acc[0][ni]);
        };
        // One K step (32 channels x 1 tap): 2 groups x 6 piece products x NW tiles = 12*NW MFMAs per wave.
        // A 32x32x16 bf16 MFMA is 32 cycles, so a step is only 384*NW cycles: every non-MFMA instruction
        // that is not issued in the shadow of a running MFMA shows.  Pipeline (pinned): the reads of group 1
        // are interleaved 1:2 with group 0's MFMAs, the reads of the NEXT step's group 0 with group 1's;
        // only the chunk's first step reads its own group 0 (after the hand-over barrier).
        auto step = [&](auto ROLE, int slab, int chunk, int q) {
            constexpr bool kB = decltype(ROLE)::value;
            const uint4* Xb = xaddr(chunk, q);
#if DR_ABLATE != 1          // measurement build 1: no A loads at all
            if constexpr (kB) wA = load_a3(min(slab + 1, NS - 1));
            else wB = load_a3(min(slab + 1, NS - 1));
#endif
            b1 = rd3(Xb, 1);
            mma6(kB ? wB.v[0] : wA.v[0], kB ? wB.v[2] : wA.v[2], kB ? wB.v[4] : wA.v[4], b0);
            b0 = rd3(xaddr(chunk, min(q + 1, per_chunk - 1)), 0);
            mma6(kB ? wB.v[6] : wA.v[6], kB ? wB.v[8] : wA.v[8], kB ? wB.v[10] : wA.v[10], b1);
#if DR_ABLATE == 1
            sgb_mix<3 * NW, 0>();
#else
            sgb_mix<3 * NW, 6>();           // + the 6 A-fragment loads of the next step, one per MFMA pair
#endif
            sgb_mix<3 * NW, 0>();
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        for (int chunk = c0; chunk < c1; ++chunk) {
            auto at = [&](auto R, int q) {
                const int j = q / KS, sub = q - j * KS;
                step(R, (chunk * KS + sub) * a.taps + j, chunk, q);
            };
            __syncthreads();
            b0 = rd3(xaddr(chunk, 0), 0);
            int q = 0;
            for (; q + 2 <= per_chunk; q += 2) {
                at(F_{}, q);
                at(T_{}, q + 1);
            }
            if (q < per_chunk) {
                at(F_{}, q);
                wA = wB;
            }
        }
    } else {
    // A fragments through buffer loads: the M tile's weight panel is the resource (SGPRs), the per-lane
    // byte offset is fixed for the whole kernel and the per-step offset is scalar - no vector address
    // arithmetic in the K loop (with flat 64-bit addresses it was ~12 exposed VALU instructions per step).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Wp + (long)mt * NS * 4096), 0, (unsigned)NS * 16384u, 0x00020000);
    const int wvo = (hi * 128 + wr * WROWS + r) * 16;
    auto load_a = [&](int slab) -> A8 {
        A8 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            DR_CHECK(slab >= 0 && wvo + slab * 16384 + g * 4096 + 16 <= NS * 16384, 105, slab, NS);
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, slab * 16384 + g * 4096, 0);
            o.v[g * 2] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
        return o;
    };
    static_assert(MI == 1, "one 32-row MFMA tile per consumer wave");

    A8 wA = load_a(c0 * KS * a.taps), wB;
    const int cen = (a.taps - 1) >> 1;

    // B fragments of one 8-channel group: NW float4 (one per 32-frame MFMA tile), conflict-free ds_read_b128
    struct BF { float4 v[NW]; };
    BF b0, b1;                                   // groups 0/2 and 1/3 of the step in flight
    auto xaddr = [&](int chunk, int q) -> const float4* {     // X tile address of step q of a chunk
        const int j = q / KS, sub = q - j * KS;
        return Xs + (((chunk - c0) & 1) * XP + sub * 8 + hi) * FW + halo + (j - cen) * a.dil + wc * WFR + r;
    };
    auto rd = [&](const float4* Xb, int g) -> BF {
        BF o;
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            DR_CHECK_LDS(Xb + g * 2 * FW + ni * 32, xs0, xs1, 106);
            o.v[ni] = Xb[g * 2 * FW + ni * 32];
        }
        return o;
    };
    // 8 channels of K for every frame tile: consecutive MFMAs go to different accumulators (the pinned schedule
    // keeps program order inside its small groups: a chain of dependent back-to-back MFMAs costs ~8 cycles each)
    auto mma4 = [&](const float4 af, const BF& bf) {
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.v[ni].x, acc[0][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.v[ni].y, acc[0][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.v[ni].z, acc[0][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.v[ni].w, acc[0][ni], 0, 0, 0);
    };

    // One K step (32 channels x 1 tap): 16*NW MFMAs per wave in 4 groups of 8 channels.  Software pipeline,
    // pinned with sched_group_barrier: the fragment reads of group g+1 are issued before group g's MFMAs,
    // and the reads of the NEXT step's group 0 before this step's group 3 - so inside a chunk no MFMA ever
    // waits for LDS; only the chunk's first step reads its own group 0, right after the hand-over barrier
    // (the last step of a chunk prefetches a valid but unused address: the next tile is not staged yet).
    // ROLE (compile time) selects which of the two A-fragment register sets is consumed; the other receives
    // the next step's fragments (prefetch distance one step), requested at the top of the step.  Roles
    // alternate statically: no per-step register copies.
    const int per_chunk = a.taps * KS;
    auto step = [&](auto ROLE, int slab, int chunk, int q) {
        constexpr bool kB = decltype(ROLE)::value;
        const float4* Xb = xaddr(chunk, q);
        if constexpr (kB) wA = load_a(min(slab + 1, NS - 1));
        else wB = load_a(min(slab + 1, NS - 1));
        b1 = rd(Xb, 1);
        mma4(kB ? wB.v[0] : wA.v[0], b0);
        b0 = rd(Xb, 2);
        mma4(kB ? wB.v[2] : wA.v[2], b1);
        b1 = rd(Xb, 3);
        mma4(kB ? wB.v[4] : wA.v[4], b0);
        b0 = rd(xaddr(chunk, min(q + 1, per_chunk - 1)), 0);
        mma4(kB ? wB.v[6] : wA.v[6], b1);
        // pinned schedule: the 4 A-fragment loads ride inside group 0's MFMAs (one per NW MFMAs: issued in the
        // shadow of a running MFMA instead of as a burst with the matrix pipe idle; hipcc on its own sinks them
        // to their first use and exposes the whole L2 latency once per step)
        sgb<0x100, NW>(); sgb_spread<4, NW>();
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>();
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>();
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // Steps of a chunk, q = 0 .. per_chunk-1, in memory order of the slabs ([32-channel kchunk][tap]):
    // tap-major, sub-chunk minor.  Roles alternate A,B,A,...; a chunk always starts in role A (one
    // register copy per chunk when per_chunk is odd).
    for (int chunk = c0; chunk < c1; ++chunk) {
        auto at = [&](auto R, int q) {
            const int j = q / KS, sub = q - j * KS;
            step(R, (chunk * KS + sub) * a.taps + j, chunk, q);
        };
        __syncthreads();   // X tile #chunk staged by the producers (matches their hand-over barrier)
        if (a.dbg && blockIdx.x == 0 && tid == 0 && chunk < 14) a.dbg[2 + chunk] = clock64() - tick0;
        b0 = rd(xaddr(chunk, 0), 0);
        int q = 0;
        for (; q + 2 <= per_chunk; q += 2) {
            at(F_{}, q);
            at(T_{}, q + 1);
        }
        if (q < per_chunk) {
            at(F_{}, q);
            wA = wB;
        }
    }

    }

    const long long tick1 = a.dbg ? clock64() : 0;
    // ----------------------------------------------------------------------------------------
    // split-K reduction, per consumer wave (no block-level synchronisation): every wave parks its partial
    // accumulators in the workspace (fragment order: 1 KiB per store instruction), then takes a ticket on
    // the counter of its (tile, wave) region.  The wave that draws the last ticket re-reads ALL ksplit
    // partials in split order - so the sum does not depend on arrival order: results are deterministic -
    // and runs the epilogue; the others are done.  It also re-arms the counter for the next launch.
    // The XCDs' L2s are not coherent with each other, so the partials travel with agent-coherent cache
    // policy (sc0 sc1: write-through stores, L2-bypassing loads) and the ordering is
    // stores -> s_waitcnt vmcnt(0) -> ticket (agent-scope atomic) -> loads; a full __threadfence() here
    // costs an L2-wide write-back + invalidate per wave (measured: 25 us per launch).
    // ----------------------------------------------------------------------------------------
    if (a.ksplit > 1) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int WQ = NW * 4;                                   // float4 per lane per wave region
        constexpr int WSCOH = 17;                                    // buffer cache policy: sc0 | sc1
        const int tile = nt * a.MT + mt;
        const __amdgpu_buffer_rsrc_t wsr =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, 0, (unsigned)(a.ws_floats * 4), 0x00020000);
        const int sstride = 4 * WQ * 64 * 16;                        // bytes between consecutive splits
        const int base = ((tile * a.ksplit * 4 + wave) * (WQ * 64) + lane) * 16;
#pragma unroll
        for (int ni = 0; ni < NW; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = {__float_as_uint(acc[0][ni][4 * q]), __float_as_uint(acc[0][ni][4 * q + 1]),
                                 __float_as_uint(acc[0][ni][4 * q + 2]), __float_as_uint(acc[0][ni][4 * q + 3])};
                DR_CHECK((size_t)(base + ks * sstride + (ni * 4 + q) * 1024) + 16 <= a.ws_floats * 4, 107, base + ks * sstride, a.ws_floats);
                __builtin_amdgcn_raw_buffer_store_b128(v, wsr, base + ks * sstride + (ni * 4 + q) * 1024, 0, WSCOH);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // partials are out before the ticket is drawn
        unsigned ticket = 0;
        DR_CHECK((size_t)(tile * 4 + wave) < a.ws_cnt_n, 108, tile * 4 + wave, a.ws_cnt_n);
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.ws_cnt + tile * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != (unsigned)(a.ksplit - 1)) return;
        if (lane == 0) __hip_atomic_store(a.ws_cnt + tile * 4 + wave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ni = 0; ni < NW; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][ni][e] = 0.f;
        // U splits in flight per wait (the loads are pure latency: ~1.5 us each when taken one by one);
        // the adds stay in split order whatever U is
        auto reduce = [&](auto UU, int sp0) {
            constexpr int U = decltype(UU)::value;
            u32x4 v[U][WQ];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < WQ; ++i) {
                    DR_CHECK((size_t)(base + (sp0 + u) * sstride + i * 1024) + 16 <= a.ws_floats * 4, 109, base + (sp0 + u) * sstride, a.ws_floats);
                    v[u][i] = __builtin_amdgcn_raw_buffer_load_b128(wsr, base + (sp0 + u) * sstride + i * 1024, 0, WSCOH);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int ni = 0; ni < NW; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[0][ni][4 * q + e] += __uint_as_float(v[u][ni * 4 + q][e]);
        };
        constexpr int UMAX = (NW == 2) ? 4 : 2;
        if (a.ksplit % UMAX == 0) {
            for (int sp = 0; sp < a.ksplit; sp += UMAX) reduce(std::integral_constant<int, UMAX>{}, sp);
        } else {
            for (int sp = 0; sp < a.ksplit; sp += 2) reduce(std::integral_constant<int, 2>{}, sp);
        }
    }
    // ----------------------------------------------------------------------------------------
    // epilogue.  C/D fragment of 32x32: column = lane&31 (frame), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // => per register quad q a lane owns 4 consecutive rows 8q+4hi..+3 = one float4 of the P4 layout.
    // ----------------------------------------------------------------------------------------
    auto f4arr = [](const float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; };
    // Dual-output mode (EPI_GATE, a.dual = B > 0): classifier-free guidance feeds the SAME x_t to the conditional
    // and the unconditional evaluation, so in the first residual layer the dilated conv of sample b and of
    // sample b + B is the same contraction - it is done once, and the epilogue runs twice (conditioner of
    // sample b / constant unconditional bias) writing both samples' gated outputs.
    const int npass = (EPI == EPI_GATE && a.dual > 0) ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int be = b + pass * a.dual;
        float4 ebias[MI][4];                        // [mi][q]
        float4 ed2[MI][4];                          // second-output offset (step embedding of the next conv)
        {
            const float* bsrc = a.bias;
            if constexpr (EPI == EPI_GATE) bsrc = (be < a.n_cond) ? a.bias : a.bias2;
    #pragma unroll
            for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                for (int q = 0; q < 4; ++q)
                    ebias[mi][q] = *reinterpret_cast<const float4*>(bsrc + mt * 128 + wr * WROWS + mi * 32 + 8 * q + 4 * hi);
            if constexpr (EPI == EPI_RELU || EPI == EPI_RES_SKIP) {
    #pragma unroll
                for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // residual rows only exist below y_rows; the clamp keeps the (unused) skip-row loads in range
                        const int p0 = min(mt * 128 + wr * WROWS + mi * 32 + 8 * q + 4 * hi, a.y_rows - 4);
                        ed2[mi][q] = *reinterpret_cast<const float4*>(a.d2 + (a.tsel ? (long)a.tsel[be] * a.d2_ts : 0) + p0);
                    }
            }
        }
    #pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            const int t = t0 + wc * WFR + ni * 32 + r;
            if constexpr (EPI == EPI_RES_SKIP) {
    #pragma unroll
                for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        DR_CHECK_LDS(Rs + (wr * (WROWS / 4) + mi * 8 + 2 * q + hi) * BN + wc * WFR + ni * 32 + r, xs1, rs1, 110);
                        eop[mi][ni][q] = Rs[(wr * (WROWS / 4) + mi * 8 + 2 * q + hi) * BN + wc * WFR + ni * 32 + r];
                    }
            }
            if constexpr (EPI == EPI_GATE) {
                // conditioner quads of this frame column: one unconditional batch (unconditional samples read
                // sample 0's tensor - valid memory - and ignore it)
                const int tc = min(t, a.T - 1);
                // samples >= n_cond: the shared learned unconditional conditioner (condition='trainable_spec') when
                // there is one, else sample 0's tensor as a readable dummy
                const float* cb = ((be < a.n_cond || !a.cond2) ? a.cond + (long)(be < a.n_cond ? be : 0) * a.c_bs : a.cond2) + (long)tc * 4;
    #pragma unroll
                for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int p0 = mt * 128 + wr * WROWS + mi * 32 + 8 * q + 4 * hi;
                        eop[mi][ni][q] = *reinterpret_cast<const float4*>(cb + (long)(p0 >> 2) * a.T * 4);
                    }
            }
            if (t >= a.T) continue;
    #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rq = 8 * q + 4 * hi;   // row offset inside a 32-row MFMA tile
                if constexpr (PAIRED) {
                    if (q >= 2) continue;                     // quads 0,1 = gate / cos rows, quads 2,3 = their partners
                    const int c0 = mt * 64 + wr * 16 + rq;    // output channel of the quad (rq = 8q + 4hi < 16)
                    if (c0 >= a.y_rows) continue;
                    float v0[4], v1[4], o[4];
    #pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = acc[0][ni][4 * q + e]; v1[e] = acc[0][ni][4 * (q + 2) + e]; }
                    if constexpr (EPI == EPI_GATE) {
                        // y = conv + b_conv + (Wc spec + bc)   [model/diffwave.py:143-144]; unconditional samples
                        // carry the constant conditioner inside bias2
                        float b0[4], b1[4], c0v[4], c1v[4];
                        f4arr(ebias[0][q], b0); f4arr(ebias[0][q + 2], b1);
                        f4arr(eop[0][ni][q], c0v); f4arr(eop[0][ni][q + 2], c1v);
                        const bool has_c = be < a.n_cond || a.cond2 != nullptr;
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a0 = has_c ? b0[e] + c0v[e] : b0[e];
                            const float a1 = has_c ? b1[e] + c1v[e] : b1[e];
                            o[e] = gatef_(v0[e] + a0, v1[e] + a1);   // gate = first half, filter = second (:146-147)
                        }
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = v0[e] * v0[e] + v1[e] * v1[e];
                    }
                    if (a.out_s3 & 1) {   // g for the split-bf16 1x1 kernel
                        store_s3_quad(a.Y + (long)be * a.y_bs, o, c0, t, a.T, a.y_rows >> 3);
                    } else {
                        float* dst = a.Y + (long)be * a.y_bs + (long)(c0 >> 2) * a.y_ps + (long)t * a.y_fs;
                        store_f4<COH>(dst, make_float4(o[0], o[1], o[2], o[3]), a.wt_store);
                    }
                } else {
    #pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int p0 = mt * 128 + wr * WROWS + mi * 32 + rq;
                        float v[4], bb[4], o[4];
    #pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * q + e];
                        f4arr(ebias[mi][q], bb);
                        if constexpr (EPI == EPI_RES_SKIP) {
                            float pv[4];
                            f4arr(eop[mi][ni][q], pv);
                            // packed rows [0, y_rows) are the residual half, [y_rows, 2*y_rows) the skip half
                            // (y_rows is a multiple of 64, so the branch is wave-uniform)
                            if (p0 < a.y_rows) {   // h = (h + (acc + be)) / sqrt(2), in place (:151)
                                float* dst = a.Y + (long)be * a.y_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
    #pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = div_sqrt2(pv[e] + (v[e] + bb[e]));
                                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                                if (a.Y2) {        // hd = h + d_{l+1}: the next dilated conv's input (:139)
                                    float dd[4];
                                    f4arr(ed2[mi][q], dd);
                                    const float o2[4] = {o[0] + dd[0], o[1] + dd[1], o[2] + dd[2], o[3] + dd[3]};
                                    if (a.out_s3 & 2) {
                                        store_s3_quad(a.Y2 + (long)be * a.y2_bs, o2, p0, t, a.T, a.y_rows >> 3);
                                    } else {
                                        float* dst2 = a.Y2 + (long)be * a.y2_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                                        *reinterpret_cast<float4*>(dst2) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                                    }
                                }
                            } else {               // skip (+)= acc + be (:680)
                                float* dst = a.skip + (long)be * a.s_bs + ((long)((p0 - a.y_rows) >> 2) * a.T + t) * 4;
    #pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = a.skip_init ? v[e] + bb[e] : (v[e] + bb[e]) + pv[e];
                                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                            }
                        } else {
                            if (p0 >= a.y_rows) continue;
    #pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if constexpr (EPI == EPI_PLAIN) o[e] = a.alpha * v[e] + bb[e];
                                else if constexpr (EPI == EPI_RELU) o[e] = fmaxf(a.alpha * v[e] + bb[e], 0.f);
                                else if constexpr (EPI == EPI_SILU) { const float z = v[e] + bb[e]; o[e] = z * sigmoidf_(z); }
                                else o[e] = logf(v[e] + 1e-6f);
                            }
                            float* dst = a.Y + (long)be * a.y_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                            if constexpr (EPI == EPI_RELU) {
                                if (a.Y2) {    // hd = h + d_0 for the first dilated conv
                                    float dd[4];
                                    f4arr(ed2[mi][q], dd);
                                    const float o2[4] = {o[0] + dd[0], o[1] + dd[1], o[2] + dd[2], o[3] + dd[3]};
                                    if (a.out_s3 & 2) {
                                        store_s3_quad(a.Y2 + (long)be * a.y2_bs, o2, p0, t, a.T, a.y_rows >> 3);
                                    } else {
                                        float* dst2 = a.Y2 + (long)be * a.y2_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                                        *reinterpret_cast<float4*>(dst2) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (a.dbg && blockIdx.x == 0 && tid == 0) {
        a.dbg[0] = tick1 - tick0;           // main loop
        a.dbg[1] = clock64() - tick0;       // whole block
    }
}

template <int NI, int KS, int EPI, int PREC>
__global__ __launch_bounds__(512) void gemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // blockIdx.x % MT = M tile: with MT == 8 each XCD (block b runs on XCD b % 8) streams exactly
    // one 128-row weight panel, which then stays resident in that XCD's private L2.
    // xcd_n != 0 (X-heavy 1x1 GEMMs: small weights, big activations): the 8 M tiles of one frame tile
    // run on the SAME XCD instead, so the X tile is fetched from HBM once per XCD and hits L2 for the
    // other M tiles, while the (small) weight matrix is L2-resident in every XCD.
    // Split-K (ksplit > 1, under-filled launches only: few samples / narrow GEMMs): ksplit blocks share
    // one output tile, each contracts a contiguous range of the K chunks; see the reduction in gemm_body.
    int mt, nt, ks;
    if (a.xcd_n) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % a.MT;
        const int rest = idx / a.MT;
        ks = rest % a.ksplit;
        nt = (rest / a.ksplit) * 8 + xcd;
    } else {
        mt = blockIdx.x % a.MT;
        const int rest = blockIdx.x / a.MT;
        ks = rest % a.ksplit;
        nt = rest / a.ksplit;
    }
    gemm_body<NI, KS, EPI, PREC, 0>(a, smem, mt, nt, ks);
}

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1) GEMM with BOTH operands straight from L2: the 1x1 output projection + residual / skip
// update (model/diffwave.py:149-151, :680).  A 1x1 GEMM has no tap reuse, so staging X through LDS buys
// nothing - and costs a hand-over barrier per chunk plus the LDS-DMA traffic, during which the consumers of
// gemm_kernel were measured to run ~30 % slower (87 vs 66.7 ticks per MFMA).  Here a block is just four
// consumer waves, 4 (M) x 1 (N), wave tile 32 rows x 32*NW frames: per 32-channel K step a wave loads its
// 4 A fragments (packed weights, fragment order) and 4*NW B fragments (P4 activations: one float4 = the 4
// K values of a lane for 4 consecutive MFMAs) one step ahead into the alternate register set, then
// issues 16*NW MFMAs.  No LDS, no barriers, no producers.  Frames beyond T are clamped (their columns are
// never written).
// ---------------------------------------------------------------------------------------------
//   COH = 1 (fused residual-stack kernel): the second output Y2 (hd, read by other workgroups of the same launch)
//   is stored write-through (sc1) unless a.wt_store == 0; X (g, written by other workgroups) is read with plain
//   loads AFTER an agent-scope acquire in the preceding group barrier; the read-modify-write tiles h and skip
//   belong to this workgroup alone for the whole launch.
//   Run by waves 0-3 of the block (wave = wave index); mt / nt = M tile and frame tile.
//   RLDS = 1 (fused kernel; Rs = the tile): the block's read-modify-write tile (its 128 rows of h / skip x BN frames,
//   [32 planes][BN] float4) is RESIDENT IN LDS for the whole launch instead of being re-read from and re-written
//   to global memory by every layer: the epilogue reads and updates it there (no operand registers are held
//   across the last K steps - with them the fused kernel spilled), only hd goes to global.
//   TBN / foff (fused kernel, 160-frame tiles): the block's frame tile is TBN frames wide and this call covers its
//   32*NW frames starting at foff - the 160-frame 1x1 runs as a 96-frame and a 64-frame pass, because one pass
//   with 5 frame tiles per wave does not fit the fused kernel's register budget without spilling in the K loop.
//   EPI (default EPI_RES_SKIP; the tail kernel also runs EPI_RELU / EPI_PLAIN): the plain epilogues y = [relu](alpha acc
//   + bias) of the skip / output projection (model/diffwave.py:682-685) with the output strides of GemmArgs (the
//   output projection writes the (B, T, 88) roll layout); their output is stored write-through under COH.
template <int NW, int COH, int RLDS, int TBN = 32 * NW, int EPI = EPI_RES_SKIP>     // 32-frame MFMA tiles per wave: 128 rows x 32*NW frames per call
DR_DEVINL void pw_body(const GemmArgs& a, const int mt, const int nt, const int wave, float4* Rs = nullptr, const int foff = 0) {
    static_assert(EPI == EPI_RES_SKIP || EPI == EPI_RELU || EPI == EPI_PLAIN, "pw_body epilogues");
    static_assert(EPI == EPI_RES_SKIP || !RLDS, "the LDS-resident tile belongs to the residual / skip epilogue");
    constexpr int BN = TBN;
    // X loads are PLAIN also in the fused kernel: the four waves of a block read the same B fragments, and only
    // the CU's L1 turns that into one L2 request instead of four (measured: with L1-bypassing sc1 loads the phase
    // ran 2x slower) - the fused kernel therefore invalidates the L1 once, in the barrier before this phase.
    constexpr int XAUX = 0;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, hi = lane >> 5;
    const long long tick0 = a.dbg ? clock64() : 0;

    const int tps = (a.T + BN - 1) / BN;
    const int b = nt / tps;
    const int t0 = (nt % tps) * BN + foff;
    const int NS = a.kchunks;

    // Both operands through buffer loads: resource = (this M tile's weight panel | this sample's X tensor) in
    // SGPRs, per-lane byte offset fixed for the whole kernel (1 + NW VGPRs), per-step offset scalar - so a
    // K step costs no vector address arithmetic (flat 64-bit addressing cost ~70 VALU ops per step, i.e.
    // ~12 of 76 ticks per MFMA, all exposed: nothing else runs on the SIMD).  Out-of-range reads (K padding
    // planes) return 0 from the bounds check; frames past T read the next plane's data (finite, never used).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Wp + (long)mt * NS * 4096), 0, (unsigned)NS * 16384u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.X + (long)b * a.x_bs), 0, (unsigned)(a.x_planes * a.x_ps * 4), 0x00020000);
    const int wvo = (hi * 128 + wave * 32 + r) * 16;
    const int xps = (int)a.x_ps * 4;                          // bytes per plane
    int xvo[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) xvo[ni] = hi * xps + min(t0 + ni * 32 + r, a.T - 1) * 16;

    struct AF { float4 v[4]; };
    struct BF { float4 v[4][NW]; };
    auto asf4 = [](const u32x4 u) { return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)); };
    auto load_a = [&](int slab) -> AF {
        AF o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            DR_CHECK(slab >= 0 && wvo + slab * 16384 + g * 4096 + 16 <= NS * 16384, 120, slab, NS);
            o.v[g] = asf4(__builtin_amdgcn_raw_buffer_load_b128(wr, wvo, slab * 16384 + g * 4096, 0));
        }
        return o;
    };
    auto load_b = [&](int slab) -> BF {
        BF o;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni)
                o.v[g][ni] = asf4(__builtin_amdgcn_raw_buffer_load_b128(xr, xvo[ni], (slab * 8 + g * 2) * xps, XAUX));
        return o;
    };

    f32x16 acc[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;

    AF aA = load_a(0), aB;
    BF bA = load_b(0), bB;
    auto step = [&](auto ROLE, int slab) {
        constexpr bool kB = decltype(ROLE)::value;
        const int nxt = min(slab + 1, NS - 1);
        if constexpr (kB) { aA = load_a(nxt); bA = load_b(nxt); }
        else { aB = load_a(nxt); bB = load_b(nxt); }
        // consecutive MFMAs go to different accumulators (the pinned schedule keeps program order)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 af = kB ? aB.v[g] : aA.v[g];
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, (kB ? bB.v[g][ni] : bA.v[g][ni]).x, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, (kB ? bB.v[g][ni] : bA.v[g][ni]).y, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, (kB ? bB.v[g][ni] : bA.v[g][ni]).z, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, (kB ? bB.v[g][ni] : bA.v[g][ni]).w, acc[ni], 0, 0, 0);
        }
        // pinned schedule: the 4 + 4*NW prefetch loads are spread through the MFMA stream (one per PER MFMAs),
        // so each is issued in the shadow of a running MFMA; issued as one burst at the top of the step they
        // cost 16 TA cycles each with the matrix pipe idle (71.8 vs 67.6 ticks per MFMA)
        constexpr int NLD = 4 + 4 * NW, PER = (16 * NW) / NLD;
        sgb_spread<NLD, PER>();
        sgb<0x8, 16 * NW - NLD * PER>();
    };
    // epilogue operands (EPI_RES_SKIP): rows [0, y_rows) h = (h + acc + b) / sqrt(2) in place (+ hd = h + d_next),
    // rows [y_rows, 2 y_rows) skip (+)= acc + b.  The read-modify-write tile comes straight from global and is
    // requested BEFORE the last two K steps, so its latency hides behind their MFMAs.
    auto f4arr = [](const float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; };
    const int rowb = mt * 128 + wave * 32 + 4 * hi;           // + 8q
    const bool res_rows = rowb < a.y_rows;                     // wave-uniform (y_rows is a multiple of 64)
    float4 ebias[4], ed2[4], eop[RLDS ? 1 : NW][4];
    auto load_epilogue = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ebias[q] = *reinterpret_cast<const float4*>(a.bias + rowb + 8 * q);
            if constexpr (EPI == EPI_RES_SKIP)
                ed2[q] = *reinterpret_cast<const float4*>(a.d2 + (a.tsel ? (long)a.tsel[b] * a.d2_ts : 0) + min(rowb + 8 * q, a.y_rows - 4));
        }
        if constexpr (RLDS || EPI != EPI_RES_SKIP) return;              // the tile waits in LDS / there is none
        const float* base = res_rows ? a.Y + (long)b * a.y_bs + (long)(rowb >> 2) * a.y_ps
                                     : a.skip + (long)b * a.s_bs + (long)((rowb - a.y_rows) >> 2) * a.T * 4;
        const long qs = res_rows ? 2 * a.y_ps : (long)2 * a.T * 4;      // 8 rows = 2 planes further
        const long fs = res_rows ? a.y_fs : 4;
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            const int tc = min(t0 + ni * 32 + r, a.T - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) eop[ni][q] = *reinterpret_cast<const float4*>(base + q * qs + tc * fs);
        }
    };
    int slab = 0;
    const int pre = max(NS - 2, 0) & ~1;                      // even number of steps before the prefetch point
    for (; slab < pre; slab += 2) {
        step(std::false_type{}, slab);
        step(std::true_type{}, slab + 1);
        if (a.dbg && slab == 0 && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[2] = clock64() - tick0;   // first two steps done
    }
    if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[3] = clock64() - tick0;                    // before the RMW-tile request
    // (RLDS: only the bias / step-embedding rows are left to load - 8 L2-hot float4 per lane, requested after the
    // loop: holding them across the last K steps is what pushed the fused kernel over its register budget)
    if constexpr (!RLDS) load_epilogue();
    __builtin_amdgcn_sched_barrier(0);
    for (; slab + 2 <= NS; slab += 2) {
        step(std::false_type{}, slab);
        step(std::true_type{}, slab + 1);
    }
    if (slab < NS) step(std::false_type{}, slab);
    const long long tick1 = a.dbg ? clock64() : 0;
    if constexpr (RLDS) load_epilogue();

#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        const int t = t0 + ni * 32 + r;
        if (t >= a.T) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p0 = rowb + 8 * q;
            float v[4], bb[4], pv[4], o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[ni][4 * q + e];
            if constexpr (EPI != EPI_RES_SKIP) {
                if (p0 >= a.y_rows) continue;
                f4arr(ebias[q], bb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = a.alpha * v[e] + bb[e];
                    if constexpr (EPI == EPI_RELU) o[e] = fmaxf(o[e], 0.f);
                }
                float* dst = a.Y + (long)b * a.y_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                store_f4<COH>(dst, make_float4(o[0], o[1], o[2], o[3]), a.wt_store);
                continue;
            }
            float4* rs = nullptr;                                                   // this quad in the LDS tile (RLDS)
            if constexpr (RLDS) rs = Rs + (wave * 8 + 2 * q + hi) * BN + foff + ni * 32 + r;
#ifdef DR_BOUNDS
            if constexpr (RLDS) DR_CHECK_LDS(rs, lds_off(Rs), lds_off(Rs) + 32u * BN * 16u, 121);
#endif
            f4arr(ebias[q], bb);
            if constexpr (RLDS) f4arr(*rs, pv);
            else f4arr(eop[ni][q], pv);
            if (res_rows) {
                float* dst = a.Y + (long)b * a.y_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = div_sqrt2(pv[e] + (v[e] + bb[e]));
                if constexpr (RLDS) *rs = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                if (a.Y2) {
                    float dd[4];
                    f4arr(ed2[q], dd);
                    const float o2[4] = {o[0] + dd[0], o[1] + dd[1], o[2] + dd[2], o[3] + dd[3]};
                    if (a.out_s3 & 2) {
                        store_s3_quad(a.Y2 + (long)b * a.y2_bs, o2, p0, t, a.T, a.y_rows >> 3);
                    } else {
                        float* dst2 = a.Y2 + (long)b * a.y2_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                        store_f4<COH>(dst2, make_float4(o2[0], o2[1], o2[2], o2[3]), a.wt_store);
                    }
                }
            } else {
                float* dst = a.skip + (long)b * a.s_bs + ((long)((p0 - a.y_rows) >> 2) * a.T + t) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a.skip_init ? v[e] + bb[e] : (v[e] + bb[e]) + pv[e];
                if constexpr (RLDS) *rs = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) {
        a.dbg[0] = tick1 - tick0;
        a.dbg[1] = clock64() - tick0;
    }
}

template <int NW>
__global__ __launch_bounds__(256) void pw_kernel(const GemmArgs a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int mt, nt;
    if (a.xcd_n) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % a.MT;
        nt = (idx / a.MT) * 8 + xcd;
    } else {
        mt = blockIdx.x % a.MT;
        nt = blockIdx.x / a.MT;
    }
    mt += a.mt0;    // launches over a sub-range of the M tiles (the last layer only needs its skip rows)
    pw_body<NW, 0, 0>(a, mt, nt, wave);
}

// ---------------------------------------------------------------------------------------------
// The 1x1 residual / skip GEMM for launches that cannot fill the chip (single clips: 2 evaluations x 125 frames are 32
// of pw_kernel's tiles).  Block = 4 waves on ONE 32-row x 32*NW-frame output tile, the waves splitting K in-block: wave
// w contracts the w-th quarter of the channel slabs (k ascending inside it) with both operands straight from L2, the
// four partial tiles meet in LDS and are added in wave order ((p0 + p1) + p2) + p3 - deterministic, independent of
// timing - and wave w runs the epilogue of register quad w (8 rows x 4... = one float4 of the P4 layout per lane and
// frame tile).  256 blocks at config 1 with a K loop of 64 MFMAs per wave instead of 32 tiles x 8 K slices exchanged
// through a workspace with tickets: 13.9 -> ~8 us per launch.  Same epilogue arithmetic as pw_body (EPI_RES_SKIP).
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void pwk_kernel(const GemmArgs a) {
    __shared__ float4 part[4][NW][4][64];                      // [wave][frame tile][quad][lane]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, hi = lane >> 5;
    constexpr int BN = 32 * NW;
    const int RT = a.MT * 4;                                   // 32-row tiles
    const int rt = blockIdx.x % RT + a.mt0 * 4, nt = blockIdx.x / RT;
    const int mt = rt >> 2, sr = rt & 3;                       // 128-row weight panel, 32-row sub-tile
    const int tps = (a.T + BN - 1) / BN;
    const int b = nt / tps;
    const int t0 = (nt % tps) * BN;
    const int NS = a.kchunks, NSW = NS >> 2;                   // slabs (32 channels) in all / per wave
    const int s0 = wave * NSW;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Wp + (long)mt * NS * 4096), 0, (unsigned)NS * 16384u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.X + (long)b * a.x_bs), 0, (unsigned)(a.x_planes * a.x_ps * 4), 0x00020000);
    const int wvo = (hi * 128 + sr * 32 + r) * 16;
    const int xps = (int)a.x_ps * 4;
    int xvo[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) xvo[ni] = hi * xps + min(t0 + ni * 32 + r, a.T - 1) * 16;
    auto asf4 = [](const u32x4 u) { return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)); };
    struct AF { float4 v[4]; };
    struct BF { float4 v[4][NW]; };
    auto load_a = [&](int slab) -> AF {
        AF o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            DR_CHECK(slab >= 0 && wvo + slab * 16384 + g * 4096 + 16 <= NS * 16384, 140, slab, NS);
            o.v[g] = asf4(__builtin_amdgcn_raw_buffer_load_b128(wr, wvo, slab * 16384 + g * 4096, 0));
        }
        return o;
    };
    auto load_b = [&](int slab) -> BF {
        BF o;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni)
                o.v[g][ni] = asf4(__builtin_amdgcn_raw_buffer_load_b128(xr, xvo[ni], (slab * 8 + g * 2) * xps, 0));
        return o;
    };
    // the epilogue operands of THIS wave's quad (q = wave): requested first, their latency hides behind the K loop
    const int p0 = mt * 128 + sr * 32 + 8 * wave + 4 * hi;     // first of the lane's 4 packed rows
    const bool res_rows = p0 < a.y_rows;                       // wave-uniform (y_rows is a multiple of 64)
    const float4 ebias = *reinterpret_cast<const float4*>(a.bias + p0);
    const float4 ed2 = *reinterpret_cast<const float4*>(a.d2 + (a.tsel ? (long)a.tsel[b] * a.d2_ts : 0) + min(p0, a.y_rows - 4));
    float4 eop[NW];
    {
        const float* base = res_rows ? a.Y + (long)b * a.y_bs + (long)(p0 >> 2) * a.y_ps
                                     : a.skip + (long)b * a.s_bs + (long)((p0 - a.y_rows) >> 2) * a.T * 4;
        const long fs = res_rows ? a.y_fs : 4;
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) eop[ni] = *reinterpret_cast<const float4*>(base + (long)min(t0 + ni * 32 + r, a.T - 1) * fs);
    }

    f32x16 acc[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
    AF aA = load_a(s0), aB;
    BF bA = load_b(s0), bB;
    auto mma = [&](const AF& af, const BF& bf) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.v[g].x, bf.v[g][ni].x, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.v[g].y, bf.v[g][ni].y, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.v[g].z, bf.v[g][ni].z, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.v[g].w, bf.v[g][ni].w, acc[ni], 0, 0, 0);
        }
    };
    int slab = 0;
    for (; slab + 2 <= NSW; slab += 2) {          // two register sets, prefetch distance one slab
        aB = load_a(s0 + slab + 1); bB = load_b(s0 + slab + 1);
        mma(aA, bA);
        const int nx = s0 + min(slab + 2, NSW - 1);
        aA = load_a(nx); bA = load_b(nx);
        mma(aB, bB);
    }
    if (slab < NSW) mma(aA, bA);

    // the four partial tiles meet in LDS; wave w sums register quad w of every frame tile in wave order
#pragma unroll
    for (int ni = 0; ni < NW; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            part[wave][ni][q][lane] = make_float4(acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]);
    __syncthreads();
    auto f4arr = [](const float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; };
    float bb[4], dd[4];
    f4arr(ebias, bb); f4arr(ed2, dd);
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        const int t = t0 + ni * 32 + r;
        if (t >= a.T) continue;
        float v[4], pv[4], o[4], u[4];
        f4arr(part[0][ni][wave][lane], v);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            f4arr(part[w][ni][wave][lane], u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += u[e];
        }
        f4arr(eop[ni], pv);
        if (res_rows) {          // h = (h + (acc + b)) / sqrt(2) in place, hd = h + d_{l+1}   (model/diffwave.py:151, :139)
            float* dst = a.Y + (long)b * a.y_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = div_sqrt2(pv[e] + (v[e] + bb[e]));
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            if (a.Y2) {
                float* dst2 = a.Y2 + (long)b * a.y2_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                *reinterpret_cast<float4*>(dst2) = make_float4(o[0] + dd[0], o[1] + dd[1], o[2] + dd[2], o[3] + dd[3]);
            }
        } else {                 // skip (+)= acc + b   (model/diffwave.py:680)
            float* dst = a.skip + (long)b * a.s_bs + ((long)((p0 - a.y_rows) >> 2) * a.T + t) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = a.skip_init ? v[e] + bb[e] : (v[e] + bb[e]) + pv[e];
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <int NW>
static hipError_t launch_pwk_t(const GemmArgs& a, hipStream_t s) {
    const int BN = 32 * NW;
    const int NT = a.NB * ((a.T + BN - 1) / BN);
    DR_CHECK_EXTENTS(a, EPI_RES_SKIP, 0, "pwk_kernel");
    hipLaunchKernelGGL((pwk_kernel<NW>), dim3((unsigned)(a.MT * 4 * NT)), dim3(256), 0, s, a);
    return hipGetLastError();
}
// 1x1 EPI_RES_SKIP GEMM of an under-filled launch, fp32: block = 32 rows x 32*NW frames, K split over its 4 waves
hipError_t launch_pointwise_ksplit(const GemmArgs& a, int NW, hipStream_t s) {
    if (a.taps != 1 || a.kchunks < 4 || (a.kchunks & 3) || a.x_fs != 4 || a.out_s3) return hipErrorInvalidValue;
    return NW == 1 ? launch_pwk_t<1>(a, s) : NW == 2 ? launch_pwk_t<2>(a, s) : hipErrorInvalidValue;
}

// Block -> XCD mapping of a per-phase GEMM launch (see gemm_kernel): 0 = one weight panel (M tile) per XCD, 1 = the M
// tiles of a frame tile share an XCD.  Chosen by the bytes each choice pulls through the XCDs' L2s (what the FETCH
// counters see): with mapping 0 every XCD streams ITS panel once (L2-resident if it fits) and all of X; with mapping 1
// every XCD streams its share of X once and ALL panels - once if the whole weight matrix fits its L2, else once per
// round of concurrently resident frame tiles.  (Until round 3 the rule was "xbytes > wbytes", which picked mapping 1
// for 5-round launches of big convs - 640-frame generation batches - and paid 13.9x the algorithmic traffic.)
static int pick_xcd_mapping(int MT, int NT, double wbytes, double xbytes) {
    if (MT <= 1 || NT % 8 != 0) return 0;
    static const int force = getenv("DR_XCD_N") ? atoi(getenv("DR_XCD_N")) : -1;      // tuning experiments
    if (force >= 0) return force;
    static const int model = getenv("DR_XCD_MODEL") ? atoi(getenv("DR_XCD_MODEL")) : 1;
    if (!model) return xbytes > wbytes ? 1 : 0;
    const double l2 = 4.0 * 1024 * 1024, cus_per_xcd = 32.0;
    const double conc = cus_per_xcd / MT < 1.0 ? 1.0 : cus_per_xcd / MT;               // frame tiles resident per XCD (mapping 1)
    const double rounds1 = wbytes <= l2 ? 1.0 : ((NT / 8.0) / conc < 1.0 ? 1.0 : (NT / 8.0) / conc);
    const double cost1 = 8.0 * rounds1 * wbytes + xbytes;
    const double panel = wbytes / MT;
    const double rounds0 = panel <= l2 ? 1.0 : (double)((long)MT * NT + 255) / 256;    // a panel that does not fit is re-streamed per round
    const double cost0 = rounds0 * wbytes + 8.0 * xbytes;
    return cost1 < cost0 ? 1 : 0;
}

template <int NW>
static hipError_t launch_pw_t(const GemmArgs& a, hipStream_t s) {
    const int BN = 32 * NW;
    const int NT = a.NB * ((a.T + BN - 1) / BN);
    GemmArgs b = a;
    const double wbytes = 4.0 * 128.0 * a.MT * 32.0 * a.kchunks, xbytes = 4.0 * (double)NT * BN * 32.0 * a.kchunks;
    b.xcd_n = pick_xcd_mapping(a.MT, NT, wbytes, xbytes);
    DR_CHECK_EXTENTS(b, EPI_RES_SKIP, 0, "pw_kernel");
    hipLaunchKernelGGL((pw_kernel<NW>), dim3((unsigned)(a.MT * NT)), dim3(256), 0, s, b);
    return hipGetLastError();
}
// 1x1 EPI_RES_SKIP GEMM, fp32, operands direct from L2; block = 128 rows x 32*NW frames, NW in {2,3,4,5}
hipError_t launch_pointwise(const GemmArgs& a, int NW, hipStream_t s) {
    if (a.taps != 1 || a.kchunks < 1 || a.x_fs != 4) return hipErrorInvalidValue;
    switch (NW) {
        case 2: return launch_pw_t<2>(a, s);
        case 3: return launch_pw_t<3>(a, s);
        case 4: return launch_pw_t<4>(a, s);
        case 5: return launch_pw_t<5>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Fused residual stack (model/diffwave.py:134-151 x residual_layers, the loop at :678-681): a persistent kernel
// that walks a range of the 2L phases - phase 2l = gemm_body<EPI_GATE> of layer l (dilated conv + conditioner +
// gate -> g), phase 2l+1 = pw_body of layer l (1x1 -> h, hd = h + d_{l+1}, skip) - with every block keeping its
// (M tile, frame tile) for the whole launch.  Same device code, same MFMA order, same epilogue arithmetic as the
// per-phase launches: results are bit-identical to them.
//
// Why it is legal without a grid barrier: a clip evaluation never reads another clip evaluation's activations
// (no cross-sample operation on the path), so only the blocks of ONE sample - its M tiles x its
// frame tiles, the GROUP - exchange data: g (written per M tile, read by every 1x1 block of the group) and hd
// (written by the residual-row blocks, read with its halo by every conv block of the group).  h and skip tiles
// are read-modify-written by the same block in every layer.  Between phases the group meets at a counter.
//
// Hand-off form (MI355X_MICROARCH.md "inter-workgroup visibility", valid under any block -> XCD placement):
// producers store g / hd write-through (sc1) -> every wave drains (s_waitcnt vmcnt(0)) -> __syncthreads() ->
// one lane arrives on the group counter (relaxed, agent scope) and polls it -> __syncthreads() -> consumers read hd
// with sc1 LDS-DMA loads (L1 bypassed) and g with plain loads - through an L1 that a producer wave invalidated
// (agent-scope acquire) during the preceding conv phase, after the CU's last read of the previous g; the XCD's L2 is never left with a stale copy: a write-through
// store drops / invalidates it.  Counters are re-armed by the last block of the group to leave the launch, so a
// replayed graph needs no memset node.  Spins are bounded: a wait that runs into the bound sets *err and
// carries on (wrong data, but no hung queue).
// ---------------------------------------------------------------------------------------------
// L2 warm-up by a wave that has nothing else to do (the conv's producer waves once their last X tile is staged;
// all four of them during a 1x1 phase): touch one dword of every 128-byte line of [base, base + bytes) so that the
// consumers' first fragment / epilogue-operand loads of the NEXT phase hit the XCD's L2 instead of HBM.  part / parts
// split the range over the helper waves.  The loaded values are folded into a register the compiler must
// materialise (the empty asm), nothing else depends on them.
DR_DEVINL void l2_touch(const float* base, const unsigned bytes, const int part, const int parts) {
    if (!base || !bytes) return;
    typedef unsigned u32;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    u32 acc = 0;
#pragma unroll 8
    for (unsigned off = (unsigned)part * 8192u + (unsigned)lane * 128u; off < bytes; off += (unsigned)parts * 8192u)
        acc ^= __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
    asm volatile("" ::"v"(acc));
}

// Spins are bounded.  A wait that runs into the bound (~1 s of polling; a phase lasts < 1 ms) raises BOTH flags -
// *err (host-mapped: the host sees it without a copy, dr_finish / dr_stack_status) and *derr (device memory: what
// the kernels themselves poll) - and carries on with wrong data; every other wait of the launch then gives up
// within ~64 polls (it looks at *derr after 64 polls and every 4096 after), and every later fused launch of the
// engine returns at its first instruction (stack_kernel) until the host has cleared the condition.
template <bool ACQUIRE>
DR_DEVINL void group_barrier(unsigned* ctr, const unsigned target, unsigned* err, unsigned* derr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY wave: its stores (incl. the asm sc1 ones) are out
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            ++spins;
            if (spins > (1u << 20) ||
                ((spins == 64u || (spins & 4095u) == 0) && __hip_atomic_load(derr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(derr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        // ONE agent-scope acquire per block (buffer_inv sc1: drops this CU's L1 lines) after the match, so that the
        // PLAIN loads of the next phase cannot hit a line cached before the producers rewrote it
        if constexpr (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int NJ, int KS, int EPI, int COH>
DR_DEVINL void gemm16_body(const GemmArgs& a, char* smem, const int mt, const int nt);     // defined below

// FL = frame-tile flavour: 1 / 2 = 64 / 128 frames per block on the 32x32x2 MFMA (gemm_body<FL>), 5 = 160 frames on
// the 16x16x4 MFMA (gemm16_body<5>: 640-frame clips fill the chip's 256 CUs exactly with 4 x 160-frame tiles)
template <int FL>
__global__ __launch_bounds__(512) void stack_kernel(const StackArgs s_by_value) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The arguments are read through the kernarg segment pointer (constant address space: scalar loads at the
    // point of use).  Indexing the by-value struct with the run-time layer index made the compiler copy it to
    // scratch, after which every per-layer pointer lived in VGPRs and each buffer load was wrapped in a
    // waterfall loop.
    (void)s_by_value;
    typedef const __attribute__((address_space(4))) StackArgs* KernArgs;
    const KernArgs sp = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) StackArgs& s = *sp;
    constexpr int BN = (FL == 5) ? 160 : 64 * FL;
    constexpr int RWL = (BN + 63) / 64;                    // 64-frame segments of a tile row
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int MT = s.Cp >> 6;
    const int tps = (s.T + BN - 1) / BN;
    const unsigned gsize = (unsigned)(MT * tps);          // blocks per group
    int mt, nt, grp, member;
    if (s.xcd_n) {       // all blocks of a group on one XCD (block b is dispatched to XCD b % 8): the group shares an L2
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        member = idx % (int)gsize;
        grp = (idx / (int)gsize) * 8 + xcd;
    } else {             // one weight panel per XCD, as the per-phase conv launches
        member = blockIdx.x % (int)gsize;
        grp = blockIdx.x / (int)gsize;
    }
    // (a launch whose evaluations are not a multiple of 8 is padded with idle groups - launch_stack - so that the
    // group-per-XCD dealing stays whole: their blocks have nothing to do and touch no counter)
    if (grp >= s.NB) return;
    mt = member % MT;                                     // sample (clip evaluation) grp = barrier group
    nt = grp * tps + member / MT;
    // A time-out of an earlier launch of this engine that the host has not cleared yet (dr_finish / dr_stack_status):
    // do nothing at all - the chain's remaining launches drain in microseconds and the caller re-runs the sample on
    // the per-phase kernels.  (Written by an EARLIER kernel of the stream: visible across the kernel boundary.)
    // (the flag is requested here and tested after the resident tile's loads are issued: its latency hides there)
    const unsigned pending_timeout = __hip_atomic_load(s.derr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DR_CHECK(grp >= 0 && grp < 512 && (long)grp * gsize + member < 1024, 130, grp, member);       // counters / tag words
    unsigned* ctr = s.bar + 4 * grp;                      // {arrivals, departures, generation, -}
    const long act_bs = (long)s.Cp * s.T;
    const int P = s.Cp >> 2;
    unsigned episode = 0;
    // Store mode of the tensors handed to other workgroups (g, hd).  Until the group has PROVED that all its
    // blocks run on one XCD (same L2) they are stored write-through (sc1), which is valid under any placement;
    // each block publishes its XCC id before the first barrier and compares the group's ids after it - when they
    // all agree the remaining phases use plain stores (the lines stay in the shared L2, where the sc1 loads of the
    // consumers find them: ~2x faster 1x1 phases).  Placement is never ASSUMED.
    // The block's read-modify-write tile - its 128 packed rows of the 1x1 output (h rows for the residual M tiles,
    // skip rows for the others) x its BN frames - lives in LDS for the whole launch: [32 planes][BN] float4 behind
    // the conv's X tiles.  Loaded here (LDS-DMA, frames >= T read 0), written back after the last phase.
    float4* Rs = reinterpret_cast<float4*>(smem + s.rs_off);
    const int b_ = nt / tps, t0_ = (nt % tps) * BN;
    auto tile_plane = [&](int pl, bool& is_res) -> float* {     // global address of plane pl (4 rows) of the tile, frame 0
        const int row0 = mt * 128 + pl * 4;
        is_res = row0 < s.Cp;
        return is_res ? s.h + (long)b_ * act_bs + (long)(row0 >> 2) * s.T * 4
                      : s.skip + (long)b_ * act_bs + (long)((row0 - s.Cp) >> 2) * s.T * 4;
    };
    {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const int lane = threadIdx.x & 63;
        for (int i = wave; i < 32 * RWL; i += 8) {
            const int pl = i / RWL, seg = i - pl * RWL;
            bool is_res;
            const float* src = tile_plane(pl, is_res);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)s.T * 16u, 0x00020000);
#ifdef DR_BOUNDS
            if (seg * 64 + lane < BN) DR_CHECK_LDS(Rs + pl * BN + seg * 64 + lane, s.rs_off, s.rs_off + 32 * BN * 16, 131);
            if (threadIdx.x == 0 && i == 0) DR_CHECK(s.rs_off + 32 * BN * 16 + 16 <= s.lds_bytes && lds_off(smem) == 0u, 132, s.rs_off, s.lds_bytes);
#endif
            if (seg * 64 + lane < BN)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(Rs + pl * BN + seg * 64), 16, (t0_ + seg * 64 + lane) * 16, 0, 0, 0);
        }
        if (pending_timeout) return;
        // (the first group barrier - or the end of a one-phase launch - drains these loads: s_waitcnt vmcnt(0)
        // + __syncthreads(); a launch that STARTS with a 1x1 phase waits right here)
        if (s.p0 & 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // The published word carries the group's GENERATION (a counter the last block to leave a launch advances, read
    // here with a returning agent-scope atomic, i.e. at the coherence point): a word left behind by any earlier
    // launch - whatever cache it might be served from - can never compare equal to this launch's, it reads as
    // "not my XCD" and the group keeps the write-through stores that are valid under every placement.
    int wt_store = 1;
    unsigned my_tag = 0;
    unsigned& same_xcd_s = *reinterpret_cast<unsigned*>(Rs + 32 * BN);      // one word behind the resident tile
    if (threadIdx.x == 0) {
        unsigned my_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        const unsigned gen = __hip_atomic_fetch_add(ctr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        my_tag = (gen << 4) | (my_xcc & 0xfu);
        __hip_atomic_store(s.xid + (long)grp * gsize + member, my_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

#pragma unroll 1
    for (int p = s.p0; p < s.p1; ++p) {
        const int l = p >> 1;
        const __attribute__((address_space(4))) StackLayer& ly = sp->layer[l];
        if (s.dbg && blockIdx.x == 0 && threadIdx.x == 0) {
            s.dbg[p - s.p0] = clock64();
            if (p == s.p0) s.dbg[120] = wall_clock64();          // constant 100 MHz: gives the shader clock the launch ran at
        }
        GemmArgs a{};
        a.d2 = s.zero;
        a.lds_bytes = s.lds_bytes;
        a.wt_store = wt_store;
        a.MT = MT; a.NB = s.NB; a.T = s.T; a.alpha = 1.f; a.ksplit = 1;
        a.x_bs = act_bs; a.x_ps = (long)s.T * 4; a.x_fs = 4; a.x_planes = P; a.kchunks = s.Cp >> 5;
        a.y_bs = act_bs; a.y_ps = (long)s.T * 4; a.y_fs = 4; a.y_rows = s.Cp;
        if ((p & 1) == 0) {
            a.Wp = ly.conv_w; a.bias = ly.conv_b; a.bias2 = ly.conv_b2;
            a.X = s.hd; a.taps = s.taps; a.dil = ly.dil;
            a.cond = ly.cond; a.cond2 = ly.cond2; a.c_bs = s.c_bs; a.n_cond = s.n_cond;
            a.Y = s.g;
            if (s.dbg && p + 2 >= s.p1) a.dbg = s.dbg + 64;       // last conv phase: body tick marks of block 0
            if constexpr (FL == 5) gemm16_body<5, 1, EPI_GATE, 1>(a, smem, mt, nt);
            else gemm_body<FL, 1, EPI_GATE, 0, 1>(a, smem, mt, nt, 0);
            // The agent-scope acquire the NEXT phase needs (the 1x1 reads g, written by other workgroups, with plain
            // loads through this CU's L1): one producer wave issues it here, while the consumers still contract the
            // last chunk, instead of everyone waiting ~1.7 us for it behind the barrier.  It is valid anywhere
            // between the previous 1x1 phase's last g load and the next one's first: no wave of this CU reads a g
            // address in a conv phase (weights and the conditioner are read-only, hd comes through L1-bypassing sc1
            // LDS-DMA), so no g line can re-enter the L1 after this invalidate; the conv's own weight-fragment loads
            // are used once each and lose nothing.
            if (wave == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (wave >= 4 && s.warm) {
                // the producers are back while the consumers still contract the last chunk (~16 us at k = 9): warm
                // the L2 with what comes next - this block's conditioner tile (read by the gate epilogue: 32 planes x
                // T x 16 B, contiguous) and the weight panel of the 1x1 phase that follows (128 rows x Cp x 4 B)
                const int be = nt / tps;
                if (be < s.n_cond)
                    l2_touch(ly.cond + (long)be * s.c_bs + (long)mt * 32 * s.T * 4, (unsigned)(32 * s.T * 16), wave - 4, 4);
                l2_touch(ly.out_w + (long)mt * (s.Cp >> 5) * 4096, (unsigned)((s.Cp >> 5) * 16384), wave - 4, 4);
            }
        } else {
            a.Wp = ly.out_w; a.bias = ly.out_b;
            a.X = s.g; a.taps = 1; a.dil = 1;
            a.Y = s.h;
            const bool last = (l + 1 == s.L);
            if (!last) {     // hd = h + d_{l+1}: the next dilated conv's input
                a.Y2 = s.hd; a.y2_bs = act_bs;
                a.d2 = s.d2 + (long)(l + 1) * s.Cp; a.tsel = s.tsel; a.d2_ts = s.d2_ts;
            }
            a.skip = s.skip; a.s_bs = act_bs; a.skip_init = (l == 0);
            // the last layer's residual output is never read (model/diffwave.py:678-682): its residual-only M
            // tiles have nothing to do
            const bool idle = last && mt < (s.Cp >> 7);
            if (s.dbg && p + 3 == s.p1) a.dbg = s.dbg + 96;       // second-to-last 1x1 phase (block 0 works in it)
            // 128- / 160-frame flavours: all eight waves contract the 1x1 - the producers have nothing to stage in this
            // phase, so waves w and w + 4 share the rows of wave w and split its frames (two MFMA streams per SIMD
            // cover each other's fragment waits and epilogue; 64 fewer live registers in the merged kernel: 204
            // instead of 256 + 16 B of scratch).  Same k order per output: bit-identical.  Measured at config 2:
            // 1x1 phase 76.9 k -> 73.7 k cycles, chain 883.1 -> 879.3 ms.  The 64-frame flavour keeps four waves:
            // with one 32-frame tile per wave the phase takes the same cycles at a lower clock (485.4 vs 483.3 ms).
            if constexpr (FL == 5) {
                if (!idle) {
                    if (wave < 4) pw_body<3, 1, 1, 160>(a, mt, nt, wave, Rs, 0);
                    else pw_body<2, 1, 1, 160>(a, mt, nt, wave - 4, Rs, 96);
                }
            } else if constexpr (FL == 2) {
                if (!idle) pw_body<2, 1, 1, 128>(a, mt, nt, wave & 3, Rs, (wave >> 2) * 64);
            } else {
                if (wave < 4 && !idle) pw_body<BN / 32, 1, 1>(a, mt, nt, wave, Rs);
            }
            if (wave >= 4 && s.warm && !last) {
                // idle for the whole 1x1 phase: fetch the first two chunks (2 x taps slabs of 16 KB) of the next
                // layer's conv weight panel of this M tile
                const __attribute__((address_space(4))) StackLayer& nx = sp->layer[l + 1];
                l2_touch(nx.conv_w + (long)mt * (s.Cp >> 5) * s.taps * 4096, (unsigned)(2 * s.taps * 16384), wave - 4, 4);
            }
        }
        if (p + 1 < s.p1) {
            // (no acquire here: a conv phase reads hd with L1-bypassing sc1 LDS-DMA loads, and the 1x1 phase's L1
            // invalidate was issued by a producer wave during the conv phase, above)
            // (s.fault: test hook - one arrival more than the group has is awaited, so every wait runs into its bound)
            group_barrier<false>(ctr, ++episode * (gsize + (unsigned)s.fault), s.err, s.derr);
            if (episode == 1) {      // every block of the group has published its (generation, XCC id): one L2 for all?
                if (threadIdx.x == 0) {
                    unsigned same = 1;
                    for (unsigned i = 0; i < gsize; ++i)
                        same &= (__hip_atomic_load(s.xid + (long)grp * gsize + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_tag);
                    same_xcd_s = same;
                }
                __syncthreads();
                wt_store = __builtin_amdgcn_readfirstlane(same_xcd_s ? 0 : 1);     // block-uniform
            }
        }
    }
    if (s.dbg && blockIdx.x == 0 && threadIdx.x == 0) {
        s.dbg[s.p1 - s.p0] = clock64();
        s.dbg[121] = wall_clock64();
    }
    // write the resident tile back: skip always (the skip projection reads it next), h only when layers remain
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int lane = threadIdx.x & 63;
        for (int i = wave; i < 32 * RWL; i += 8) {
            const int pl = i / RWL, seg = i - pl * RWL;
            bool is_res;
            float* dst = tile_plane(pl, is_res);
            const int t = t0_ + seg * 64 + lane;
#ifdef DR_BOUNDS
            if (seg * 64 + lane < BN) DR_CHECK_LDS(Rs + pl * BN + seg * 64 + lane, s.rs_off, s.rs_off + 32 * BN * 16, 133);
#endif
            if (seg * 64 + lane < BN && t < s.T && (!is_res || s.p1 < 2 * s.L))
                *reinterpret_cast<float4*>(dst + (long)t * 4) = Rs[pl * BN + seg * 64 + lane];
        }
    }
    // leave: the last block of the group to get here re-arms both counters for the next launch (nobody of this
    // group polls any more: everyone passed its last barrier before arriving here)
    if (threadIdx.x == 0) {
        const unsigned left = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gsize - 1) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // next generation
        }
    }
}

// Group-per-XCD dealing of a persistent launch (block b is dispatched to XCD b % 8): groups g, g + 8, g + 16, ... share
// XCD g, so it needs every XCD's share - ceil(NB / 8) groups - to fit that XCD's CUs.  When NB is not a multiple of 8
// the grid is padded with idle groups (their blocks exit at once) if that still holds; else the launch falls back to
// the spread mapping (groups across all XCDs, write-through hand-offs).
static int xcd_padded_groups(int NB, int gsize, int* xcd_n) {
    if (!*xcd_n) return NB;
    static const int cus = [] {                         // (thread-safe one-time initialisation)
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int per_xcd = (NB + 7) / 8;
    if (NB % 8 == 0) return NB;
    if (per_xcd * gsize * 8 <= cus) return per_xcd * 8;
    *xcd_n = 0;
    return NB;
}

hipError_t launch_stack(const StackArgs& s, int FL, int max_dil, hipStream_t st) {
    if (FL != 1 && FL != 2 && FL != 5) return hipErrorInvalidValue;
    if (s.L < 1 || s.L > DR_STACK_MAX_LAYERS || s.p0 < 0 || s.p1 > 2 * s.L || s.p0 >= s.p1 || (s.Cp & 63)) return hipErrorInvalidValue;
    const int BN = stack_tile_frames(FL), tps = (s.T + BN - 1) / BN, MT = s.Cp >> 6;
    const size_t lds = stack_lds_bytes(FL, s.taps, max_dil);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    StackArgs b = s;
    b.rs_off = (int)(lds - 16 - (size_t)32 * BN * 16);
    b.lds_bytes = (int)lds;
    const int NBp = xcd_padded_groups(s.NB, MT * tps, &b.xcd_n);
#ifdef DR_BOUNDS
    for (int l = 0; l < s.L; ++l) {
        GemmArgs g{};
        g.MT = MT; g.NB = s.NB; g.T = s.T; g.taps = s.taps; g.kchunks = s.Cp >> 5; g.y_rows = s.Cp;
        g.x_bs = g.y_bs = g.y2_bs = g.s_bs = (long)s.Cp * s.T; g.x_ps = g.y_ps = (long)s.T * 4; g.x_fs = g.y_fs = 4; g.x_planes = s.Cp >> 2;
        g.Wp = s.layer[l].conv_w; g.bias = s.layer[l].conv_b; g.bias2 = s.layer[l].conv_b2; g.X = s.hd; g.Y = s.g;
        g.cond = s.layer[l].cond; g.cond2 = s.layer[l].cond2; g.c_bs = s.c_bs; g.n_cond = s.n_cond;
        check_gemm_extents(g, EPI_GATE, 0, "stack_kernel (conv phase)");
        g.taps = 1; g.Wp = s.layer[l].out_w; g.bias = s.layer[l].out_b; g.X = s.g; g.Y = s.h; g.Y2 = s.hd; g.skip = s.skip;
        check_gemm_extents(g, EPI_RES_SKIP, 0, "stack_kernel (1x1 phase)");
    }
    host_extent(s.bar, (size_t)4 * 512 * 4, "group counters", "stack_kernel");
    host_extent(s.xid, (size_t)1024 * 4, "tag words", "stack_kernel");
#endif
    const dim3 grid((unsigned)(MT * tps * NBp));
    if (FL == 1) hipLaunchKernelGGL((stack_kernel<1>), grid, dim3(512), lds, st, b);
    else if (FL == 2) hipLaunchKernelGGL((stack_kernel<2>), grid, dim3(512), lds, st, b);
    else hipLaunchKernelGGL((stack_kernel<5>), grid, dim3(512), lds, st, b);
    return hipGetLastError();
}
int stack_tile_frames(int FL) { return FL == 5 ? 160 : 64 * FL; }
// the conv's double-buffered X tiles + the resident h / skip tile
size_t stack_lds_bytes(int FL, int taps, int max_dil) {
    const int BN = stack_tile_frames(FL), halo = ((taps - 1) / 2) * max_dil;
    return (size_t)2 * 8 * (BN + 2 * halo) * 16 + (size_t)32 * BN * 16 + 16;     // + one flag word (16-byte slot)
}

// ---------------------------------------------------------------------------------------------
// Flexible-width variant on v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue): 16-frame column tiles,
// so a block covers BN = 32*NJ frames (96, 160).  Used when 64/128-frame tiles
// quantise badly over the 256 CUs (config 5: 8 x 640 frames -> 4 x 160-frame tiles per clip = exactly
// 256 blocks instead of 2.5 rounds of 64-frame blocks).  Same packed weights (the slab is
// [channel/4][row][4], which serves both MFMA shapes), same LDS-DMA producers, same P4 layouts.
//   consumers 4 (M) x 1 (N): wave tile = 2 row tiles (paired epilogue: gate 16 / filter 16 of the same
//   channels) x 2*NJ column tiles = 20 accumulators x 4 registers at NJ = 5.
// ---------------------------------------------------------------------------------------------
//   COH = 1 (fused residual-stack kernel): as gemm_body - X (hd) through sc1 LDS-DMA loads, the gated output
//   (g) stored write-through unless a.wt_store == 0.
template <int NJ, int KS, int EPI, int COH>
DR_DEVINL void gemm16_body(const GemmArgs& a, char* smem, const int mt, const int nt) {
    static_assert(EPI == EPI_GATE || EPI == EPI_RES_SKIP, "16x16 variant: hot kernels only");
    constexpr int BN = 32 * NJ;
    constexpr int XP = 8 * KS;
    constexpr int WNC = 1;                          // 4 (M) x 1 (N) consumers, as in gemm_kernel
    constexpr int RT = 2;                           // 16-row tiles per wave (paired: tile 0 gate, tile 1 filter)
    constexpr int CT = 2 * NJ;                      // 16-frame tiles per wave
    constexpr int WROWS = RT * 16, WFR = CT * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo = ((a.taps - 1) >> 1) * a.dil;
    const int FW = BN + 2 * halo;
    float4* Xs = reinterpret_cast<float4*>(smem);   // [2][XP][FW]
    float4* Rs = Xs + 2 * XP * FW;                  // EPI_RES_SKIP: [32 planes][BN]
#ifdef DR_BOUNDS
    const unsigned xs0 = lds_off(Xs), xs1 = xs0 + 2u * XP * FW * 16u, rs1 = xs1 + (EPI == EPI_RES_SKIP ? 32u * BN * 16u : 0u);
    if (tid == 0 && !COH) DR_CHECK(rs1 <= (unsigned)a.lds_bytes, 140, rs1, a.lds_bytes);
#endif

    const int tps = (a.T + BN - 1) / BN;
    const int b = nt / tps;
    const int t0 = (nt % tps) * BN;
    const int NS = a.kchunks * a.taps;
    const int nchunks = a.kchunks / KS;

    if (wave >= 4) {   // producers: identical to gemm_kernel's (LDS-DMA, hardware zero padding)
        const int pw = wave - 4;
        const int bx = a.x_bmod ? (b % a.x_bmod) : b;
        const float* Xg = a.X + (long)bx * a.x_bs;
        const int last_plane = a.x_planes - 1;
        const unsigned recs = ((unsigned)(a.T - 1) * (unsigned)a.x_fs + 4u) * 4u;
        const int wl = (FW + 63) >> 6;
        const int total = XP * wl;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        auto issue = [&](int chunk) {
            for (int i = pw; i < total; i += 4) {
                const int pl = i / wl, seg = i - pl * wl;
                const int f = seg * 64 + lane;
                const float* src = Xg + (long)min(chunk * XP + pl, last_plane) * a.x_ps;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, recs, 0x00020000);
                const int voff = (t0 - halo + f) * (int)a.x_fs * 4;
                float4* dst = Xs + ((chunk & 1) * XP + pl) * FW + seg * 64;
                if (f < FW) DR_CHECK_LDS(dst + lane, xs0, xs1, 141);
                if (f < FW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)dst, 16, voff, 0, 0, COH ? 16 : 0);
            }
        };
        if constexpr (EPI == EPI_RES_SKIP) {
            const unsigned rrecs = (unsigned)a.T * 16u;
            constexpr int RWL = (BN + 63) / 64;
            for (int i = pw; i < 32 * RWL; i += 4) {
                const int pl = i / RWL, seg = i - pl * RWL;
                const int row0 = mt * 128 + pl * 4;
                const float* src = (row0 < a.y_rows)
                    ? a.Y + (long)b * a.y_bs + (long)(row0 >> 2) * a.y_ps
                    : a.skip + (long)b * a.s_bs + (long)((row0 - a.y_rows) >> 2) * a.T * 4;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rrecs, 0x00020000);
                const int f = seg * 64 + lane;
                if (f < BN) DR_CHECK_LDS(Rs + pl * BN + seg * 64 + lane, xs1, rs1, 142);
                if (f < BN)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(Rs + pl * BN + seg * 64), 16, (t0 + f) * 16, 0, 0, 0);
            }
        }
        issue(0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile #chunk has landed (see gemm_body)
            __syncthreads();
            if (chunk + 1 < nchunks) issue(chunk + 1);
        }
        return;
    }

    // consumers.  16x16x4: A lane (i = l&15, kq = l>>4) holds W[row i][4 channels kq*4..+3 of a 16-channel
    // group] (one per MFMA), B lane (j = l&15, kq) the matching X values; C/D: column = l&15,
    // rows (l>>4)*4 + reg -> one float4 of the P4 layout per tile.
    const int wr = wave / WNC, wc = wave % WNC;
    const int li = lane & 15, kq = lane >> 4;
    float4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f4zero();

    // A fragments through buffer loads with scalar per-step offsets (as in gemm_kernel)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Wp + (long)mt * NS * 4096), 0, (unsigned)NS * 16384u, 0x00020000);
    const int wvo = (kq * 128 + wr * WROWS + li) * 16;
    struct AF { float4 v[2 * RT]; };   // [g16][rt]
    auto load_a = [&](int slab) -> AF {
        AF o;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                DR_CHECK(slab >= 0 && wvo + slab * 16384 + g * 8192 + rt * 256 + 16 <= NS * 16384, 143, slab, NS);
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, slab * 16384 + g * 8192 + rt * 256, 0);
                o.v[g * RT + rt] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
            }
        return o;
    };
    AF wA = load_a(0), wB;
    const int cen = (a.taps - 1) >> 1;
    const int per_chunk = a.taps * KS;
    typedef float v4f __attribute__((ext_vector_type(4)));

    // B fragments of one 16-channel group: CT float4 (one per 16-frame column tile)
    struct BF { float4 v[CT]; };
    BF b0, b1;
    auto xaddr = [&](int chunk, int q) -> const float4* {
        const int j = q / KS, sub = q - j * KS;
        return Xs + ((chunk & 1) * XP + sub * 8 + kq) * FW + halo + (j - cen) * a.dil + wc * WFR + li;
    };
    auto rd = [&](const float4* Xb, int g) -> BF {
        BF o;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            DR_CHECK_LDS(Xb + g * 4 * FW + ct * 16, xs0, xs1, 144);
            o.v[ct] = Xb[g * 4 * FW + ct * 16];
        }
        return o;
    };
    auto mma = [&](const float4 af, const BF& bf, int rt) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            v4f c = {acc[rt][ct].x, acc[rt][ct].y, acc[rt][ct].z, acc[rt][ct].w};
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.v[ct].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.v[ct].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.v[ct].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.v[ct].w, c, 0, 0, 0);
            acc[rt][ct] = make_float4(c[0], c[1], c[2], c[3]);
        }
    };
    // K step = 2 groups of 16 channels; group 1's fragments are read before group 0's MFMAs, the NEXT step's
    // group 0 before group 1's MFMAs (cross-step prefetch inside a chunk, as in gemm_kernel)
    auto step = [&](auto ROLE, int slab, int chunk, int q) {
        constexpr bool kB = decltype(ROLE)::value;
        const float4* Xb = xaddr(chunk, q);
        if constexpr (kB) wA = load_a(min(slab + 1, NS - 1));
        else wB = load_a(min(slab + 1, NS - 1));
        b1 = rd(Xb, 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mma(kB ? wB.v[rt] : wA.v[rt], b0, rt);
        b0 = rd(xaddr(chunk, min(q + 1, per_chunk - 1)), 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mma(kB ? wB.v[RT + rt] : wA.v[RT + rt], b1, rt);
        // the 2*RT A-fragment loads of the next step ride inside group 0's MFMAs
        sgb<0x100, CT>(); sgb_spread<2 * RT, (4 * RT * CT) / (2 * RT)>();
        sgb<0x100, CT>(); sgb<0x8, 4 * RT * CT>();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        auto at = [&](auto R, int q) {
            const int j = q / KS, sub = q - j * KS;
            step(R, (chunk * KS + sub) * a.taps + j, chunk, q);
        };
        __syncthreads();
        b0 = rd(xaddr(chunk, 0), 0);
        int q = 0;
        for (; q + 2 <= per_chunk; q += 2) {
            at(F_{}, q);
            at(T_{}, q + 1);
        }
        if (q < per_chunk) {
            at(F_{}, q);
            wA = wB;
        }
    }

    // epilogue: per (row tile, column tile) a lane owns rows rowbase + kq*4 .. +3 of frame column li
    auto f4arr = [](const float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; };
    const int rowb = mt * 128 + wr * WROWS + kq * 4;     // + rt*16
    float4 ebias[RT], ed2[RT];
    {
        const float* bsrc = a.bias;
        if constexpr (EPI == EPI_GATE) bsrc = (b < a.n_cond) ? a.bias : a.bias2;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            ebias[rt] = *reinterpret_cast<const float4*>(bsrc + rowb + rt * 16);
            if constexpr (EPI == EPI_RES_SKIP)
                ed2[rt] = *reinterpret_cast<const float4*>(a.d2 + (a.tsel ? (long)a.tsel[b] * a.d2_ts : 0) + min(rowb + rt * 16, a.y_rows - 4));
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int t = t0 + wc * WFR + ct * 16 + li;
        if constexpr (EPI == EPI_GATE) {
            float4 cnd[RT];
            const int tc = min(t, a.T - 1);
            const float* cb = ((b < a.n_cond || !a.cond2) ? a.cond + (long)(b < a.n_cond ? b : 0) * a.c_bs : a.cond2) + (long)tc * 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) cnd[rt] = *reinterpret_cast<const float4*>(cb + (long)((rowb + rt * 16) >> 2) * a.T * 4);
            if (t >= a.T) continue;
            const bool has_c = b < a.n_cond || a.cond2 != nullptr;
            {                                    // tile 0 = gate rows, tile 1 = filter rows of the same 16 channels
                const int c0 = mt * 64 + wr * 16 + kq * 4;
                if (c0 >= a.y_rows) continue;
                float v0[4], v1[4], b0[4], b1[4], c0v[4], c1v[4], o[4];
                f4arr(acc[0][ct], v0); f4arr(acc[1][ct], v1);
                f4arr(ebias[0], b0); f4arr(ebias[1], b1);
                f4arr(cnd[0], c0v); f4arr(cnd[1], c1v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = has_c ? b0[e] + c0v[e] : b0[e];
                    const float a1 = has_c ? b1[e] + c1v[e] : b1[e];
                    o[e] = gatef_(v0[e] + a0, v1[e] + a1);
                }
                if (a.out_s3 & 1) {
                    store_s3_quad(a.Y + (long)b * a.y_bs, o, c0, t, a.T, a.y_rows >> 3);
                } else {
                    float* dst = a.Y + (long)b * a.y_bs + (long)(c0 >> 2) * a.y_ps + (long)t * a.y_fs;
                    store_f4<COH>(dst, make_float4(o[0], o[1], o[2], o[3]), a.wt_store);
                }
            }
        } else {
            float4 pv4[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                DR_CHECK_LDS(Rs + ((wr * WROWS + rt * 16) / 4 + kq) * BN + wc * WFR + ct * 16 + li, xs1, rs1, 145);
                pv4[rt] = Rs[((wr * WROWS + rt * 16) / 4 + kq) * BN + wc * WFR + ct * 16 + li];
            }
            if (t >= a.T) continue;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int p0 = rowb + rt * 16;
                float v[4], bb[4], pv[4], o[4];
                f4arr(acc[rt][ct], v); f4arr(ebias[rt], bb); f4arr(pv4[rt], pv);
                if (p0 < a.y_rows) {
                    float* dst = a.Y + (long)b * a.y_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = div_sqrt2(pv[e] + (v[e] + bb[e]));
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    if (a.Y2) {
                        float dd[4];
                        f4arr(ed2[rt], dd);
                        const float o2[4] = {o[0] + dd[0], o[1] + dd[1], o[2] + dd[2], o[3] + dd[3]};
                        if (a.out_s3 & 2) {
                            store_s3_quad(a.Y2 + (long)b * a.y2_bs, o2, p0, t, a.T, a.y_rows >> 3);
                        } else {
                            float* dst2 = a.Y2 + (long)b * a.y2_bs + (long)(p0 >> 2) * a.y_ps + (long)t * a.y_fs;
                            *reinterpret_cast<float4*>(dst2) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                        }
                    }
                } else {
                    float* dst = a.skip + (long)b * a.s_bs + ((long)((p0 - a.y_rows) >> 2) * a.T + t) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = a.skip_init ? v[e] + bb[e] : (v[e] + bb[e]) + pv[e];
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}

template <int NJ, int KS, int EPI>
__global__ __launch_bounds__(512) void gemm16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int mt, nt;
    if (a.xcd_n) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % a.MT;
        nt = (idx / a.MT) * 8 + xcd;
    } else {
        mt = blockIdx.x % a.MT;
        nt = blockIdx.x / a.MT;
    }
    gemm16_body<NJ, KS, EPI, 0>(a, smem, mt, nt);
}

template <int NJ, int KS, int EPI>
static hipError_t launch_gemm16_t(const GemmArgs& a, hipStream_t s) {
    const int BN = 32 * NJ;
    const int halo = ((a.taps - 1) / 2) * a.dil;
    const size_t lds = (size_t)2 * 8 * KS * (BN + 2 * halo) * 16 + (EPI == EPI_RES_SKIP ? (size_t)32 * BN * 16 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int NT = a.NB * ((a.T + BN - 1) / BN);
    GemmArgs b = a;
    const double wbytes = 4.0 * 128.0 * a.MT * 32.0 * a.kchunks * a.taps, xbytes = 4.0 * (double)NT * BN * 32.0 * a.kchunks;
    b.xcd_n = pick_xcd_mapping(a.MT, NT, wbytes, xbytes);
    b.lds_bytes = (int)lds;
    DR_CHECK_EXTENTS(b, EPI, 0, "gemm16_kernel");
    hipLaunchKernelGGL((gemm16_kernel<NJ, KS, EPI>), dim3((unsigned)(a.MT * NT)), dim3(512), lds, s, b);
    return hipGetLastError();
}
// frames per block = 32 * NJ; NJ in {3, 5}: 96 / 160 (64 and 128 are served by gemm_kernel; 192 does not fit the
// 256-register budget of a 512-thread block without spilling)
hipError_t launch_gemm16(const GemmArgs& a, int epi, int NJ, hipStream_t s) {
    if (a.kchunks < 1) return hipErrorInvalidValue;
    if (epi == EPI_GATE) {
        if (NJ == 3) return launch_gemm16_t<3, 1, EPI_GATE>(a, s);
        if (NJ == 5) return launch_gemm16_t<5, 1, EPI_GATE>(a, s);
    } else if (epi == EPI_RES_SKIP && a.taps == 1 && a.kchunks % 2 == 0) {
        if (NJ == 3) return launch_gemm16_t<3, 2, EPI_RES_SKIP>(a, s);
        if (NJ == 5) return launch_gemm16_t<5, 2, EPI_RES_SKIP>(a, s);
    }
    return hipErrorInvalidValue;
}
static hipError_t init_gemm16() {
    hipError_t e;
#define DR_INIT16(NJ, KS, EPI)                                                                           \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_kernel<NJ, KS, EPI>),            \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) \
        return e;
    DR_INIT16(3, 1, EPI_GATE) DR_INIT16(5, 1, EPI_GATE)
    DR_INIT16(3, 2, EPI_RES_SKIP) DR_INIT16(5, 2, EPI_RES_SKIP)
#undef DR_INIT16
    return hipSuccess;
}

size_t gemm_lds_bytes(int NI, int KS, int taps, int dil, int prec, int epi) {
    const int halo = ((taps - 1) / 2) * dil;
    const int FW = 64 * NI + 2 * halo;
    return (size_t)2 * (prec ? 12 : 8) * KS * FW * 16 + (epi == EPI_RES_SKIP ? (size_t)32 * 64 * NI * 16 : 0);
}

template <int NI, int KS, int EPI, int PREC>
static hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    const int BN = 64 * NI;
    const int tps = (a.T + BN - 1) / BN;
    const size_t lds = gemm_lds_bytes(NI, KS, a.taps, a.dil, PREC, EPI);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int NT = a.NB * tps;
    GemmArgs b = a;
    // Split-K: launches that cannot fill the chip, and (round 3) launches that fill it unevenly.  The ticket reduction
    // needs no co-residency (nobody spins), so a launch may be cut into MORE blocks than the chip holds: 10 evaluations
    // of 125 frames are 160 tiles = one round of full-K blocks on 62 % of the CUs; cut 4x in K they are 640 blocks =
    // 3 rounds of quarter-length blocks: 136 -> 107 us per conv launch, 2410 -> 1996 us per reverse step.  Cost model in
    // us per launch (fp32), fitted to 3..8 guided clips of 125 frames (tools/small_batch_ab.py: equal blocks run in
    // lockstep rounds - 160 / 192 / 224 tiles cut 4x took 107 / 109 / 143 us): rounds x t_full / ks + exchange, with
    // t_full = a full-K tile (MFMA count x 69 cycles) and the exchange (store, ticket, the last arriver's re-read of
    // ks partials) ~(4 + ks) us.
    b.ksplit = 1;
    if (a.ws && a.ws_cnt) {
        static const int ks_max = getenv("DR_KSPLIT_MAX") ? atoi(getenv("DR_KSPLIT_MAX")) : 16;   // tuning experiments
        static const long max_blocks = getenv("DR_KSPLIT_BLOCKS") ? atol(getenv("DR_KSPLIT_BLOCKS")) : (PREC ? 256 : 2048);
        const int nchunks = a.kchunks / KS;
        const long tiles = (long)a.MT * NT;
        const double t_full = (double)a.kchunks * a.taps * 16.0 * (2 * NI) * 69.0 / 2400.0;
        auto cost = [&](int ks) {
            return (double)((tiles * ks + 255) / 256) * t_full / ks + (ks > 1 ? 4.0 + ks : 0.0);
        };
        double best = cost(1);
        for (int ks = 2; ks <= ks_max && ks <= 16; ks *= 2) {
            if (tiles * ks > max_blocks || nchunks % ks != 0) break;
            if ((size_t)tiles * ks * 128 * BN > a.ws_floats || (size_t)tiles * 4 > a.ws_cnt_n) break;
            const double c = cost(ks);
            // (inside one resident round more slices are taken as before; beyond it a split must win by 3 %)
            if (tiles * ks <= 256 || c < 0.97 * best) { best = std::min(best, c); b.ksplit = ks; }
        }
    }
    b.lds_bytes = (int)lds;
    const dim3 grid((unsigned)(a.MT * NT * b.ksplit));
    // weights: MT*128 rows x 32*kchunks*taps floats; activations: NT*BN frames x 32*kchunks floats
    const double wbytes = 4.0 * 128.0 * a.MT * 32.0 * a.kchunks * a.taps, xbytes = 4.0 * (double)NT * BN * 32.0 * a.kchunks;
    b.xcd_n = pick_xcd_mapping(a.MT, NT, wbytes, xbytes);
    DR_CHECK_EXTENTS(b, EPI, PREC, "gemm_kernel");
    hipLaunchKernelGGL((gemm_kernel<NI, KS, EPI, PREC>), grid, dim3(512), lds, s, b);
    return hipGetLastError();
}

template <int NI, int KS, int EPI, int PREC>
static hipError_t init_gemm_t() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<NI, KS, EPI, PREC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int NI, int KS>
static hipError_t init_gemm_ni() {
    hipError_t e;
    if ((e = init_gemm_t<NI, KS, EPI_PLAIN, 0>()) != hipSuccess) return e;
    if ((e = init_gemm_t<NI, KS, EPI_RELU, 0>()) != hipSuccess) return e;
    if ((e = init_gemm_t<NI, KS, EPI_SILU, 0>()) != hipSuccess) return e;
    if ((e = init_gemm_t<NI, KS, EPI_GATE, 0>()) != hipSuccess) return e;
    if ((e = init_gemm_t<NI, KS, EPI_RES_SKIP, 0>()) != hipSuccess) return e;
    if ((e = init_gemm_t<NI, KS, EPI_POWER, 0>()) != hipSuccess) return e;
    return init_gemm_t<NI, KS, EPI_LOG, 0>();
}
__global__ void tail_kernel(const TailArgs s);      // below
__global__ void stft_power_kernel(const float* __restrict__ wav_pad, const float* __restrict__ win, const float2* __restrict__ tw,
                                  float* __restrict__ power, int Lp, int TF, int N, int hop, int bins_p, float norm);   // below
// allow > 64 KiB of dynamic LDS for every instantiation; call once per process before any launch
// (and never inside a stream capture)
hipError_t init_kernels() {
    hipError_t e;
    if ((e = init_gemm_ni<1, 1>()) != hipSuccess) return e;
    if ((e = init_gemm_ni<2, 1>()) != hipSuccess) return e;
    if ((e = init_gemm_ni<1, 2>()) != hipSuccess) return e;
    if ((e = init_gemm_ni<2, 2>()) != hipSuccess) return e;
    if ((e = init_gemm_ni<1, 4>()) != hipSuccess) return e;
    if ((e = init_gemm_ni<2, 4>()) != hipSuccess) return e;
    // split-bf16 instantiations: dilated conv (KS = 1) and 1x1 (NI = 1: KS = 4)
    if ((e = init_gemm_t<1, 1, EPI_GATE, 1>()) != hipSuccess) return e;
    if ((e = init_gemm_t<2, 1, EPI_GATE, 1>()) != hipSuccess) return e;
    if ((e = init_gemm_t<1, 4, EPI_RES_SKIP, 1>()) != hipSuccess) return e;
    if ((e = init_gemm_t<1, 1, EPI_RES_SKIP, 1>()) != hipSuccess) return e;
    // the FFT front-end holds two n_fft/2-point complex buffers: n_fft * 8 bytes (128 KiB at its largest size, 16384)
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_power_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    return init_gemm16();
}

template <int NI, int KS>
static hipError_t launch_gemm_ni(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_PLAIN: return launch_gemm_t<NI, KS, EPI_PLAIN, 0>(a, s);
        case EPI_RELU: return launch_gemm_t<NI, KS, EPI_RELU, 0>(a, s);
        case EPI_SILU: return launch_gemm_t<NI, KS, EPI_SILU, 0>(a, s);
        case EPI_GATE: return launch_gemm_t<NI, KS, EPI_GATE, 0>(a, s);
        case EPI_RES_SKIP: return launch_gemm_t<NI, KS, EPI_RES_SKIP, 0>(a, s);
        case EPI_POWER: return launch_gemm_t<NI, KS, EPI_POWER, 0>(a, s);
        case EPI_LOG: return launch_gemm_t<NI, KS, EPI_LOG, 0>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm(const GemmArgs& a, int epi, int NI, hipStream_t s, int prec) {
    if (a.kchunks < 1) return hipErrorInvalidValue;   // the X tile width is only bounded by LDS (checked per launch)
    if (prec == 1) {   // split-bf16 input: only the two hot kernels exist in this precision
        if (epi == EPI_GATE) return NI == 1 ? launch_gemm_t<1, 1, EPI_GATE, 1>(a, s) : launch_gemm_t<2, 1, EPI_GATE, 1>(a, s);
        if (epi == EPI_RES_SKIP && a.taps == 1)
            return a.kchunks % 4 == 0 ? launch_gemm_t<1, 4, EPI_RES_SKIP, 1>(a, s) : launch_gemm_t<1, 1, EPI_RES_SKIP, 1>(a, s);
        return hipErrorInvalidValue;
    }
    // 1x1 GEMMs restage X every step: take up to 128 channels per chunk there (fewer hand-overs);
    // EPI_RES_SKIP also keeps its read-modify-write tile in LDS, which leaves room for 64 channels at NI = 2
    int KS = a.taps != 1 ? 1 : (a.kchunks % 4 == 0 ? 4 : (a.kchunks % 2 == 0 ? 2 : 1));
    if (epi == EPI_RES_SKIP && NI == 2 && KS == 4) KS = 2;
    static const int ks_force = getenv("DR_1X1_KS") ? atoi(getenv("DR_1X1_KS")) : 0;   // tuning experiments
    if (ks_force && a.taps == 1 && a.kchunks % ks_force == 0) KS = ks_force;
    if (NI == 1) {
        if (KS == 4) return launch_gemm_ni<1, 4>(a, epi, s);
        return KS == 2 ? launch_gemm_ni<1, 2>(a, epi, s) : launch_gemm_ni<1, 1>(a, epi, s);
    }
    if (NI == 2) {
        if (KS == 4) return launch_gemm_ni<2, 4>(a, epi, s);
        return KS == 2 ? launch_gemm_ni<2, 2>(a, epi, s) : launch_gemm_ni<2, 1>(a, epi, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: z ~ N(0,1), keyed by (seed, global sample, step, element/4) so the
// noise of a sample does not depend on how the batch is sharded over GPUs.
// ---------------------------------------------------------------------------------------------
DR_DEVINL void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

DR_DEVINL void box_muller(uint32_t u0, uint32_t u1, float& z0, float& z1) {
    const float a = ((float)(u0 >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float bb = (float)(u1 >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float rad = sqrtf(-2.0f * logf(a));
    float sn, cs;
    sincosf(6.283185307179586f * bb, &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

// Classifier-free combine + x0-prediction posterior update of ONE float4 (4 consecutive elements, index i4) of the
// roll.  Same operation order as task/diffusion.py:953 and :957-967; contraction off so that no FMA is formed where
// the reference rounds twice.  Shared by update_kernel and the tail kernel (identical arithmetic).
DR_DEVINL float4 update_quad(const UpdateArgs& a, const long i4) {
#pragma clang fp contract(off)
    const float4 xc = reinterpret_cast<const float4*>(a.x0c)[i4];
    float x0[4] = {xc.x, xc.y, xc.z, xc.w};
    // per-call scalars: by value (eager launches) or from the device block (captured chain)
    const float gw = a.dyn ? a.dyn->w : a.w, g1pw = a.dyn ? a.dyn->onepw : a.onepw;
    const uint64_t seed = a.dyn ? a.dyn->seed : a.seed;
    const int first_sample = a.dyn ? a.dyn->first_sample : a.first_sample;
    if (a.x0u) {
        const float4 xu = reinterpret_cast<const float4*>(a.x0u)[i4];
        const float u[4] = {xu.x, xu.y, xu.z, xu.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) x0[e] = g1pw * x0[e] - gw * u[e];
    }
    const float c0 = a.coef[0], c1 = a.coef[1], c2 = a.coef[2], c3 = a.coef[3], c4 = a.coef[4];
    float o[4];
    // which updates draw noise at t > 0: x0 DDPM (0), eps ddpm (2), eps ddim2ddpm (4)
    const bool noisy = (a.mode == 0 || a.mode == 2 || a.mode == 4) && a.t > 0;
    float x[4] = {0.f, 0.f, 0.f, 0.f}, z[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.t > 0 || a.mode >= 2) {
        const float4 xv = reinterpret_cast<const float4*>(a.x)[i4];
        x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w;
    }
    if (noisy) {
        if (a.noise) {
            const float4 zv = reinterpret_cast<const float4*>(a.noise)[i4];
            z[0] = zv.x; z[1] = zv.y; z[2] = zv.z; z[3] = zv.w;
        } else {
            const long e0 = i4 * 4;
            const long smp = e0 / a.per_sample;
            const long within = (e0 - smp * a.per_sample) >> 2;
            uint32_t rnd[4];
            philox4x32_10((uint32_t)within, (uint32_t)(within >> 32), (uint32_t)a.t,
                          (uint32_t)(first_sample + smp), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
            box_muller(rnd[0], rnd[1], z[0], z[1]);
            box_muller(rnd[2], rnd[3], z[2], z[3]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float y = x0[e];   // network output: x0 prediction (modes 0/1) or epsilon (modes 2-4)
        if (a.mode <= 1) {
            // ddpm_x0 family :957-967 / ddim_x0 family :864-873 (c4 = sigma = 0, the 0*z term is dropped)
            if (a.t == 0) o[e] = y / c2;
            else {
                const float t1 = c0 * y;
                const float t2 = (c1 * (x[e] - c2 * y)) / c3;
                o[e] = (a.mode == 0) ? (t1 + t2) + c4 * z[e] : (t1 + t2);
            }
        } else if (a.mode == 2) {
            // ddpm :820-829: sqrt_recip_alphas_t * (x - betas_t * eps / sqrt_1m_acp_t) [+ sqrt(post_var_t) * z]
            const float m = c0 * (x[e] - (c1 * y) / c2);
            o[e] = (a.t == 0) ? m : m + c3 * z[e];
        } else {
            // ddim :885-890 / ddim2ddpm :902-909
            const float xe = (x[e] - c3 * y) / c2;
            if (a.t == 0) o[e] = xe;
            else if (a.mode == 3) o[e] = c0 * xe + c1 * y;
            else o[e] = (c0 * xe + c1 * y) + c4 * z[e];
        }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void update_kernel(const UpdateArgs a) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= a.n) return;
    reinterpret_cast<float4*>(a.x)[i4] = update_quad(a, i4);
}

// ---------------------------------------------------------------------------------------------
// Tail of a reverse step as ONE persistent launch (model/diffwave.py:682-686 + task/diffusion.py:953-967 + the next
// step's model/diffwave.py:667-668): what used to be four launches - skip projection, output projection, posterior
// update, input projection of the next step - with the same grid, block -> (group, member) mapping and
// counter protocol as stack_kernel (a GROUP = the blocks of one clip evaluation; under classifier-free guidance the
// conditional group b and the unconditional group b + B form a PAIR):
//   T1  skip projection + relu: relu(W_s skip / sqrt(L) + b_s) -> tmp.  Items = (128-row tile, 64-frame chunk) of the
//       group's clip, dealt over the group's blocks; both operands straight from L2 (pw_body<2>); output write-through.
//   --  group barrier (+ L1 invalidate: T2 reads tmp with plain loads)
//   T2  output projection -> x0 in the (B, T, 88) roll layout.  Items = 32-frame chunks (pw_body<1>; 88 rows: three
//       of the four waves).  Output write-through.
//   --  pair barrier (group barrier without guidance) + L1 invalidate
//   T3  per item (128-row tile of the input projection, 32-frame chunk) of the pair's clip: the classifier-free
//       combine + posterior update of those 32 frames x 88 keys (update_quad: the arithmetic of update_kernel) into
//       LDS - the item with row tile 0 also writes x_{t-1} back - then, when a step follows, its input projection
//       h = relu(W_in x_{t-1} + b_in), hd = h + d_0[t-1] from that LDS tile, written for BOTH samples of the pair
//       (the conditional and the unconditional evaluation start from the same x).  No barrier inside T3: the 4 row
//       tiles of a chunk recompute the same (tiny) update instead of exchanging it.
//   --  pair barrier (hd is stored write-through and read by T4 with L1-bypassing LDS-DMA)
//   T4  (guided chains only) the next step's FIRST-LAYER dilated conv + conditioner + gate: both evaluations of a pair
//       convolve the same h + d_0, so the contraction is done once per pair - items = (M tile, 64-frame chunk) of the
//       pair's clip dealt over the pair's blocks, gemm_body's dual epilogue writes both samples' g - exactly what the
//       separate layer-0 launch did; the following stack launch starts at phase 1.
// Same MFMA order (k ascending, one accumulator per output) and the same epilogue expressions as the per-phase
// kernels without split-K: bit-identical to them.  512 threads (T1-T3: waves 0-3 contract, T4: 4 consumer + 4
// producer waves); dynamic LDS = max(the 24-plane x 32-frame x tile of T3 (12 KiB), T4's X tiles).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void tail_kernel(const TailArgs s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* XT = reinterpret_cast<float4*>(smem);         // [plane = key / 4][frame of the chunk]: x_{t-1}, planes 22-23 zero
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int MT = s.Cp >> 6;                             // blocks per frame tile, as in stack_kernel
    const int tps = (s.T + s.BN - 1) / s.BN;
    const unsigned gsize = (unsigned)(MT * tps);
    int grp, member;
    if (s.xcd_n) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        member = idx % (int)gsize;
        grp = (idx / (int)gsize) * 8 + xcd;
    } else {
        member = blockIdx.x % (int)gsize;
        grp = blockIdx.x / (int)gsize;
    }
    if (grp >= s.NB) return;                                                               // padding group (launch_tail)
    if (__hip_atomic_load(s.derr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // see stack_kernel
    DR_CHECK(grp >= 0 && grp < 512, 150, grp, member);
    unsigned* ctr = s.bar + 4 * grp;
    const bool paired = s.dual > 0;
    const int pair_i = paired ? (grp < s.dual ? grp : grp - s.dual) : grp;
    const int pair_half = (paired && grp >= s.dual) ? 1 : 0;
    unsigned* pctr = s.pbar + 4 * pair_i;
    const int P = s.Cp >> 2;
    const long act_bs = (long)s.Cp * s.T;
    int mark_i = 0;
    auto mark = [&]() { if (s.dbg && blockIdx.x == 0 && threadIdx.x == 0) s.dbg[mark_i] = clock64(); ++mark_i; };
    mark();

    // ---- T1: skip projection (C x C) + relu, alpha = 1 / sqrt(L)
    {
        GemmArgs a{};
        a.d2 = s.zero; a.wt_store = 1;
        a.Wp = s.skip_w; a.bias = s.skip_b; a.MT = (s.Cp + 127) >> 7; a.NB = s.NB; a.T = s.T; a.alpha = s.alpha;
        a.X = s.skip; a.x_bs = act_bs; a.x_ps = (long)s.T * 4; a.x_fs = 4; a.x_planes = P; a.kchunks = s.Cp >> 5;
        a.taps = 1; a.dil = 1; a.ksplit = 1;
        a.Y = s.tmp; a.y_bs = act_bs; a.y_ps = (long)s.T * 4; a.y_fs = 4; a.y_rows = s.Cp;
        const int tps64 = (s.T + 63) >> 6;
        const int n1 = a.MT * tps64;
        if (wave < 4)
            for (int it = member; it < n1; it += (int)gsize) pw_body<2, 1, 0, 64, EPI_RELU>(a, it % a.MT, grp * tps64 + it / a.MT, wave);
    }
    mark();
    group_barrier<true>(ctr, gsize + (unsigned)s.fault, s.err, s.derr);
    mark();
    // ---- T2: output projection (88 x C) into the roll layout
    const int tps32 = (s.T + 31) >> 5;
    {
        GemmArgs a{};
        a.d2 = s.zero; a.wt_store = 1;
        a.Wp = s.outp_w; a.bias = s.outp_b; a.MT = 1; a.NB = s.NB; a.T = s.T; a.alpha = 1.f;
        a.X = s.tmp; a.x_bs = act_bs; a.x_ps = (long)s.T * 4; a.x_fs = 4; a.x_planes = P; a.kchunks = s.Cp >> 5;
        a.taps = 1; a.dil = 1; a.ksplit = 1;
        a.Y = s.x0; a.y_bs = (long)s.T * 88; a.y_ps = 4; a.y_fs = 88; a.y_rows = 88;
        if (wave < 3)      // rows 96 .. 127 of the tile do not exist
            for (int it = member; it < tps32; it += (int)gsize) pw_body<1, 1, 0, 32, EPI_PLAIN>(a, 0, grp * tps32 + it, wave);
    }
    mark();
    if (paired) group_barrier<true>(pctr, 2u * gsize + (unsigned)s.fault, s.err, s.derr);
    else group_barrier<true>(ctr, 2u * (gsize + (unsigned)s.fault), s.err, s.derr);
    mark();
    // ---- T3: combine + update (+ the next step's input projection) per (row tile, 32-frame chunk) of the pair's clip
    if (pair_i < s.u_B) {       // (groups without a roll of their own - none today - would skip)
        const int MTi = s.in_w ? ((s.Cp + 127) >> 7) : 1;
        const int n3 = MTi * tps32;
        const int stride = (int)gsize * (paired ? 2 : 1);
        const long roll0 = (long)pair_i * s.T * 88;                   // first element of this clip's roll
        for (int it = pair_half * (int)gsize + member; it < n3; it += stride) {
            const int mti = it % MTi, ck = it / MTi;
            const int f0 = ck * 32;
            __syncthreads();                                          // the previous item's readers of XT are done
            for (int q = threadIdx.x; q < 24 * 32; q += 512) {
                const int fi = q / 24, pl = q - fi * 24;              // 22 consecutive threads walk one frame's 88 keys
                float4 v = f4zero();
                if (pl < 22 && f0 + fi < s.T) {
                    const long i4 = (roll0 + (long)(f0 + fi) * 88 + pl * 4) >> 2;
                    v = update_quad(s.u, i4);
                    if (mti == 0) reinterpret_cast<float4*>(s.x_out)[i4] = v;
                }
                XT[pl * 32 + fi] = v;
            }
            if (!s.in_w) continue;
            __syncthreads();
            if (wave >= 4) continue;
            // input projection of the next step: 128 rows (this row tile) x 32 frames, K = 88 -> 96 (3 K steps); A
            // fragments from the packed weights (as pw_body), B fragments from XT (conflict-free ds_read_b128)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int r = lane & 31, hi = lane >> 5;
            const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)(s.in_w + (long)mti * 3 * 4096), 0, 3u * 16384u, 0x00020000);
            const int wvo = (hi * 128 + wave * 32 + r) * 16;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 3; ++kc)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wr, wvo, kc * 16384 + g * 4096, 0);
                    const float4 bf = XT[(kc * 8 + g * 2 + hi) * 32 + r];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(u.x), bf.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(u.y), bf.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(u.z), bf.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(u.w), bf.w, acc, 0, 0, 0);
                }
            const int t = f0 + r;
            if (t >= s.T) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p0 = mti * 128 + wave * 32 + 8 * q + 4 * hi;
                if (p0 >= s.Cp) continue;
                const float4 bb = *reinterpret_cast<const float4*>(s.in_b + p0);
                const float4 dd = *reinterpret_cast<const float4*>(s.d2_next + p0);
                const float4 h4 = make_float4(fmaxf(acc[4 * q] + bb.x, 0.f), fmaxf(acc[4 * q + 1] + bb.y, 0.f),
                                              fmaxf(acc[4 * q + 2] + bb.z, 0.f), fmaxf(acc[4 * q + 3] + bb.w, 0.f));
                const float4 hd4 = make_float4(h4.x + dd.x, h4.y + dd.y, h4.z + dd.z, h4.w + dd.w);
                const long off = ((long)(p0 >> 2) * s.T + t) * 4;
                // (hd of the conditional sample is read by T4 of this launch, by other blocks: write-through)
                *reinterpret_cast<float4*>(s.h + (long)pair_i * act_bs + off) = h4;
                store_f4<1>(s.hd + (long)pair_i * act_bs + off, hd4, s.conv_w != nullptr);
                if (paired) {
                    *reinterpret_cast<float4*>(s.h + (long)(pair_i + s.dual) * act_bs + off) = h4;
                    *reinterpret_cast<float4*>(s.hd + (long)(pair_i + s.dual) * act_bs + off) = hd4;
                }
            }
        }
    }
    mark();
    // ---- T4: the next step's shared first-layer conv (pairs only)
    if (paired && s.conv_w) {
        group_barrier<false>(pctr, 2u * (2u * gsize + (unsigned)s.fault), s.err, s.derr);
        mark();
        GemmArgs a{};
        a.d2 = s.zero; a.wt_store = 0;                    // g is consumed by the NEXT launch: plain stores
        a.lds_bytes = s.lds_bytes;
        a.MT = MT; a.NB = s.dual; a.T = s.T; a.alpha = 1.f; a.ksplit = 1;
        a.x_bs = act_bs; a.x_ps = (long)s.T * 4; a.x_fs = 4; a.x_planes = P; a.kchunks = s.Cp >> 5;
        a.y_bs = act_bs; a.y_ps = (long)s.T * 4; a.y_fs = 4; a.y_rows = s.Cp;
        a.Wp = s.conv_w; a.bias = s.conv_b; a.bias2 = s.conv_b2;
        a.X = s.hd; a.taps = s.taps; a.dil = s.dil;
        a.cond = s.cond; a.cond2 = s.cond2; a.c_bs = s.c_bs; a.n_cond = s.dual;
        a.dual = s.dual;
        a.Y = s.g;
        const int tps64 = (s.T + 63) >> 6;
        const int n4 = MT * tps64;
        for (int it = pair_half * (int)gsize + member; it < n4; it += 2 * (int)gsize) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                              // (the LDS tiles of the previous item / of T3 are free)
            gemm_body<1, 1, EPI_GATE, 0, 1>(a, smem, it % MT, pair_i * tps64 + it / MT, 0);
        }
    }
    if (s.dbg && blockIdx.x == 0 && threadIdx.x == 0) s.dbg[7] = clock64();
    // leave: re-arm the counters (nobody polls them any more: everyone passed its last barrier before arriving here)
    if (threadIdx.x == 0) {
        const unsigned left = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gsize - 1) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (paired) {
            const unsigned pleft = __hip_atomic_fetch_add(pctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pleft == 2u * gsize - 1u) {
                __hip_atomic_store(pctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

hipError_t launch_tail(const TailArgs& s, hipStream_t st) {
    if (s.BN != 64 && s.BN != 128) return hipErrorInvalidValue;
    if ((s.Cp & 63) || s.NB < 1 || s.T < 1) return hipErrorInvalidValue;
    if (s.dual > 0 && s.NB != 2 * s.dual) return hipErrorInvalidValue;
    if (!s.x_out || s.x_out == s.u.x) return hipErrorInvalidValue;
    const int tps = (s.T + s.BN - 1) / s.BN, MT = s.Cp >> 6;
    TailArgs b = s;
    const int NBp = xcd_padded_groups(s.NB, MT * tps, &b.xcd_n);       // idle padding groups, as launch_stack
    size_t lds = 24 * 32 * 16;
    if (s.conv_w) {
        if (s.dual <= 0 || (s.taps & 1) == 0) return hipErrorInvalidValue;
        lds = std::max(lds, gemm_lds_bytes(1, 1, s.taps, s.dil, 0, EPI_GATE));
        if (lds > 160 * 1024) return hipErrorInvalidValue;
    }
    b.lds_bytes = (int)lds;
    hipLaunchKernelGGL(tail_kernel, dim3((unsigned)(MT * tps * NBp)), dim3(512), lds, st, b);
    return hipGetLastError();
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

hipError_t launch_reflect_pad(const float* wav, float* out, int B, int L, int pad, hipStream_t s) {
    const int Lp = (L + 2 * pad + 3) & ~3;
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

// per-sample min / max (model/utils.py:25-26) over the n_rows x TF valid values; wavefront shuffles
// then one LDS hop across the 4 waves.
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, float* __restrict__ mm,
                                                     int planes, int TF, int n_rows) {
    const int b = blockIdx.x;
    const float4* xb = reinterpret_cast<const float4*>(x) + (long)b * planes * TF;
    const int vplanes = (n_rows + 3) >> 2;
    float mn = INFINITY, mx = -INFINITY;
    for (long i = threadIdx.x; i < (long)vplanes * TF; i += 256) {
        const int pl = (int)(i / TF);
        const float4 v = xb[i];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (pl * 4 + e < n_rows) { mn = fminf(mn, vv[e]); mx = fmaxf(mx, vv[e]); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[b * 2 + 0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mm[b * 2 + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
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

hipError_t launch_minmax(const float* logmel, float* mm, int B, int planes, int TF, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)B), dim3(256), 0, s, logmel, mm, planes, TF, n_rows);
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
