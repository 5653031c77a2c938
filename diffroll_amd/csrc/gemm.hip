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
#include "gemm_body.h"

namespace dr {

Tuning& tuning() { static Tuning t; return t; }
std::atomic<unsigned>& tuning_epoch() { static std::atomic<unsigned> n{0}; return n; }

DR_BOUNDS_TU(gemm)
hipError_t read_bounds(unsigned long long* out4) {
    unsigned long long t[4];
    hipError_t e;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    hipError_t (*readers[])(unsigned long long*) = {read_bounds_gemm, read_bounds_stack, read_bounds_tail};
    for (auto rd : readers) {
        if ((e = rd(t)) != hipSuccess) return e;
        if (!out4[0] && t[0]) { out4[0] = t[0]; out4[1] = t[1]; out4[2] = t[2]; }
        out4[3] += t[3];
    }
    return hipSuccess;
}
hipError_t reset_bounds() {
    hipError_t e;
    if ((e = reset_bounds_gemm()) != hipSuccess) return e;
    if ((e = reset_bounds_stack()) != hipSuccess) return e;
    return reset_bounds_tail();
}

template <int NI, int KS, int EPI, int PREC, int FOLDP = (NI == 1)>
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
    gemm_body<NI, KS, EPI, PREC, 0, FOLDP>(a, smem, mt, nt, ks);
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
    if (tuning().xcd_n >= 0) return tuning().xcd_n;      // A/B experiments
    if (!tuning().xcd_model) return xbytes > wbytes ? 1 : 0;
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
    const int BN = gemm_block_frames(NI);
    const int FW = BN + 2 * halo;
    return (size_t)2 * (prec ? 12 : 8) * KS * FW * 16 + (epi == EPI_RES_SKIP ? (size_t)32 * BN * 16 : 0);
}

KSplitPlan plan_ksplit(long tiles, int nchunks, int kchunks, int taps, int NI, int prec, size_t ws_floats, size_t ws_cnt_n) {
    const int ks_max = tuning().ksplit_max;
    const long forced_blocks = tuning().ksplit_blocks.load();
    const long max_blocks = forced_blocks ? forced_blocks : (prec ? 256 : 2048);
    const int BN = gemm_block_frames(NI);
    const double t_full = (double)kchunks * taps * 16.0 * (BN / 32) * 69.0 / 2400.0;
    auto cost = [&](int ks) {
        return (double)((tiles * ks + 255) / 256) * t_full / ks + (ks > 1 ? 4.0 + ks : 0.0);
    };
    KSplitPlan p{1, cost(1), cost(1)};
    for (int ks = 2; ks <= ks_max && ks <= 16; ks *= 2) {
        if (tiles * ks > max_blocks || nchunks % ks != 0) break;
        if ((size_t)tiles * ks * 128 * BN > ws_floats || (size_t)tiles * 4 > ws_cnt_n) break;
        const double c = cost(ks);
        if (tiles * ks <= 256 || c < 0.97 * p.us) { p.us = std::min(p.us, c); p.ks = ks; }
    }
    return p;
}

template <int NI, int KS, int EPI, int PREC>
static hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    const int BN = gemm_block_frames(NI);
    const int tps = (a.T + BN - 1) / BN;
    const size_t lds = gemm_lds_bytes(NI, KS, a.taps, a.dil, PREC, EPI);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int NT = a.NB * tps;
    GemmArgs b = a;
    // Split-K: launches that cannot fill the chip, and launches that fill it unevenly.  The ticket reduction needs no
    // co-residency (nobody spins), so a launch may be cut into MORE blocks than the chip holds: 10 evaluations of 125
    // frames are 160 tiles = one round of full-K blocks on 62 % of the CUs; cut 4x in K they are 640 blocks = 3 rounds
    // of quarter-length blocks: 136 -> 107 us per conv launch.  The decision is plan_ksplit's (shared with the engine).
    b.ksplit = 1;
    if (a.ws && a.ws_cnt) b.ksplit = plan_ksplit((long)a.MT * NT, a.kchunks / KS, a.kchunks, a.taps, NI, PREC, a.ws_floats, a.ws_cnt_n).ks;
    b.lds_bytes = (int)lds;
    const dim3 grid((unsigned)(a.MT * NT * b.ksplit));
    // weights: MT*128 rows x 32*kchunks*taps floats; activations: NT*BN frames x 32*kchunks floats
    const double wbytes = 4.0 * 128.0 * a.MT * 32.0 * a.kchunks * a.taps, xbytes = 4.0 * (double)NT * BN * 32.0 * a.kchunks;
    b.xcd_n = pick_xcd_mapping(a.MT, NT, wbytes, xbytes);
    DR_CHECK_EXTENTS(b, EPI, PREC, "gemm_kernel");
    // the 128-frame gated conv exists with and without blocked accumulation (GemmArgs::fold128)
    if constexpr (NI == 2 && KS == 1 && EPI == EPI_GATE) {
        if (a.fold128) {
            hipLaunchKernelGGL((gemm_kernel<2, 1, EPI_GATE, PREC, 1>), grid, dim3(512), lds, s, b);
            return hipGetLastError();
        }
    }
    if constexpr (NI == 1 && KS == 1 && EPI == EPI_GATE && PREC == 0) {
        if (a.nofold64 && b.ksplit == 1) {    // (see GemmArgs::nofold64)
            hipLaunchKernelGGL((gemm_kernel<1, 1, EPI_GATE, 0, 0>), grid, dim3(512), lds, s, b);
            return hipGetLastError();
        }
    }
    if constexpr (NI == 3 || NI == 5) {       // 96 / 160-frame blocks: the gated conv with blocked accumulation only
        hipLaunchKernelGGL((gemm_kernel<NI, 1, EPI_GATE, 0, 1>), grid, dim3(512), lds, s, b);
        return hipGetLastError();
    } else {
        hipLaunchKernelGGL((gemm_kernel<NI, KS, EPI, PREC>), grid, dim3(512), lds, s, b);
        return hipGetLastError();
    }
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
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<2, 1, EPI_GATE, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<1, 1, EPI_GATE, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<2, 1, EPI_GATE, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<5, 1, EPI_GATE, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<3, 1, EPI_GATE, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = init_frontend_kernels()) != hipSuccess) return e;
    if ((e = init_update_kernels()) != hipSuccess) return e;
    if ((e = init_tail_kernels()) != hipSuccess) return e;
    if ((e = init_stack_kernels()) != hipSuccess) return e;
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
    const int ks_force = tuning().one_ks;
    if (ks_force && a.taps == 1 && a.kchunks % ks_force == 0) KS = ks_force;
    if (NI == 1) {
        if (KS == 4) return launch_gemm_ni<1, 4>(a, epi, s);
        return KS == 2 ? launch_gemm_ni<1, 2>(a, epi, s) : launch_gemm_ni<1, 1>(a, epi, s);
    }
    if (NI == 2) {
        if (KS == 4) return launch_gemm_ni<2, 4>(a, epi, s);
        return KS == 2 ? launch_gemm_ni<2, 2>(a, epi, s) : launch_gemm_ni<2, 1>(a, epi, s);
    }
    if (NI == 5 && epi == EPI_GATE && KS == 1) return launch_gemm_t<5, 1, EPI_GATE, 0>(a, s);
    if (NI == 3 && epi == EPI_GATE && KS == 1) return launch_gemm_t<3, 1, EPI_GATE, 0>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace dr
