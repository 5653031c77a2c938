// The three GEMM bodies every contraction of the path is made of (device code, templates): gemm_body (LDS-staged X,
// 32x32x2 fp32 / 32x32x16 split-bf16 MFMA), pw_body (1x1, both operands straight from L2) and gemm16_body (16x16x4 MFMA,
// 96 / 160-frame blocks).  Included by gemm.hip (one launch per phase), stack.hip and tail.hip (persistent kernels): the
// same device code everywhere, which is what makes the fused kernels bit-identical to the per-phase launches.
#pragma once
#include "device_common.h"

namespace dr {

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
//
//   Accumulation order (the contraction's numerics): every 32*KS-channel chunk (x all taps: 288 terms of the k = 9
//   conv) is contracted as one k-ordered fp32 MFMA chain starting from ZERO, and the chunk sums are added up in chunk
//   order in a second set of registers ("outer") - blocked accumulation, as the CPU libraries' K-blocked GEMMs do it.
//   One 4608-term chain per output (rounds 1-3) had 3-3.9x the rounding error of the CPU fp32 reference against
//   float64 in the trained-weight regime (profiles/r03_parity_margins.txt); with 288-term blocks a CPU emulation
//   gives 2.1e-5 rms on unit-variance data against 8.3e-5 for the single chain and 2.1e-5 for torch's own matmul.
//   The fold (NW x 16 adds per chunk) sits behind the hand-over barrier, under the latency of the chunk's first
//   fragment reads; the chunk's first MFMAs take a zero C operand, so nothing is cleared.
//
//   FOLDP: blocked accumulation wanted.  It applies to the gated conv (EPI_GATE: the K = taps x C contraction; every other
//   GEMM of the path has K <= 1056 and keeps one chain, bit-identical to pw_body).  Cost, with the fold pinned behind the
//   hand-over barrier (see fold()): +0.6 % per config-3 chain on 64-frame blocks, +0.2-0.6 % per config-2 chain on 128-row x
//   128-frame blocks (208 registers with the in-place fragment refresh below) - on by default in both (engine option
//   "blocked_accumulation": 1 = 128-frame blocks keep one chain).  profiles/r04_conv_flavour_ab.txt.
template <int NI, int KS, int EPI, int PREC, int COH, int FOLDP = (NI == 1)>
DR_DEVINL void gemm_body(const GemmArgs& a, char* smem, const int mt, const int nt, const int ks) {
    // NI = 1 / 2: 64 / 128-frame blocks (2 / 4 32-frame MFMA tiles per consumer wave); NI = 3 / 5: THREE / FIVE tiles = 96 /
    // 160-frame blocks, the dilated conv only (640-frame geometries: 4 x 160-frame blocks per clip, 8 evaluations x 4 x 8 M
    // tiles = 256 blocks; ragged lengths: 96) - fp32 with blocked accumulation, in place of the 16x16-MFMA kernels of those
    // widths, which have no blocked form
    constexpr int NW = (NI == 3 || NI == 5) ? NI : 2 * NI;    // 32-frame MFMA tiles per wave
    constexpr int BN = 32 * NW;
    static_assert(NI == 1 || NI == 2 || ((NI == 3 || NI == 5) && EPI == EPI_GATE && PREC == 0 && KS == 1 && FOLDP == 1),
                  "block widths: 64, 128; 96, 160 (conv)");
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
    constexpr int WROWS = MI * 32;                // rows per wave
    constexpr int WFR = NW * 32;                  // frames per wave

    int tid = threadIdx.x;
    // COH (the persistent kernels call this body once per phase inside a loop): everything this call derives from the lane
    // index must be computed INSIDE the call.  Without the opaque barrier the compiler hoists the per-lane address
    // arithmetic of every phase body out of the phase loop - dozens of registers that then live through the K loops of the
    // other phases (stack_kernel<2> with blocked accumulation: 256 VGPRs + 232 B of scratch before).
    if constexpr (COH) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long tick0 = a.dbg ? clock64() : 0;   // measurement hook (null in production launches)

    const int halo = ((a.taps - 1) >> 1) * a.dil;
    const int FW = BN + 2 * halo;
    float4* Xs = reinterpret_cast<float4*>(smem);   // [2 buffers][XP][FW]
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
    const int cps = a.kchunks / KS / a.ksplit;      // chunks (hand-overs) of this block: [c0, c1)
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
                // (COH: the tile was written by THIS workgroup in an earlier phase of the same launch - bypass the L1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(Rs + pl * BN + seg * 64), 16, voff, 0, 0, COH ? 16 : 0);
            }
        }
        issue(c0);
        for (int chunk = c0; chunk < c1; ++chunk) {
            // hand-over #chunk: this wave's DMA of tile #chunk must have LANDED before the barrier releases the
            // consumers - barriers do not drain VMEM, and hipcc does not reliably insert the wait for a
            // __syncthreads() behind LDS-DMA builtins (it did in the stand-alone kernels and did NOT in the fused
            // ones: tools/isa_audit.py; the consumers then read the previous occupant of the buffer whenever the tile
            // was slower than their own first weight fragments - observed with cross-XCD hand-offs).  Hence explicit.
            // The consumers' matching barrier opens their chunk; only then may the OTHER buffer be refilled (the
            // consumers finished reading it before they arrived here).
#if DR_FAULT != 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#else                       // litmus build 1: the hand-over WITHOUT the wait - a bare s_barrier, so that hipcc's own fence
            __builtin_amdgcn_s_barrier();       // handling cannot put it back (which kernels of rounds 1-2 had it was luck)
#endif
            if (chunk + 1 < c1) issue(chunk + 1);
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
    // blocked accumulation (see the header): acc = the running chunk's chain, outer = the sum of the finished chunks
    static_assert(MI == 1, "one 32-row MFMA tile per consumer wave");
    f32x16 outer[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) outer[ni][e] = 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr bool FOLD = FOLDP && EPI == EPI_GATE;
    auto fold = [&]() {
        if constexpr (FOLD) {
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) outer[ni] = outer[ni] + acc[0][ni];
            // The adds happen HERE.  Their results are only needed after the K loop, and left alone the compiler sinks
            // them to the bottom of the chunk - which keeps the previous chain alive through the whole chunk in a THIRD
            // register set (16*NW copies per chunk behind the last MFMAs; at NW = 4 also what pushed the kernel into
            // scratch).  An opaque use pins them in front of the chunk's first MFMAs, which then reuse acc's registers.
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) asm volatile("" : "+v"(outer[ni]));
        }
    };

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
                DR_CHECK(slab >= 0 && wvo + slab * 24576 + gp * 4096 + 16 <= NS * 24576, 103, slab, NS);
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, slab * 24576 + gp * 4096, 0);
                o.v[gp * 2] = make_uint4(u.x, u.y, u.z, u.w);
            }
            return o;
        };
        A12 wA = load_a3(c0 * KS * a.taps), wB;
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
        auto mma6 = [&](auto FIRST, const uint4 a0, const uint4 a1, const uint4 a2, const BF3& bf) {
            constexpr bool kFirst = FOLD && decltype(FIRST)::value;      // the chunk's first products: C = 0 (a new chain)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[0][ni] = mma_bf16(a2, bf.v[0][ni], kFirst ? zero16 : acc[0][ni]);
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
        auto step = [&](auto ROLE, auto FIRST, int slab, int chunk, int q) {
            constexpr bool kB = decltype(ROLE)::value;
            const uint4* Xb = xaddr(chunk, q);
            if constexpr (kB) wA = load_a3(min(slab + 1, NS - 1));
            else wB = load_a3(min(slab + 1, NS - 1));
            b1 = rd3(Xb, 1);
            mma6(FIRST, kB ? wB.v[0] : wA.v[0], kB ? wB.v[2] : wA.v[2], kB ? wB.v[4] : wA.v[4], b0);
            b0 = rd3(xaddr(chunk, min(q + 1, per_chunk - 1)), 0);
            mma6(std::false_type{}, kB ? wB.v[6] : wA.v[6], kB ? wB.v[8] : wA.v[8], kB ? wB.v[10] : wA.v[10], b1);
            sgb_mix<3 * NW, 6>();           // + the 6 A-fragment loads of the next step, one per MFMA pair
            sgb_mix<3 * NW, 0>();
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        for (int chunk = c0; chunk < c1; ++chunk) {
            auto at = [&](auto R, auto FIRST, int q) {
                const int j = q / KS, sub = q - j * KS;
                step(R, FIRST, (chunk * KS + sub) * a.taps + j, chunk, q);
            };
            __syncthreads();
            b0 = rd3(xaddr(chunk, 0), 0);
            fold();                       // the previous chunk's chain joins the outer sum (zeros the first time)
            at(F_{}, T_{}, 0);
            int q = 1;
            for (; q + 2 <= per_chunk; q += 2) {
                at(T_{}, F_{}, q);
                at(F_{}, F_{}, q + 1);
            }
            if (q < per_chunk) at(T_{}, F_{}, q);
            else wA = wB;
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
    auto mma4 = [&](auto FIRST, const float4 af, const BF& bf) {
        constexpr bool kFirst = FOLD && decltype(FIRST)::value;  // the chunk's first MFMAs: C = 0 (a new chain)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.v[ni].x, kFirst ? zero16 : acc[0][ni], 0, 0, 0);
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
    auto step = [&](auto ROLE, auto FIRST, int slab, int chunk, int q) {
        constexpr bool kB = decltype(ROLE)::value;
        const float4* Xb = xaddr(chunk, q);
        if constexpr (kB) wA = load_a(min(slab + 1, NS - 1));
        else wB = load_a(min(slab + 1, NS - 1));
        b1 = rd(Xb, 1);
        mma4(FIRST, kB ? wB.v[0] : wA.v[0], b0);
        b0 = rd(Xb, 2);
        mma4(std::false_type{}, kB ? wB.v[2] : wA.v[2], b1);
        b1 = rd(Xb, 3);
        mma4(std::false_type{}, kB ? wB.v[4] : wA.v[4], b0);
        b0 = rd(xaddr(chunk, min(q + 1, per_chunk - 1)), 0);
        mma4(std::false_type{}, kB ? wB.v[6] : wA.v[6], b1);
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
    // 128-frame blocks (NI = 2) refresh the A fragments IN PLACE instead: group g's fragment register quad receives the
    // next step's group g right behind the group's last MFMA (one buffer load per group: prefetch distance three
    // groups = 48 MFMAs) - ONE fragment set instead of two, which is what lets the second accumulator set of the
    // blocked accumulation (64 registers at NW = 4) live in the K loop without spilling; no roles, no per-chunk copy.
    constexpr bool AINP = (NI == 2 || NI == 5) && FOLD;
    auto load_ag = [&](int slab, int g) -> float4 {
        DR_CHECK(slab >= 0 && wvo + slab * 16384 + g * 4096 + 16 <= NS * 16384, 112, slab, NS);
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, slab * 16384 + g * 4096, 0);
        return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    };
    auto stepi = [&](auto FIRST, int slab, int chunk, int q) {
        const float4* Xb = xaddr(chunk, q);
        const int nx = min(slab + 1, NS - 1);
        b1 = rd(Xb, 1);
        mma4(FIRST, wA.v[0], b0);
        wA.v[0] = load_ag(nx, 0);
        b0 = rd(Xb, 2);
        mma4(F_{}, wA.v[2], b1);
        wA.v[2] = load_ag(nx, 1);
        b1 = rd(Xb, 3);
        mma4(F_{}, wA.v[4], b0);
        wA.v[4] = load_ag(nx, 2);
        b0 = rd(xaddr(chunk, min(q + 1, per_chunk - 1)), 0);
        mma4(F_{}, wA.v[6], b1);
        wA.v[6] = load_ag(nx, 3);
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>(); sgb<0x20, 1>();
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>(); sgb<0x20, 1>();
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>(); sgb<0x20, 1>();
        sgb<0x100, NW>(); sgb<0x8, 4 * NW>(); sgb<0x20, 1>();
    };

    // Steps of a chunk, q = 0 .. per_chunk-1, in memory order of the slabs ([32-channel kchunk][tap]):
    // tap-major, sub-chunk minor.  Roles alternate A,B,A,...; a chunk always starts in role A (one
    // register copy per chunk when per_chunk is odd).
    for (int chunk = c0; chunk < c1; ++chunk) {
        auto slab_of = [&](int q) {
            const int j = q / KS, sub = q - j * KS;
            return (chunk * KS + sub) * a.taps + j;
        };
        auto at = [&](auto R, auto FIRST, int q) { step(R, FIRST, slab_of(q), chunk, q); };
        __syncthreads();   // X tile #chunk staged by the producers (matches their hand-over barrier)
        if (a.dbg && blockIdx.x == 0 && tid == 0 && chunk < 14) a.dbg[2 + chunk] = clock64() - tick0;
        b0 = rd(xaddr(chunk, 0), 0);
        fold();                           // the previous chunk's chain joins the outer sum (zeros the first time)
        // (the first step's pinned schedule starts HERE: without the fence its first sched group adopts the reads of b0
        // above, and every fragment read of the step is issued one group late - right in front of the MFMAs that wait for it)
        if constexpr (FOLD) __builtin_amdgcn_sched_barrier(0);
        if constexpr (AINP) {
            stepi(T_{}, slab_of(0), chunk, 0);      // C = 0: a new chain
            int q = 1;
            for (; q + 2 <= per_chunk; q += 2) {
                stepi(F_{}, slab_of(q), chunk, q);
                stepi(F_{}, slab_of(q + 1), chunk, q + 1);
            }
            if (q < per_chunk) stepi(F_{}, slab_of(q), chunk, q);
        } else {
            at(F_{}, T_{}, 0);                // C = 0: a new chain
            int q = 1;
            for (; q + 2 <= per_chunk; q += 2) {
                at(T_{}, F_{}, q);
                at(F_{}, F_{}, q + 1);
            }
            if (q < per_chunk) at(T_{}, F_{}, q);
            else wA = wB;                     // (an odd number of steps ends in role A: the next fragments sit in set B.
                                              // Swapping the roles across chunks instead - two copies of the chunk body -
                                              // was measured: 505 vs 483 ms per config-3 chain, spills at 128 frames)
        }
    }

    }
    // the last chunk's chain; from here on acc holds the block's (partial) sums
    if constexpr (FOLD) {
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[0][ni] = outer[ni] + acc[0][ni];
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
                        store_s3_quad<COH>(a.Y + (long)be * a.y_bs, o, c0, t, a.T, a.y_rows >> 3, a.wt_store);
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
                                        store_s3_quad<COH>(a.Y2 + (long)be * a.y2_bs, o2, p0, t, a.T, a.y_rows >> 3, a.wt_store);
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
                                        store_s3_quad<COH>(a.Y2 + (long)be * a.y2_bs, o2, p0, t, a.T, a.y_rows >> 3, a.wt_store);
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
    int lane = threadIdx.x & 63;
    if constexpr (COH) asm volatile("" : "+v"(lane));      // (as gemm_body: no per-lane value of this call may be hoisted out of a phase loop)
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

}  // namespace dr
