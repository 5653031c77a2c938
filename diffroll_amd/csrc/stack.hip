// gfx950 (MI355X, CDNA4): the fused residual stack - all residual layers of one network evaluation as ONE persistent
// launch (model/diffwave.py:134-151 x residual_layers, the loop at :678-681).  Device code of the phases: gemm_body.h.
#include "gemm_body.h"
#include "persistent.h"

namespace dr {

DR_BOUNDS_TU(stack)

// FL = block flavour (on the 32x32x2 MFMA): 1 / 2 = 128 packed rows x 64 / 128 frames (gemm_body<FL>); 5 = 128 packed rows x
// 160 frames (gemm_body<5>: five 32-frame MFMA tiles per consumer wave, blocked accumulation only) for the 640-frame
// geometries - 8 evaluations x 4 frame tiles x 8 M tiles = 256 blocks = one resident round, one 32-block group per XCD; its
// 1x1 phases run on all eight waves, waves 0-3 on frames [0, 96) and waves 4-7 on frames [96, 160) of the block's tile.
// (Two more flavours were built, measured and removed: 160-frame blocks on the 16x16x4 MFMA for 640-frame clips, rounds
// 2-3 - never faster than the per-phase launches there, 92 spilled registers; and HALF tiles of 64 packed rows x 128
// frames with K split over the block's wave pairs, round 4, for BASELINE config 3's 16 evaluations x 125 frames - every
// wave then runs the 128-frame instruction stream, but a 64-row block stages the same X tile for half the MFMAs and the
// doubled LDS-DMA traffic makes it 4 % slower than 64-frame blocks: profiles/r04_conv_flavour_ab.txt.)
// FOLDP: blocked accumulation in the conv phases (gemm_body.h) - always for flavour 1, on request for flavour 2.
// PREC = 1: the split-bf16 ("S3") precision - s.hd / s.g are the S3 tensors hd3 / g3 ([sample][piece][channel/8][frame][8
// bf16]); the conv phase is gemm_body<.., PREC = 1> (S3 in, S3 out), the 1x1 phase the LDS-staged S3 GEMM (gemm_body<1, 4,
// EPI_RES_SKIP, 1>) on the block's 64-frame tile(s), which re-reads its h / skip tile from global every layer (nothing is
// LDS-resident here: the S3 X tiles of a 128-channel 1x1 chunk need the space); same device code as the per-phase launches
// of that precision: bit-identical to them.  Needs Cp % 128 == 0.
template <int FL, int FOLDP = 1, int PREC = 0>
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
    static_assert(FL == 1 || FL == 2 || (FL == 5 && FOLDP == 1 && PREC == 0), "block flavours");
    constexpr int BN = FL == 5 ? 160 : 64 * FL;
    constexpr int RP = 32;                                 // planes (4 rows each) of the block's resident tile
    constexpr int RWL = (BN + 63) / 64;                    // 64-frame segments of a tile row
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int MT = s.Cp >> 6;                              // 128-row M tiles of the 2 Cp packed rows
    const int MB = MT;                                     // blocks per frame tile
    const int tps = (s.T + BN - 1) / BN;
    const unsigned gsize = (unsigned)(MB * tps);          // blocks per group
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
    mt = member % MB;                                     // sample (clip evaluation) grp = barrier group
    nt = grp * tps + member / MB;
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
    if constexpr (PREC) {
        if (pending_timeout) return;
    } else {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const int lane = threadIdx.x & 63;
        for (int i = wave; i < RP * RWL; i += 8) {
            const int pl = i / RWL, seg = i - pl * RWL;
            bool is_res;
            const float* src = tile_plane(pl, is_res);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)s.T * 16u, 0x00020000);
#ifdef DR_BOUNDS
            if (seg * 64 + lane < BN) DR_CHECK_LDS(Rs + pl * BN + seg * 64 + lane, s.rs_off, s.rs_off + RP * BN * 16, 131);
            if (threadIdx.x == 0 && i == 0) DR_CHECK(s.rs_off + RP * BN * 16 + 16 <= s.lds_bytes && lds_off(smem) == 0u, 132, s.rs_off, s.lds_bytes);
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
    unsigned& same_xcd_s = *reinterpret_cast<unsigned*>(Rs + RP * BN);      // one word behind the resident tile
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
        if constexpr (PREC) {
            const long s3_bs = act_bs + act_bs / 2;          // per-sample size of an S3 tensor (4-byte units)
            if ((p & 1) == 0) {
                a.Wp = ly.conv_w; a.bias = ly.conv_b; a.bias2 = ly.conv_b2;
                a.X = s.hd; a.x_bs = s3_bs; a.x_piece = (long)(s.Cp >> 3) * s.T * 4; a.x_planes = s.Cp >> 3;
                a.taps = s.taps; a.dil = ly.dil;
                a.cond = ly.cond; a.cond2 = ly.cond2; a.c_bs = s.c_bs; a.n_cond = s.n_cond;
                a.Y = s.g; a.y_bs = s3_bs; a.out_s3 = 1;
                gemm_body<FL, 1, EPI_GATE, 1, 1, FOLDP>(a, smem, mt, nt, 0);
            } else {
                a.Wp = ly.out_w; a.bias = ly.out_b;
                a.X = s.g; a.x_bs = s3_bs; a.x_piece = (long)(s.Cp >> 3) * s.T * 4; a.x_planes = s.Cp >> 3;
                a.taps = 1; a.dil = 1;
                a.Y = s.h;
                const bool last = (l + 1 == s.L);
                if (!last) {
                    a.Y2 = s.hd; a.y2_bs = s3_bs; a.out_s3 = 2;
                    a.d2 = s.d2 + (long)(l + 1) * s.Cp; a.tsel = s.tsel; a.d2_ts = s.d2_ts;
                }
                a.skip = s.skip; a.s_bs = act_bs; a.skip_init = (l == 0);
                const bool idle = last && mt < (s.Cp >> 7);
                // the block's BN-frame tile as 64-frame tiles of the S3 1x1 body (one hand-over chunk = 128 channels)
                const int tps64 = (s.T + 63) >> 6;
                const int b_ = nt / tps, ti = nt % tps;
#pragma unroll 1
                for (int half = 0; half < BN / 64; ++half) {
                    const int t64 = ti * (BN / 64) + half;
                    if (half) {          // (the LDS regions of the previous call are free once every wave is past its epilogue)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                    }
                    if (!idle && t64 < tps64) gemm_body<1, 4, EPI_RES_SKIP, 1, 1>(a, smem, mt, b_ * tps64 + t64, 0);
                }
            }
        } else
        if ((p & 1) == 0) {
            a.Wp = ly.conv_w; a.bias = ly.conv_b; a.bias2 = ly.conv_b2;
            a.X = s.hd; a.taps = s.taps; a.dil = ly.dil;
            a.cond = ly.cond; a.cond2 = ly.cond2; a.c_bs = s.c_bs; a.n_cond = s.n_cond;
            a.Y = s.g;
            if (s.dbg && p + 2 >= s.p1) a.dbg = s.dbg + 64;       // last conv phase: body tick marks of block 0
            gemm_body<FL, 1, EPI_GATE, 0, 1, FOLDP>(a, smem, mt, nt, 0);
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
            // 128-frame flavour: all eight waves contract the 1x1 - the producers have nothing to stage in this phase, so
            // waves w and w + 4 share the rows of wave w and split its frames (two MFMA streams per SIMD cover each
            // other's fragment waits and epilogue; 64 fewer live registers in the merged kernel: 204 instead of 256 + 16 B
            // of scratch).  Same k order per output: bit-identical.  Measured at config 2: 1x1 phase 76.9 k -> 73.7 k
            // cycles, chain 883.1 -> 879.3 ms.  The 64-frame flavour keeps four waves: with one 32-frame tile per wave
            // the phase takes the same cycles at a lower clock (485.4 vs 483.3 ms).
            if constexpr (FL == 2) {
                if (!idle) pw_body<2, 1, 1, 128>(a, mt, nt, wave & 3, Rs, (wave >> 2) * 64);
            } else if constexpr (FL == 5) {
                // 160-frame flavour: waves 0-3 take frames [0, 96) (three MFMA tiles), waves 4-7 frames [96, 160) (two) of
                // the same rows - five tiles per SIMD as in the conv phase, shared by two MFMA streams
                if (!idle) {
                    if (wave < 4) pw_body<3, 1, 1, 160>(a, mt, nt, wave, Rs, 0);
                    else pw_body<2, 1, 1, 160>(a, mt, nt, wave - 4, Rs, 96);
                }
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
            group_barrier<false>(ctr, ++episode * (gsize + DR_FAULT_EXTRA(s)), s.err, s.derr);
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
    if constexpr (!PREC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int lane = threadIdx.x & 63;
        for (int i = wave; i < RP * RWL; i += 8) {
            const int pl = i / RWL, seg = i - pl * RWL;
            bool is_res;
            float* dst = tile_plane(pl, is_res);
            const int t = t0_ + seg * 64 + lane;
#ifdef DR_BOUNDS
            if (seg * 64 + lane < BN) DR_CHECK_LDS(Rs + pl * BN + seg * 64 + lane, s.rs_off, s.rs_off + RP * BN * 16, 133);
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

hipError_t launch_stack(const StackArgs& s, int FL, int max_dil, hipStream_t st, int prec) {
    if (FL != 1 && FL != 2 && FL != 5) return hipErrorInvalidValue;
    if (prec && ((s.Cp & 127) || FL == 5)) return hipErrorInvalidValue;
    if (s.L < 1 || s.L > DR_STACK_MAX_LAYERS || s.p0 < 0 || s.p1 > 2 * s.L || s.p0 >= s.p1 || (s.Cp & 63)) return hipErrorInvalidValue;
    const int BN = stack_tile_frames(FL), MT = s.Cp >> 6, gsize = stack_group_blocks(FL, s.Cp, s.T);
    const size_t lds = prec ? stack3_lds_bytes(FL, s.taps, max_dil) : stack_lds_bytes(FL, s.taps, max_dil);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    StackArgs b = s;
    b.rs_off = (int)(lds - 16 - (size_t)32 * BN * 16);
    b.lds_bytes = (int)lds;
    const int NBp = xcd_padded_groups(s.NB, gsize, &b.xcd_n);
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
    (void)MT;
    const dim3 grid((unsigned)(gsize * NBp));
    if (prec) {
        if (FL == 1) hipLaunchKernelGGL((stack_kernel<1, 1, 1>), grid, dim3(512), lds, st, b);
        else if (s.fold128) hipLaunchKernelGGL((stack_kernel<2, 1, 1>), grid, dim3(512), lds, st, b);
        else hipLaunchKernelGGL((stack_kernel<2, 0, 1>), grid, dim3(512), lds, st, b);
        return hipGetLastError();
    }
    if (FL == 1) hipLaunchKernelGGL((stack_kernel<1>), grid, dim3(512), lds, st, b);
    else if (FL == 5) hipLaunchKernelGGL((stack_kernel<5>), grid, dim3(512), lds, st, b);
    else if (FL == 2 && s.fold128) hipLaunchKernelGGL((stack_kernel<2, 1>), grid, dim3(512), lds, st, b);
    else if (FL == 2) hipLaunchKernelGGL((stack_kernel<2, 0>), grid, dim3(512), lds, st, b);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
int stack_tile_frames(int FL) { return FL == 5 ? 160 : 64 * FL; }
// split-bf16 flavour: max over its two phase bodies (conv: S3 X tiles; 1x1: 128-channel S3 X tiles + the 64-frame h / skip tile)
size_t stack3_lds_bytes(int FL, int taps, int max_dil) {
    return std::max(gemm_lds_bytes(FL, 1, taps, max_dil, 1, EPI_GATE), gemm_lds_bytes(1, 4, 1, 1, 1, EPI_RES_SKIP)) + 16;
}
// blocks of one clip evaluation (= one barrier group): M tiles x frame tiles
int stack_group_blocks(int FL, int Cp, int T) {
    const int BN = stack_tile_frames(FL);
    return (Cp >> 6) * ((T + BN - 1) / BN);
}
// the conv's double-buffered X tiles + the resident h / skip tile
size_t stack_lds_bytes(int FL, int taps, int max_dil) {
    const int BN = stack_tile_frames(FL), halo = ((taps - 1) / 2) * max_dil;
    return (size_t)2 * 8 * (BN + 2 * halo) * 16 + (size_t)32 * BN * 16 + 16;     // + one flag word (16-byte slot)
}

hipError_t init_stack_kernels() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<2, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_kernel<2, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace dr
