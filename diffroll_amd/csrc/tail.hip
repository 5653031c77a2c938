// gfx950 (MI355X, CDNA4): the tail of a reverse step as one persistent launch (model/diffwave.py:682-686 +
// task/diffusion.py:953-967 + the next step's model/diffwave.py:667-668 and shared first-layer conv).
#include "gemm_body.h"
#include "persistent.h"
#include "update_quad.h"

namespace dr {

DR_BOUNDS_TU(tail)

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
    group_barrier<true>(ctr, gsize + DR_FAULT_EXTRA(s), s.err, s.derr);
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
    if (paired) group_barrier<true>(pctr, 2u * gsize + DR_FAULT_EXTRA(s), s.err, s.derr);
    else group_barrier<true>(ctr, 2u * (gsize + DR_FAULT_EXTRA(s)), s.err, s.derr);
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
        group_barrier<false>(pctr, 2u * (2u * gsize + DR_FAULT_EXTRA(s)), s.err, s.derr);
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
        // item width (launch_tail): 64 frames, or 96 where that needs fewer x narrower rounds over the pair's blocks (640-frame
        // clips in 32-block groups: 56 items of 96 frames = one round instead of 80 items of 64 frames = two)
        const int bn4 = s.t4_ni == 3 ? 96 : 64;
        const int tps4 = (s.T + bn4 - 1) / bn4;
        const int n4 = MT * tps4;
        for (int it = pair_half * (int)gsize + member; it < n4; it += 2 * (int)gsize) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                              // (the LDS tiles of the previous item / of T3 are free)
            // (same accumulation order as the stack launch that consumes g: blocked unless that is the unblocked 128-frame flavour)
            if (s.t4_ni == 3) gemm_body<3, 1, EPI_GATE, 0, 1, 1>(a, smem, it % MT, pair_i * tps4 + it / MT, 0);
            else if (s.fold) gemm_body<1, 1, EPI_GATE, 0, 1, 1>(a, smem, it % MT, pair_i * tps4 + it / MT, 0);
            else gemm_body<1, 1, EPI_GATE, 0, 1, 0>(a, smem, it % MT, pair_i * tps4 + it / MT, 0);
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
    if (s.BN != 64 && s.BN != 128 && s.BN != 160) return hipErrorInvalidValue;
    if ((s.Cp & 63) || s.NB < 1 || s.T < 1) return hipErrorInvalidValue;
    if (s.dual > 0 && s.NB != 2 * s.dual) return hipErrorInvalidValue;
    if (!s.x_out || s.x_out == s.u.x) return hipErrorInvalidValue;
    const int tps = (s.T + s.BN - 1) / s.BN, MT = s.Cp >> 6;
    TailArgs b = s;
    const int NBp = xcd_padded_groups(s.NB, MT * tps, &b.xcd_n);       // idle padding groups, as launch_stack
    size_t lds = 24 * 32 * 16;
    if (s.conv_w) {
        if (s.dual <= 0 || (s.taps & 1) == 0) return hipErrorInvalidValue;
        // T4's item width: rounds over the pair's 2 * gsize blocks x frames per item, 64 unless 96 is strictly cheaper (the
        // 96-frame body exists with blocked accumulation only)
        const long pair_blocks = 2L * MT * tps;
        const long n64 = (long)MT * ((s.T + 63) / 64), n96 = (long)MT * ((s.T + 95) / 96);
        const long c64 = (n64 + pair_blocks - 1) / pair_blocks * 64, c96 = (n96 + pair_blocks - 1) / pair_blocks * 96;
        b.t4_ni = (s.fold && s.t4_ni != 1 && (c96 < c64 || s.t4_ni == 3)) ? 3 : 1;
        lds = std::max(lds, gemm_lds_bytes(b.t4_ni, 1, s.taps, s.dil, 0, EPI_GATE));
        if (lds > 160 * 1024) return hipErrorInvalidValue;
    }
    b.lds_bytes = (int)lds;
    hipLaunchKernelGGL(tail_kernel, dim3((unsigned)(MT * tps * NBp)), dim3(512), lds, st, b);
    return hipGetLastError();
}

hipError_t init_tail_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace dr
