// Shared by the two persistent kernels (stack.hip, tail.hip): the group barrier, the L2 warm-up helper and the
// group-per-XCD grid padding.
#pragma once
#include "device_common.h"

// Test hook of the time-out path (option "stack_fault_test" of dr_debug_set_option): only libraries built with -DDR_FAULT_HOOK
// (variant "hook", diffroll_amd/build.py) let a launch ask its group barriers for one arrival more than a group has - the
// production library compiles the term away and knows no such option.
#ifdef DR_FAULT_HOOK
#define DR_FAULT_EXTRA(s) ((unsigned)(s).fault)
#else
#define DR_FAULT_EXTRA(s) 0u
#endif

namespace dr {

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

// Group-per-XCD dealing of a persistent launch (block b is dispatched to XCD b % 8): groups g, g + 8, g + 16, ... share
// XCD g, so it needs every XCD's share - ceil(NB / 8) groups - to fit that XCD's CUs.  When NB is not a multiple of 8
// the grid is padded with idle groups (their blocks exit at once) if that still holds; else the launch falls back to
// the spread mapping (groups across all XCDs, write-through hand-offs).
static inline int xcd_padded_groups(int NB, int gsize, int* xcd_n) {
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

}  // namespace dr
