/*
 * diffroll_amd_debug.h - the LAB side of libdiffroll_amd.so: measurement, checker and test entry points.
 *
 * Nothing here is part of the drop-in boundary (that is diffroll_amd.h, which a maintainer binding the engine under
 * ClassifierFreeDiffRoll reads on its own); bench.py's roofline pass, tools/ and the test-suite use these.  Same
 * conventions as diffroll_amd.h (plain C, borrowed device pointers, 0 or a negative DR_E* code, dr_last_error()).
 */
#ifndef DIFFROLL_AMD_DEBUG_H
#define DIFFROLL_AMD_DEBUG_H

#include "diffroll_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: the FFT stage of dr_frontend on its own - reflect padding + windowed FFT + / sqrt(sum w^2) + |.|^2
 * (torchaudio Spectrogram(center, reflect, normalized=True, power=2) = torch.stft + those two steps;
 * model/diffwave.py:635,643) - d_wav (B, L) -> d_power_out (B, L / hop + 1, n_fft / 2 + 1) row-major.  Lets a test
 * hold the FFT kernel to torch.stft directly.  n_fft must be a power of two.  Synchronises `stream`. */
int dr_debug_stft_power(dr_engine* e, const float* d_wav, int B, int L, float* d_power_out, void* stream);

/* Checker builds only (csrc compiled with -DDR_BOUNDS, tools/checked_build.sh): every hand-computed LDS address and
 * in-range buffer offset of the GEMM kernels and of the fused residual-stack kernel is compared at run time with the
 * region it must stay inside, and every tensor extent a launch will touch with the device allocation it lives in.
 * out4 = {code of the first violated check (0 = none), two details, number of violations} since the last reset.
 * A production build returns DR_ESTATE.  Synchronises the device. */
int dr_debug_bounds(int64_t* out4, int reset);

/* Test hooks of the co-tenant detection (csrc/tenants.h).
 * dr_debug_tenants scans a KFD sysfs tree rooted at kfd_root (the real one is /sys/class/kfd/kfd) for the GPU at PCI
 * (domain, bus, device): out4 = {the driver's gpu_id or -1, processes holding a compute queue on it, the sum of their
 * cu_occupancy, 1 if the proc directory was readable}.  No engine, no GPU.
 * dr_debug_kfd_root points the ENGINES of this process at another tree from their next look on (NULL: the real one) -
 * how a test shows an engine a busy co-tenant that does not exist. */
int dr_debug_tenants(const char* kfd_root, int pci_domain, int pci_bus, int pci_device, int64_t* out4);
int dr_debug_kfd_root(const char* kfd_root);

/*
 * Options outside the boundary (dr_set_option handles the product's own):
 *   "fused_stack_xcd"  [1] block mapping of the fused kernels: 1 = the blocks of a clip share an XCD (and its L2),
 *                          0 = one weight panel per XCD.  Performance only.
 *   "fused_stack_warm" [0] idle waves of the fused stack touch the next phase's weights / conditioner tile so that
 *                          they are L2-resident when needed.  Performance only (measured: 888.6 vs 889.3 ms per
 *                          config-2 chain, i.e. nothing, and 511.6 vs 486.3 ms at config 3).
 *   "stack_ticks"      [0] block 0 records s_memtime at every phase start (dr_stack_status).
 *   "stack_fault_test" [0] ONLY in libraries built with -DDR_FAULT_HOOK (variant "hook" of diffroll_amd/build.py; the
 *                          production library answers DR_ENAME and contains no trace of it): the persistent kernels'
 *                          group barriers await one arrival more than a group has, so the first wait runs into its spin
 *                          bound (~1 s) - the launch ends, flags the time-out, later fused launches return at once,
 *                          dr_finish reports DR_ETIMEOUT and heals.
 *   "tune.<field>"         A/B knobs of the tile / split-K / fused-stack planners and the launchers, PROCESS-wide (they
 *                          apply to every engine of the process from its next launch on; every engine drops its
 *                          captured chain at its next dr_sample): tune.tile (3201 / 3202 / 3203 / 3205 / 1603 / 1605 = MFMA
 *                          size and frame tiles per wave; 0 = cost model), tune.pw, tune.pw_nw, tune.pwk, tune.ksplit_max,
 *                          tune.ksplit_blocks, tune.one_ks, tune.stack3, tune.stack_fl, tune.tail_t4, tune.xcd_n,
 *                          tune.xcd_model, tune.pack_threads, tune.s3_eager, tune.debug_chunks - fields and defaults:
 *                          csrc/kernels.h `Tuning`.  What tools/ and the bit-identity tests pin kernel flavours with
 *                          (tools/tuning_env.py); the library reads NO environment variable.
 * Names of dr_set_option are accepted too.
 */
int dr_debug_set_option(dr_engine* e, const char* name, int value);

/* Synchronises the device.  *timed_out != 0: a group barrier of the fused kernel ran into its spin bound (results
 * of that launch are invalid; never observed in a healthy run) - the counters are reset and the condition cleared WITHOUT
 * the healing dr_finish does.  *launches: fused-kernel launches issued so far (a captured chain counts once, when it is
 * captured).  ticks (optional, n_ticks <= 128): the phase tick marks of the last launch recorded with "stack_ticks". */
int dr_stack_status(dr_engine* e, int32_t* timed_out, int64_t* launches, int64_t* ticks, int n_ticks);

/* Start-up costs of this engine, seconds (a one-shot process - sampling.py: load checkpoint, one batch - pays them once):
 * out5 = {host-side weight packing of the last dr_commit, its uploads, its device-built tables (step embedding),
 * capture + instantiation of the last chain graph, kernel nodes of that graph}. */
int dr_cold_times(dr_engine* e, double* out5);

/* Timing of the dominant kernel inside dr_sample, measured with HIP events on the launch stream when enabled: launches
 * and total milliseconds since the last reset; _ex: plus the ALGORITHMIC FLOPs of the timed launches (SURVEY.md 8d
 * per-frame figures x the frames each launch processed) and the name of the timed kernel - the fused residual-stack
 * kernel when the launch geometry allows it, else the dilated conv + gate kernel. */
int dr_profile_enable(dr_engine* e, int on);
int dr_profile_read(dr_engine* e, int64_t* launches, double* total_ms, int reset);
int dr_profile_read_ex(dr_engine* e, int64_t* launches, double* total_ms, double* total_flops, char* name,
                       size_t name_len, int reset);

/* Standalone launch of the dilated-conv + gate kernel of layer `layer` on the engine's workspace activations
 * (micro-benchmarks / roofline), and the same for the 1x1 output projection + residual / skip kernel (in place on the
 * workspace: repeated launches keep rescaling h, which is harmless for timing). */
int dr_bench_layer(dr_engine* e, int layer, int NB, int T, int t, int n_cond, void* stream);
int dr_bench_pointwise(dr_engine* e, int layer, int NB, int T, void* stream);
/* s_memtime ticks (shader clock) block 0 of the last dr_bench_layer launch spent in its K loop / in total: with the
 * wall time this gives the effective clock the kernel ran at. */
int dr_debug_ticks(dr_engine* e, int64_t* loop_ticks, int64_t* block_ticks);

#ifdef __cplusplus
}
#endif
#endif /* DIFFROLL_AMD_DEBUG_H */
