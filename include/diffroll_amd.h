/*
 * diffroll_amd.h - C-ABI of the MI355X-native DiffRoll sampling engine.
 *
 * The reference (sony/DiffRoll) has NO plugin / FFI layer: its sampling path sits behind the
 * Python methods of a LightningModule.  This header is therefore the boundary a maintainer
 * would bind UNDER those methods (ctypes stub: INTEGRATION.md).  Each entry point names the
 * reference interface it replaces (paths relative to the reference checkout):
 *
 *   dr_create / dr_set_param / dr_commit   ClassifierFreeDiffRoll.__init__ + load_from_checkpoint
 *                                          (model/diffwave.py:580-635, sampling.py:54-65) and the
 *                                          schedule of SpecRollDiffusion.__init__
 *                                          (task/diffusion.py:239-256)
 *   dr_frontend                            mel_layer -> log -> normalize_spec -> inpainting mask ->
 *                                          trim (model/diffwave.py:643-662, model/utils.py:21-32)
 *   dr_forward                             ClassifierFreeDiffRoll.forward after the front-end
 *                                          (model/diffwave.py:664-686, ResidualBlock :134-151)
 *   dr_step                                cfdg_ddpm_x0 / generation_ddpm_x0 / inpainting_ddpm_x0 /
 *                                          ddpm_x0 (task/diffusion.py:943-1025, :831-853)
 *   dr_sample / dr_sample_checked          the loop of predict_step / sampling
 *                                          (task/diffusion.py:528-534, :779-788)
 *
 * Conventions: plain C, no torch types.  All tensor arguments are BORROWED device pointers to
 * contiguous fp32 (hipMalloc'd / torch ROCm memory on the engine's device); outputs are written
 * into caller-allocated buffers.  `stream` is a hipStream_t passed as void* (e.g.
 * torch.cuda.current_stream().cuda_stream).  Every function returns 0 on success or a negative
 * DR_E* code and never throws; the message is available from dr_last_error().  An engine handle
 * is not re-entrant: one handle per (device, stream), one host thread at a time.
 *
 * This header is the WHOLE boundary (30 functions).  Measurement, checker and test entry points of the same library
 * (dr_profile_*, dr_bench_*, dr_debug_*, dr_stack_status, dr_cold_times, the A/B options "tune.*") are declared in
 * diffroll_amd_debug.h; nothing on the sampling path needs them.
 */
#ifndef DIFFROLL_AMD_H
#define DIFFROLL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_ABI_VERSION 10

enum {
    DR_OK = 0,
    DR_EINVAL = -1,   /* bad argument / shape */
    DR_ESTATE = -2,   /* call order (e.g. dr_forward before dr_commit / dr_frontend) */
    DR_EHIP = -3,     /* a HIP runtime call failed (message has the HIP error string) */
    DR_ENOMEM = -4,
    DR_ENAME = -5,    /* unknown parameter name / wrong shape in dr_set_param */
    DR_ETIMEOUT = -6  /* a group barrier of the fused residual-stack kernel ran into its spin bound: the results
                         computed since the last dr_finish are invalid (see dr_finish / dr_sample_checked) */
};

/* samplers: task/diffusion.py, bound at :255 by hparams.sampling.type */
enum {
    DR_SAMPLER_DDPM_X0 = 0,        /* :831-853  one conditional evaluation            */
    DR_SAMPLER_CFDG_DDPM_X0 = 1,   /* :943-969  conditional + unconditional, weight w */
    DR_SAMPLER_GENERATION_DDPM_X0 = 2, /* :971-997  one unconditional evaluation (spec = -1) */
    DR_SAMPLER_INPAINTING_DDPM_X0 = 3, /* :999-1025 as cfdg; spectrogram frames/bins masked by
                                          dr_frontend's mask arguments */
    /* the remaining samplers = the same kernels with other per-step coefficients */
    DR_SAMPLER_DDIM_X0 = 4,            /* :855-875   x0 update with sigma = 0                      */
    DR_SAMPLER_CFDG_DDIM_X0 = 5,       /* :1027-1055 as ddim_x0 with guidance; its second branch is
                                          forward(zero waveform) WITHOUT sampling=True: spec == 0  */
    DR_SAMPLER_DDPM_EPS = 6,           /* :804-829   network output is epsilon ("ddpm")            */
    DR_SAMPLER_DDIM_EPS = 7,           /* :877-892   ("ddim")                                      */
    DR_SAMPLER_DDIM2DDPM_EPS = 8       /* :894-911   ("ddim2ddpm")                                 */
};

/* coefficient families of dr_set_tables' `coef` argument */
enum {
    DR_COEF_DDPM_X0 = 0,   /* [sqrt_acp[t-1], sqrt(1 - sqrt_acp[t-1]^2 - sigma^2), sqrt_acp[t], sqrt_1m_acp[t], sigma] */
    DR_COEF_DDIM_X0 = 1,   /* same with sigma = 0                                                          */
    DR_COEF_DDPM_EPS = 2,  /* [sqrt_recip_alphas[t], betas[t], sqrt_1m_acp[t], sqrt(posterior_variance[t]), 0] */
    DR_COEF_DDIM_EPS = 3,  /* [sqrt_acp[t-1], sqrt_1m_acp[t-1], sqrt_acp[t], sqrt_1m_acp[t], 0]            */
    DR_COEF_DDIM2DDPM_EPS = 4, /* [sqrt_acp[t-1], sqrt(1 - sqrt_acp[t-1]^2 - sigma^2), sqrt_acp[t], sqrt_1m_acp[t], sigma] */
    DR_COEF_FAMILIES = 5
};

/* arithmetic of the two hot contractions (dilated conv, 1x1 output projection); everything else is fp32 */
enum {
    DR_PRECISION_F32 = 0,     /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): the default                     */
    DR_PRECISION_BF16X3 = 1   /* opt-in: each fp32 operand is split EXACTLY into three bf16 pieces and a
                                 product is formed from the six piece products with i + j <= 2 on the bf16
                                 MFMA with fp32 accumulation (drops terms <= 2^-24 |ab|); fp32-level error
                                 at 2.67x the matrix rate                                                  */
};

/* which spectrogram a dr_forward evaluation sees (model/diffwave.py:656-660) */
enum {
    DR_COND_SPEC = 0,    /* the spectrogram of the last dr_frontend call   */
    DR_COND_UNCOND = 1   /* sampling=True: spectrogram == -1 everywhere     */
};

/* hyper-parameters: config/model/ClassifierFreeDiffRoll.yaml:1-15, config/task/<task>.yaml,
 * config/spec/mel.yaml:1-10, config/sampling.yaml:1-4 */
typedef struct dr_config {
    int32_t abi_version;        /* DR_ABI_VERSION */
    int32_t device;             /* HIP device ordinal */
    int32_t residual_channels;  /* 512 (multiple of 64) */
    int32_t residual_layers;    /* 15 */
    int32_t kernel_size;        /* odd: 3 / 9 / 15 */
    int32_t dilation_base;      /* 2 */
    int32_t dilation_bound;     /* 4 */
    int32_t n_mels;             /* 229 */
    int32_t timesteps;          /* 200 */
    int32_t sample_rate;        /* 16000 */
    int32_t n_fft;              /* 2048 (multiple of 32) */
    int32_t hop_length;         /* 512  (multiple of 4) */
    float f_min;                /* 0 */
    float f_max;                /* 8000 */
    float beta_start;           /* 1e-4  (informational; the coefficient table is passed in) */
    float beta_end;             /* 0.02 */
} dr_config;

typedef struct dr_engine dr_engine;

/* version of the loaded library (== DR_ABI_VERSION of the header it was built with) */
int dr_abi_version(void);

int dr_create(dr_engine** out, const dr_config* cfg);
void dr_destroy(dr_engine* e);
const char* dr_last_error(const dr_engine* e);   /* e may be NULL: error of the last dr_create */

/*
 * Hand over one parameter tensor by its reference state_dict name, in the
 * reference's own layout, from HOST memory (copied):
 *   input_projection.{weight (C,88,1), bias (C)}
 *   diffusion_embedding.projection1.{weight (512,128), bias}, .projection2.{weight (512,512), bias}
 *   residual_layers.<i>.dilated_conv.{weight (2C,C,k), bias (2C)}
 *   residual_layers.<i>.diffusion_projection.{weight (C,512), bias (C)}
 *   residual_layers.<i>.conditioner_projection.{weight (2C,n_mels,1), bias (2C)}
 *   residual_layers.<i>.output_projection.{weight (2C,C,1), bias (2C)}
 *   skip_projection.{weight (C,C,1), bias (C)},  output_projection.{weight (88,C,1), bias (88)}
 * numel must match the shape implied by the config.  Unknown names -> DR_ENAME.
 */
int dr_set_param(dr_engine* e, const char* name, const float* host_data, size_t numel);

/*
 * Host-built tables (same torch expressions as the reference, so bit-equal):
 *   embedding  (timesteps, 128)  DiffusionEmbedding._build_embedding (model/diffwave.py:83-88)
 *   coef       (DR_COEF_FAMILIES, timesteps, 5)  per-step scalars of the samplers' updates
 *                                (task/diffusion.py:957-967 and :804-911), one table per coefficient
 *                                family, columns as listed at the DR_COEF_* enum; row 0 of the x0
 *                                families only uses column 2 (x = x0 / sqrt_acp[0]).
 */
int dr_set_tables(dr_engine* e, const float* host_embedding, const float* host_coef);

/*
 * Constants of the mel front-end built by the caller WITH THE REFERENCE'S ARITHMETIC (torchaudio 0.11
 * MelSpectrogram, model/diffwave.py:635): host_window (n_fft) = torch.hann_window(n_fft), window_norm =
 * window.pow(2).sum().sqrt() (normalized=True divides the spectrum by it), host_fb (n_fft/2+1, n_mels) row-major =
 * torchaudio.functional.melscale_fbanks(..., norm=None, mel_scale='htk').  torchaudio evaluates these in fp32 and
 * the rounding of the filterbank is visible (2e-5 in the normalised log-mel), so parity needs the same tables:
 * diffroll_amd/frontend_tables.py builds them with the same torch expressions.  Optional: without this call (or
 * with NULL tables) dr_commit evaluates the published formulas itself, in double precision.
 */
int dr_set_frontend_tables(dr_engine* e, const float* host_window, float window_norm, const float* host_fb);

/* Pack weights for the kernels, upload, and build the hoisted tables on the device (the
 * (timesteps, layers, C) step-embedding projections; the unconditional conditioner constants).
 * Requires every parameter and both tables.  Synchronises `stream`. */
int dr_commit(dr_engine* e, void* stream);

/*
 * Front-end, once per clip batch.  d_wav (B, L) -> d_spec_out (B, n_mels, T) with
 * T = min(T_roll, L / hop + 1); also builds the per-layer conditioner tensors the residual
 * blocks consume, for this batch.  Mask: spectrogram[f0:f1, t0:t1] = -1 after normalisation
 * (pass t0 = t1 = -1 / f0 = f1 = -1 for "no mask on that axis"; model/diffwave.py:649-654).
 * d_spec_out may be NULL.
 */
int dr_frontend(dr_engine* e, const float* d_wav, int B, int L, int T_roll,
                int mask_t0, int mask_t1, int mask_f0, int mask_f1,
                float* d_spec_out, void* stream);

/* One network evaluation at diffusion step t: d_x (B, T, 88) [the reference's (B,1,T,88)] ->
 * d_x0_out (B, T, 88).  cond = DR_COND_SPEC needs a preceding dr_frontend with the same B, T. */
int dr_forward(dr_engine* e, const float* d_x, int B, int T, int t, int cond,
               float* d_x0_out, void* stream);

/* The same with one diffusion step PER SAMPLE (host_t: B ints on the host), the general form of the reference's
 * forward(x_t, waveform, diffusion_step (B,)) as its training / validation step() calls it; the samplers always
 * pass one step for the whole batch.  Synchronises `stream` (the step vector is uploaded). */
int dr_forward_steps(dr_engine* e, const float* d_x, int B, int T, const int32_t* host_t, int cond,
                     float* d_x0_out, void* stream);

/* One reverse-diffusion step t (in place on d_x).  d_noise (B, T, 88) is the z of that step
 * (ignored at t == 0); NULL -> on-device Philox keyed by (seed, first_sample + b, t). */
int dr_step(dr_engine* e, int sampler, float* d_x, const float* d_noise, int B, int T, int t,
            float w, uint64_t seed, int first_sample, void* stream);

/*
 * The whole reverse chain t = timesteps-1 .. 0, in place on d_x, no host synchronisation.
 * d_noise: (timesteps, B, T, 88) injected noise (row t used at step t >= 1) or NULL for Philox.
 * use_graph != 0: the chain is captured once into a hipGraph and replayed; the graph is cached per
 * (sampler, B, T, d_noise): the chain runs on an engine-owned copy of d_x, and w, seed and first_sample are
 * read from a device block at run time, so new buffers / values re-use the instantiated graph.  Needs dr_frontend first unless sampler == DR_SAMPLER_GENERATION_DDPM_X0.
 */
int dr_sample(dr_engine* e, int sampler, float* d_x, const float* d_noise, int B, int T,
              float w, uint64_t seed, int first_sample, int use_graph, void* stream);

/*
 * Consume point of asynchronous results.  The fused residual-stack kernel (option "fused_stack") assumes that all
 * its workgroups are resident on the device at once; when something else holds CUs while it runs (a second engine,
 * stream or process computing on the same device) a group barrier can run into its spin bound - the launch then
 * carries on with wrong data, raises a flag, and every later fused launch of the engine returns immediately.
 * (An engine avoids most of these BEFORE it launches.  Engines of one process take turns on a per-device slot: the one
 * that finds another engine's fused work still in flight orders its own launches behind it on the device (a stream
 * wait on an event: no host wait, nobody gives up fusing).  And dr_create / dr_sample look for another PROCESS computing
 * on the GPU in the kernel driver's process list (/sys/class/kfd/kfd/proc: csrc/tenants.h): if there is one the engine
 * YIELDS - one launch per phase from then on, same results, one line on stderr, counted in dr_launch_state - and goes
 * back to fused launches once two looks in a row, in front of later chains, find the GPU its own again.  The spin bound
 * remains the backstop for what those checks cannot see: a tenant that arrives in the middle of a chain.)
 * dr_finish synchronises `stream` and checks that flag:
 *   DR_OK        everything issued on this engine since the last check is valid;
 *   DR_ETIMEOUT  it is NOT: recompute it.  The condition has been cleared and the engine switched to one launch per
 *                phase (fused_stack = 0: bit-identical results, no residency assumption), so the recomputation
 *                cannot time out again; re-enable with dr_set_option when the device is the engine's own again.
 * Call it before a roll produced by dr_forward / dr_step / dr_sample is used (copied to the host, written as MIDI,
 * gathered).  While an unchecked time-out is pending every entry point that COMPUTES (dr_forward, dr_forward_steps,
 * dr_step, dr_sample) refuses to start, and every entry point that CONSUMES a roll given an engine handle
 * (dr_note_runs, dr_frame_counts, dr_q_sample / dr_extract_x0, dr_gather) first does what dr_pending_timeout does -
 * it synchronises the stream the fused launches ran on (and `stream`) if any have been issued since the last check - and
 * returns DR_ETIMEOUT instead of working on an invalid roll; dr_gather still takes part in the collective first (a time-out
 * is a per-rank event: a rank that stayed out would leave its peers blocked) and reports DR_ETIMEOUT afterwards - ON EVERY
 * RANK (a status word travels with the rolls): every rank must gather again after the invalid shard has been recomputed.
 * Only dr_finish clears the condition.
 */
int dr_finish(dr_engine* e, void* stream);
/* The check alone: DR_ETIMEOUT when a fused launch issued on this engine has timed out and dr_finish has not been called
 * since; DR_OK otherwise.  Synchronises `stream` only when fused launches have been issued since the last check (so that
 * the flag is final); does not heal, does not clear.  `e` may be NULL (DR_OK). */
int dr_pending_timeout(dr_engine* e, void* stream);
/* dr_sample + dr_finish + (on a time-out) the re-run of the chain from the same x_T on the per-phase kernels:
 * returns DR_OK only with the correct roll in d_x.  Synchronous.  *recovered (optional) = 1 when the re-run was
 * needed.  This is what a one-shot caller (sampling.py, predict_step) should use: task/diffusion.py:528-538 returns
 * a finished roll, never a silently invalid one. */
int dr_sample_checked(dr_engine* e, int sampler, float* d_x, const float* d_noise, int B, int T,
                      float w, uint64_t seed, int first_sample, int use_graph, int32_t* recovered, void* stream);
/*
 * How this engine launches the residual layers, and what has happened to that decision - the record a measurement
 * must check (bench.py refuses to print a line when `fallbacks` or `yields` moved during its timed region):
 *   mode          DR_MODE_* of the most recently planned network evaluation (a captured chain: at capture)
 *   fused_enabled the current value of option "fused_stack" (0 while the engine runs per-phase launches after a
 *                 time-out or a yield)
 *   fallbacks     time-outs dr_finish has detected and healed (each switched the engine to per-phase launches)
 *   yields        times the engine gave up fusing BEFORE launching because another process was found computing on
 *                 its GPU (csrc/tenants.h); same results, one launch per phase from then on
 *   rearms        times fused launches were switched back on (after "fused_rearm" clean chains behind a time-out, or
 *                 two clean looks behind a yield)
 *   stack_launches / tail_launches   persistent launches issued so far (a captured chain counts once, at capture)
 */
enum { DR_MODE_NONE = 0, DR_MODE_PER_PHASE = 1, DR_MODE_FUSED_STACK = 2, DR_MODE_FUSED_STACK_TAIL = 3 };
typedef struct dr_launch_info {
    int32_t mode;
    int32_t fused_enabled;
    int64_t fallbacks;
    int64_t yields;
    int64_t rearms;
    int64_t stack_launches;
    int64_t tail_launches;
} dr_launch_info;
int dr_launch_state(dr_engine* e, dr_launch_info* out);

/*
 * Roll -> notes, the scan of extract_notes_wo_velocity (task/diffusion.py:1185-1233) as the reference's
 * drivers call it (onsets == frames == the roll, one threshold, rule1): d_note_end (B, T, 88) int32
 * receives, at every (frame, pitch) where a note STARTS, the frame index at which it ends (exclusive),
 * and 0 elsewhere.  np.nonzero() of that tensor enumerates the notes in the reference's order.
 */
int dr_note_runs(dr_engine* e, const float* d_roll, int B, int T, float threshold, int32_t* d_note_end,
                 void* stream);

/*
 * Frame-level evaluation of test_step (task/diffusion.py:381-383): confusion counts of
 * (d_pred > threshold) against the binary label roll over n elements, the integers sklearn's
 * precision_recall_fscore_support(average='binary') is computed from.  host_counts = {TP, FP, FN}.
 * Synchronises `stream`.
 */
int dr_frame_counts(dr_engine* e, const float* d_pred, const float* d_label, size_t n, float threshold,
                    int64_t* host_counts, void* stream);

/*
 * The forward-process arithmetic of task/diffusion.py (free functions, used by step() around the network):
 *   dr_q_sample   (:31-46)  out = sqrt_alphas_cumprod[t_b] * x_start + sqrt_one_minus_alphas_cumprod[t_b] * noise
 *   dr_extract_x0 (:49-64)  out = (x_t - sqrt_one_minus_alphas_cumprod[t_b] * epsilon) / sqrt_alphas_cumprod[t_b]
 * d_t (B,) int64 per-sample step indices, d_sac / d_s1m the two schedule vectors (n_steps,) - all on the
 * device; tensors are (B, per_sample) contiguous fp32.  Same operation order and roundings as the reference's
 * broadcasted torch expression (bit-exact).  Step indices are clamped to [0, n_steps).  `e` may be NULL (no
 * engine state is involved: current device, error text via dr_last_error(NULL)).
 */
int dr_q_sample(dr_engine* e, const float* d_x_start, const float* d_noise, const int64_t* d_t,
                const float* d_sac, const float* d_s1m, int n_steps, int B, size_t per_sample, float* d_out,
                void* stream);
int dr_extract_x0(dr_engine* e, const float* d_x_t, const float* d_epsilon, const int64_t* d_t,
                  const float* d_sac, const float* d_s1m, int n_steps, int B, size_t per_sample, float* d_out,
                  void* stream);

/* Spectrogram normalisation of the following dr_frontend calls: the mode of Normalization(0, 1, norm_args[2])
 * (model/diffwave.py:632, model/utils.py:10-32) - min-max per clip ("imagewise", the default and the released
 * configs) or per frame over the frequency bins ("framewise"). */
#define DR_NORM_IMAGEWISE 0
#define DR_NORM_FRAMEWISE 1
int dr_set_spec_norm(dr_engine* e, int mode);

/* Select DR_PRECISION_* for subsequent dr_forward / dr_step / dr_sample calls (default F32).
 * Drops a captured chain. */
int dr_set_precision(dr_engine* e, int mode);

/*
 * Integer options (defaults in brackets).  Changing one drops a captured chain.
 *   "blocked_accumulation" [2] accumulation order of the dilated conv's K = taps x channels contraction
 *                          (model/diffwave.py:144 as a CPU library executes it: K-blocked): 2 = one fp32 MFMA chain per
 *                          32-channel block, block sums added up in block order, in every fp32 kernel flavour (against
 *                          float64 the error is 1.0-1.6x the CPU fp32 reference's in the trained-weight regime);
 *                          1 = the 128-frame blocks and the 96 / 160-frame flavours contract all of K as ONE chain (the
 *                          numerics of ABI <= 7: 0.2-0.8 % faster, 3.0-3.9x; 1.0 % faster in the split-bf16 precision).
 *   "fused_rearm"      [0] n > 0: after a time-out has switched this engine to per-phase launches, go back to the fused
 *                          kernels once n chains in a row have finished cleanly (a time-out caused by a transient
 *                          tenant - a profiler, a second process that has left - then costs n chains at the per-phase
 *                          pace instead of the rest of the engine's life).  0 = stay on per-phase launches until
 *                          "fused_stack" is set again.  Seeded results after a recovery can differ from a healthy fused
 *                          run in the last bits (the per-phase launches split K where the fused kernel does not).
 *   "fused_stack"      [1] the residual layers of an evaluation (model/diffwave.py:678-681: 15 x ResidualBlock.forward,
 *                          :134-151) run as ONE persistent launch whenever samples x frame tiles x M tiles fits the
 *                          chip's CUs in one resident round (the BASELINE configurations 2-5 do: 64 / 128 / 160-frame blocks); 0 = one launch per
 *                          dilated conv and per 1x1 (bit-identical results, 2 x residual_layers launches); 2 = fuse
 *                          also launches that fill less than half the chip (tests).
 *   "fused_tail"       [1] where the evaluation is one fused launch, the REST of a reverse
 *                          step is fused too (model/diffwave.py:667-668, :682-686; task/diffusion.py:953-967): skip
 *                          projection, output projection, combine + posterior update, the NEXT step's input projection
 *                          and - under classifier-free guidance - the next step's first-layer dilated conv (the same
 *                          contraction for the conditional and the unconditional evaluation: done once per pair) run as
 *                          one persistent "tail" launch, and the following stack launch starts at that layer's 1x1:
 *                          2 launches per reverse step instead of 6 (the first step of a chain still runs its input
 *                          projection and first-layer conv as launches of their own).  0 = separate launches
 *                          (bit-identical without split-K).
 * Unknown names -> DR_ENAME.  (The A/B and test knobs - "tune.*", "fused_stack_xcd", "fused_stack_warm", "stack_ticks" -
 * are set with dr_debug_set_option, diffroll_amd_debug.h.)
 */
int dr_set_option(dr_engine* e, const char* name, int value);
/*
 * Multi-GPU: the path shards by clips - one process per GPU, every rank runs its contiguous shard of
 * the batch with dr_sample (first_sample = the shard's global offset, so Philox noise does not depend on the world
 * size) and the ONLY collective is one all-gather of the finished rolls over xGMI.  These entry points give a
 * caller without torch.distributed that collective: RCCL (librccl, looked up with dlopen at first use).
 * The reference reaches N GPUs through Lightning's Trainer(gpus=N) (sampling.py:70) and never gathers.
 *   dr_comm_unique_id   rank 0: 128 bytes (ncclUniqueId) to hand to every rank by any side channel
 *   dr_comm_create      every rank, collectively: ncclCommInitRank on `device`
 *   dr_comm_info        ranks / rank of a communicator and the version code of the loaded librccl (any pointer may be
 *                       NULL; c == NULL: the version alone)
 *   dr_gather           d_shard (B_local, T, 88) of every rank -> d_full (n_ranks * B_local, T, 88), rank-major, on
 *                       `stream` (ncclAllGather; equal B_local on all ranks - pad uneven shards, see
 *                       diffroll_amd/distributed.py).  `e` may be NULL.  SYNCHRONOUS, and the verdict is COLLECTIVE: behind the
 *                       rolls every rank also gathers one status word (0 = my shard is valid, 1 = it came out of a fused
 *                       launch that timed out: dr_pending_timeout), and EVERY rank returns DR_ETIMEOUT when any shard was
 *                       invalid - nobody is handed a d_full that holds a bad shard together with DR_OK.  After
 *                       DR_ETIMEOUT: the rank(s) whose dr_finish also reports it recompute their shard, then all ranks
 *                       gather again.
 * Errors of these functions: dr_comm_last_error().
 */
typedef struct dr_comm dr_comm;
int dr_comm_unique_id(char* id_out /* 128 bytes */);
int dr_comm_create(dr_comm** out, const char* id /* 128 bytes */, int n_ranks, int rank, int device);
void dr_comm_destroy(dr_comm* c);
int dr_comm_info(const dr_comm* c, int* n_ranks, int* rank, int* rccl_version);
const char* dr_comm_last_error(void);
int dr_gather(dr_engine* e, dr_comm* comm, const float* d_shard, float* d_full, int B_local, int T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFROLL_AMD_H */
