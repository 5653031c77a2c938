// Internal launch interface between the host translation units (pack / plan / abi / debug_abi .hip: host logic, C-ABI) and the kernel translation units gemm / stack / tail / update / frontend .hip (gfx950 kernels).
// Not part of the public ABI (that is include/diffroll_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace dr {

// ---------------------------------------------------------------------------------------------
// Data layouts in HBM
//
//  "P4" activation layout: a (batch, channels, frames) tensor is stored as
//        [batch][plane = channel/4][frame][4]        (fp32; one float4 per (plane, frame))
//  so that (a) a wavefront's 64 lanes reading consecutive frames of a plane move 1 KiB
//  contiguously, (b) the float4 is exactly the 4 consecutive K-values an MFMA lane consumes
//  over 4 consecutive v_mfma_f32_32x32x2_f32 issues, and (c) the MFMA C/D fragment (4 consecutive
//  rows per lane per register quad) is written back as one float4 with no shuffles.
//  The roll (B, T, 88) is addressed with the same (batch, plane, frame) strides
//  (plane stride 4, frame stride 88), so no transposed copy is ever made.
//
//  Packed weights: for a GEMM  Y[M][N] = W[M][K] X[K][N]  with K = taps x Cin,
//        Wp[mtile][kchunk][tap][g = 0..3][hi = 0..1][row = 0..127][4]
//  = W[orig_row(mtile,row)][channel = kchunk*32 + g*8 + hi*4 + i][tap]; one (mtile,kchunk,tap)
//  slab is 16 KiB, contiguous, and is the exact LDS image the kernel reads (linear copy).
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// A/B knobs of the planners and launchers (process-wide; defaults = what ships).  They are set through
// dr_set_option(e, "tune.<field>", value) by tools/ and tests - the library reads no environment variable.
// ---------------------------------------------------------------------------------------------
// Fields are atomics (relaxed loads / stores are all that is needed: independent integers): dr_debug_set_option may run on
// one host thread while another engine's planner reads them.  Every change bumps tuning_epoch(); an engine compares it in
// dr_sample and drops a chain it captured under older values.
struct Tuning {
    std::atomic<int> pack_threads{16};   // host threads packing the layers at dr_commit (1 = serial)
    std::atomic<int> tile{0};            // force a conv / LDS-staged 1x1 tile: 3201 / 3202 / 3203 / 3205 (32x32 MFMA, NI) or 1603 / 1605 (16x16, NJ); 0 = cost model
    std::atomic<int> pw{1};              // 1x1 residual/skip GEMM: 1 = operands direct from L2 (pw_kernel), 0 = the LDS-staged kernels
    std::atomic<int> pw_nw{0};           // force 32 * pw_nw frames per pw_kernel block (2..5); 0 = cost model
    std::atomic<int> pwk{1};             // under-filled 1x1 launches: K split over the block's waves (pwk_kernel)
    std::atomic<int> ksplit_max{16};     // largest split-K factor of gemm_kernel launches (1 = never split)
    std::atomic<long> ksplit_blocks{0};  // cap on tiles x ksplit (0 = 2048 fp32 / 256 split-bf16)
    std::atomic<int> one_ks{0};          // force the 1x1 GEMMs' channels-per-hand-over (KS = 1 / 2 / 4); 0 = by divisibility
    std::atomic<int> stack3{1};          // the split-bf16 flavour of the fused residual stack
    std::atomic<int> stack_fl{0};        // force the fused stack's block flavour (1 / 2 / 5 = 64 / 128 / 160-frame blocks); -5 = never the 160-frame one; 0 = cost model
    std::atomic<int> tail_t4{0};         // force the item width of the tail kernel's first-layer conv (1 = 64, 3 = 96 frames); 0 = by rounds x width
    std::atomic<int> xcd_n{-1};          // force the block -> XCD mapping of per-phase GEMM launches (0 / 1); -1 = traffic model
    std::atomic<int> xcd_model{1};       // 0 = the rounds 1-2 rule (activation bytes > weight bytes)
    std::atomic<int> s3_eager{0};        // build the split-bf16 packings at every dr_commit (rounds 1-3 behaviour)
    std::atomic<int> debug_chunks{0};    // dr_debug_ticks prints the chunk-start tick marks
};
Tuning& tuning();            // gemm.hip
std::atomic<unsigned>& tuning_epoch();

enum Epilogue : int {
    EPI_PLAIN = 0,     // y = alpha*acc + bias
    EPI_RELU = 1,      // y = relu(alpha*acc + bias)
    EPI_SILU = 2,      // y = silu(acc + bias)
    EPI_GATE = 3,      // rows paired (gate, filter): y = sigmoid(a0 + c0) * tanh(a1 + c1)
    EPI_RES_SKIP = 4,  // first half of M: h = (h + acc + b)/sqrt(2) in place; second half: skip (+)= acc + b
    EPI_POWER = 5,     // rows paired (cos, sin): y = a0^2 + a1^2
    EPI_LOG = 6        // y = log(acc + 1e-6)
};

struct GemmArgs {
    // A operand
    const float* Wp;      // packed weights
    const float* bias;    // [MT*128] in packed-row order; never null (pass a zero vector: loads are unconditional)
    const float* bias2;   // EPI_GATE: bias for samples >= n_cond (unconditional: conv bias + cond const)
    // B operand
    const float* X;
    long x_bs, x_ps, x_fs;   // batch / plane / frame strides in floats
    long x_piece;            // S3 input only: stride between the three bf16 pieces (4-byte units)
    int x_planes;            // valid planes (Cin/4); planes beyond re-read the last one (their weights are zero)
    int x_bmod;              // sample index is taken modulo this (0 = no modulo)
    int NB, T;               // samples, frames per sample
    int taps, dil;           // conv taps (odd) and dilation; halo = (taps-1)/2*dil
    int kchunks;             // ceil(Cin/32)
    int MT;                  // M tiles of 128 packed rows
    int mt0;                 // pw_kernel only: first M tile of the launch (tiles mt0 .. mt0 + MT - 1 are computed)
    // output
    float* Y;
    long y_bs, y_ps, y_fs;
    long y2_bs;              // batch stride of Y2 (4-byte units)
    int out_s3;              // bit 0: Y (EPI_GATE) is written in S3 split-bf16 layout; bit 1: Y2 is
    float* Y2;               // EPI_RELU / residual rows of EPI_RES_SKIP: optional second output Y2 = Y + d2[row]
                             // (same strides): the (h + step embedding) tensor the next dilated conv reads
    const float* d2;         // [>= y_rows] never null (zero vector when unused)
    const int* tsel;         // optional per-sample row selector of d2: sample b adds d2[tsel[b] * d2_ts + row] (per-sample
    long d2_ts;              // diffusion steps of forward(); null in the samplers: one step for the whole batch)
    int y_rows;              // valid output rows (quads starting at >= y_rows are not written)
    // epilogue extras
    const float* cond;       // EPI_GATE: [n_cond][MT*32 planes][T][4] in packed-row order; must point at valid memory
                             // of at least one sample even when n_cond == 0 (prefetched unconditionally, then ignored)
    long c_bs;
    const float* cond2;      // EPI_GATE, optional: one conditioner tensor [MT*32 planes][T][4] shared by all samples >= n_cond
                             // (the learned unconditional spectrogram of condition='trainable_spec'); null: they use bias2 only
    int n_cond;
    int dual;                // EPI_GATE, gemm_kernel only: > 0 = also write sample b + dual from the same accumulators
                             // with the unconditional bias (first layer under classifier-free guidance); NB counts b only
    float* skip;             // EPI_RES_SKIP: P4 [NB][MT/2*32 planes][T][4]
    long s_bs;
    int skip_init;           // 1: skip = value, 0: skip += value
    float alpha;
    int xcd_n;               // set by the launcher: block -> (M tile, frame tile) mapping, see gemm_kernel
    // split-K (gemm_kernel only; optional): partial-accumulator workspace + zero-initialised ticket counters.
    // The launcher picks ksplit (1 when ws is null or the launch already fills the chip).
    float* ws;               // >= tiles * ksplit * 128 * BN floats
    unsigned* ws_cnt;        // >= tiles * 4 counters, all zero between launches
    size_t ws_floats, ws_cnt_n;
    int ksplit;              // set by the launcher
    long long* dbg;          // measurement hook: block 0 writes {main-loop ticks, block ticks} (s_memtime); null normally
    int lds_bytes;           // set by the launcher: dynamic LDS of the launch (read by DR_BOUNDS checker builds only)
    int fold128;             // EPI_GATE on 128-frame blocks (NI = 2, fp32): 1 = the blocked-accumulation instantiation
    int nofold64;            // EPI_GATE on 64-frame blocks (NI = 1, fp32, no split-K): 1 = ONE chain per output instead of the blocked
                             // accumulation every other 64-frame launch uses - the shared first-layer conv of an engine whose fused
                             // 128-frame stack runs single-chain (blocked_accumulation = 1), so that the per-phase launch and the
                             // tail kernel's copy of that conv (TailArgs::fold = 0) produce the same bits
    int wt_store;            // fused residual stack only (COH bodies): 1 = the tensors handed to other workgroups are
                             // stored write-through (sc1), 0 = plain stores (every workgroup of the group shares one
                             // XCD's L2, verified at run time)
};

// gemm_kernel: block tile = 128 packed rows x 64*NI frames (NI in {1, 2}), 512 threads (4 consumer + 4 producer
// waves); call init_kernels() once per process before any launch.
hipError_t init_kernels();
// DR_BOUNDS checker builds: {code of the first violated check, detail, detail, number of violations} / reset;
// hipErrorNotSupported in production builds
hipError_t read_bounds(unsigned long long* out4);
hipError_t reset_bounds();
// prec = 0: fp32 X / weights; 1: split-bf16 ("S3") X / weights (EPI_GATE and 1x1 EPI_RES_SKIP only)
hipError_t launch_gemm(const GemmArgs& a, int epi, int NI, hipStream_t s, int prec = 0);
// flexible-width variant (16x16x4 MFMA): block = 128 rows x 32*NJ frames, NJ in {3,5}; fp32, EPI_GATE / 1x1 EPI_RES_SKIP
hipError_t launch_gemm16(const GemmArgs& a, int epi, int NJ, hipStream_t s);
// 1x1 EPI_RES_SKIP GEMM with both operands direct from L2 (no LDS): block = 128 rows x 32*NW frames
// (NW in {2,3,4,5}), 256 threads
hipError_t launch_pointwise(const GemmArgs& a, int NW, hipStream_t s);
hipError_t launch_pointwise_ksplit(const GemmArgs& a, int NW, hipStream_t s);   // under-filled launches: 32-row tiles, K split over the block's waves
// frames per block of gemm_kernel<NI, ...>: 64 / 128 (NI = 1 / 2), 96 / 160 (NI = 3 / 5: that many 32-frame MFMA tiles per
// consumer wave, the gated conv only)
inline int gemm_block_frames(int NI) { return (NI == 3 || NI == 5) ? 32 * NI : 64 * NI; }
size_t gemm_lds_bytes(int NI, int KS, int taps, int dil, int prec, int epi);
// THE split-K decision of gemm_kernel launches (the launcher takes it; the engine's tile choice prices a launch with it,
// so the estimate and the launch cannot disagree): for `tiles` output tiles of 128 rows x 64 NI frames, `nchunks` hand-over
// chunks of K (kchunks / KS), a workspace of ws_floats / ws_cnt_n: the number of K slices and the modelled time.
// Model (fp32, fitted to 3..8 guided clips of 125 frames, tools/lab/small_batch_ab.py): equal blocks run in lockstep rounds
// over the 256 CUs - rounds x t_full / ks + exchange, t_full = a full-K tile (MFMA count x 69 cycles at 2.4 GHz), the
// exchange (store, ticket, the last arriver's ordered re-read) ~(4 + ks) us.  Inside one resident round more slices are
// always taken; beyond it a split must win by 3 %.
struct KSplitPlan { int ks; double us; double us_unsplit; };
KSplitPlan plan_ksplit(long tiles, int nchunks, int kchunks, int taps, int NI, int prec, size_t ws_floats, size_t ws_cnt_n);

// ---------------------------------------------------------------------------------------------
// Fused residual stack: ONE persistent launch runs a range of the 2L phases of the residual layers
// (phase 2l = dilated conv + conditioner + gate of layer l, phase 2l+1 = its 1x1 + residual / skip), instead of
// one launch per phase.  grid = (samples x frame tiles x M tiles) <= the number of CUs, every block owns the same
// (M tile, frame tile) in every phase; the blocks of one sample (= one clip evaluation: M tiles x its frame tiles)
// form a GROUP that synchronises on a device counter between phases - nothing is exchanged between groups, so
// there is no grid-wide barrier.  See stack_kernel in stack.hip.
// ---------------------------------------------------------------------------------------------
constexpr int DR_STACK_MAX_LAYERS = 30;
struct StackLayer {
    const float *conv_w, *conv_b, *conv_b2;   // packed dilated-conv weights; bias of samples < n_cond / >= n_cond
    const float *cond, *cond2;                // conditioner tensors of this layer (see GemmArgs)
    const float *out_w, *out_b;               // packed 1x1 weights / bias
    int dil, pad_;
};
struct StackArgs {
    float *h, *hd, *g, *skip;                 // P4 activations [NB][Cp/4][T][4]
    const float* d2;                          // step-embedding rows: layer l+1's row is d2 + (l + 1) * Cp (+ tsel[b] * d2_ts)
    const int* tsel;
    long d2_ts;
    const float* zero;                        // device zero vector (never-null operands)
    int NB, T, Cp, taps, n_cond, L;
    long c_bs;
    int p0, p1;                               // phases [p0, p1)
    int xcd_n;                                // block -> (M tile, frame tile) mapping, as in gemm_kernel
    int rs_off;                               // set by the launcher: LDS byte offset of the resident h / skip tile
    int lds_bytes;                            // set by the launcher: dynamic LDS of the launch (DR_BOUNDS checker builds)
    int fault;                                // test hook: barriers wait for one arrival too many (exercises the spin bound)
    int fold128;                              // FL = 2: 1 = the instantiation with blocked accumulation in its conv phases
    int warm;                                 // idle waves warm the L2 with the next phase's weights / conditioner tile
    unsigned* xid;                            // [grid] scratch: the XCC each block runs on (rewritten by every launch)
    unsigned* bar;                            // [groups][4] {arrivals, departures, generation, -}; the first two are zero
                                              // between launches, the generation advances by one per launch
    unsigned* err;                            // host-mapped: set to 1 if a barrier wait ran into its spin bound (never in a
                                              // healthy run)
    unsigned* derr;                           // device copy of that flag: polled by the waits, and a launch that finds it set
                                              // returns at once (cleared by the host together with *err)
    long long* dbg;                           // optional: block 0 writes s_memtime at every phase start (dbg[p - p0]) and at the end
    StackLayer layer[DR_STACK_MAX_LAYERS];
};
// FL = block flavour: 1 / 2 = 128 packed rows x 64 / 128 frames.  The caller guarantees
// NB * stack_group_blocks(FL, Cp, T) <= #CUs and stack_lds_bytes(..) <= 160 KiB.
// prec = 1: the split-bf16 flavour (s.hd / s.g = the S3 tensors; Cp % 128 == 0; LDS: stack3_lds_bytes)
hipError_t launch_stack(const StackArgs& s, int FL, int max_dil, hipStream_t st, int prec = 0);
size_t stack3_lds_bytes(int FL, int taps, int max_dil);
int stack_tile_frames(int FL);
int stack_group_blocks(int FL, int Cp, int T);      // blocks per clip evaluation
size_t stack_lds_bytes(int FL, int taps, int max_dil);

// Per-call scalars of the update that must not be baked into a captured graph: the chain graph reads them from
// this device block, which a one-thread kernel rewrites (stream-ordered) before every graph launch - a new
// seed / batch offset / guidance weight re-uses the instantiated graph.
struct DynParams {
    unsigned long long seed;
    int first_sample;
    float w, onepw;
};
hipError_t launch_set_dyn(DynParams* d, unsigned long long seed, int first_sample, float w, float onepw, hipStream_t s);

struct UpdateArgs {
    float* x;              // (B, T, 88) in/out
    const float* x0c;      // (B, T, 88) conditional (or the only) prediction
    const float* x0u;      // unconditional prediction or null
    const float* noise;    // (B, T, 88) or null -> philox
    const float* coef;     // device pointer to this step's 5 coefficients
    int t;                 // step index
    int mode;              // coefficient family (DR_COEF_*): 0/1 x0 update, 2 eps ddpm, 3 eps ddim, 4 eps ddim2ddpm
    long n;                // B*T*88
    long per_sample;       // T*88
    float w, onepw;
    uint64_t seed;
    int first_sample;
    const DynParams* dyn;  // non-null: w / onepw / seed / first_sample are read from here instead
};
hipError_t launch_update(const UpdateArgs& a, hipStream_t s);

// Tail of a reverse step as one persistent launch (tail_kernel in tail.hip): skip projection -> output projection ->
// classifier-free combine + posterior update -> input projection of the next step.  Same grid / grouping as the
// stack_kernel launch it follows: NB samples (first `dual` conditional, next `dual` unconditional when dual > 0) x
// ceil(T / BN) frame tiles x Cp / 64 blocks, all resident at once.
struct TailArgs {
    int NB, T, Cp, BN;                        // BN = frames per block of the preceding stack launch (64 / 128 / 160): grouping only
    int dual;                                 // B > 0: classifier-free pairs (sample b, sample b + B); 0: every sample on its own
    int u_B;                                  // rolls to update (B)
    int xcd_n, fault;
    float alpha;                              // 1 / sqrt(residual_layers)
    const float* skip;                        // P4 [NB][Cp/4][T][4]
    float* tmp;                               // P4 workspace, same shape
    float* x0;                                // (NB, T, 88): network outputs (u.x0c / u.x0u point into it)
    const float *skip_w, *skip_b, *outp_w, *outp_b, *zero;
    UpdateArgs u;                             // the update of this step; u.x = x_t is only READ here
    float* x_out;                             // x_{t-1} is written here (never u.x: other blocks still read x_t)
    const float *in_w, *in_b, *d2_next;       // input projection of the NEXT step (in_w null: the chain ends here / single step)
    float *h, *hd;                            // its outputs, P4 [NB][Cp/4][T][4]
    // the next step's shared first-layer conv (dual > 0 and in_w set; conv_w null: none): layer 0 as in StackLayer
    const float *conv_w, *conv_b, *conv_b2, *cond, *cond2;
    long c_bs;
    int taps, dil;
    int fold;                                 // blocked accumulation in that conv (gemm_body.h), as the stack launch that follows
    int t4_ni;                                // in: 0 = the launcher chooses T4's item width, 1 / 3 = force 64 / 96 frames; the launcher sets 1 or 3
    float* g;                                 // its output, P4 [NB][Cp/4][T][4]
    int lds_bytes;                            // set by the launcher
    long long* dbg;                           // optional: block 0 writes s_memtime at the start and after T1, its barrier, T2, its
                                              // barrier, T3, its barrier, T4 (8 marks)
    unsigned *bar, *pbar, *err, *derr;        // counters as in StackArgs (own arrays), time-out flags (shared with the stack)
};
hipError_t launch_tail(const TailArgs& s, hipStream_t st);

hipError_t launch_reflect_pad(const float* wav, float* out, int B, int L, int pad, hipStream_t s);
// STFT power by FFT (N a power of two): wav_pad (B, Lp) -> power (B, TF, bins_p) row-major; win (N) the window,
// tw (N complex) = exp(-2 pi i k / N), norm = sqrt(sum win^2) (the spectrum is divided by it)
hipError_t launch_stft_power(const float* wav_pad, const float* win, const float* tw, float* power, int B, int Lp, int TF,
                             int N, int hop, int bins_p, float norm, hipStream_t s);
// per-sample min/max of logmel P4 [B][planes][TF][4] over rows < n_rows -> mm[B][2]; scratch = minmax_scratch_floats(B)
// floats of device memory, zero before its first use (partials + one ticket word per sample, left zero by the kernel)
hipError_t launch_minmax(const float* logmel, float* mm, float* scratch, int B, int planes, int TF, int n_rows, hipStream_t s);
size_t minmax_scratch_floats(int B);
// normalise, mask, trim -> spec P4 [B][planes_out][T][4] (rows >= n_rows zero) and optional plain (B, n_rows, T)
// framewise = 1: mm holds one (min, max) per (sample, frame) [B][TF][2] (launch_minmax_frame) instead of per sample
hipError_t launch_minmax_frame(const float* logmel, float* mm, int B, int planes, int TF, int n_rows, hipStream_t s);
hipError_t launch_normalize(const float* logmel, const float* mm, float* specP4, float* spec_plain,
                            int B, int planes_in, int planes_out, int TF, int T, int n_rows,
                            int mt0, int mt1, int mf0, int mf1, hipStream_t s, int framewise = 0);
hipError_t launch_fill(float* p, float v, long n, hipStream_t s);
// q_sample / extract_x0 of task/diffusion.py:31-64 (mode 0 / 1): per-sample schedule lookup by t[b], elementwise
hipError_t launch_noise_mix(int mode, const float* x, const float* y, const int64_t* t, const float* sac,
                            const float* s1m, int n_steps, int B, long per_sample, float* out, hipStream_t s);
// note_end[b][t][p] = offset frame (exclusive) of the note that STARTS at frame t on pitch p, else 0:
// roll (B, T, 88) thresholded at thr; a note = maximal run of frames above the threshold (rule1 with
// onsets == frames, task/diffusion.py:1185-1233)
hipError_t launch_note_runs(const float* roll, int* note_end, int B, int T, float thr, hipStream_t s);
// work[0..2] = {TP, FP, FN} of (pred > thr) against (label > 0.5) over n elements (exact integers); work = device
// scratch of frame_counts_work_words() 64-bit words, zero before its first use (the kernel leaves its ticket word zero)
hipError_t launch_frame_counts(const float* pred, const float* label, float thr, long n,
                               unsigned long long* work, hipStream_t s);
size_t frame_counts_work_words();
hipError_t init_update_kernels();

}  // namespace dr
