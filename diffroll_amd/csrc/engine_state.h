// Host side of the DiffRoll sampling engine, shared declarations: the engine object behind the C-ABI handle and the
// helpers the four host translation units use -
//   pack.hip       weight packing, staged uploads, dr_set_param / dr_set_tables / dr_commit (+ the split-bf16 packings)
//   plan.hip       tile / split-K / fused-stack planning and the launch sequences of one evaluation and one reverse step
//   abi.hip        the C-ABI of include/diffroll_amd.h: life cycle, front-end, forward / step / sample (hipGraph), time-outs
//   debug_abi.hip  measurement and checker entry points (dr_bench_*, dr_debug_*, dr_profile_*)
// Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/diffroll_amd_debug.h"      // (includes the boundary, diffroll_amd.h)
#include "kernels.h"

namespace drh {
using namespace dr;

extern thread_local std::string g_create_error;      // dr_last_error(NULL): errors of calls that have no engine

struct LayerW {
    float* conv_w = nullptr;     // packed (paired rows) [MTc][kch][k] slabs
    float* conv_b = nullptr;     // packed-row bias (conditional samples; cond tensor carries bc)
    float* conv_b_u = nullptr;   // packed-row bias for unconditional samples: b_conv + (bc - sum_m Wc)
    float* conv_b_z = nullptr;   // ... for spec == 0 samples (cfdg_ddim_x0's second branch): b_conv + bc
    float* conv_w3 = nullptr;    // split-bf16 ("S3") packing of conv_w  [MTc][kch][k] slabs of 24 KiB
    float* out_w3 = nullptr;     // split-bf16 packing of out_w
    float* out_w = nullptr;      // packed (natural halves) 1x1
    float* out_b = nullptr;
    float* cond_w = nullptr;     // packed (paired rows) conditioner 1x1
    float* cond_b = nullptr;
    int dil = 1;
};

// What a captured chain bakes in: sampler, shape, the engine's work buffer and the injected-noise address (test
// mode).  Seed, batch offset and guidance weight live in the DynParams device block; the caller's roll buffer
// is copied into / out of the work buffer around the launch.
struct GraphKey {
    int sampler = -1, B = 0, T = 0;
    float* x = nullptr;
    const float* noise = nullptr;
    bool w_zero = false;        // guidance weight 0 captures a different (conditional-only) chain
    bool operator==(const GraphKey& o) const {
        return sampler == o.sampler && B == o.B && T == o.T && x == o.x && noise == o.noise && w_zero == o.w_zero;
    }
};

}  // namespace drh

struct dr_engine {
    dr_config cfg{};
    int C = 0, Cp = 0, L = 0, K = 0, S = 0, NM = 0;
    int n_bins = 0, bins_p = 0;          // n_fft/2+1 and its 64-multiple padding
    std::string err;
    std::map<std::string, std::vector<float>> params;
    std::vector<float> h_emb, h_coef;
    std::vector<float> h_win, h_fb;      // optional caller-built front-end tables (dr_set_frontend_tables)
    float h_win_norm = 0.f;
    bool committed = false;

    // device constants
    float* d_coef = nullptr;   // (DR_COEF_FAMILIES, S, 5)
    float* d_dtab = nullptr;   // (S, L, Cp)   hoisted diffusion_projection(diffusion_embedding(t))
    std::vector<drh::LayerW> layers;
    float *in_w = nullptr, *in_b = nullptr, *skip_w = nullptr, *skip_b = nullptr, *outp_w = nullptr, *outp_b = nullptr;
    float *dft_w = nullptr, *mel_w = nullptr;
    float *fft_win = nullptr, *fft_tw = nullptr;     // FFT front-end: window (n_fft), roots of unity (n_fft complex)
    float fft_norm = 1.f;                            // the spectrum is divided by it (normalized=True)
    bool use_fft = false;
    std::vector<void*> owned;   // every constant allocation, for dr_destroy

    // activation workspace (sized for ws_NB samples x ws_T frames)
    int ws_NB = 0, ws_T = 0;
    // split-K workspace (partials) and ticket counters, see gemm_kernel
    float* sk_ws = nullptr;
    unsigned* sk_cnt = nullptr;
    static constexpr size_t SK_WS_FLOATS = (size_t)16 << 20, SK_CNT_N = 4096;     // 64 MiB: up to 1024 partial tiles of 128 x 128
    float *h = nullptr, *hd = nullptr, *g = nullptr, *skip = nullptr, *tmp = nullptr, *x0buf = nullptr;
    float* xwork = nullptr;                // the captured chain runs in place on this engine-owned roll buffer
    float *hd3 = nullptr, *g3 = nullptr;   // split-bf16 (S3) versions of hd and g: 1.5x the fp32 size
    int prec = 0;                          // 0: exact fp32 MFMA, 1: split-bf16 (bf16x3, 6 products)
    bool s3_ready = false;                 // the split-bf16 packings exist (built on first use: ensure_s3)
    double t_pack_s = 0.0, t_upload_s = 0.0, t_tables_s = 0.0, t_capture_s = 0.0;      // dr_cold_times
    int norm_framewise = 0;                // spectrogram normalisation: 0 imagewise, 1 framewise (norm_args[2])
    // conditioner tensors of the last dr_frontend: [L][fe_B][2Cp/4][fe_T][4]
    int fe_B = 0, fe_T = 0;
    size_t cond_cap = 0;
    float* cond = nullptr;
    float* cond_dummy = nullptr;   // one sample of readable memory for generation (no dr_frontend): never used
    // condition='trainable_spec': per-layer conditioner of the learned unconditional spectrogram, [L][2Cp/4][T][4]
    float* cond_tr = nullptr;
    int cond_tr_T = 0;
    // front-end workspace
    size_t fe_cap_wav = 0, fe_cap_pow = 0, fe_cap_log = 0, fe_cap_spec = 0, fe_cap_mm = 0;
    float *wav_pad = nullptr, *power = nullptr, *logmel = nullptr, *specP4 = nullptr, *mm = nullptr;

    // graph cache
    drh::GraphKey gkey;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    long long* dbg_ticks = nullptr;     // dr_bench_layer measurement hook
    unsigned long long* d_counts = nullptr;   // dr_frame_counts: result words, ticket, per-block partials (update.hip)
    size_t mm_scratch_off = 0;                // floats into `mm` where the multi-block min-max keeps its partials / tickets
    hipStream_t cap_stream = nullptr;   // capture happens here (the caller's stream may be the null stream)
    dr::DynParams* d_dyn = nullptr;         // per-call scalars of the captured chain (seed, batch offset, guidance weight)
    int* d_tsel = nullptr;              // per-sample steps of dr_forward_steps
    size_t tsel_cap = 0;
    bool use_dyn = false;               // set while the chain is being captured: run_step points the update at d_dyn

    // fused residual stack (stack_kernel): one persistent launch for the residual layers when every block of the
    // launch is resident at once; opt_stack 0 = always one launch per phase
    int opt_stack = 1;
    int opt_stack_xcd = 1;              // group-per-XCD block mapping (0: weight-panel-per-XCD)
    int opt_stack_fault = 0;            // test hook (option "stack_fault_test")
    int opt_stack_warm = 0;             // idle waves of the fused kernel warm the L2 for the next phase (measured: +-0)
    int n_cus = 0;
    unsigned* stack_bar = nullptr;      // [STACK_GROUPS][4] {arrivals, departures, generation, -}: the first two zero between launches
    unsigned* stack_err = nullptr;      // device address of the time-out flag (host-mapped memory)
    volatile unsigned* stack_err_host = nullptr;
    unsigned* stack_derr = nullptr;     // the same flag in device memory (what the kernels poll / test at launch start)
    unsigned* stack_xid = nullptr;      // [1024] (generation, XCC id) tags published by the blocks of the last launch
    unsigned* tail_bar = nullptr;       // group / pair counters of the tail kernel (own arrays, same protocol)
    unsigned* tail_pbar = nullptr;
    int opt_tail = 1;                   // fused step: layer 0's shared conv inside the stack launch + the tail kernel
    int64_t tail_launches = 0;
    float* xalt = nullptr;              // the tail kernel writes x_{t-1} here (it must not update x_t in place: other
                                        // blocks still read it); the chain ping-pongs between this and its roll buffer
    int64_t stack_fallbacks = 0;        // time-outs detected by dr_finish: each one switched this engine to per-phase launches
    int64_t stack_yields = 0;           // times this engine gave up fusing because another process was computing on the device (no time-out)
    int64_t stack_rearms = 0;           // times fused launches were switched back on (after a time-out: fused_rearm; after a yield: clean looks)
    int yielded_from = 0;               // the fused_stack value a yield switched off (0: none pending)
    int yield_clean = 0;                // looks in a row, in front of later chains, that found the GPU exclusive again
    int last_mode = 0;                  // DR_MODE_* of the most recently planned evaluation (dr_launch_state)
    unsigned tuning_epoch = 0;          // tuning_epoch() when the cached chain was captured
    long kfd_gpu_id = -1;               // the driver's id of this GPU in /sys/class/kfd (tenants.h); -1: unknown, no scans
    double last_tenant_scan_s = -1.0;
    hipStream_t graph_stream = nullptr; // where the captured chain was last launched
    bool graph_stream_set = false;
    bool unverified = false;            // persistent launches have been issued since the last check of the time-out flag
    hipStream_t fused_stream = nullptr; // ... on this stream (the last one): what a check synchronises before it reads the flag
    int opt_blocked = 2;                // option "blocked_accumulation": 2 (default) = every fp32 flavour that has a blocked form, 1 = 128-frame blocks keep one chain per output (-0.5 % per chain, 2-3x the rounding error)
    int opt_rearm = 0;                  // option "fused_rearm": clean chains after a time-out before fusing again (0: never)
    int healed_from = 0;                // the fused_stack value a time-out switched off (0: none pending re-arm)
    int clean_chains = 0;               // chains finished cleanly since that time-out
    float* xsave = nullptr;             // dr_sample_checked: copy of x_T, so that a timed-out chain can be re-run
    size_t xsave_cap = 0;
    long long* stack_dbg = nullptr;     // phase tick marks of block 0 (dr_debug_stack_ticks)
    int stack_dbg_on = 0;
    int64_t stack_launches = 0;         // fused-kernel launches issued (captured launches count once, at capture)
    static constexpr int STACK_GROUPS = 512;

    // profiling of the dominant kernel
    double prof_flops = 0.0;            // algorithmic FLOPs of the timed launches
    std::string prof_name;
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
};

namespace drh {

inline int fail(dr_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(e, expr)                                                                         \
    do {                                                                                        \
        hipError_t _st = (expr);                                                                \
        if (_st != hipSuccess)                                                                  \
            return drh::fail((e), DR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_st),   \
                             __FILE__, __LINE__);                                               \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Every entry point runs on the engine's device and leaves the caller's current device as it found it (a process
// that drives several GPUs must not have its device switched by constructing or calling an engine).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// roctx ranges around the host-side phases (rocprofv3 --marker-trace shows them next to the kernel trace).  The
// marker library is looked up at run time: no link-time dependency, silent no-ops when it is absent.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
inline Roctx& roctx() { static Roctx r; return r; }
struct Range {
    explicit Range(const char* name) { if (roctx().push) roctx().push(name); }
    ~Range() { if (roctx().pop) roctx().pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

// ---- abi.hip
int clear_stack_timeout(dr_engine* e);
int set_option(dr_engine* e, const char* name, int value, bool lab);      // lab: the names of dr_debug_set_option too
void set_kfd_root(const char* root);                                      // dr_debug_kfd_root

// ---- pack.hip
int dev_alloc(dr_engine* e, float** p, size_t floats, bool zero = true);
const std::vector<float>* find_param(dr_engine* e, const std::string& name);
size_t expected_numel(const dr_engine* e, const std::string& name);
int ensure_s3(dr_engine* e);            // the split-bf16 packings, built on first use
int commit(dr_engine* e, hipStream_t st);

// ---- plan.hip
struct Tile { int flavor, n; };
Tile pick_tile(int MT, int NB, int T, int taps, int dil, int prec, int epi, bool allow16, bool wide32 = false);
int pick_ni(int MT, int NB, int T, int taps, int dil, int prec = 0);
hipError_t launch_tiled(const GemmArgs& a, int epi, Tile t, hipStream_t s, int prec);
Tile pick_pointwise_tile(int MT, int NB, int T, int prec, int kchunks = 0);
void allow_splitk(const dr_engine* e, GemmArgs& a);
constexpr int MAX_DEVICES = 64;
extern const float* g_zero_vecs[MAX_DEVICES];
const float* zero_vec();
GemmArgs p4_gemm(const float* Wp, const float* bias, int MT, const float* X, int planes, int NB, int T);
void p4_out(GemmArgs& a, float* Y, int planes, int T, int rows);
int build_trainable_cond(dr_engine* e, int T);
void drop_graph(dr_engine* e);
int ensure_workspace(dr_engine* e, int NB, int T);
// What run_step offers run_network so that a whole reverse step becomes TWO launches (the residual stack incl. layer 0's
// shared contraction + the tail kernel: skip / output projection, update, next input projection) where the fused
// kernel applies; run_network reports back what it took.
struct TailPlan {
    UpdateArgs u{};            // this step's update (x = x_t, read only by the tail kernel)
    float* x_out = nullptr;    // where the tail kernel writes x_{t-1}
    int u_B = 0;               // rolls
    int next_t = -1;           // >= 0: the chain continues with step next_t (its input projection joins the tail)
    bool skip_inproj = false;  // h / hd of THIS step (and, guided, layer 0's g) were written by the previous step's tail
    bool done = false;         // out: the tail kernel ran (update included, result in x_out)
    bool inproj_done = false;  // out: ... and it wrote the next step's h / hd (and layer 0's g for a guided pair)
};
int run_network(dr_engine* e, const float* xin, int bmod, int NB, int n_cond, int T, int t, float* x0_out,
                hipStream_t st, bool zero_spec = false, const int* tsel = nullptr, TailPlan* tail = nullptr);
int sampler_shape(int sampler, int B, int& NB, int& n_cond, int& family, bool& zero_spec);
int sampler_shape(int sampler, int B, int& NB, int& n_cond);
struct ChainState { bool inproj_ready = false; int next_t = -1; };
int run_step(dr_engine* e, int sampler, float* x, const float* noise, int B, int T, int t, float w, uint64_t seed,
             int first_sample, hipStream_t st, float** result = nullptr, ChainState* chain = nullptr);

}  // namespace drh
