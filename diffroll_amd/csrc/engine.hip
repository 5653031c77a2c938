// Host side of the DiffRoll sampling engine: weight packing, hoisted tables, per-step launch
// sequences, hipGraph capture of the reverse chain, and the C-ABI of include/diffroll_amd.h.
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

#include "../../include/diffroll_amd.h"
#include "kernels.h"

using namespace dr;

namespace {

thread_local std::string g_create_error;

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

}  // namespace

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
    std::vector<LayerW> layers;
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
    GraphKey gkey;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    long long* dbg_ticks = nullptr;     // dr_bench_layer measurement hook
    unsigned long long* d_counts = nullptr;   // dr_frame_counts accumulator
    hipStream_t cap_stream = nullptr;   // capture happens here (the caller's stream may be the null stream)
    DynParams* d_dyn = nullptr;         // per-call scalars of the captured chain (seed, batch offset, guidance weight)
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
    bool unverified = false;            // persistent launches have been issued since the last check of the time-out flag
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

namespace {

int fail(dr_engine* e, int code, const char* fmt, ...) {
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
            return fail((e), DR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_st),   \
                        __FILE__, __LINE__);                                                    \
    } while (0)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Host-side packing of the layers is embarrassingly parallel (one task per residual layer): dr_commit is on the
// critical path of a one-shot process (sampling.py: load checkpoint -> one batch), where it used to cost more than
// the whole 50-step chain of a single clip.  DR_PACK_THREADS caps the worker count (1 = serial).
template <class F>
void parallel_for(int n, F fn) {
    static const int cap = getenv("DR_PACK_THREADS") ? atoi(getenv("DR_PACK_THREADS")) : 16;
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = std::max(1, std::min(std::min(n, cap), hw > 0 ? hw : 1));
    if (nt == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
    for (auto& t : th) t.join();
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
Roctx& roctx() { static Roctx r; return r; }
struct Range {
    explicit Range(const char* name) { if (roctx().push) roctx().push(name); }
    ~Range() { if (roctx().pop) roctx().pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

// ---- weight packing (layout: kernels.h) -------------------------------------------------------
// get(prow, ch, tap) returns the (zero-padded) weight for packed row prow, input channel ch.
template <class F>
std::vector<float> pack_weights(int MT, int kchunks, int taps, F get) {
    std::vector<float> out((size_t)MT * kchunks * taps * 4096);
    size_t o = 0;
    for (int mt = 0; mt < MT; ++mt)
        for (int kc = 0; kc < kchunks; ++kc)
            for (int j = 0; j < taps; ++j)
                for (int gq = 0; gq < 4; ++gq)
                    for (int hi = 0; hi < 2; ++hi)
                        for (int row = 0; row < 128; ++row)
                            for (int i = 0; i < 4; ++i)
                                out[o++] = get(mt * 128 + row, kc * 32 + gq * 8 + hi * 4 + i, j);
    return out;
}

// split-bf16 packing: [mtile][kchunk32][tap][g16 = 2][piece = 3][kq = 2][row = 128][8 bf16]
inline uint16_t bf16_rne(float x) {
    uint32_t v;
    memcpy(&v, &x, 4);
    return (uint16_t)((v + 0x7FFFu + ((v >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(uint16_t b) {
    const uint32_t v = (uint32_t)b << 16;
    float f;
    memcpy(&f, &v, 4);
    return f;
}
template <class F>
std::vector<uint16_t> pack_weights_s3(int MT, int kchunks, int taps, F get) {
    std::vector<uint16_t> out((size_t)MT * kchunks * taps * 12288);
    size_t slab = 0;
    for (int mt = 0; mt < MT; ++mt)
        for (int kc = 0; kc < kchunks; ++kc)
            for (int j = 0; j < taps; ++j, ++slab)
                for (int g = 0; g < 2; ++g)
                    for (int kq = 0; kq < 2; ++kq)
                        for (int row = 0; row < 128; ++row)
                            for (int i = 0; i < 8; ++i) {
                                const float w = get(mt * 128 + row, kc * 32 + g * 16 + kq * 8 + i, j);
                                const uint16_t p0 = bf16_rne(w);
                                const float r1 = w - bf16_to_f32(p0);
                                const uint16_t p1 = bf16_rne(r1);
                                const uint16_t p2 = bf16_rne(r1 - bf16_to_f32(p1));
                                const uint16_t pc[3] = {p0, p1, p2};
                                for (int pz = 0; pz < 3; ++pz)
                                    out[slab * 12288 + ((((size_t)g * 3 + pz) * 2 + kq) * 128 + row) * 8 + i] = pc[pz];
                            }
    return out;
}

// paired row map: packed row -> (which half mi, channel c); a 128-row tile = 4 consumer waves x
// [16 gate (cos) rows, 16 filter (sin) rows] of the same 16 channels (gemm_body.h: pairing inside one MFMA tile)
inline void paired_row(int prow, int& mi, int& c) {
    const int mt = prow >> 7, rr = prow & 127;
    const int w = rr >> 5, r = rr & 31;
    mi = r >> 4;
    c = mt * 64 + w * 16 + (r & 15);
}

// Host -> device copies of the packed constants go through two pinned staging buffers (a pageable hipMemcpy of the 347 MB
// of a full-size network ran at 1.8 GB/s - 0.2 s of a one-shot process's start-up; staged it is a host memcpy overlapped
// with a DMA at link rate).  One stager per process and device thread; small copies (< 64 KiB) take the plain path.
struct Stager {
    static constexpr size_t CHUNK = (size_t)16 << 20;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    bool busy[2] = {false, false};
    bool ok = false;
    Stager() {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
        for (int i = 0; i < 2; ++i)
            if (hipHostMalloc(&pin[i], CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return;
        ok = true;
    }
    hipError_t copy(void* dst, const void* src, size_t bytes) {
        if (!ok || bytes < (64u << 10)) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
        int b = 0;
        for (size_t off = 0; off < bytes; off += CHUNK, b ^= 1) {
            const size_t n = std::min(CHUNK, bytes - off);
            hipError_t e;
            if (busy[b] && (e = hipEventSynchronize(done[b])) != hipSuccess) return e;
            memcpy(pin[b], (const char*)src + off, n);
            if ((e = hipMemcpyAsync((char*)dst + off, pin[b], n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
            if ((e = hipEventRecord(done[b], st)) != hipSuccess) return e;
            busy[b] = true;
        }
        return hipSuccess;          // (in flight: drain() before the data is used)
    }
    hipError_t drain() {
        busy[0] = busy[1] = false;
        return ok ? hipStreamSynchronize(st) : hipSuccess;
    }
};
Stager& stager() {          // one per (host thread, device): its stream and pinned buffers belong to the device current at creation
    static thread_local std::map<int, Stager*> per_device;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Stager*& s = per_device[dev];
    if (!s) s = new Stager();
    return *s;
}

int upload_bytes(dr_engine* e, const void* data, size_t bytes, float** out) {
    void* p = nullptr;
    HIPCHK(e, hipMalloc(&p, std::max<size_t>(bytes, 16)));
    e->owned.push_back(p);
    HIPCHK(e, stager().copy(p, data, bytes));
    *out = (float*)p;
    return DR_OK;
}

int upload(dr_engine* e, const std::vector<float>& v, float** out) {
    return upload_bytes(e, v.data(), v.size() * sizeof(float), out);
}

int dev_alloc(dr_engine* e, float** p, size_t floats, bool zero = true) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    void* q = nullptr;
    HIPCHK(e, hipMalloc(&q, std::max<size_t>(floats, 4) * sizeof(float)));
    if (zero) HIPCHK(e, hipMemset(q, 0, std::max<size_t>(floats, 4) * sizeof(float)));
    *p = (float*)q;
    return DR_OK;
}

const std::vector<float>* find_param(dr_engine* e, const std::string& name) {
    auto it = e->params.find(name);
    return it == e->params.end() ? nullptr : &it->second;
}

size_t expected_numel(const dr_engine* e, const std::string& name) {
    const size_t C = e->C, K = e->K, NM = e->NM;
    if (name == "input_projection.weight") return C * 88;
    if (name == "input_projection.bias") return C;
    if (name == "diffusion_embedding.projection1.weight") return 512 * 128;
    if (name == "diffusion_embedding.projection1.bias") return 512;
    if (name == "diffusion_embedding.projection2.weight") return 512 * 512;
    if (name == "diffusion_embedding.projection2.bias") return 512;
    if (name == "skip_projection.weight") return C * C;
    if (name == "skip_projection.bias") return C;
    if (name == "output_projection.weight") return 88 * C;
    if (name == "output_projection.bias") return 88;
    if (name == "trainable_parameters") return NM * 641;      // condition='trainable_spec' (model/diffwave.py:601)
    const std::string pre = "residual_layers.";
    if (name.compare(0, pre.size(), pre) == 0) {
        const size_t dot = name.find('.', pre.size());
        if (dot == std::string::npos) return 0;
        const int li = atoi(name.substr(pre.size(), dot - pre.size()).c_str());
        if (li < 0 || li >= e->L) return 0;
        const std::string rest = name.substr(dot + 1);
        if (rest == "dilated_conv.weight") return 2 * C * C * K;
        if (rest == "dilated_conv.bias") return 2 * C;
        if (rest == "diffusion_projection.weight") return C * 512;
        if (rest == "diffusion_projection.bias") return C;
        if (rest == "conditioner_projection.weight") return 2 * C * NM;
        if (rest == "conditioner_projection.bias") return 2 * C;
        if (rest == "output_projection.weight") return 2 * C * C;
        if (rest == "output_projection.bias") return 2 * C;
    }
    return 0;
}

// Frame-tile size (NI = 1: 64 frames, 2: 128 frames per block) for a GEMM of MT row tiles over NB samples
// of T frames: minimise (block rounds over the 256 CUs) x (tile cost); 128-frame tiles win ties (half the
// weight traffic per MFMA).  One block per CU is resident (LDS / 512-thread blocks).
// Frame-tile choice for a GEMM of MT row tiles over NB samples of T frames.  flavor 0: gemm_kernel
// (32x32 MFMA) with NI = n (64*n frames per block); flavor 1: gemm16_kernel (16x16 MFMA) with NJ = n
// (32*n frames per block, fp32 hot kernels only).  Cost = (block rounds over the 256 CUs, one block per
// CU) x (frames per block); 16x16 tiles carry a small penalty (more operand reads per MFMA), 128-frame
// 32x32 tiles win ties.
struct Tile { int flavor, n; };
// The 32x32 conv kernels may be cut in K into more blocks than CUs (launch_gemm's split-K cost model: equal blocks run
// in lockstep rounds, the exchange costs ~(4 + ks) us): a width whose tile count fills the chip unevenly can still win
// that way - 2 guided 640-frame clips: 320 64-frame tiles cut 4x, 3209 vs 3592 us per step on 224 96-frame tiles of
// the 16x16 kernel, which has no split.  Cost in the units of pick_tile (block rounds x frames per block x penalty) of
// the best split that needs MORE than one resident round, with a 5 % handicap; 1e30 if there is none.
double split_cost(long blocks, int bn, double pen, int MT, int taps) {
    const int nchunks = 2 * MT;                                                  // 32-channel chunks of K (convs: KS = 1)
    // the launcher's own decision and price (plan_ksplit, gemm.hip): what it WILL do with this launch
    const KSplitPlan p = plan_ksplit(blocks, nchunks, nchunks, taps, bn / 64, 0, dr_engine::SK_WS_FLOATS, dr_engine::SK_CNT_N);
    if (p.ks <= 1 || blocks * p.ks <= 256) return 1e30;                          // (one resident round: priced by the caller)
    const double us_per_frame = p.us_unsplit / ((double)((blocks + 255) / 256) * bn);      // us of one frame column of a full-K tile
    return 1.05 * pen * p.us / us_per_frame;
}
// wide32: the 96 / 160-frame flavours of the 32x32 conv kernel (n = 3 / 5; fp32 gated conv with blocked accumulation) may be
// used - they take the place of the 16x16 kernels of those widths, which have no blocked form (option blocked_accumulation = 2)
Tile pick_tile(int MT, int NB, int T, int taps, int dil, int prec, int epi, bool allow16, bool wide32 = false) {
    static const char* forced = getenv("DR_TILE");     // tuning experiments: "32:2", "16:5", ... (if it fits)
    const int halo = ((taps - 1) / 2) * dil;
    struct Cand { int flavor, n, bn; double pen; };
    const Cand cands[] = {{0, 2, 128, 1.0}, {0, 5, 160, 1.04}, {1, 5, 160, 1.04}, {0, 3, 96, 1.04}, {1, 3, 96, 1.04}, {0, 1, 64, 1.0}};
    auto feasible = [&](const Cand& c) {
        if (c.flavor == 1 && (!allow16 || prec != 0)) return false;
        if (c.flavor == 0 && (c.n == 3 || c.n == 5) && (!wide32 || prec != 0 || epi != EPI_GATE || taps == 1)) return false;
        const int ks = (taps == 1) ? 2 : 1;
        const size_t lds = (c.flavor == 0)
            ? gemm_lds_bytes(c.n, (taps == 1 && c.n == 1) ? 4 : ks, taps, dil, prec, epi)
            : (size_t)2 * 8 * ks * (c.bn + 2 * halo) * 16 + (epi == EPI_RES_SKIP ? (size_t)32 * c.bn * 16 : 0);
        return lds <= 160 * 1024;
    };
    if (forced && strlen(forced) >= 4) {
        const int ff = forced[0] == '1' ? 1 : 0, fn = atoi(forced + 3);
        for (const Cand& c : cands)
            if (c.flavor == ff && c.n == fn && feasible(c)) return Tile{ff, fn};
    }
    Tile best{0, 1};
    double best_cost = 1e30;
    for (const Cand& c : cands) {
        if (!feasible(c)) continue;
        const long blocks = (long)MT * NB * ((T + c.bn - 1) / c.bn);
        double cost = (double)((blocks + 255) / 256) * c.bn * c.pen;
        if (c.flavor == 0 && c.n <= 2 && prec == 0 && epi == EPI_GATE && taps > 1 && allow16)
            cost = std::min(cost, split_cost(blocks, c.bn, c.pen, MT, taps));
        if (cost < best_cost - 1e-9) { best_cost = cost; best = Tile{c.flavor, c.n}; }
    }
    return best;
}
int pick_ni(int MT, int NB, int T, int taps, int dil, int prec = 0) {
    return pick_tile(MT, NB, T, taps, dil, prec, EPI_GATE, false).n;
}
hipError_t launch_tiled(const GemmArgs& a, int epi, Tile t, hipStream_t s, int prec) {
    if (t.flavor == 3) return launch_pointwise_ksplit(a, t.n, s);
    if (t.flavor == 2) return launch_pointwise(a, t.n, s);
    return t.flavor == 1 ? launch_gemm16(a, epi, t.n, s) : launch_gemm(a, epi, t.n, s, prec);
}
// tile of the 1x1 residual/skip GEMM: flavor 2 = operands direct from L2 (pw_kernel), fp32 only
Tile pick_pointwise_tile(int MT, int NB, int T, int prec, int kchunks = 0) {      // kchunks: 32-channel slabs of K (default: MT tiles cover all rows)
    if (prec) return Tile{0, 1};
    static const int pw = getenv("DR_PW") ? atoi(getenv("DR_PW")) : 1;          // tuning experiments: 0 = LDS-staged kernels
    static const int pw_ni = getenv("DR_PW_NW") ? atoi(getenv("DR_PW_NW")) : 0;    // force 32*NW-frame blocks
    if (!pw) return pick_tile(MT, NB, T, 1, 1, 0, EPI_RES_SKIP, true);
    if (pw_ni) return Tile{2, pw_ni};
    // launches that cannot fill half the chip even with 64-frame blocks (single clips): 32-row x 32-frame tiles whose
    // four waves split K in-block (flavor 3, pwk_kernel: 256 blocks at config 1, 13.9 -> 8 us per launch); without it
    // (DR_PWK=0, K splitting pinned off, a channel count that is not a multiple of 128) the LDS-staged kernel with
    // split-K through the workspace.  (Measured and rejected in round 3: 32-frame blocks of the direct kernel instead -
    // 64 blocks at config 1 - 35.2 vs 34.0 ms per chain.)
    if ((long)MT * NB * ((T + 63) / 64) <= 128) {
        static const int pwk = getenv("DR_PWK") ? atoi(getenv("DR_PWK")) : 1;
        static const int ks_max = getenv("DR_KSPLIT_MAX") ? atoi(getenv("DR_KSPLIT_MAX")) : 16;
        if (pwk && ks_max > 1 && (kchunks ? kchunks : 2 * MT) % 4 == 0)
            return Tile{3, (long)4 * MT * NB * ((T + 31) / 32) <= 512 ? 1 : 2};
        return pick_tile(MT, NB, T, 1, 1, 0, EPI_RES_SKIP, true);
    }
    // cost = block rounds over the 256 CUs x frames per block; 64-frame blocks carry a measured 7 % penalty
    // (twice the operand loads per MFMA)
    struct Cand { int nw; double pen; };
    const Cand cands[] = {{4, 1.0}, {5, 1.0}, {3, 1.02}, {2, 1.07}};
    Tile best{2, 4};
    double best_cost = 1e30;
    for (const Cand& c : cands) {
        const int bn = 32 * c.nw;
        const long blocks = (long)MT * NB * ((T + bn - 1) / bn);
        const double cost = (double)((blocks + 255) / 256) * bn * c.pen;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = Tile{2, c.nw}; }
    }
    return best;
}
// let the launcher split K when the launch under-fills the chip (single clips, narrow projections)
void allow_splitk(const dr_engine* e, GemmArgs& a) {
    a.ws = e->sk_ws; a.ws_cnt = e->sk_cnt;
    a.ws_floats = dr_engine::SK_WS_FLOATS; a.ws_cnt_n = dr_engine::SK_CNT_N;
}

// common GemmArgs for a P4 activation input [NB][planes][T][4]
// Device zero vector (a never-null bias / d2 operand: epilogue loads are unconditional), one per device, shared by
// the engines of the process on that device and never freed.
constexpr int MAX_DEVICES = 64;
const float* g_zero_vecs[MAX_DEVICES] = {};
const float* zero_vec() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (dev >= 0 && dev < MAX_DEVICES) ? g_zero_vecs[dev] : nullptr;
}

GemmArgs p4_gemm(const float* Wp, const float* bias, int MT, const float* X, int planes, int NB, int T) {
    GemmArgs a{};
    a.d2 = zero_vec();
    a.Wp = Wp; a.bias = bias ? bias : zero_vec(); a.MT = MT;
    a.X = X; a.x_bs = (long)planes * T * 4; a.x_ps = (long)T * 4; a.x_fs = 4; a.x_planes = planes;
    a.kchunks = (planes + 7) / 8;
    a.NB = NB; a.T = T; a.taps = 1; a.dil = 1; a.alpha = 1.f;
    return a;
}
void p4_out(GemmArgs& a, float* Y, int planes, int T, int rows) {
    a.Y = Y; a.y_bs = (long)planes * T * 4; a.y_ps = (long)T * 4; a.y_fs = 4; a.y_rows = rows;
}

// condition='trainable_spec' (model/diffwave.py:600-606, :656-658): the unconditional branch feeds the learned
// (n_mels, 641) spectrogram, trimmed to the roll length, through every layer's conditioner projection.  Like the
// conditional tensors it is hoisted: [L][2Cp/4][T][4], rebuilt when T changes (one-time, null stream).
int build_trainable_cond(dr_engine* e, int T) {
    const std::vector<float>* P = find_param(e, "trainable_parameters");
    if (!P) return DR_OK;
    if (e->cond_tr && e->cond_tr_T == T) return DR_OK;
    if (T > 641) return fail(e, DR_EINVAL, "condition='trainable_spec' holds 641 frames, roll has %d", T);
    const int NM = e->NM, Cp = e->Cp, mel_planes = (NM + 3) / 4;
    std::vector<float> sp((size_t)mel_planes * T * 4, 0.f);      // P4 image of P[:, :T]
    for (int m = 0; m < NM; ++m)
        for (int t = 0; t < T; ++t) sp[((size_t)(m >> 2) * T + t) * 4 + (m & 3)] = (*P)[(size_t)m * 641 + t];
    float* d_sp = nullptr;
    int rc;
    if ((rc = dev_alloc(e, &d_sp, sp.size(), false))) return rc;
    HIPCHK(e, hipMemcpy(d_sp, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = dev_alloc(e, &e->cond_tr, (size_t)e->L * 2 * Cp * T))) { (void)hipFree(d_sp); return rc; }
    for (int l = 0; l < e->L; ++l) {
        const LayerW& w = e->layers[l];
        GemmArgs a = p4_gemm(w.cond_w, w.cond_b, Cp / 64, d_sp, mel_planes, 1, T);
        p4_out(a, e->cond_tr + (size_t)l * 2 * Cp * T, 2 * Cp / 4, T, 2 * Cp);
        HIPCHK(e, launch_gemm(a, EPI_PLAIN, 2, nullptr));
    }
    HIPCHK(e, hipDeviceSynchronize());
    (void)hipFree(d_sp);
    e->cond_tr_T = T;
    return DR_OK;
}

void drop_graph(dr_engine* e);

int ensure_workspace(dr_engine* e, int NB, int T) {
    if (NB <= e->ws_NB && T == e->ws_T) return DR_OK;
    drop_graph(e);       // a captured chain holds the addresses of the buffers that are about to be replaced
    const int nb = std::max(NB, e->ws_T == T ? e->ws_NB : 0);
    const size_t act = (size_t)nb * e->Cp * T;
    int rc;
    if ((rc = dev_alloc(e, &e->h, act))) return rc;
    if ((rc = dev_alloc(e, &e->hd, act))) return rc;
    if ((rc = dev_alloc(e, &e->hd3, act + act / 2))) return rc;
    if ((rc = dev_alloc(e, &e->g3, act + act / 2))) return rc;
    if ((rc = dev_alloc(e, &e->g, act))) return rc;
    if ((rc = dev_alloc(e, &e->skip, act))) return rc;
    if ((rc = dev_alloc(e, &e->tmp, act))) return rc;
    if ((rc = dev_alloc(e, &e->x0buf, (size_t)nb * T * 88))) return rc;
    if ((rc = dev_alloc(e, &e->xwork, (size_t)nb * T * 88))) return rc;
    if ((rc = dev_alloc(e, &e->xalt, (size_t)nb * T * 88))) return rc;
    if ((rc = dev_alloc(e, &e->cond_dummy, (size_t)2 * e->Cp * T))) return rc;
    e->ws_NB = nb;
    e->ws_T = T;
    if ((rc = build_trainable_cond(e, T))) return rc;
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    e->gkey = GraphKey{};
    return DR_OK;
}

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

// One network evaluation for NB samples (first n_cond conditional) at step t.
//   xin (B,T,88) rows are used modulo bmod (classifier-free batching: 2B evaluations of B inputs).
int run_network(dr_engine* e, const float* xin, int bmod, int NB, int n_cond, int T, int t, float* x0_out,
                hipStream_t st, bool zero_spec = false, const int* tsel = nullptr, TailPlan* tail = nullptr) {
    // tsel (device, NB ints): per-sample diffusion steps (forward() with a (B,) step tensor); else step t for all
    const int Cp = e->Cp, P = Cp / 4, L = e->L;
    const int prec = e->prec;
    const long act_bs = (long)Cp * T, s3_bs = act_bs + act_bs / 2;   // per-sample sizes (4-byte units)
    // S3 input description of an activation tensor with Cp channels
    auto s3_in = [&](GemmArgs& a, const float* X) {
        a.X = X; a.x_bs = s3_bs; a.x_piece = (long)(Cp / 8) * T * 4; a.x_ps = (long)T * 4; a.x_fs = 4;
        a.x_planes = Cp / 8; a.kchunks = Cp / 32;
    };
    // ---- fused residual stack: the layers as ONE persistent launch when all its blocks are resident at once ----
    // (exact fp32 only; the first layer's conv stays a launch of its own under classifier-free guidance, where it
    // is contracted once per (conditional, unconditional) pair)
    int stack_from = -1;                   // first phase run by the fused kernel (-1: none)
    int stack_ni = 0, stack_chunks = 1;    // flavour, and how many sample chunks the evaluation is launched in
    bool fused_step = false;
    // (the split-bf16 precision has its own flavour of the kernel: 128-channel S3 chunks in the 1x1 phases need Cp % 128 == 0)
    static const int stack3 = getenv("DR_STACK3") ? atoi(getenv("DR_STACK3")) : 1;
    if (e->opt_stack && (prec == 0 || (stack3 && Cp % 128 == 0)) && L <= DR_STACK_MAX_LAYERS && e->n_cus > 0) {
        int maxdil = 1;
        for (int l = 0; l < L; ++l) maxdil = std::max(maxdil, e->layers[l].dil);
        // Flavours 1 / 2 (128 packed rows x 64 / 128 frames per block) are chosen automatically; DR_STACK_FL=n pins one
        // (tests / measurements).
        static const int fl_force = getenv("DR_STACK_FL") ? atoi(getenv("DR_STACK_FL")) : 0;
        // A launch must be ONE resident round (groups spin on each other), so an evaluation with more samples than
        // fit is launched in balanced CHUNKS of samples, one fused launch after the other (samples are independent).
        // Cost model per frame-tile width, as pick_tile's: (block rounds over the CUs) x (frames per block) x a
        // per-width penalty (64-frame blocks load twice the weight fragments per MFMA; 16x16 tiles more operands) -
        // for the fused kernel rounds = chunks, minus what fusing was measured to save; fused wins if its best width
        // costs no more than the per-phase launches' best width.
        const int MT = Cp / 64;
        auto per_phase_cost = [&]() {
            double best = 1e30;
            const struct { int bn; double pen; } cands[] = {{64, 1.0 / 0.93}, {96, 1.04}, {128, 1.0}, {160, 1.04}};
            for (const auto& c : cands) {
                const long blocks = (long)MT * NB * ((T + c.bn - 1) / c.bn);
                best = std::min(best, (double)((blocks + e->n_cus - 1) / e->n_cus) * c.bn * c.pen);
                // (the 32x32 widths may split K beyond one round: 20 guided clips, 640 64-frame tiles cut 2x, 6166 us
                // per step against 7074 as three fused launches of 13-14 evaluations)
                if (c.bn == 64 || c.bn == 128) best = std::min(best, split_cost(blocks, c.bn, c.pen, MT, e->K));
            }
            return best;
        };
        double best = 1e30;
        for (int fl : {1, 2}) {
            if (fl_force && fl != fl_force) continue;
            const int bn = stack_tile_frames(fl);
            const long gsize = stack_group_blocks(fl, Cp, T);                           // blocks per sample
            const long cap = std::min<long>(e->n_cus, 1024) / gsize;                    // samples per launch
            if (cap < 1 || (prec ? stack3_lds_bytes(fl, e->K, maxdil) : stack_lds_bytes(fl, e->K, maxdil)) > 160 * 1024) continue;
            const long chunks = (NB + cap - 1) / cap;
            if ((NB + chunks - 1) / chunks > dr_engine::STACK_GROUPS) continue;
            // (what fusing saves is per-launch overhead, which the per-phase launches amortise over their rounds:
            // measured +2.5 % at one round, +1.1 % at two (B = 32 guided clips per GPU), nothing at four)
            const double cost = (1.0 - 0.025 / chunks) * chunks * bn * (fl == 1 ? 1.0 / 0.93 : 1.0);
            // (a single launch that leaves more than a fifth of the CUs idle is better served by the per-phase kernels'
            // split-K, which this cost model does not see: they cut the same work into many short blocks that balance
            // over all CUs - 8 evaluations x 125 frames (half the chip): 1365 vs 2422 us per step, 10 / 12 evaluations
            // (62 / 75 %): 1994 / 2022 vs 2425, 14 (87 %): 2526 vs 2424; opt_stack == 2 fuses regardless: tests)
            const bool ok = e->opt_stack == 2 || (chunks == 1 ? 5 * NB * gsize > 4 * (long)e->n_cus : true);
            if (ok && cost < best) { best = cost; stack_ni = fl; stack_chunks = (int)chunks; }
        }
        if (stack_ni && e->opt_stack != 2 && best > per_phase_cost()) stack_ni = 0;
        const bool dual0 = (bmod > 0 && NB == 2 * bmod && n_cond == bmod);
        if (stack_ni) stack_from = dual0 ? 1 : 0;
        // fused step (option "fused_tail"): everything behind the stack launch - skip / output projection, update, and
        // for a chain the next step's input projection and (guided) shared first-layer conv - is one tail launch,
        // when the evaluation is ONE fused launch of the 32x32-MFMA flavours
        fused_step = stack_ni && stack_chunks == 1 && e->opt_tail && !tsel && prec == 0;      // (the tail kernel is fp32 only)
    }
    const bool use_tail = fused_step && tail != nullptr;
    // input projection + relu (model/diffwave.py:667-668) - unless the previous step's tail kernel already wrote h / hd
    if (!(use_tail && tail->skip_inproj)) {
        GemmArgs a{};
        a.Wp = e->in_w; a.bias = e->in_b; a.MT = (Cp + 127) / 128;
        a.X = xin; a.x_bs = (long)T * 88; a.x_ps = 4; a.x_fs = 88; a.x_planes = 22; a.x_bmod = bmod;
        a.kchunks = 3; a.NB = NB; a.T = T; a.taps = 1; a.dil = 1; a.alpha = 1.f;
        p4_out(a, e->h, P, T, Cp);
        // hd = h + d_0 (model/diffwave.py:138-139), fp32 P4 or split-bf16 for the first dilated conv
        a.d2 = e->d_dtab + (tsel ? 0 : (size_t)t * L * Cp);
        a.tsel = tsel; a.d2_ts = (long)L * Cp;
        if (prec) { a.Y2 = e->hd3; a.y2_bs = s3_bs; a.out_s3 = 2; }
        else { a.Y2 = e->hd; a.y2_bs = act_bs; }
        allow_splitk(e, a);
        HIPCHK(e, launch_gemm(a, EPI_RELU, pick_ni(a.MT, NB, T, 1, 1), st));
    }

    auto launch_stack_range = [&](int p0, int p1) -> int {
        int maxdil = 1;
        for (int l = 0; l < L; ++l) maxdil = std::max(maxdil, e->layers[l].dil);
        const long act_n = (long)Cp * T, c_bs = (long)2 * Cp * T;
        int b0 = 0;
        for (int ck = 0; ck < stack_chunks; ++ck) {
            const int nb = NB / stack_chunks + (ck < NB % stack_chunks ? 1 : 0);      // balanced chunk sizes
            StackArgs sa{};
            sa.h = e->h + b0 * act_n; sa.hd = e->hd + b0 * act_n; sa.g = e->g + b0 * act_n; sa.skip = e->skip + b0 * act_n;
            if (prec) { sa.hd = e->hd3 + b0 * (act_n + act_n / 2); sa.g = e->g3 + b0 * (act_n + act_n / 2); }      // the S3 tensors
            sa.d2 = e->d_dtab + (tsel ? 0 : (size_t)t * L * Cp);
            sa.tsel = tsel ? tsel + b0 : nullptr; sa.d2_ts = (long)L * Cp;
            sa.zero = zero_vec();
            sa.NB = nb; sa.T = T; sa.Cp = Cp; sa.taps = e->K; sa.L = L;
            sa.n_cond = std::max(0, std::min(nb, n_cond - b0));
            sa.c_bs = c_bs;
            sa.p0 = p0; sa.p1 = p1;
            sa.xcd_n = e->opt_stack_xcd;
            sa.warm = e->opt_stack_warm;
            sa.fault = e->opt_stack_fault;
            sa.fold128 = e->opt_blocked >= 2;
            sa.bar = e->stack_bar; sa.err = e->stack_err; sa.derr = e->stack_derr; sa.xid = e->stack_xid;
            sa.dbg = e->stack_dbg_on ? e->stack_dbg : nullptr;
            for (int l = 0; l < L; ++l) {
                const LayerW& w = e->layers[l];
                StackLayer& y = sa.layer[l];
                y.conv_w = prec ? w.conv_w3 : w.conv_w; y.conv_b = w.conv_b;
                y.conv_b2 = zero_spec ? w.conv_b_z : w.conv_b_u;
                y.cond2 = nullptr;
                if (e->cond_tr && !zero_spec) { y.cond2 = e->cond_tr + (size_t)l * 2 * Cp * T; y.conv_b2 = w.conv_b; }
                // conditional samples of this chunk start at sample b0 of the layer's conditioner tensor (a chunk
                // without any keeps a readable pointer: the kernel prefetches, then ignores it)
                y.cond = e->cond ? e->cond + (size_t)l * e->fe_B * 2 * Cp * T + (b0 < n_cond ? (size_t)b0 * c_bs : 0)
                                 : e->cond_dummy;
                y.out_w = prec ? w.out_w3 : w.out_w; y.out_b = w.out_b; y.dil = w.dil;
            }
            const bool timed = e->prof && e->prof_used < e->prof_events.size();
            if (timed) HIPCHK(e, hipEventRecord(e->prof_events[e->prof_used].first, st));
            HIPCHK(e, launch_stack(sa, stack_ni, maxdil, st, prec));
            e->stack_launches += 1;
            e->unverified = true;
            if (timed) {
                HIPCHK(e, hipEventRecord(e->prof_events[e->prof_used++].second, st));
                const double C = e->C, fr = (double)nb * T;
                // executed work only: the last layer's 1x1 computes its skip half alone (the residual half is never read)
                for (int p = p0; p < p1; ++p)
                    e->prof_flops += fr * 2.0 * C * 2.0 * C * ((p & 1) ? (p == 2 * L - 1 ? 0.5 : 1.0) : (double)e->K);
                e->prof_name = "stack_kernel<" + std::to_string(stack_ni) + "> (fused residual stack: dilated conv k=" +
                               std::to_string(e->K) + " + conditioner + gate and 1x1 + residual/skip, phases " +
                               std::to_string(p0) + ".." + std::to_string(p1 - 1) + " of " + std::to_string(2 * L) +
                               (stack_chunks > 1 ? ", " + std::to_string(stack_chunks) + " sample chunks" : "") +
                               (prec ? ((stack_ni != 2 || e->opt_blocked >= 2) ? ", split-bf16, blocked accumulation" : ", split-bf16, one chain per output")
                                     : ((stack_ni != 2 || e->opt_blocked >= 2) ? ", blocked accumulation" : ", one fp32 chain per output")) + ")";
            }
            b0 += nb;
        }
        return DR_OK;
    };
    if (stack_from == 0) {
        int rc = launch_stack_range(0, 2 * L);
        if (rc) return rc;
    }
    for (int l = 0; l < L && stack_from != 0; ++l) {
        const LayerW& w = e->layers[l];
        // (layer 0's shared conv of a guided step was already done by the previous step's tail kernel)
        if (!(l == 0 && stack_from == 1 && use_tail && tail->skip_inproj)) {   // dilated conv of (h + d_l) + conditioner, gate (model/diffwave.py:138-147)
            GemmArgs a = p4_gemm(prec ? w.conv_w3 : w.conv_w, w.conv_b, Cp / 64, e->hd, P, NB, T);
            if (prec) s3_in(a, e->hd3);
            a.bias2 = zero_spec ? w.conv_b_z : w.conv_b_u;   // samples >= n_cond: spec == 0 or spec == -1
            if (e->cond_tr && !zero_spec) {                  // ... or the learned unconditional spectrogram
                a.cond2 = e->cond_tr + (size_t)l * 2 * Cp * T;
                a.bias2 = w.conv_b;
            }
            a.taps = e->K; a.dil = w.dil;
            a.fold128 = e->opt_blocked >= 2;
            a.cond = e->cond ? e->cond + (size_t)l * e->fe_B * 2 * Cp * T : e->cond_dummy;
            a.c_bs = (long)2 * Cp * T;
            a.n_cond = n_cond;
            p4_out(a, e->g, P, T, Cp);
            if (prec) { a.Y = e->g3; a.y_bs = s3_bs; a.out_s3 = 1; }
            allow_splitk(e, a);
            // Classifier-free guidance evaluates the same x_t twice (samples b and b + bmod): in the first layer
            // both halves convolve the same h + d_0, so the contraction is done once per pair and the epilogue
            // writes both gated outputs (conditioner of b / constant unconditional bias).  Bit-identical.
            const bool dual = (l == 0 && bmod > 0 && NB == 2 * bmod && n_cond == bmod);
            if (dual) { a.NB = bmod; a.dual = bmod; }
            const Tile tile = dual ? pick_tile(Cp / 64, bmod, T, e->K, w.dil, prec, EPI_GATE, false, e->opt_blocked >= 2)
                                   : pick_tile(Cp / 64, NB, T, e->K, w.dil, prec, EPI_GATE, true, e->opt_blocked >= 2);
            if (e->stack_dbg_on && l + 1 == L) a.dbg = e->stack_dbg + 64;
            const bool timed = e->prof && !dual && stack_from < 0 && e->prof_used < e->prof_events.size();
            if (timed) HIPCHK(e, hipEventRecord(e->prof_events[e->prof_used].first, st));
            HIPCHK(e, launch_tiled(a, EPI_GATE, tile, st, prec));
            if (timed) {
                HIPCHK(e, hipEventRecord(e->prof_events[e->prof_used++].second, st));
                e->prof_flops += (double)NB * T * 2.0 * e->C * 2.0 * e->C * e->K;
                e->prof_name = "gemm_kernel<EPI_GATE> (dilated conv k=" + std::to_string(e->K) + " + conditioner + gate)";
            }
        }
        if (stack_from == 1) {             // layer 0's conv ran above; everything from its 1x1 on is one launch
            int rc = launch_stack_range(1, 2 * L);
            if (rc) return rc;
            break;
        }
        {   // 1x1 output projection, residual and skip (model/diffwave.py:149-151, :680)
            GemmArgs a = p4_gemm(prec ? w.out_w3 : w.out_w, w.out_b, Cp / 64, e->g, P, NB, T);
            if (prec) s3_in(a, e->g3);
            p4_out(a, e->h, P, T, Cp);
            if (l + 1 < L) {
                a.d2 = e->d_dtab + ((tsel ? 0 : (size_t)t * L) + l + 1) * Cp;
                a.tsel = tsel; a.d2_ts = (long)L * Cp;
                if (prec) { a.Y2 = e->hd3; a.y2_bs = s3_bs; a.out_s3 = 2; }
                else { a.Y2 = e->hd; a.y2_bs = act_bs; }
            }
            a.skip = e->skip; a.s_bs = (long)Cp * T; a.skip_init = (l == 0);
            allow_splitk(e, a);
            if (e->stack_dbg_on && l + 2 == L) a.dbg = e->stack_dbg + 96;     // same tick marks as the fused kernel's
            Tile tile = pick_pointwise_tile(Cp / 64, NB, T, prec);
            // the last layer's residual output is never read (model/diffwave.py:678-682 only uses the skip sum
            // after the loop): launch the skip half of the M tiles only
            if (l + 1 == L && tile.flavor >= 2) {
                // packed rows [0, Cp) are the residual half: the first 128-row tile holding a skip row is Cp / 128
                // (when Cp is not a multiple of 128 that tile also recomputes a few residual rows: harmless)
                const int first = Cp / 128, count = Cp / 64 - first;
                const Tile half = pick_pointwise_tile(count, NB, T, prec, Cp / 32);
                if (half.flavor == tile.flavor) { tile = half; a.MT = count; a.mt0 = first; }
            }
            HIPCHK(e, launch_tiled(a, EPI_RES_SKIP, tile, st, prec));
        }
    }
    if (use_tail) {
        // the rest of the step in one launch: skip projection, output projection, combine + update, next input projection
        TailArgs ta{};
        ta.NB = NB; ta.T = T; ta.Cp = Cp; ta.BN = stack_tile_frames(stack_ni);
        ta.dual = (bmod > 0 && NB == 2 * bmod) ? bmod : 0;
        ta.u_B = tail->u_B;
        ta.xcd_n = e->opt_stack_xcd; ta.fault = e->opt_stack_fault;
        ta.alpha = (float)(1.0 / std::sqrt((double)L));
        ta.skip = e->skip; ta.tmp = e->tmp; ta.x0 = x0_out;
        ta.skip_w = e->skip_w; ta.skip_b = e->skip_b; ta.outp_w = e->outp_w; ta.outp_b = e->outp_b; ta.zero = zero_vec();
        ta.u = tail->u; ta.x_out = tail->x_out;
        if (tail->next_t >= 0) {
            ta.in_w = e->in_w; ta.in_b = e->in_b; ta.d2_next = e->d_dtab + (size_t)tail->next_t * L * Cp;
            ta.h = e->h; ta.hd = e->hd;
            if (ta.dual > 0 && n_cond == bmod) {        // the next step's shared first-layer conv (as the dual launch above)
                const LayerW& w0 = e->layers[0];
                ta.conv_w = w0.conv_w; ta.conv_b = w0.conv_b;
                ta.conv_b2 = zero_spec ? w0.conv_b_z : w0.conv_b_u;
                if (e->cond_tr && !zero_spec) { ta.cond2 = e->cond_tr; ta.conv_b2 = w0.conv_b; }
                ta.cond = e->cond ? e->cond : e->cond_dummy;
                ta.c_bs = (long)2 * Cp * T;
                ta.taps = e->K; ta.dil = w0.dil;
                ta.fold = (stack_ni != 2 || e->opt_blocked >= 2);
                ta.g = e->g;
            }
        }
        ta.bar = e->tail_bar; ta.pbar = e->tail_pbar; ta.err = e->stack_err; ta.derr = e->stack_derr;
        // ticks 112..119 of dr_stack_status: the last tail launch of a chain that has a next step (all its parts run)
        ta.dbg = (e->stack_dbg_on && tail->next_t >= 0) ? e->stack_dbg + 112 : nullptr;
        HIPCHK(e, launch_tail(ta, st));
        e->tail_launches += 1;
        e->unverified = true;
        tail->done = true;
        tail->inproj_done = tail->next_t >= 0;
        return DR_OK;
    }
    {   // skip / sqrt(L) -> skip_projection -> relu (model/diffwave.py:682-684)
        GemmArgs a = p4_gemm(e->skip_w, e->skip_b, (Cp + 127) / 128, e->skip, P, NB, T);
        a.alpha = (float)(1.0 / std::sqrt((double)L));
        p4_out(a, e->tmp, P, T, Cp);
        allow_splitk(e, a);
        HIPCHK(e, launch_gemm(a, EPI_RELU, 1, st));   // M = C only: 64-frame tiles to fill more CUs
    }
    {   // output projection, written straight into the (B,T,88) roll layout (:685-686)
        GemmArgs a = p4_gemm(e->outp_w, e->outp_b, 1, e->tmp, P, NB, T);
        a.Y = x0_out; a.y_bs = (long)T * 88; a.y_ps = 4; a.y_fs = 88; a.y_rows = 88;
        allow_splitk(e, a);
        HIPCHK(e, launch_gemm(a, EPI_PLAIN, 1, st));  // M = 88 (one row tile): 64-frame tiles
    }
    return DR_OK;
}

int sampler_shape(int sampler, int B, int& NB, int& n_cond, int& family, bool& zero_spec) {
    zero_spec = false;
    switch (sampler) {
        case DR_SAMPLER_DDPM_X0: NB = B; n_cond = B; family = DR_COEF_DDPM_X0; return DR_OK;
        case DR_SAMPLER_CFDG_DDPM_X0:
        case DR_SAMPLER_INPAINTING_DDPM_X0: NB = 2 * B; n_cond = B; family = DR_COEF_DDPM_X0; return DR_OK;
        case DR_SAMPLER_GENERATION_DDPM_X0: NB = B; n_cond = 0; family = DR_COEF_DDPM_X0; return DR_OK;
        case DR_SAMPLER_DDIM_X0: NB = B; n_cond = B; family = DR_COEF_DDIM_X0; return DR_OK;
        case DR_SAMPLER_CFDG_DDIM_X0: NB = 2 * B; n_cond = B; family = DR_COEF_DDIM_X0; zero_spec = true; return DR_OK;
        case DR_SAMPLER_DDPM_EPS: NB = B; n_cond = B; family = DR_COEF_DDPM_EPS; return DR_OK;
        case DR_SAMPLER_DDIM_EPS: NB = B; n_cond = B; family = DR_COEF_DDIM_EPS; return DR_OK;
        case DR_SAMPLER_DDIM2DDPM_EPS: NB = B; n_cond = B; family = DR_COEF_DDIM2DDPM_EPS; return DR_OK;
    }
    return DR_EINVAL;
}
int sampler_shape(int sampler, int B, int& NB, int& n_cond) {
    int fam; bool z;
    return sampler_shape(sampler, B, NB, n_cond, fam, z);
}

// One reverse step.  The result is written in place on x - or, when the fused step ran (tail kernel), into e->xalt:
// *result tells which; chain (optional) carries "h / hd of this step are already there" from step to step.
struct ChainState { bool inproj_ready = false; int next_t = -1; };
int run_step(dr_engine* e, int sampler, float* x, const float* noise, int B, int T, int t, float w, uint64_t seed,
             int first_sample, hipStream_t st, float** result = nullptr, ChainState* chain = nullptr) {
    int NB, n_cond, family;
    bool zero_spec;
    if (sampler_shape(sampler, B, NB, n_cond, family, zero_spec)) return fail(e, DR_EINVAL, "unknown sampler %d", sampler);
    // Guidance weight 0: x0 = (1 + 0) c - 0 u = c (task/diffusion.py:953) - the unconditional evaluation is
    // multiplied by zero, so it is not run (half the work; the w = 0 points of the paper's guidance sweeps).
    if (w == 0.f && NB == 2 * B) { NB = B; n_cond = B; }
    UpdateArgs u{};
    u.x = x; u.x0c = e->x0buf; u.x0u = (NB == 2 * B) ? e->x0buf + (size_t)B * T * 88 : nullptr;
    u.noise = noise; u.coef = e->d_coef + ((size_t)family * e->S + t) * 5; u.t = t; u.mode = family;
    u.n = (long)B * T * 88; u.per_sample = (long)T * 88;
    u.w = w; u.onepw = (float)(1.0 + (double)w);
    u.seed = seed; u.first_sample = first_sample;
    u.dyn = e->use_dyn ? e->d_dyn : nullptr;
    TailPlan plan;
    plan.u = u; plan.x_out = e->xalt; plan.u_B = B;
    plan.next_t = chain ? chain->next_t : -1;
    plan.skip_inproj = chain && chain->inproj_ready;
    // (x and the tail kernel's output buffer must differ: a caller that hands us xalt itself gets the unfused tail)
    TailPlan* offer = (result && x != e->xalt) ? &plan : nullptr;
    if (chain) chain->inproj_ready = false;
    int rc = run_network(e, x, B, NB, n_cond, T, t, e->x0buf, st, zero_spec, nullptr, offer);
    if (rc) return rc;
    if (offer && plan.done) {
        *result = e->xalt;
        if (chain) chain->inproj_ready = plan.inproj_done;
        return DR_OK;
    }
    if (result) *result = x;
    HIPCHK(e, launch_update(u, st));
    return DR_OK;
}

// After a barrier time-out (device idle): re-arm the group counters, forget the published XCC tags, lower both flags.
int clear_stack_timeout(dr_engine* e) {
    HIPCHK(e, hipMemset(e->stack_bar, 0, (size_t)(12 * dr_engine::STACK_GROUPS) * sizeof(unsigned)));    // all three counter arrays
    HIPCHK(e, hipMemset(e->stack_xid, 0xFF, 1024 * sizeof(unsigned)));
    HIPCHK(e, hipMemset(e->stack_derr, 0, 16 * sizeof(unsigned)));
    *e->stack_err_host = 0;
    return DR_OK;
}

void drop_graph(dr_engine* e) {
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    e->gkey = GraphKey{};
}

// The split-bf16 ("S3") packings of the two hot GEMMs (same row maps and zero padding as the fp32 ones): only the opt-in
// precision reads them, so they are built on first use - at dr_set_precision(BF16X3) after a commit, or at the end of a
// commit made in that mode - instead of costing every start-up 0.4 s of packing and 520 MB of uploads.
int ensure_s3(dr_engine* e) {
    if (e->s3_ready || !e->committed) return DR_OK;
    const int C = e->C, Cp = e->Cp, L = e->L, K = e->K;
    std::vector<std::vector<uint16_t>> c3(L), o3(L);
    parallel_for(L, [&](int l) {
        const std::string pre = "residual_layers." + std::to_string(l) + ".";
        const auto& Wd = *find_param(e, pre + "dilated_conv.weight");
        const auto& Wo = *find_param(e, pre + "output_projection.weight");
        const int MTc = Cp / 64;
        c3[l] = pack_weights_s3(MTc, Cp / 32, K, [&](int pr, int ch, int j) {
            int mi, c; paired_row(pr, mi, c);
            return (c < C && ch < C) ? Wd[((size_t)(mi * C + c) * C + ch) * K + j] : 0.f;
        });
        o3[l] = pack_weights_s3(MTc, Cp / 32, 1, [&](int pr, int ch, int) {
            const int half = pr >= Cp, c = pr - half * Cp;
            return (c < C && ch < C) ? Wo[(size_t)(half * C + c) * C + ch] : 0.f;
        });
    });
    for (int l = 0; l < L; ++l) {
        int rc;
        if ((rc = upload_bytes(e, c3[l].data(), c3[l].size() * 2, &e->layers[l].conv_w3)) ||
            (rc = upload_bytes(e, o3[l].data(), o3[l].size() * 2, &e->layers[l].out_w3)))
            return rc;
        c3[l] = {}; o3[l] = {};
    }
    HIPCHK(e, stager().drain());
    e->s3_ready = true;
    return DR_OK;
}

int check_ready(dr_engine* e, int sampler, int B, int T) {
    if (!e->committed) return fail(e, DR_ESTATE, "dr_commit has not been called");
    if (e->stack_err_host && *e->stack_err_host)
        return fail(e, DR_ETIMEOUT, "a group barrier of an earlier fused residual-stack launch timed out (the results since the "
                                    "last dr_finish are invalid): is another stream / engine computing on this device at the "
                                    "same time? call dr_finish (or dr_stack_status) to clear the condition and recompute - "
                                    "dr_finish also switches this engine to per-phase launches; dr_sample_checked does all of that");
    if (B <= 0 || T <= 0) return fail(e, DR_EINVAL, "bad shape B=%d T=%d", B, T);
    if (sampler != DR_SAMPLER_GENERATION_DDPM_X0 && (e->fe_B != B || e->fe_T != T))
        return fail(e, DR_ESTATE, "dr_frontend(B=%d,T=%d) must precede a conditional evaluation with B=%d,T=%d",
                    e->fe_B, e->fe_T, B, T);
    return DR_OK;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int dr_abi_version(void) { return DR_ABI_VERSION; }

const char* dr_last_error(const dr_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int dr_create(dr_engine** out, const dr_config* cfg) {
    if (!out || !cfg) return fail(nullptr, DR_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != DR_ABI_VERSION)
        return fail(nullptr, DR_EINVAL, "ABI version mismatch: header %d, library %d", cfg->abi_version, DR_ABI_VERSION);
    if (cfg->residual_channels <= 0 || cfg->residual_channels % 4 || cfg->residual_layers <= 0 ||
        cfg->kernel_size <= 0 || cfg->kernel_size % 2 == 0 || cfg->n_mels <= 0 || cfg->timesteps <= 0 ||
        cfg->n_fft <= 0 || cfg->n_fft % 32 || cfg->hop_length <= 0 || cfg->hop_length % 4 ||
        cfg->dilation_base <= 0 || cfg->dilation_bound <= 0)
        return fail(nullptr, DR_EINVAL, "unsupported configuration");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DR_EHIP, "no HIP device available: the engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, DR_EINVAL, "device %d out of range", cfg->device);
    DeviceGuard guard(cfg->device);
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) return fail(nullptr, DR_EHIP, "hipSetDevice failed");
    }
    {
        hipError_t ie = init_kernels();
        if (ie != hipSuccess) return fail(nullptr, DR_EHIP, "kernel init failed: %s", hipGetErrorString(ie));
    }
    if (cfg->device >= MAX_DEVICES) return fail(nullptr, DR_EINVAL, "device %d out of range", cfg->device);
    if (!g_zero_vecs[cfg->device]) {
        void* z = nullptr;
        const size_t zn = 1 << 16;   // floats; covers every Cin on the path (n_fft, bins, channels)
        if (hipMalloc(&z, zn * sizeof(float)) != hipSuccess || hipMemset(z, 0, zn * sizeof(float)) != hipSuccess)
            return fail(nullptr, DR_EHIP, "allocating the zero vector failed");
        g_zero_vecs[cfg->device] = (const float*)z;
    }
    if (cfg->n_fft > (1 << 16) || cfg->residual_channels > (1 << 15))
        return fail(nullptr, DR_EINVAL, "configuration too large");
    dr_engine* e = new dr_engine();
    e->cfg = *cfg;
    e->C = cfg->residual_channels;
    e->Cp = round_up(e->C, 64);
    e->L = cfg->residual_layers;
    e->K = cfg->kernel_size;
    e->S = cfg->timesteps;
    e->NM = cfg->n_mels;
    if (const char* v = getenv("DR_STACK")) e->opt_stack = atoi(v);           // tuning / A-B experiments
    if (const char* v = getenv("DR_STACK_XCD")) e->opt_stack_xcd = atoi(v);
    if (const char* v = getenv("DR_TAIL")) e->opt_tail = atoi(v);
    if (const char* v = getenv("DR_BLOCKED")) e->opt_blocked = atoi(v);
    e->n_bins = cfg->n_fft / 2 + 1;
    e->bins_p = round_up(e->n_bins, 64);
    int maxdil = 1;
    for (int i = 0; i < e->L; ++i) {
        int d = 1;
        for (int q = 0; q < i % cfg->dilation_bound; ++q) d *= cfg->dilation_base;
        maxdil = std::max(maxdil, d);
    }
    if (gemm_lds_bytes(1, 1, e->K, maxdil, 1, EPI_GATE) > 160 * 1024) {
        const int rf = (e->K - 1) * maxdil;
        delete e;
        return fail(nullptr, DR_EINVAL, "receptive halo (k-1)*dil = %d does not fit the 160 KiB LDS tile", rf);
    }
    *out = e;
    return DR_OK;
}

void dr_destroy(dr_engine* e) {
    if (!e) return;
    DeviceGuard guard(e->cfg.device);
    (void)hipDeviceSynchronize();
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->dbg_ticks) (void)hipFree(e->dbg_ticks);
    if (e->d_counts) (void)hipFree(e->d_counts);
    if (e->d_dyn) (void)hipFree(e->d_dyn);
    if (e->stack_bar) (void)hipFree(e->stack_bar);
    if (e->xsave) (void)hipFree(e->xsave);
    if (e->stack_err_host) (void)hipHostFree((void*)e->stack_err_host);
    if (e->stack_dbg) (void)hipFree(e->stack_dbg);
    if (e->sk_cnt) (void)hipFree(e->sk_cnt);
    if (e->d_tsel) (void)hipFree(e->d_tsel);
    for (auto& p : e->prof_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (void* p : e->owned) (void)hipFree(p);
    float* bufs[] = {e->d_coef, e->d_dtab, e->h, e->hd, e->hd3, e->g3, e->g, e->skip, e->tmp, e->x0buf, e->cond, e->cond_dummy,
                     e->wav_pad, e->power, e->logmel, e->specP4, e->mm, e->sk_ws, e->xwork, e->cond_tr, e->xalt};
    for (float* p : bufs) if (p) (void)hipFree(p);
    delete e;
}

int dr_set_param(dr_engine* e, const char* name, const float* host_data, size_t numel) {
    if (!e || !name || !host_data) return fail(e, DR_EINVAL, "null argument");
    const size_t want = expected_numel(e, name);
    if (want == 0) return fail(e, DR_ENAME, "unknown parameter '%s'", name);
    if (want != numel) return fail(e, DR_ENAME, "parameter '%s': expected %zu elements, got %zu", name, want, numel);
    e->params[name].assign(host_data, host_data + numel);
    e->committed = false;
    return DR_OK;
}

int dr_set_tables(dr_engine* e, const float* host_embedding, const float* host_coef) {
    if (!e || !host_embedding || !host_coef) return fail(e, DR_EINVAL, "null argument");
    e->h_emb.assign(host_embedding, host_embedding + (size_t)e->S * 128);
    e->h_coef.assign(host_coef, host_coef + (size_t)DR_COEF_FAMILIES * e->S * 5);
    e->committed = false;
    return DR_OK;
}

int dr_set_frontend_tables(dr_engine* e, const float* host_window, float window_norm, const float* host_fb) {
    if (!e) return DR_EINVAL;
    e->h_win.clear(); e->h_fb.clear(); e->h_win_norm = 0.f;
    if (host_window) {
        if (!(window_norm > 0.f)) return fail(e, DR_EINVAL, "window_norm must be positive");
        e->h_win.assign(host_window, host_window + e->cfg.n_fft);
        e->h_win_norm = window_norm;
    }
    if (host_fb) e->h_fb.assign(host_fb, host_fb + (size_t)e->n_bins * e->NM);
    e->committed = false;
    return DR_OK;
}

int dr_commit(dr_engine* e, void* stream) {
    if (!e) return DR_EINVAL;
    Range range("dr_commit: pack + upload weights, hoisted tables");
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(e->cfg.device);
    if (e->h_emb.empty() || e->h_coef.empty()) return fail(e, DR_ESTATE, "dr_set_tables has not been called");
    const int C = e->C, Cp = e->Cp, L = e->L, K = e->K, NM = e->NM, S = e->S;
    // all parameters present?
    {
        std::vector<std::string> names = {"input_projection.weight", "input_projection.bias",
                                          "diffusion_embedding.projection1.weight", "diffusion_embedding.projection1.bias",
                                          "diffusion_embedding.projection2.weight", "diffusion_embedding.projection2.bias",
                                          "skip_projection.weight", "skip_projection.bias",
                                          "output_projection.weight", "output_projection.bias"};
        for (int l = 0; l < L; ++l)
            for (const char* r : {"dilated_conv.weight", "dilated_conv.bias", "diffusion_projection.weight",
                                  "diffusion_projection.bias", "conditioner_projection.weight",
                                  "conditioner_projection.bias", "output_projection.weight", "output_projection.bias"})
                names.push_back("residual_layers." + std::to_string(l) + "." + r);
        for (auto& n : names)
            if (!find_param(e, n)) return fail(e, DR_ESTATE, "parameter '%s' was never set", n.c_str());
    }
    for (void* p : e->owned) (void)hipFree(p);
    e->owned.clear();
    e->layers.assign(L, LayerW{});
    e->cond_tr_T = 0;       // rebuilt from the new conditioner weights / trainable_parameters at the next use
    e->ws_T = 0;            // (ensure_workspace is where that happens)
    int rc;
    auto P = [&](const std::string& n) -> const std::vector<float>& { return *find_param(e, n); };

    // ---- network weights --------------------------------------------------------------------
    {   // input projection (C,88,1): natural rows
        const auto& W = P("input_projection.weight");
        const auto& Bv = P("input_projection.bias");
        const int MT = (Cp + 127) / 128;
        auto pk = pack_weights(MT, 3, 1, [&](int r, int ch, int) { return (r < C && ch < 88) ? W[(size_t)r * 88 + ch] : 0.f; });
        std::vector<float> bb(MT * 128, 0.f);
        for (int r = 0; r < C; ++r) bb[r] = Bv[r];
        if ((rc = upload(e, pk, &e->in_w)) || (rc = upload(e, bb, &e->in_b))) return rc;
    }
    struct LayerPack { std::vector<float> pconv, pcond, bconv, bconv_u, bconv_z, bcond, pout, bout; };
    std::vector<LayerPack> packs(L);
    const double tp0 = now_s();
    parallel_for(L, [&](int l) {
        LayerPack& k = packs[l];
        const std::string pre = "residual_layers." + std::to_string(l) + ".";
        const auto& Wd = P(pre + "dilated_conv.weight");
        const auto& Bd = P(pre + "dilated_conv.bias");
        const auto& Wc = P(pre + "conditioner_projection.weight");
        const auto& Bc = P(pre + "conditioner_projection.bias");
        const auto& Wo = P(pre + "output_projection.weight");
        const auto& Bo = P(pre + "output_projection.bias");
        const int MTc = Cp / 64;   // 2*Cp rows
        k.pconv = pack_weights(MTc, Cp / 32, K, [&](int pr, int ch, int j) {
            int mi, c; paired_row(pr, mi, c);
            return (c < C && ch < C) ? Wd[((size_t)(mi * C + c) * C + ch) * K + j] : 0.f;
        });
        k.pcond = pack_weights(MTc, (NM + 31) / 32, 1, [&](int pr, int ch, int) {
            int mi, c; paired_row(pr, mi, c);
            return (c < C && ch < NM) ? Wc[(size_t)(mi * C + c) * NM + ch] : 0.f;
        });
        k.bconv.assign(MTc * 128, 0.f); k.bconv_u.assign(MTc * 128, 0.f); k.bconv_z.assign(MTc * 128, 0.f); k.bcond.assign(MTc * 128, 0.f);
        for (int pr = 0; pr < MTc * 128; ++pr) {
            int mi, c; paired_row(pr, mi, c);
            if (c >= C) continue;
            const int o = mi * C + c;
            double sw = 0.0;
            for (int m = 0; m < NM; ++m) sw += (double)Wc[(size_t)o * NM + m];
            const float cu = (float)((double)Bc[o] - sw);   // conditioner of spec == -1 (model/diffwave.py:660)
            k.bconv[pr] = Bd[o];
            k.bconv_u[pr] = Bd[o] + cu;
            k.bconv_z[pr] = Bd[o] + Bc[o];                  // conditioner of spec == 0 is its bias
            k.bcond[pr] = Bc[o];
        }
        // 1x1 output projection (2C,C,1): packed rows [0,Cp) residual, [Cp,2Cp) skip
        k.pout = pack_weights(MTc, Cp / 32, 1, [&](int pr, int ch, int) {
            const int half = pr >= Cp, c = pr - half * Cp;
            return (c < C && ch < C) ? Wo[(size_t)(half * C + c) * C + ch] : 0.f;
        });
        k.bout.assign(MTc * 128, 0.f);
        for (int pr = 0; pr < 2 * Cp; ++pr) {
            const int half = pr >= Cp, c = pr - half * Cp;
            if (c < C) k.bout[pr] = Bo[half * C + c];
        }
    });
    e->t_pack_s = now_s() - tp0;
    const double tu0 = now_s();
    for (int l = 0; l < L; ++l) {
        LayerW& lw = e->layers[l];
        lw.dil = 1;
        for (int q = 0; q < l % e->cfg.dilation_bound; ++q) lw.dil *= e->cfg.dilation_base;
        LayerPack& k = packs[l];
        if ((rc = upload(e, k.pconv, &lw.conv_w)) || (rc = upload(e, k.bconv, &lw.conv_b)) ||
            (rc = upload(e, k.bconv_u, &lw.conv_b_u)) || (rc = upload(e, k.bconv_z, &lw.conv_b_z)) ||
            (rc = upload(e, k.pcond, &lw.cond_w)) ||
            (rc = upload(e, k.bcond, &lw.cond_b)) || (rc = upload(e, k.pout, &lw.out_w)) ||
            (rc = upload(e, k.bout, &lw.out_b)))
            return rc;
        k = LayerPack{};
    }
    HIPCHK(e, stager().drain());
    e->t_upload_s = now_s() - tu0;
    e->s3_ready = false;      // the split-bf16 packings (opt-in precision) are built when that mode is first used
    {   // skip projection (C,C,1) and output projection (88,C,1): natural rows
        const auto& Ws = P("skip_projection.weight");
        const auto& Bs = P("skip_projection.bias");
        const int MT = (Cp + 127) / 128;
        auto pk = pack_weights(MT, Cp / 32, 1, [&](int r, int ch, int) { return (r < C && ch < C) ? Ws[(size_t)r * C + ch] : 0.f; });
        std::vector<float> bb(MT * 128, 0.f);
        for (int r = 0; r < C; ++r) bb[r] = Bs[r];
        if ((rc = upload(e, pk, &e->skip_w)) || (rc = upload(e, bb, &e->skip_b))) return rc;
        const auto& Wo = P("output_projection.weight");
        const auto& Bo = P("output_projection.bias");
        auto pk2 = pack_weights(1, Cp / 32, 1, [&](int r, int ch, int) { return (r < 88 && ch < C) ? Wo[(size_t)r * C + ch] : 0.f; });
        std::vector<float> b2(128, 0.f);
        for (int r = 0; r < 88; ++r) b2[r] = Bo[r];
        if ((rc = upload(e, pk2, &e->outp_w)) || (rc = upload(e, b2, &e->outp_b))) return rc;
    }
    // ---- front-end constants: windowed DFT matrix and HTK mel filterbank -----------------------
    {
        const int N = e->cfg.n_fft, nb = e->n_bins, bp = e->bins_p;
        std::vector<double> win(N);
        double s2 = 0.0;
        for (int k = 0; k < N; ++k) {
            win[k] = e->h_win.empty() ? 0.5 - 0.5 * std::cos(2.0 * M_PI * k / N) : (double)e->h_win[k];
            s2 += win[k] * win[k];
        }
        // normalized=True: / sqrt(sum(window^2)) - the caller's fp32 value when the window is the caller's
        const double wnorm = e->h_win.empty() ? std::sqrt(s2) : (double)e->h_win_norm;
        const double norm = 1.0 / wnorm;
        // cos/sin via an exact-phase table (k*bin mod N) to keep the twiddles accurate
        std::vector<double> ct(N), sn(N);
        for (int k = 0; k < N; ++k) { ct[k] = std::cos(2.0 * M_PI * k / N); sn[k] = std::sin(2.0 * M_PI * k / N); }
        e->use_fft = (N >= 8 && N <= 16384 && (N & (N - 1)) == 0);
        e->dft_w = e->fft_win = e->fft_tw = nullptr;
        if (e->use_fft) {
            // FFT front-end (the released configuration, n_fft = 2048): window and roots of unity exp(-2 pi i k / N),
            // rounded once from double; the window normalisation is applied to the spectrum as the reference does
            std::vector<float> wf(N), tf(2 * (size_t)N);
            for (int k = 0; k < N; ++k) { wf[k] = (float)win[k]; tf[2 * k] = (float)ct[k]; tf[2 * k + 1] = (float)(-sn[k]); }
            e->fft_norm = (float)wnorm;
            if ((rc = upload(e, wf, &e->fft_win)) || (rc = upload(e, tf, &e->fft_tw))) return rc;
        } else {
            // any other n_fft: the windowed DFT as a GEMM (cos rows / sin rows paired, |.|^2 in the epilogue)
            auto pk = pack_weights(bp / 64, N / 32, 1, [&](int pr, int k, int) {
                int mi, bin; paired_row(pr, mi, bin);
                if (bin >= nb) return 0.f;
                const int ph = (int)(((long long)k * bin) % N);
                return (float)(win[k] * norm * (mi == 0 ? ct[ph] : sn[ph]));
            });
            if ((rc = upload(e, pk, &e->dft_w))) return rc;
        }
        // torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, 'htk')
        const double fmin = e->cfg.f_min, fmax = e->cfg.f_max;
        auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
        const double m_min = hz2mel(fmin), m_max = hz2mel(fmax);
        std::vector<double> fpts(NM + 2);
        for (int i = 0; i < NM + 2; ++i) {
            const double m = m_min + (m_max - m_min) * i / (NM + 1);
            fpts[i] = 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
        }
        const double nyq = (double)(e->cfg.sample_rate / 2);
        auto fbv = [&](int bin, int mel) {
            if (!e->h_fb.empty()) return (double)e->h_fb[(size_t)bin * NM + mel];      // the caller's (reference-rounded) table
            const double f = nyq * bin / (nb - 1);
            const double down = (f - fpts[mel]) / (fpts[mel + 1] - fpts[mel]);
            const double up = (fpts[mel + 2] - f) / (fpts[mel + 2] - fpts[mel + 1]);
            return std::max(0.0, std::min(down, up));
        };
        const int MTm = (NM + 127) / 128;
        auto pm = pack_weights(MTm, bp / 32, 1, [&](int r, int ch, int) {
            return (r < NM && ch < nb) ? (float)fbv(ch, r) : 0.f;
        });
        if ((rc = upload(e, pm, &e->mel_w))) return rc;
    }
    HIPCHK(e, stager().drain());       // every staged constant has landed
    // ---- tables ------------------------------------------------------------------------------
    if ((rc = dev_alloc(e, &e->d_coef, (size_t)DR_COEF_FAMILIES * S * 5))) return rc;
    HIPCHK(e, hipMemcpy(e->d_coef, e->h_coef.data(), (size_t)DR_COEF_FAMILIES * S * 5 * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = dev_alloc(e, &e->d_dtab, (size_t)S * L * Cp))) return rc;
    if ((rc = dev_alloc(e, &e->sk_ws, dr_engine::SK_WS_FLOATS, false))) return rc;
    if (!e->sk_cnt) {       // ticket counters: zero between launches (the kernels re-arm them)
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, dr_engine::SK_CNT_N * sizeof(unsigned)));
        HIPCHK(e, hipMemset(q, 0, dr_engine::SK_CNT_N * sizeof(unsigned)));
        e->sk_cnt = (unsigned*)q;
    }
    if (!e->d_dyn) { void* q = nullptr; HIPCHK(e, hipMalloc(&q, sizeof(DynParams))); e->d_dyn = (DynParams*)q; }
    if (!e->stack_bar) {     // group counters of the fused residual stack: zero between launches (re-armed in-kernel)
        void* q = nullptr;
        const size_t G4 = (size_t)4 * dr_engine::STACK_GROUPS;
        const size_t nb = (3 * G4 + 1024 + 16) * sizeof(unsigned);
        HIPCHK(e, hipMalloc(&q, nb));
        HIPCHK(e, hipMemset(q, 0, nb));
        e->stack_bar = (unsigned*)q;                                     // [bar][tail bar][tail pair bar][xid][derr]
        e->tail_bar = e->stack_bar + G4;
        e->tail_pbar = e->stack_bar + 2 * G4;
        e->stack_xid = e->stack_bar + 3 * G4;                            // one word per block (<= 1024 CUs)
        e->stack_derr = e->stack_xid + 1024;
        HIPCHK(e, hipMemset(e->stack_xid, 0xFF, 1024 * sizeof(unsigned)));   // no tag of a launch ever equals 0xFFFFFFFF
        // the "a barrier wait gave up" flag lives in host-visible memory: every later API call sees it without a
        // synchronisation and fails loudly instead of returning rolls computed from a broken hand-off
        void* hf = nullptr;
        HIPCHK(e, hipHostMalloc(&hf, 64, hipHostMallocMapped));
        memset(hf, 0, 64);
        e->stack_err_host = (volatile unsigned*)hf;
        void* df = nullptr;
        HIPCHK(e, hipHostGetDevicePointer(&df, hf, 0));
        e->stack_err = (unsigned*)df;
        void* d = nullptr;
        HIPCHK(e, hipMalloc(&d, 128 * sizeof(long long)));
        HIPCHK(e, hipMemset(d, 0, 128 * sizeof(long long)));
        e->stack_dbg = (long long*)d;
        hipDeviceProp_t prop;
        HIPCHK(e, hipGetDeviceProperties(&prop, e->cfg.device));
        e->n_cus = prop.multiProcessorCount;
    }
    {   // hoisted step embedding: table -> Linear+silu -> Linear+silu -> per-layer Linear, with
        // "frames" = diffusion steps (model/diffwave.py:65-74, :126,:138).  Built on the device by
        // the same GEMM kernel; result d_dtab[t][l][c].
        std::vector<float> embP4((size_t)32 * S * 4);
        for (int t = 0; t < S; ++t)
            for (int c = 0; c < 128; ++c) embP4[((size_t)(c / 4) * S + t) * 4 + (c % 4)] = e->h_emb[(size_t)t * 128 + c];
        float *d_emb = nullptr, *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
        float *a1 = nullptr, *a2 = nullptr;
        if ((rc = upload(e, embP4, &d_emb))) return rc;
        const auto& W1 = P("diffusion_embedding.projection1.weight");
        const auto& W2 = P("diffusion_embedding.projection2.weight");
        auto p1 = pack_weights(4, 4, 1, [&](int r, int ch, int) { return W1[(size_t)r * 128 + ch]; });
        auto p2 = pack_weights(4, 16, 1, [&](int r, int ch, int) { return W2[(size_t)r * 512 + ch]; });
        if ((rc = upload(e, p1, &w1)) || (rc = upload(e, P("diffusion_embedding.projection1.bias"), &b1)) ||
            (rc = upload(e, p2, &w2)) || (rc = upload(e, P("diffusion_embedding.projection2.bias"), &b2)))
            return rc;
        if ((rc = dev_alloc(e, &a1, (size_t)512 * S)) || (rc = dev_alloc(e, &a2, (size_t)512 * S))) return rc;
        HIPCHK(e, stager().drain());
        GemmArgs g1 = p4_gemm(w1, b1, 4, d_emb, 32, 1, S);
        p4_out(g1, a1, 128, S, 512);
        HIPCHK(e, launch_gemm(g1, EPI_SILU, 2, st));
        GemmArgs g2 = p4_gemm(w2, b2, 4, a1, 128, 1, S);
        p4_out(g2, a2, 128, S, 512);
        HIPCHK(e, launch_gemm(g2, EPI_SILU, 2, st));
        const int MT = (Cp + 127) / 128;
        for (int l = 0; l < L; ++l) {
            const std::string pre = "residual_layers." + std::to_string(l) + ".";
            const auto& Wd = P(pre + "diffusion_projection.weight");
            const auto& Bd = P(pre + "diffusion_projection.bias");
            auto pd = pack_weights(MT, 16, 1, [&](int r, int ch, int) { return r < C ? Wd[(size_t)r * 512 + ch] : 0.f; });
            std::vector<float> bb(MT * 128, 0.f);
            for (int r = 0; r < C; ++r) bb[r] = Bd[r];
            float *wd = nullptr, *bd = nullptr;
            if ((rc = upload(e, pd, &wd)) || (rc = upload(e, bb, &bd))) return rc;
            HIPCHK(e, stager().drain());
            GemmArgs g3 = p4_gemm(wd, bd, MT, a2, 128, 1, S);
            g3.Y = e->d_dtab + (size_t)l * Cp; g3.y_bs = 0; g3.y_ps = 4; g3.y_fs = (long)L * Cp; g3.y_rows = Cp;
            HIPCHK(e, launch_gemm(g3, EPI_PLAIN, 2, st));
        }
        HIPCHK(e, hipStreamSynchronize(st));
        (void)hipFree(a1);
        (void)hipFree(a2);
    }
    e->committed = true;
    e->fe_B = e->fe_T = 0;
    e->t_tables_s = now_s() - tu0 - e->t_upload_s;
    // (DR_S3_EAGER=1: build the split-bf16 packings at every commit, as rounds 1-3 did - the "before" of the cold-start report)
    static const bool s3_eager = getenv("DR_S3_EAGER") && atoi(getenv("DR_S3_EAGER"));
    if (e->prec || s3_eager) return ensure_s3(e);
    return DR_OK;
}

int dr_frontend(dr_engine* e, const float* d_wav, int B, int L, int T_roll, int mask_t0, int mask_t1, int mask_f0,
                int mask_f1, float* d_spec_out, void* stream) {
    if (!e || !d_wav) return fail(e, DR_EINVAL, "null argument");
    if (!e->committed) return fail(e, DR_ESTATE, "dr_commit has not been called");
    Range range("dr_frontend: mel + conditioner projections");
    DeviceGuard guard(e->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const int N = e->cfg.n_fft, hop = e->cfg.hop_length, pad = N / 2;
    if (B <= 0 || L <= pad || T_roll <= 0) return fail(e, DR_EINVAL, "bad front-end shape B=%d L=%d T=%d", B, L, T_roll);
    const int TF = L / hop + 1;
    const int T = std::min(T_roll, TF);
    const int Lp = (L + 2 * pad + 3) & ~3;
    const int Cp = e->Cp, NM = e->NM, bp = e->bins_p;
    const int mel_planes = (NM + 3) / 4;
    int rc;
    {   // buffers below are reallocated only when a shape grows: synchronise just then (a previous call may still
        // be reading them), not on every call
        const size_t mm_need0 = e->norm_framewise ? (size_t)B * TF * 2 : (size_t)B * 2;
        const bool grow = (size_t)B * Lp > e->fe_cap_wav || (size_t)B * bp * TF > e->fe_cap_pow ||
                          (size_t)B * mel_planes * 4 * TF > e->fe_cap_log || (size_t)B * mel_planes * 4 * T > e->fe_cap_spec ||
                          mm_need0 > e->fe_cap_mm || (size_t)e->L * B * 2 * Cp * T > e->cond_cap;
        if (grow) HIPCHK(e, hipDeviceSynchronize());
    }
    if ((size_t)B * Lp > e->fe_cap_wav) { if ((rc = dev_alloc(e, &e->wav_pad, (size_t)B * Lp))) return rc; e->fe_cap_wav = (size_t)B * Lp; }
    if ((size_t)B * bp * TF > e->fe_cap_pow) { if ((rc = dev_alloc(e, &e->power, (size_t)B * bp * TF))) return rc; e->fe_cap_pow = (size_t)B * bp * TF; }
    if ((size_t)B * mel_planes * 4 * TF > e->fe_cap_log) { if ((rc = dev_alloc(e, &e->logmel, (size_t)B * mel_planes * 4 * TF))) return rc; e->fe_cap_log = (size_t)B * mel_planes * 4 * TF; }
    if ((size_t)B * mel_planes * 4 * T > e->fe_cap_spec) { if ((rc = dev_alloc(e, &e->specP4, (size_t)B * mel_planes * 4 * T))) return rc; e->fe_cap_spec = (size_t)B * mel_planes * 4 * T; }
    const size_t mm_need = e->norm_framewise ? (size_t)B * TF * 2 : (size_t)B * 2;
    if (mm_need > e->fe_cap_mm) { if ((rc = dev_alloc(e, &e->mm, mm_need))) return rc; e->fe_cap_mm = mm_need; }
    const size_t cond_need = (size_t)e->L * B * 2 * Cp * T;
    bool cond_moved = false;
    if (cond_need > e->cond_cap) { if ((rc = dev_alloc(e, &e->cond, cond_need))) return rc; e->cond_cap = cond_need; cond_moved = true; }
    if (cond_moved || B != e->fe_B || T != e->fe_T) {
        // a captured chain bakes the conditioner pointers / strides: drop it when they change
        if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
        if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
        e->gkey = GraphKey{};
    }

    // 1. center / reflect padding
    HIPCHK(e, launch_reflect_pad(d_wav, e->wav_pad, B, L, pad, st));
    // 2. STFT power spectrum: frames are read straight out of the padded waveform (frame stride hop) - no framed
    //    copy.  FFT per frame (n_fft a power of two), else the windowed DFT as a GEMM.
    if (e->use_fft) {
        HIPCHK(e, launch_stft_power(e->wav_pad, e->fft_win, e->fft_tw, e->power, B, Lp, TF, N, hop, bp, e->fft_norm, st));
    } else {
        GemmArgs a{};
        a.Wp = e->dft_w; a.MT = bp / 64; a.bias = zero_vec();
        a.X = e->wav_pad; a.x_bs = Lp; a.x_ps = 4; a.x_fs = hop; a.x_planes = N / 4; a.kchunks = N / 32;
        a.NB = B; a.T = TF; a.taps = 1; a.dil = 1; a.alpha = 1.f;
        a.d2 = zero_vec();
        p4_out(a, e->power, bp / 4, TF, bp);
        HIPCHK(e, launch_gemm(a, EPI_POWER, 2, st));
    }
    // 3. mel filterbank + log(. + 1e-6)   (model/diffwave.py:644)
    {
        GemmArgs a = p4_gemm(e->mel_w, nullptr, (NM + 127) / 128, e->power, bp / 4, B, TF);
        if (e->use_fft) {     // power is (B, TF, bins) row-major: 4 bins per plane at stride 4, frames at stride bins
            a.x_bs = (long)TF * bp; a.x_ps = 4; a.x_fs = bp;
        }
        p4_out(a, e->logmel, mel_planes, TF, mel_planes * 4);
        HIPCHK(e, launch_gemm(a, EPI_LOG, 2, st));
    }
    // 4. imagewise min-max over the untrimmed TF frames, mask, trim
    if (e->norm_framewise) HIPCHK(e, launch_minmax_frame(e->logmel, e->mm, B, mel_planes, TF, NM, st));
    else HIPCHK(e, launch_minmax(e->logmel, e->mm, B, mel_planes, TF, NM, st));
    HIPCHK(e, launch_normalize(e->logmel, e->mm, e->specP4, d_spec_out, B, mel_planes, mel_planes, TF, T, NM,
                               mask_t0, mask_t1, mask_f0, mask_f1, st, e->norm_framewise));
    // 5. hoisted conditioner projections, one (B, 2C, T) tensor per layer (model/diffwave.py:143)
    for (int l = 0; l < e->L; ++l) {
        const LayerW& w = e->layers[l];
        GemmArgs a = p4_gemm(w.cond_w, w.cond_b, Cp / 64, e->specP4, mel_planes, B, T);
        p4_out(a, e->cond + (size_t)l * B * 2 * Cp * T, 2 * Cp / 4, T, 2 * Cp);
        HIPCHK(e, launch_gemm(a, EPI_PLAIN, 2, st));
    }
    e->fe_B = B;
    e->fe_T = T;
    return DR_OK;
}

int dr_forward(dr_engine* e, const float* d_x, int B, int T, int t, int cond, float* d_x0_out, void* stream) {
    if (!e || !d_x || !d_x0_out) return fail(e, DR_EINVAL, "null argument");
    DeviceGuard guard(e->cfg.device);
    const int sampler = cond == DR_COND_UNCOND ? DR_SAMPLER_GENERATION_DDPM_X0 : DR_SAMPLER_DDPM_X0;
    int rc = check_ready(e, sampler, B, T);
    if (rc) return rc;
    if (t < 0 || t >= e->S) return fail(e, DR_EINVAL, "step %d out of range", t);
    if ((rc = ensure_workspace(e, B, T))) return rc;
    return run_network(e, d_x, 0, B, cond == DR_COND_UNCOND ? 0 : B, T, t, d_x0_out, (hipStream_t)stream);
}

int dr_forward_steps(dr_engine* e, const float* d_x, int B, int T, const int32_t* host_t, int cond, float* d_x0_out,
                     void* stream) {
    if (!e || !d_x || !d_x0_out || !host_t) return fail(e, DR_EINVAL, "null argument");
    DeviceGuard guard(e->cfg.device);
    const int sampler = cond == DR_COND_UNCOND ? DR_SAMPLER_GENERATION_DDPM_X0 : DR_SAMPLER_DDPM_X0;
    int rc = check_ready(e, sampler, B, T);
    if (rc) return rc;
    for (int b = 0; b < B; ++b)
        if (host_t[b] < 0 || host_t[b] >= e->S) return fail(e, DR_EINVAL, "step %d of sample %d out of range", host_t[b], b);
    if ((rc = ensure_workspace(e, B, T))) return rc;
    if ((size_t)B > e->tsel_cap) {
        if (e->d_tsel) (void)hipFree(e->d_tsel);
        e->d_tsel = nullptr;
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, (size_t)B * sizeof(int)));
        e->d_tsel = (int*)q;
        e->tsel_cap = (size_t)B;
    }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipStreamSynchronize(st));       // the previous call may still be reading the step buffer
    HIPCHK(e, hipMemcpy(e->d_tsel, host_t, (size_t)B * sizeof(int), hipMemcpyHostToDevice));
    return run_network(e, d_x, 0, B, cond == DR_COND_UNCOND ? 0 : B, T, 0, d_x0_out, st, false, e->d_tsel);
}

int dr_step(dr_engine* e, int sampler, float* d_x, const float* d_noise, int B, int T, int t, float w, uint64_t seed,
            int first_sample, void* stream) {
    if (!e || !d_x) return fail(e, DR_EINVAL, "null argument");
    DeviceGuard guard(e->cfg.device);
    int rc = check_ready(e, sampler, B, T);
    if (rc) return rc;
    if (t < 0 || t >= e->S) return fail(e, DR_EINVAL, "step %d out of range", t);
    int NB, n_cond;
    if (sampler_shape(sampler, B, NB, n_cond)) return fail(e, DR_EINVAL, "unknown sampler %d", sampler);
    if ((rc = ensure_workspace(e, NB, T))) return rc;
    float* res = nullptr;
    if ((rc = run_step(e, sampler, d_x, d_noise, B, T, t, w, seed, first_sample, (hipStream_t)stream, &res))) return rc;
    if (res != d_x)      // the fused step wrote x_{t-1} into the engine's buffer: hand it back in place
        HIPCHK(e, hipMemcpyAsync(d_x, res, (size_t)B * T * 88 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DR_OK;
}

int dr_sample(dr_engine* e, int sampler, float* d_x, const float* d_noise, int B, int T, float w, uint64_t seed,
              int first_sample, int use_graph, void* stream) {
    if (!e || !d_x) return fail(e, DR_EINVAL, "null argument");
    DeviceGuard guard(e->cfg.device);
    int rc = check_ready(e, sampler, B, T);
    if (rc) return rc;
    int NB, n_cond;
    if (sampler_shape(sampler, B, NB, n_cond)) return fail(e, DR_EINVAL, "unknown sampler %d", sampler);
    if ((rc = ensure_workspace(e, NB, T))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t per = (size_t)B * T * 88;
    auto chain = [&](float* xbuf) -> int {
        // the roll ping-pongs between xbuf and e->xalt while the fused step runs (its tail kernel cannot update in
        // place), and each tail also computes the next step's input projection
        ChainState cs;
        float* cur = xbuf;
        for (int t = e->S - 1; t >= 0; --t) {
            // row t of the injected noise is the z of step t; t == 0 draws none (task/diffusion.py:957-960)
            const float* z = d_noise ? d_noise + (size_t)t * per : nullptr;
            cs.next_t = t - 1;
            float* res = nullptr;
            int r;
            if (cur == e->xalt) {      // the previous step left the roll in the engine's buffer: this one writes back into xbuf
                float* keep = e->xalt;
                e->xalt = xbuf;
                r = run_step(e, sampler, cur, z, B, T, t, w, seed, first_sample, st, &res, &cs);
                e->xalt = keep;
            } else {
                r = run_step(e, sampler, cur, z, B, T, t, w, seed, first_sample, st, &res, &cs);
            }
            if (r) return r;
            cur = res;
        }
        if (cur != xbuf) HIPCHK(e, hipMemcpyAsync(xbuf, cur, per * sizeof(float), hipMemcpyDeviceToDevice, st));
        return DR_OK;
    };
    if (!use_graph || e->prof) {
        Range range("dr_sample: eager chain");
        return chain(d_x);
    }

    GraphKey key;
    key.sampler = sampler; key.B = B; key.T = T; key.x = e->xwork; key.noise = d_noise; key.w_zero = (w == 0.f);
    if (!e->gexec || !(key == e->gkey)) {
        if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
        if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
        if (!e->cap_stream) HIPCHK(e, hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        hipStream_t user = st;
        st = e->cap_stream;   // chain() launches on `st`
        Range range("dr_sample: capture + instantiate the chain graph");
        const double tc0 = now_s();
        HIPCHK(e, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        e->use_dyn = true;
        rc = chain(e->xwork);
        e->use_dyn = false;
        hipGraph_t gr = nullptr;
        hipError_t ce = hipStreamEndCapture(st, &gr);
        st = user;
        if (rc) { if (gr) (void)hipGraphDestroy(gr); return rc; }
        if (ce != hipSuccess) return fail(e, DR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
        e->graph = gr;
        HIPCHK(e, hipGraphInstantiate(&e->gexec, e->graph, nullptr, nullptr, 0));
        e->t_capture_s = now_s() - tc0;
        e->gkey = key;
    }
    Range range("dr_sample: launch the chain graph");
    // the graph owns no caller address: x_T is copied in, the finished roll copied out (0.7 MB each way)
    HIPCHK(e, hipMemcpyAsync(e->xwork, d_x, per * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(e, launch_set_dyn(e->d_dyn, seed, first_sample, w, (float)(1.0 + (double)w), st));
    HIPCHK(e, hipGraphLaunch(e->gexec, st));
    if (e->stack_launches || e->tail_launches) e->unverified = true;      // (the captured chain may hold persistent launches)
    HIPCHK(e, hipMemcpyAsync(d_x, e->xwork, per * sizeof(float), hipMemcpyDeviceToDevice, st));
    return DR_OK;
}

int dr_finish(dr_engine* e, void* stream) {
    if (!e) return DR_EINVAL;
    DeviceGuard guard(e->cfg.device);
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    if (!e->stack_err_host || !*e->stack_err_host) {
        e->unverified = false;
        return DR_OK;
    }
    // A group barrier of the fused kernel gave up: something else held CUs while it ran (another engine / stream /
    // process on this device).  Everything computed since the last dr_finish is invalid.  Heal: wait for the
    // device, re-arm, and run this engine on the per-phase kernels from now on (bit-identical results, no
    // co-residency assumption) - the caller recomputes.
    HIPCHK(e, hipDeviceSynchronize());
    int rc = clear_stack_timeout(e);
    if (rc) return rc;
    drop_graph(e);
    e->unverified = false;
    if (e->opt_stack) e->healed_from = e->opt_stack;      // (option "fused_rearm" may restore it after clean chains)
    e->clean_chains = 0;
    e->opt_stack = 0;
    e->stack_fallbacks += 1;
    static std::atomic<bool> warned{false};           // (engines of several host threads may get here together)
    if (!warned.exchange(true)) {
        fprintf(stderr, "[diffroll_amd] a group barrier of the fused residual-stack kernel timed out (another stream, engine or "
                        "process is computing on device %d): this engine now uses one launch per phase (option fused_stack = 0); "
                        "results since the last check are recomputed\n", e->cfg.device);
    }
    return fail(e, DR_ETIMEOUT, "a fused residual-stack launch timed out: results since the last dr_finish are invalid and must be "
                                "recomputed; the engine has been switched to per-phase launches (fused_stack = 0)");
}

int dr_sample_checked(dr_engine* e, int sampler, float* d_x, const float* d_noise, int B, int T, float w, uint64_t seed,
                      int first_sample, int use_graph, int32_t* recovered, void* stream) {
    if (recovered) *recovered = 0;
    if (!e || !d_x) return fail(e, DR_EINVAL, "null argument");
    if (B <= 0 || T <= 0) return fail(e, DR_EINVAL, "bad shape B=%d T=%d", B, T);
    DeviceGuard guard(e->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const size_t per = (size_t)B * T * 88;
    const bool may_fuse = e->opt_stack != 0;
    if (may_fuse) {       // only a fused launch can time out: keep x_T so that the chain can be re-run
        if (per > e->xsave_cap) {
            HIPCHK(e, hipStreamSynchronize(st));
            int rc = dev_alloc(e, &e->xsave, per, false);
            if (rc) return rc;
            e->xsave_cap = per;
        }
        HIPCHK(e, hipMemcpyAsync(e->xsave, d_x, per * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    int rc = dr_sample(e, sampler, d_x, d_noise, B, T, w, seed, first_sample, use_graph, st);
    if (rc == DR_ETIMEOUT) {          // a flag left by unchecked earlier calls: clear it and carry on (nothing of THIS call ran)
        (void)dr_finish(e, st);
        rc = dr_sample(e, sampler, d_x, d_noise, B, T, w, seed, first_sample, use_graph, st);
    }
    if (rc) return rc;
    rc = dr_finish(e, st);
    if (rc == DR_OK && e->healed_from && e->opt_rearm > 0 && !e->opt_stack && ++e->clean_chains >= e->opt_rearm) {
        // option "fused_rearm": the tenant that caused the time-out has had opt_rearm chains to leave - fuse again
        drop_graph(e);
        e->opt_stack = e->healed_from;
        e->healed_from = 0;
        e->clean_chains = 0;
    }
    if (rc != DR_ETIMEOUT) return rc;
    if (!may_fuse) return rc;         // cannot happen: no fused launch was issued
    HIPCHK(e, hipMemcpyAsync(d_x, e->xsave, per * sizeof(float), hipMemcpyDeviceToDevice, st));
    rc = dr_sample(e, sampler, d_x, d_noise, B, T, w, seed, first_sample, use_graph, st);     // per-phase kernels now
    if (rc) return rc;
    rc = dr_finish(e, st);
    if (rc == DR_OK && recovered) *recovered = 1;
    if (rc == DR_OK) e->err.clear();
    return rc;
}

int dr_pending_timeout(dr_engine* e, void* stream) {
    if (!e || !e->stack_err_host) return DR_OK;
    if (e->unverified) {
        DeviceGuard guard(e->cfg.device);
        HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    }
    if (*e->stack_err_host)
        return fail(e, DR_ETIMEOUT, "a fused residual-stack launch issued on this engine timed out and has not been checked: the roll is "
                                    "invalid - call dr_finish (clears the condition, switches to per-phase launches) and recompute, or "
                                    "use dr_sample_checked");
    e->unverified = false;
    return DR_OK;
}

int dr_cold_times(dr_engine* e, double* out5) {
    if (!e || !out5) return DR_EINVAL;
    out5[0] = e->t_pack_s; out5[1] = e->t_upload_s; out5[2] = e->t_tables_s; out5[3] = e->t_capture_s;
    size_t nodes = 0;
    if (e->graph && hipGraphGetNodes(e->graph, nullptr, &nodes) != hipSuccess) nodes = 0;
    out5[4] = (double)nodes;
    return DR_OK;
}

int dr_stack_fallbacks(dr_engine* e, int64_t* count) {
    if (!e || !count) return DR_EINVAL;
    *count = e->stack_fallbacks;
    return DR_OK;
}

int dr_tail_launches(dr_engine* e, int64_t* count) {
    if (!e || !count) return DR_EINVAL;
    *count = e->tail_launches;
    return DR_OK;
}

int dr_note_runs(dr_engine* e, const float* d_roll, int B, int T, float threshold, int32_t* d_note_end, void* stream) {
    if (!e || !d_roll || !d_note_end) return fail(e, DR_EINVAL, "null argument");
    if (B <= 0 || T <= 0) return fail(e, DR_EINVAL, "bad shape B=%d T=%d", B, T);
    if (int rc = dr_pending_timeout(e, stream)) return rc;
    DeviceGuard guard(e->cfg.device);
    HIPCHK(e, launch_note_runs(d_roll, d_note_end, B, T, threshold, (hipStream_t)stream));
    return DR_OK;
}

int dr_frame_counts(dr_engine* e, const float* d_pred, const float* d_label, size_t n, float threshold,
                    int64_t* host_counts, void* stream) {
    if (!e || !d_pred || !d_label || !host_counts) return fail(e, DR_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = dr_pending_timeout(e, stream)) return rc;
    DeviceGuard guard(e->cfg.device);
    if (!e->d_counts) {
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, 3 * sizeof(unsigned long long)));
        e->d_counts = (unsigned long long*)q;
    }
    HIPCHK(e, hipMemsetAsync(e->d_counts, 0, 3 * sizeof(unsigned long long), st));
    HIPCHK(e, launch_frame_counts(d_pred, d_label, threshold, (long)n, e->d_counts, st));
    unsigned long long h[3];
    HIPCHK(e, hipMemcpyAsync(h, e->d_counts, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    for (int i = 0; i < 3; ++i) host_counts[i] = (int64_t)h[i];
    return DR_OK;
}

static int noise_mix(dr_engine* e, int mode, const float* a, const float* b, const int64_t* d_t, const float* d_sac,
                     const float* d_s1m, int n_steps, int B, size_t per_sample, float* d_out, void* stream) {
    // e may be NULL (free functions of the reference: no engine state is involved; current device; the error text
    // is then read with dr_last_error(NULL))
    if (!a || !b || !d_t || !d_sac || !d_s1m || !d_out) return fail(e, DR_EINVAL, "null argument");
    if (B <= 0 || n_steps <= 0 || per_sample == 0) return fail(e, DR_EINVAL, "bad shape B=%d n_steps=%d", B, n_steps);
    if (int rc = dr_pending_timeout(e, stream)) return rc;
    HIPCHK(e, launch_noise_mix(mode, a, b, d_t, d_sac, d_s1m, n_steps, B, (long)per_sample, d_out, (hipStream_t)stream));
    return DR_OK;
}
int dr_q_sample(dr_engine* e, const float* d_x_start, const float* d_noise, const int64_t* d_t, const float* d_sac,
                const float* d_s1m, int n_steps, int B, size_t per_sample, float* d_out, void* stream) {
    return noise_mix(e, 0, d_x_start, d_noise, d_t, d_sac, d_s1m, n_steps, B, per_sample, d_out, stream);
}
int dr_extract_x0(dr_engine* e, const float* d_x_t, const float* d_epsilon, const int64_t* d_t, const float* d_sac,
                  const float* d_s1m, int n_steps, int B, size_t per_sample, float* d_out, void* stream) {
    return noise_mix(e, 1, d_x_t, d_epsilon, d_t, d_sac, d_s1m, n_steps, B, per_sample, d_out, stream);
}

int dr_profile_enable(dr_engine* e, int on) {
    if (!e) return DR_EINVAL;
    DeviceGuard guard(e->cfg.device);
    e->prof = on != 0;
    if (e->prof && e->prof_events.empty()) {
        const size_t n = (size_t)e->S * e->L;
        e->prof_events.resize(n);
        for (auto& p : e->prof_events) { HIPCHK(e, hipEventCreate(&p.first)); HIPCHK(e, hipEventCreate(&p.second)); }
    }
    e->prof_used = 0;
    return DR_OK;
}

int dr_profile_read(dr_engine* e, int64_t* launches, double* total_ms, int reset) {
    if (!e) return DR_EINVAL;
    DeviceGuard guard(e->cfg.device);
    HIPCHK(e, hipDeviceSynchronize());
    for (size_t i = 0; i < e->prof_used; ++i) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->prof_events[i].first, e->prof_events[i].second));
        e->prof_ms += ms;
        e->prof_launches += 1;
    }
    e->prof_used = 0;
    if (launches) *launches = e->prof_launches;
    if (total_ms) *total_ms = e->prof_ms;
    if (reset) { e->prof_launches = 0; e->prof_ms = 0.0; }
    return DR_OK;
}

int dr_profile_read_ex(dr_engine* e, int64_t* launches, double* total_ms, double* total_flops, char* name, size_t name_len,
                       int reset) {
    if (!e) return DR_EINVAL;
    const double fl = e->prof_flops;
    int rc = dr_profile_read(e, launches, total_ms, reset);
    if (rc) return rc;
    if (total_flops) *total_flops = fl;
    if (name && name_len) {
        snprintf(name, name_len, "%s", e->prof_name.c_str());
    }
    if (reset) e->prof_flops = 0.0;
    return DR_OK;
}

int dr_set_option(dr_engine* e, const char* name, int value) {
    if (!e || !name) return fail(e, DR_EINVAL, "null argument");
    const std::string n = name;
    DeviceGuard guard(e->cfg.device);
    auto drop_graph = [&]() {
        (void)hipDeviceSynchronize();
        if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
        if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
        e->gkey = GraphKey{};
    };
    if (n == "fused_stack") { if (e->opt_stack != value) drop_graph(); e->opt_stack = value; return DR_OK; }
    if (n == "fused_stack_xcd") { if (e->opt_stack_xcd != value) drop_graph(); e->opt_stack_xcd = value; return DR_OK; }
    if (n == "fused_rearm") { e->opt_rearm = value; return DR_OK; }
    if (n == "blocked_accumulation") {
        if (value != 1 && value != 2) return fail(e, DR_EINVAL, "blocked_accumulation is 1 or 2");
        if (e->opt_blocked != value) drop_graph();
        e->opt_blocked = value;
        return DR_OK;
    }
    if (n == "fused_tail") { if (e->opt_tail != value) drop_graph(); e->opt_tail = value; return DR_OK; }
    if (n == "fused_stack_warm") { if (e->opt_stack_warm != value) drop_graph(); e->opt_stack_warm = value; return DR_OK; }
    if (n == "stack_fault_test") { if (e->opt_stack_fault != value) drop_graph(); e->opt_stack_fault = value; return DR_OK; }
    if (n == "stack_ticks") { if (e->stack_dbg_on != value) drop_graph(); e->stack_dbg_on = value; return DR_OK; }
    return fail(e, DR_ENAME, "unknown option '%s'", name);
}

int dr_stack_status(dr_engine* e, int32_t* timed_out, int64_t* launches, int64_t* ticks, int n_ticks) {
    if (!e) return DR_EINVAL;
    if (!e->stack_bar) return fail(e, DR_ESTATE, "dr_commit has not been called");
    DeviceGuard guard(e->cfg.device);
    HIPCHK(e, hipDeviceSynchronize());
    const unsigned flag = *e->stack_err_host;
    if (timed_out) *timed_out = (int32_t)flag;
    if (launches) *launches = e->stack_launches;
    if (flag) {      // a barrier wait hit its spin bound: counters may be left armed - reset everything
        int rc = clear_stack_timeout(e);
        if (rc) return rc;
    }
    if (ticks && n_ticks > 0) {
        long long h[128];
        HIPCHK(e, hipMemcpy(h, e->stack_dbg, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < n_ticks && i < 128; ++i) ticks[i] = h[i];
    }
    return DR_OK;
}

int dr_bench_layer(dr_engine* e, int layer, int NB, int T, int t, int n_cond, void* stream) {
    if (!e) return DR_EINVAL;
    if (!e->committed) return fail(e, DR_ESTATE, "dr_commit has not been called");
    if (layer < 0 || layer >= e->L || t < 0 || t >= e->S || n_cond < 0 || n_cond > NB)
        return fail(e, DR_EINVAL, "bad argument");
    if (n_cond > 0 && (e->fe_B < n_cond || e->fe_T != T)) return fail(e, DR_ESTATE, "dr_frontend needed for n_cond > 0");
    DeviceGuard guard(e->cfg.device);
    int rc = ensure_workspace(e, NB, T);
    if (rc) return rc;
    const int Cp = e->Cp, P = Cp / 4;
    const LayerW& w = e->layers[layer];
    GemmArgs a = p4_gemm(e->prec ? w.conv_w3 : w.conv_w, w.conv_b, Cp / 64, e->hd, P, NB, T);
    const long act_bs = (long)Cp * T, s3_bs = act_bs + act_bs / 2;
    if (e->prec) {
        a.X = e->hd3; a.x_bs = s3_bs; a.x_piece = (long)(Cp / 8) * T * 4; a.x_ps = (long)T * 4; a.x_fs = 4;
        a.x_planes = Cp / 8; a.kchunks = Cp / 32;
    }
    a.bias2 = w.conv_b_u;
    a.taps = e->K; a.dil = w.dil;
    (void)t;
    a.cond = e->cond ? e->cond + (size_t)layer * e->fe_B * 2 * Cp * T : e->cond_dummy;
    a.c_bs = (long)2 * Cp * T;
    a.n_cond = n_cond;
    p4_out(a, e->g, P, T, Cp);
    if (e->prec) { a.Y = e->g3; a.y_bs = s3_bs; a.out_s3 = 1; }
    if (!e->dbg_ticks) {
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, 16 * sizeof(long long)));
        HIPCHK(e, hipMemset(q, 0, 16 * sizeof(long long)));
        e->dbg_ticks = (long long*)q;
    }
    a.dbg = e->dbg_ticks;
    allow_splitk(e, a);
    HIPCHK(e, launch_tiled(a, EPI_GATE, pick_tile(Cp / 64, NB, T, e->K, w.dil, e->prec, EPI_GATE, true, e->opt_blocked >= 2), (hipStream_t)stream, e->prec));
    return DR_OK;
}

int dr_bench_pointwise(dr_engine* e, int layer, int NB, int T, void* stream) {
    if (!e) return DR_EINVAL;
    if (!e->committed) return fail(e, DR_ESTATE, "dr_commit has not been called");
    if (layer < 0 || layer >= e->L) return fail(e, DR_EINVAL, "bad argument");
    DeviceGuard guard(e->cfg.device);
    int rc = ensure_workspace(e, NB, T);
    if (rc) return rc;
    const int Cp = e->Cp, P = Cp / 4;
    const LayerW& w = e->layers[layer];
    GemmArgs a = p4_gemm(e->prec ? w.out_w3 : w.out_w, w.out_b, Cp / 64, e->g, P, NB, T);
    if (e->prec) {
        const long act_bs = (long)Cp * T;
        a.X = e->g3; a.x_bs = act_bs + act_bs / 2; a.x_piece = (long)(Cp / 8) * T * 4; a.x_ps = (long)T * 4; a.x_fs = 4;
        a.x_planes = Cp / 8; a.kchunks = Cp / 32;
    }
    p4_out(a, e->h, P, T, Cp);
    a.skip = e->skip; a.s_bs = (long)Cp * T; a.skip_init = 0;
    {   // second output as in the chain: hd = h + d_{l+1} (fp32 P4 or split-bf16)
        const long act_bs = (long)Cp * T;
        a.d2 = e->d_dtab + (size_t)((layer + 1) % e->L) * Cp;
        if (e->prec) { a.Y2 = e->hd3; a.y2_bs = act_bs + act_bs / 2; a.out_s3 = 2; }
        else { a.Y2 = e->hd; a.y2_bs = act_bs; }
    }
    if (!e->dbg_ticks) {
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, 16 * sizeof(long long)));
        HIPCHK(e, hipMemset(q, 0, 16 * sizeof(long long)));
        e->dbg_ticks = (long long*)q;
    }
    a.dbg = e->dbg_ticks;
    allow_splitk(e, a);
    HIPCHK(e, launch_tiled(a, EPI_RES_SKIP, pick_pointwise_tile(Cp / 64, NB, T, e->prec), (hipStream_t)stream, e->prec));
    return DR_OK;
}

int dr_debug_stft_power(dr_engine* e, const float* d_wav, int B, int L, float* d_power_out, void* stream) {
    if (!e || !d_wav || !d_power_out) return fail(e, DR_EINVAL, "null argument");
    if (!e->committed) return fail(e, DR_ESTATE, "dr_commit has not been called");
    if (!e->use_fft) return fail(e, DR_ESTATE, "n_fft = %d is not a power of two: the spectrum is a windowed-DFT GEMM, not the FFT kernel", e->cfg.n_fft);
    DeviceGuard guard(e->cfg.device);
    hipStream_t st = (hipStream_t)stream;
    const int N = e->cfg.n_fft, hop = e->cfg.hop_length, pad = N / 2;
    if (B <= 0 || L <= pad) return fail(e, DR_EINVAL, "bad front-end shape B=%d L=%d", B, L);
    const int TF = L / hop + 1, Lp = (L + 2 * pad + 3) & ~3, bp = e->bins_p;
    float *wp = nullptr, *pw = nullptr;        // private buffers: the engine's front-end state is left alone
    int rc;
    if ((rc = dev_alloc(e, &wp, (size_t)B * Lp, false))) return rc;
    if ((rc = dev_alloc(e, &pw, (size_t)B * bp * TF, false))) { (void)hipFree(wp); return rc; }
    hipError_t he = launch_reflect_pad(d_wav, wp, B, L, pad, st);
    if (he == hipSuccess) he = launch_stft_power(wp, e->fft_win, e->fft_tw, pw, B, Lp, TF, N, hop, bp, e->fft_norm, st);
    if (he == hipSuccess)
        he = hipMemcpy2DAsync(d_power_out, (size_t)e->n_bins * 4, pw, (size_t)bp * 4, (size_t)e->n_bins * 4, (size_t)B * TF,
                              hipMemcpyDeviceToDevice, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    (void)hipFree(wp);
    (void)hipFree(pw);
    if (he != hipSuccess) return fail(e, DR_EHIP, "dr_debug_stft_power: %s", hipGetErrorString(he));
    return DR_OK;
}

int dr_debug_bounds(int64_t* out4, int reset) {
    if (!out4) return DR_EINVAL;
    (void)hipDeviceSynchronize();
    unsigned long long v[4] = {0, 0, 0, 0};
    hipError_t he = read_bounds(v);
    if (he == hipErrorNotSupported) return fail(nullptr, DR_ESTATE, "not a checker build (compile csrc with -DDR_BOUNDS: tools/checked_build.sh)");
    if (he != hipSuccess) return fail(nullptr, DR_EHIP, "dr_debug_bounds: %s", hipGetErrorString(he));
    for (int i = 0; i < 4; ++i) out4[i] = (int64_t)v[i];
    if (reset && reset_bounds() != hipSuccess) return fail(nullptr, DR_EHIP, "dr_debug_bounds: reset failed");
    return DR_OK;
}

int dr_set_spec_norm(dr_engine* e, int mode) {
    if (!e) return DR_EINVAL;
    if (mode != DR_NORM_IMAGEWISE && mode != DR_NORM_FRAMEWISE) return fail(e, DR_EINVAL, "unknown normalisation mode %d", mode);
    e->norm_framewise = mode == DR_NORM_FRAMEWISE;
    return DR_OK;
}

int dr_set_precision(dr_engine* e, int mode) {
    if (!e) return DR_EINVAL;
    if (mode != DR_PRECISION_F32 && mode != DR_PRECISION_BF16X3) return fail(e, DR_EINVAL, "unknown precision mode %d", mode);
    if (mode != e->prec) {
        DeviceGuard guard(e->cfg.device);     // the graph may still be executing on the ENGINE's device
        (void)hipDeviceSynchronize();
        if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
        if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
        e->gkey = GraphKey{};
        e->prec = mode;
        if (mode) return ensure_s3(e);
    }
    return DR_OK;
}

int dr_debug_ticks(dr_engine* e, int64_t* loop_ticks, int64_t* block_ticks) {
    if (!e || !e->dbg_ticks) return fail(e, DR_ESTATE, "no dr_bench_layer launch yet");
    DeviceGuard guard(e->cfg.device);
    long long h[16];
    HIPCHK(e, hipDeviceSynchronize());
    HIPCHK(e, hipMemcpy(h, e->dbg_ticks, sizeof h, hipMemcpyDeviceToHost));
    if (loop_ticks) *loop_ticks = h[0];
    if (block_ticks) *block_ticks = h[1];
    if (getenv("DR_DEBUG_CHUNKS")) {
        fprintf(stderr, "[dr] chunk-start ticks since block start:");
        for (int i = 2; i < 16; ++i) fprintf(stderr, " %lld", h[i]);
        fprintf(stderr, "\n");
    }
    return DR_OK;
}

}  // extern "C"
