// Host side, part 3: the C-ABI of include/diffroll_amd.h (the boundary) - engine life cycle, front-end, forward / step / sample
// (the reverse chain as one hipGraph), time-out handling of the persistent kernels, the consumers of a finished roll, options.
#include "engine_state.h"
#include "tenants.h"

#include <condition_variable>
#include <mutex>

namespace drh {

thread_local std::string g_create_error;
void release_stager(int dev);       // pack.hip
std::atomic<int> g_engines[MAX_DEVICES];      // live engines per device: the last one out releases the upload stager

// The persistent kernels assume that all their workgroups are resident at once - true while ONE engine computes on the
// device.  Engines of one process take turns on a per-device "fused slot".  An engine that wants to issue fused launches
// while another engine of the process is issuing its own (another host thread, inside an API call) waits for that call to
// return - microseconds to milliseconds of launch overhead, no GPU wait; if that engine's last fused work is still running
// on the device, the newcomer's stream is made to wait for it ON THE DEVICE (hipStreamWaitEvent on the event recorded
// behind that work): the two engines' persistent launches then never overlap, nobody synchronises the host and nobody gives
// up fusing.  (Until round 5 the newcomer yielded to per-phase launches for the rest of its life instead; its per-phase
// blocks cannot co-reside with a persistent launch that owns every CU's LDS either, so nothing was gained by not waiting.)
// (Other PROCESSES on the device are looked for in tenants.h; the ~1 s spin bound of the barriers stays as the backstop.)
struct FusedSlot {
    std::mutex mu;
    std::condition_variable cv;
    dr_engine* owner = nullptr;
    bool claimed = false;          // the owner is issuing launches right now (its event is not recorded yet)
    hipEvent_t done = nullptr;     // recorded behind the owner's last fused work
};
FusedSlot g_slots[MAX_DEVICES];

// true: the slot is ours (after `st` has been ordered behind the previous owner's fused work, if any is in flight)
bool claim_fused_slot(dr_engine* e, hipStream_t st) {
    FusedSlot& s = g_slots[e->cfg.device];
    std::unique_lock<std::mutex> lk(s.mu);
    s.cv.wait(lk, [&] { return !s.claimed || s.owner == e; });
    if (s.owner && s.owner != e && s.done && hipEventQuery(s.done) == hipErrorNotReady) {
        if (hipStreamWaitEvent(st, s.done, 0) != hipSuccess) return false;
    }
    s.owner = e;
    s.claimed = true;
    return true;
}
void release_fused_slot(dr_engine* e, hipStream_t st) {
    FusedSlot& s = g_slots[e->cfg.device];
    {
        std::lock_guard<std::mutex> lk(s.mu);
        if (s.owner != e) return;
        if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) s.done = nullptr;
        if (s.done) (void)hipEventRecord(s.done, st);
        s.claimed = false;
    }
    s.cv.notify_all();
}
void forget_fused_slot(dr_engine* e) {
    FusedSlot& s = g_slots[e->cfg.device];
    {
        std::lock_guard<std::mutex> lk(s.mu);
        if (s.owner == e) { s.owner = nullptr; s.claimed = false; }
    }
    s.cv.notify_all();
}
// this engine stops fusing (another process is computing on the device): per-phase launches from the next launch on
int yield_fused(dr_engine* e, const char* why) {
    if (e->gexec && e->graph_stream_set) HIPCHK(e, hipStreamSynchronize(e->graph_stream));      // its captured chain may still be running
    drop_graph(e);
    e->yielded_from = e->opt_stack;     // (restored by dr_sample after two clean looks)
    e->yield_clean = 0;
    e->opt_stack = 0;
    e->stack_yields += 1;
    if (e->stack_yields <= 3)           // per engine (a measurement reads the counter: dr_launch_state); not a log flood
        fprintf(stderr, "[diffroll_amd] engine %p: %s on device %d - one launch per phase from now on (same results, no co-residency "
                        "assumption; yield #%lld of this engine; fused launches come back after two clean looks)\n",
                (void*)e, why, e->cfg.device, (long long)e->stack_yields);
    return DR_OK;
}
// Another process on this GPU (tenants.h)?  Asked at creation and in front of a chain, at most every 250 ms (a scan is
// ~0.1 ms of sysfs reads: 0.4 % of a single-clip chain if it ran every time; force: now - behind a graph capture).
// Returns 1 = yes, 0 = no, -1 = not looked (rate limit / no sysfs / undecided).  A first look that finds a second queue
// holder AND busy CUs may be seeing THIS process's own kernels - the engine's front-end, another stream of the caller, the
// previous sample's RCCL all-gather on the communicator's stream (a rank of a multi-GPU job never waits for that before it
// starts its next sample): only then - an exclusive GPU never pays for it - everything this process has in flight on
// the device is waited for (hipDeviceSynchronize: microseconds of front-end work in practice) and the look repeated, so
// that what is still busy afterwards is somebody else's.
std::mutex g_kfd_mu;
std::string g_kfd_root = "/sys/class/kfd/kfd";
std::string kfd_root() { std::lock_guard<std::mutex> lk(g_kfd_mu); return g_kfd_root; }
void set_kfd_root(const char* root) { std::lock_guard<std::mutex> lk(g_kfd_mu); g_kfd_root = root ? root : "/sys/class/kfd/kfd"; }
int shared_with_another_process(dr_engine* e, bool force = false, bool may_sync = true) {
    if (e->kfd_gpu_id < 0) return -1;
    const double now = now_s();
    if (!force && now - e->last_tenant_scan_s < 0.250) return -1;
    e->last_tenant_scan_s = now;
    const std::string root = kfd_root();
    TenantScan t = scan_tenants(root, e->kfd_gpu_id);
    if (!t.readable) return -1;
    if (!(t.holders >= 2 && t.busy_cus > 0)) return 0;
    if (!may_sync) {                    // (undecided: what is busy may be this process's own other engines)
        e->last_tenant_scan_s = 0;      // the look in front of the engine's first chain decides, whatever the rate limit says
        return -1;
    }
    (void)hipDeviceSynchronize();
    t = scan_tenants(root, e->kfd_gpu_id);
    return (t.holders >= 2 && t.busy_cus > 0) ? 1 : 0;
}

// the engine's turn on the slot for the duration of one API call that may issue fused launches
struct FusedTurn {
    dr_engine* e;
    hipStream_t st;
    bool held = false;
    int rc = DR_OK;
    FusedTurn(dr_engine* e_, hipStream_t st_) : e(e_), st(st_) {
        if (!e->opt_stack) return;
        if (claim_fused_slot(e, st)) held = true;
        else rc = fail(e, DR_EHIP, "hipStreamWaitEvent behind another engine's fused work failed");
    }
    ~FusedTurn() { if (held) release_fused_slot(e, st); }
    FusedTurn(const FusedTurn&) = delete;
    FusedTurn& operator=(const FusedTurn&) = delete;
};

// After a barrier time-out (device idle): re-arm the group counters, forget the published XCC tags, lower both flags.
int clear_stack_timeout(dr_engine* e) {
    HIPCHK(e, hipMemset(e->stack_bar, 0, (size_t)(12 * dr_engine::STACK_GROUPS) * sizeof(unsigned)));    // all three counter arrays
    HIPCHK(e, hipMemset(e->stack_xid, 0xFF, 1024 * sizeof(unsigned)));
    HIPCHK(e, hipMemset(e->stack_derr, 0, 16 * sizeof(unsigned)));
    *e->stack_err_host = 0;
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
    if (e->prec && !e->s3_ready) {       // (a commit after dr_set_precision, or a packing build that failed: never launch without them)
        int rc = ensure_s3(e);
        if (rc) return rc;
    }
    if (sampler != DR_SAMPLER_GENERATION_DDPM_X0 && (e->fe_B != B || e->fe_T != T))
        return fail(e, DR_ESTATE, "dr_frontend(B=%d,T=%d) must precede a conditional evaluation with B=%d,T=%d",
                    e->fe_B, e->fe_T, B, T);
    return DR_OK;
}

// dr_set_option (lab = false: the product's options) / dr_debug_set_option (lab = true: the A/B and test knobs too)
int set_option(dr_engine* e, const char* name, int value, bool lab) {
    if (!e || !name) return fail(e, DR_EINVAL, "null argument");
    const std::string n = name;
    DeviceGuard guard(e->cfg.device);
    auto drop = [&]() {
        (void)hipDeviceSynchronize();
        drop_graph(e);
    };
    if (n == "fused_stack") {
        if (e->opt_stack != value) drop();
        e->opt_stack = value;
        e->yielded_from = 0; e->healed_from = 0;      // the caller's word replaces any pending re-arm
        return DR_OK;
    }
    if (n == "fused_rearm") { e->opt_rearm = value; return DR_OK; }
    if (n == "blocked_accumulation") {
        if (value != 1 && value != 2) return fail(e, DR_EINVAL, "blocked_accumulation is 1 or 2");
        if (e->opt_blocked != value) drop();
        e->opt_blocked = value;
        return DR_OK;
    }
    if (n == "fused_tail") { if (e->opt_tail != value) drop(); e->opt_tail = value; return DR_OK; }
    if (!lab) return fail(e, DR_ENAME, "unknown option '%s'", name);
    if (n == "fused_stack_xcd") { if (e->opt_stack_xcd != value) drop(); e->opt_stack_xcd = value; return DR_OK; }
    if (n == "fused_stack_warm") { if (e->opt_stack_warm != value) drop(); e->opt_stack_warm = value; return DR_OK; }
#ifdef DR_FAULT_HOOK
    if (n == "stack_fault_test") { if (e->opt_stack_fault != value) drop(); e->opt_stack_fault = value; return DR_OK; }
#endif
    if (n == "stack_ticks") { if (e->stack_dbg_on != value) drop(); e->stack_dbg_on = value; return DR_OK; }
    if (n.compare(0, 5, "tune.") == 0) {      // A/B knobs of planners and launchers: PROCESS-wide (kernels.h: Tuning)
        Tuning& t = tuning();
        const std::string f = n.substr(5);
        if (f == "ksplit_blocks") {
            if (t.ksplit_blocks.load() != value) { t.ksplit_blocks.store(value); tuning_epoch().fetch_add(1); drop(); }
            return DR_OK;
        }
        std::atomic<int>* field = f == "pack_threads" ? &t.pack_threads : f == "tile" ? &t.tile : f == "pw" ? &t.pw : f == "pw_nw" ? &t.pw_nw
                   : f == "pwk" ? &t.pwk : f == "ksplit_max" ? &t.ksplit_max : f == "one_ks" ? &t.one_ks : f == "stack3" ? &t.stack3
                   : f == "stack_fl" ? &t.stack_fl : f == "tail_t4" ? &t.tail_t4 : f == "xcd_n" ? &t.xcd_n : f == "xcd_model" ? &t.xcd_model
                   : f == "s3_eager" ? &t.s3_eager : f == "debug_chunks" ? &t.debug_chunks : nullptr;
        if (!field) return fail(e, DR_ENAME, "unknown option '%s'", name);
        if (field->load() != value) {
            field->store(value);
            tuning_epoch().fetch_add(1);      // every engine drops its captured chain at its next dr_sample
            drop();
        }
        return DR_OK;
    }
    return fail(e, DR_ENAME, "unknown option '%s'", name);
}

}  // namespace drh
using namespace drh;

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
    g_engines[cfg->device].fetch_add(1);
    {   // whose GPU is it?
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess)
            e->kfd_gpu_id = kfd_gpu_id(kfd_root(), prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
        // (only the FIRST engine of the process on this device may wait for the device in a doubtful look: later ones would
        // wait for their siblings' chains - and if those are what is busy, the fused slot already handles it)
        if (shared_with_another_process(e, false, g_engines[cfg->device].load() == 1) == 1)
            (void)yield_fused(e, "another process is computing");
    }
    *out = e;
    return DR_OK;
}

void dr_destroy(dr_engine* e) {
    if (!e) return;
    DeviceGuard guard(e->cfg.device);
    (void)hipDeviceSynchronize();
    forget_fused_slot(e);
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
    if (g_engines[e->cfg.device].fetch_sub(1) == 1) release_stager(e->cfg.device);
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
    return commit(e, (hipStream_t)stream);
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
        const size_t mm_need0 = (e->norm_framewise ? (size_t)B * TF * 2 : (size_t)B * 2) + minmax_scratch_floats(B);
        const bool grow = (size_t)B * Lp > e->fe_cap_wav || (size_t)B * bp * TF > e->fe_cap_pow ||
                          (size_t)B * mel_planes * 4 * TF > e->fe_cap_log || (size_t)B * mel_planes * 4 * T > e->fe_cap_spec ||
                          mm_need0 > e->fe_cap_mm || (size_t)e->L * B * 2 * Cp * T > e->cond_cap;
        if (grow) HIPCHK(e, hipDeviceSynchronize());
    }
    if ((size_t)B * Lp > e->fe_cap_wav) { if ((rc = dev_alloc(e, &e->wav_pad, (size_t)B * Lp))) return rc; e->fe_cap_wav = (size_t)B * Lp; }
    if ((size_t)B * bp * TF > e->fe_cap_pow) { if ((rc = dev_alloc(e, &e->power, (size_t)B * bp * TF))) return rc; e->fe_cap_pow = (size_t)B * bp * TF; }
    if ((size_t)B * mel_planes * 4 * TF > e->fe_cap_log) { if ((rc = dev_alloc(e, &e->logmel, (size_t)B * mel_planes * 4 * TF))) return rc; e->fe_cap_log = (size_t)B * mel_planes * 4 * TF; }
    if ((size_t)B * mel_planes * 4 * T > e->fe_cap_spec) { if ((rc = dev_alloc(e, &e->specP4, (size_t)B * mel_planes * 4 * T))) return rc; e->fe_cap_spec = (size_t)B * mel_planes * 4 * T; }
    // (min, max) per clip / per frame, followed by the per-clip partials + ticket words of the multi-block min-max
    const size_t mm_vals = e->norm_framewise ? (size_t)B * TF * 2 : (size_t)B * 2;
    const size_t mm_need = mm_vals + minmax_scratch_floats(B);
    if (mm_need > e->fe_cap_mm) { if ((rc = dev_alloc(e, &e->mm, mm_need))) return rc; e->fe_cap_mm = mm_need; e->mm_scratch_off = mm_vals; }
    else if (e->mm_scratch_off != mm_vals) {      // same buffer, other split: the ticket words must be zero where they now lie
        HIPCHK(e, hipMemsetAsync(e->mm, 0, e->fe_cap_mm * sizeof(float), st));
        e->mm_scratch_off = mm_vals;
    }
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
    else HIPCHK(e, launch_minmax(e->logmel, e->mm, e->mm + e->mm_scratch_off, B, mel_planes, TF, NM, st));
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
    FusedTurn turn(e, (hipStream_t)stream);
    if (turn.rc) return turn.rc;
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
    FusedTurn turn(e, st);
    if (turn.rc) return turn.rc;
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
    FusedTurn turn(e, (hipStream_t)stream);
    if (turn.rc) return turn.rc;
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
    if (e->tuning_epoch != tuning_epoch().load()) {      // a tune.* knob changed (any engine, any thread): the cached chain is stale
        if (e->gexec && e->graph_stream_set) HIPCHK(e, hipStreamSynchronize(e->graph_stream));
        drop_graph(e);
        e->tuning_epoch = tuning_epoch().load();
    }
    if (e->opt_stack) {
        if (shared_with_another_process(e) == 1 && (rc = yield_fused(e, "another process is computing"))) return rc;
    } else if (e->yielded_from) {
        // a yield is a precaution, not a verdict: two looks in a row (>= 250 ms apart, in front of later chains) that find
        // the GPU exclusive again switch the fused launches back on
        const int shared = shared_with_another_process(e);
        if (shared == 1) e->yield_clean = 0;
        else if (shared == 0 && ++e->yield_clean >= 2) {
            if (e->gexec && e->graph_stream_set) HIPCHK(e, hipStreamSynchronize(e->graph_stream));
            drop_graph(e);
            e->opt_stack = e->yielded_from;
            e->yielded_from = 0;
            e->yield_clean = 0;
            e->stack_rearms += 1;
            if (e->stack_rearms <= 3)
                fprintf(stderr, "[diffroll_amd] engine %p: device %d is this process's own again - fused launches back on\n", (void*)e, e->cfg.device);
        }
    }
    FusedTurn turn(e, st);      // (released - event recorded on `st` - when this call returns, behind the chain's launches)
    if (turn.rc) return turn.rc;
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
    for (int attempt = 0; attempt < 2 && (!e->gexec || !(key == e->gkey)); ++attempt) {
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
        e->tuning_epoch = tuning_epoch().load();
        // capture + instantiation took tens of milliseconds: look again before a chain of persistent launches goes out
        if (attempt == 0 && e->opt_stack && turn.held) {
            if (shared_with_another_process(e, true) == 1) {
                if ((rc = yield_fused(e, "another process is computing"))) return rc;      // (drops the graph: captured again, per phase)
            }
        }
    }
    Range range("dr_sample: launch the chain graph");
    // the graph owns no caller address: x_T is copied in, the finished roll copied out (0.7 MB each way)
    HIPCHK(e, hipMemcpyAsync(e->xwork, d_x, per * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(e, launch_set_dyn(e->d_dyn, seed, first_sample, w, (float)(1.0 + (double)w), st));
    HIPCHK(e, hipGraphLaunch(e->gexec, st));
    e->graph_stream = st; e->graph_stream_set = true;
    if (e->stack_launches || e->tail_launches) { e->unverified = true; e->fused_stream = st; }      // (the captured chain may hold persistent launches)
    HIPCHK(e, hipMemcpyAsync(d_x, e->xwork, per * sizeof(float), hipMemcpyDeviceToDevice, st));
    return DR_OK;
}

int dr_finish(dr_engine* e, void* stream) {
    if (!e) return DR_EINVAL;
    DeviceGuard guard(e->cfg.device);
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    if (e->unverified && e->fused_stream != (hipStream_t)stream) HIPCHK(e, hipStreamSynchronize(e->fused_stream));
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
    else if (e->yielded_from) e->healed_from = e->yielded_from;
    e->clean_chains = 0;
    e->opt_stack = 0;
    e->yielded_from = 0;                                  // (a time-out outranks a pending yield: only fused_rearm re-arms now)
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
        e->stack_rearms += 1;
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
        // the flag is final once the stream the fused launches ran on has drained - which need not be the stream the
        // consumer passes (a roll sampled on one stream, scored on another)
        DeviceGuard guard(e->cfg.device);
        HIPCHK(e, hipStreamSynchronize(e->fused_stream));
        if ((hipStream_t)stream != e->fused_stream) HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    }
    if (*e->stack_err_host)
        return fail(e, DR_ETIMEOUT, "a fused residual-stack launch issued on this engine timed out and has not been checked: the roll is "
                                    "invalid - call dr_finish (clears the condition, switches to per-phase launches) and recompute, or "
                                    "use dr_sample_checked");
    e->unverified = false;
    return DR_OK;
}

int dr_launch_state(dr_engine* e, dr_launch_info* out) {
    if (!e || !out) return DR_EINVAL;
    out->mode = e->last_mode;
    out->fused_enabled = e->opt_stack;
    out->fallbacks = e->stack_fallbacks;
    out->yields = e->stack_yields;
    out->rearms = e->stack_rearms;
    out->stack_launches = e->stack_launches;
    out->tail_launches = e->tail_launches;
    return DR_OK;
}

int dr_note_runs(dr_engine* e, const float* d_roll, int B, int T, float threshold, int32_t* d_note_end, void* stream) {
    if (!e || !d_roll || !d_note_end) return fail(e, DR_EINVAL, "null argument");
    if (B <= 0 || T <= 0) return fail(e, DR_EINVAL, "bad shape B=%d T=%d", B, T);
    DeviceGuard guard(e->cfg.device);
    if (int rc = dr_pending_timeout(e, stream)) return rc;
    HIPCHK(e, launch_note_runs(d_roll, d_note_end, B, T, threshold, (hipStream_t)stream));
    return DR_OK;
}

int dr_frame_counts(dr_engine* e, const float* d_pred, const float* d_label, size_t n, float threshold,
                    int64_t* host_counts, void* stream) {
    if (!e || !d_pred || !d_label || !host_counts) return fail(e, DR_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(e->cfg.device);
    if (int rc = dr_pending_timeout(e, stream)) return rc;
    if (!e->d_counts) {
        void* q = nullptr;
        HIPCHK(e, hipMalloc(&q, frame_counts_work_words() * sizeof(unsigned long long)));
        HIPCHK(e, hipMemset(q, 0, frame_counts_work_words() * sizeof(unsigned long long)));      // (the ticket word starts at zero)
        e->d_counts = (unsigned long long*)q;
    }
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

int dr_set_option(dr_engine* e, const char* name, int value) { return drh::set_option(e, name, value, false); }

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
        if (mode) {           // the mode changes only once its packings exist (520 MB of uploads: the build can fail)
            int rc = ensure_s3(e);
            if (rc) return rc;
        }
        e->prec = mode;
    }
    return DR_OK;
}

}  // extern "C"
