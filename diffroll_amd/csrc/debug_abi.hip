// Host side, part 4: the entry points of include/diffroll_amd_debug.h - measurement, checker and test hooks (dr_profile_*,
// dr_bench_*, dr_debug_*, dr_stack_status, dr_cold_times): what bench.py's roofline pass, tools/ and the checker builds call -
// nothing on the sampling path.
#include "engine_state.h"
#include "tenants.h"

using namespace drh;

extern "C" {

int dr_debug_set_option(dr_engine* e, const char* name, int value) { return drh::set_option(e, name, value, true); }

int dr_debug_kfd_root(const char* kfd_root) {
    drh::set_kfd_root(kfd_root);
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

int dr_debug_tenants(const char* kfd_root, int pci_domain, int pci_bus, int pci_device, int64_t* out4) {
    if (!kfd_root || !out4) return DR_EINVAL;
    const long gid = kfd_gpu_id(kfd_root, pci_domain, pci_bus, pci_device);
    out4[0] = gid; out4[1] = out4[2] = out4[3] = 0;
    if (gid < 0) return DR_OK;
    const TenantScan t = scan_tenants(kfd_root, gid);
    out4[1] = t.holders; out4[2] = t.busy_cus; out4[3] = t.readable ? 1 : 0;
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

int dr_debug_ticks(dr_engine* e, int64_t* loop_ticks, int64_t* block_ticks) {
    if (!e || !e->dbg_ticks) return fail(e, DR_ESTATE, "no dr_bench_layer launch yet");
    DeviceGuard guard(e->cfg.device);
    long long h[16];
    HIPCHK(e, hipDeviceSynchronize());
    HIPCHK(e, hipMemcpy(h, e->dbg_ticks, sizeof h, hipMemcpyDeviceToHost));
    if (loop_ticks) *loop_ticks = h[0];
    if (block_ticks) *block_ticks = h[1];
    if (tuning().debug_chunks) {
        fprintf(stderr, "[dr] chunk-start ticks since block start:");
        for (int i = 2; i < 16; ++i) fprintf(stderr, " %lld", h[i]);
        fprintf(stderr, "\n");
    }
    return DR_OK;
}

}  // extern "C"
