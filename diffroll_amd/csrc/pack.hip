// Host side, part 1: weight packing into the kernels' fragment order, staged uploads, hoisted tables - dr_set_param,
// dr_set_tables, dr_set_frontend_tables, dr_commit (and the split-bf16 packings built on first use).
#include "engine_state.h"

#include <mutex>

namespace drh {

// Host-side packing of the layers is embarrassingly parallel (one task per residual layer): dr_commit is on the
// critical path of a one-shot process (sampling.py: load checkpoint -> one batch), where it used to cost more than
// the whole 50-step chain of a single clip.  tuning().pack_threads caps the worker count (1 = serial).  A worker that
// throws (bad_alloc while packing) is reported to the caller instead of terminating the process.
template <class F>
int parallel_for(dr_engine* e, int n, F fn) {
    const int cap = tuning().pack_threads;
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = std::max(1, std::min(std::min(n, cap), hw > 0 ? hw : 1));
    std::atomic<bool> failed{false};
    auto guarded = [&](int i) {
        try { fn(i); } catch (...) { failed.store(true); }
    };
    if (nt == 1) {
        for (int i = 0; i < n && !failed.load(); ++i) guarded(i);
    } else {
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        try {
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&]() { for (int i = next.fetch_add(1); i < n && !failed.load(); i = next.fetch_add(1)) guarded(i); });
        } catch (...) { failed.store(true); }
        for (auto& t : th) t.join();
    }
    return failed.load() ? fail(e, DR_ENOMEM, "out of host memory while packing the weights") : DR_OK;
}

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

// One stager per device for the whole process (its stream and pinned buffers belong to the device current at creation),
// shared by every engine and host thread under a mutex, and released when the last engine on that device is destroyed
// (release_stager) - a thread-local one leaked 32 MiB of pinned memory per short-lived host thread.
std::mutex g_stager_mu;
std::map<int, Stager*> g_stagers;
struct StagerLock {
    std::unique_lock<std::mutex> lk;
    Stager* s;
    StagerLock() : lk(g_stager_mu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        Stager*& p = g_stagers[dev];
        if (!p) p = new Stager();
        s = p;
    }
};
void release_stager(int dev) {
    std::unique_lock<std::mutex> lk(g_stager_mu);
    auto it = g_stagers.find(dev);
    if (it == g_stagers.end()) return;
    Stager* s = it->second;
    g_stagers.erase(it);
    if (s->st) { (void)hipStreamSynchronize(s->st); (void)hipStreamDestroy(s->st); }
    for (int i = 0; i < 2; ++i) {
        if (s->pin[i]) (void)hipHostFree(s->pin[i]);
        if (s->done[i]) (void)hipEventDestroy(s->done[i]);
    }
    delete s;
}

int upload_bytes(dr_engine* e, const void* data, size_t bytes, float** out) {
    void* p = nullptr;
    HIPCHK(e, hipMalloc(&p, std::max<size_t>(bytes, 16)));
    e->owned.push_back(p);
    { StagerLock sl; HIPCHK(e, sl.s->copy(p, data, bytes)); }
    *out = (float*)p;
    return DR_OK;
}

int upload(dr_engine* e, const std::vector<float>& v, float** out) {
    return upload_bytes(e, v.data(), v.size() * sizeof(float), out);
}

int dev_alloc(dr_engine* e, float** p, size_t floats, bool zero) {
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

// The split-bf16 ("S3") packings of the two hot GEMMs (same row maps and zero padding as the fp32 ones): only the opt-in
// precision reads them, so they are built on first use - at dr_set_precision(BF16X3) after a commit, or at the end of a
// commit made in that mode - instead of costing every start-up 0.4 s of packing and 520 MB of uploads.
int ensure_s3(dr_engine* e) {
    if (e->s3_ready || !e->committed) return DR_OK;
    const int C = e->C, Cp = e->Cp, L = e->L, K = e->K;
    std::vector<std::vector<uint16_t>> c3(L), o3(L);
    int prc = parallel_for(e, L, [&](int l) {
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
    if (prc) return prc;
    for (int l = 0; l < L; ++l) {
        int rc;
        if ((rc = upload_bytes(e, c3[l].data(), c3[l].size() * 2, &e->layers[l].conv_w3)) ||
            (rc = upload_bytes(e, o3[l].data(), o3[l].size() * 2, &e->layers[l].out_w3)))
            return rc;
        c3[l] = {}; o3[l] = {};
    }
    { StagerLock sl; HIPCHK(e, sl.s->drain()); }
    e->s3_ready = true;
    return DR_OK;
}

int commit(dr_engine* e, hipStream_t st) {
    Range range("dr_commit: pack + upload weights, hoisted tables");
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
    rc = parallel_for(e, L, [&](int l) {
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
    if (rc) return rc;
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
    { StagerLock sl; HIPCHK(e, sl.s->drain()); }
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
    { StagerLock sl; HIPCHK(e, sl.s->drain()); }       // every staged constant has landed
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
        { StagerLock sl; HIPCHK(e, sl.s->drain()); }
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
            { StagerLock sl; HIPCHK(e, sl.s->drain()); }
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
    // (tune.s3_eager = 1: build the split-bf16 packings at every commit, as rounds 1-3 did - the "before" of the cold-start report)
    if (e->prec || tuning().s3_eager) return ensure_s3(e);
    return DR_OK;
}

}  // namespace drh
