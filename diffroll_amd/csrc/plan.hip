// Host side, part 2: planning - which tile flavour, split-K factor and fused-stack shape a launch gets - and the launch
// sequences of one network evaluation (run_network) and one reverse step (run_step).
#include "engine_state.h"

namespace drh {

// Frame-tile size (NI = 1: 64 frames, 2: 128 frames per block) for a GEMM of MT row tiles over NB samples
// of T frames: minimise (block rounds over the 256 CUs) x (tile cost); 128-frame tiles win ties (half the
// weight traffic per MFMA).  One block per CU is resident (LDS / 512-thread blocks).
// Frame-tile choice for a GEMM of MT row tiles over NB samples of T frames.  flavor 0: gemm_kernel
// (32x32 MFMA) with NI = n (64*n frames per block); flavor 1: gemm16_kernel (16x16 MFMA) with NJ = n
// (32*n frames per block, fp32 hot kernels only).  Cost = (block rounds over the 256 CUs, one block per
// CU) x (frames per block); 16x16 tiles carry a small penalty (more operand reads per MFMA), 128-frame
// 32x32 tiles win ties.
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
Tile pick_tile(int MT, int NB, int T, int taps, int dil, int prec, int epi, bool allow16, bool wide32) {
    const int forced = tuning().tile;                   // A/B experiments: 3202, 1605, ... (if it fits)
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
    if (forced) {
        const int ff = forced / 100 == 16 ? 1 : 0, fn = forced % 100;
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
int pick_ni(int MT, int NB, int T, int taps, int dil, int prec) {
    return pick_tile(MT, NB, T, taps, dil, prec, EPI_GATE, false).n;
}
hipError_t launch_tiled(const GemmArgs& a, int epi, Tile t, hipStream_t s, int prec) {
    if (t.flavor == 3) return launch_pointwise_ksplit(a, t.n, s);
    if (t.flavor == 2) return launch_pointwise(a, t.n, s);
    return t.flavor == 1 ? launch_gemm16(a, epi, t.n, s) : launch_gemm(a, epi, t.n, s, prec);
}
// tile of the 1x1 residual/skip GEMM: flavor 2 = operands direct from L2 (pw_kernel), fp32 only
Tile pick_pointwise_tile(int MT, int NB, int T, int prec, int kchunks) {      // kchunks: 32-channel slabs of K (default: MT tiles cover all rows)
    if (prec) return Tile{0, 1};
    const int pw = tuning().pw, pw_ni = tuning().pw_nw;      // A/B experiments: 0 = LDS-staged kernels / force 32*NW-frame blocks
    if (!pw) return pick_tile(MT, NB, T, 1, 1, 0, EPI_RES_SKIP, true);
    if (pw_ni) return Tile{2, pw_ni};
    // launches that cannot fill half the chip even with 64-frame blocks (single clips): 32-row x 32-frame tiles whose
    // four waves split K in-block (flavor 3, pwk_kernel: 256 blocks at config 1, 13.9 -> 8 us per launch); without it
    // (tune.pwk = 0, K splitting pinned off, a channel count that is not a multiple of 128) the LDS-staged kernel with
    // split-K through the workspace.  (Measured and rejected in round 3: 32-frame blocks of the direct kernel instead -
    // 64 blocks at config 1 - 35.2 vs 34.0 ms per chain.)
    if ((long)MT * NB * ((T + 63) / 64) <= 128) {
        if (tuning().pwk && tuning().ksplit_max > 1 && (kchunks ? kchunks : 2 * MT) % 4 == 0)
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

// One network evaluation for NB samples (first n_cond conditional) at step t.
//   xin (B,T,88) rows are used modulo bmod (classifier-free batching: 2B evaluations of B inputs).
int run_network(dr_engine* e, const float* xin, int bmod, int NB, int n_cond, int T, int t, float* x0_out,
                hipStream_t st, bool zero_spec, const int* tsel, TailPlan* tail) {
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
    const int stack3 = tuning().stack3;
    if (e->opt_stack && (prec == 0 || (stack3 && Cp % 128 == 0)) && L <= DR_STACK_MAX_LAYERS && e->n_cus > 0) {
        int maxdil = 1;
        for (int l = 0; l < L; ++l) maxdil = std::max(maxdil, e->layers[l].dil);
        // Flavours 1 / 2 / 5 (128 packed rows x 64 / 128 / 160 frames per block) are chosen automatically; tune.stack_fl = n
        // pins one (tests / measurements); tune.stack_fl = -5 excludes the 160-frame flavour (its A/B).  Flavour 5 exists in
        // exact fp32 with blocked accumulation only (its per-phase twin is gemm_kernel<5>, which has no other form).
        const int fl_force = tuning().stack_fl;
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
        for (int fl : {1, 2, 5}) {
            if (fl_force > 0 && fl != fl_force) continue;
            if (fl == 5 && (fl_force == -5 || prec != 0 || e->opt_blocked < 2)) continue;
            const int bn = stack_tile_frames(fl);
            const long gsize = stack_group_blocks(fl, Cp, T);                           // blocks per sample
            const long cap = std::min<long>(e->n_cus, 1024) / gsize;                    // samples per launch
            if (cap < 1 || (prec ? stack3_lds_bytes(fl, e->K, maxdil) : stack_lds_bytes(fl, e->K, maxdil)) > 160 * 1024) continue;
            const long chunks = (NB + cap - 1) / cap;
            if ((NB + chunks - 1) / chunks > dr_engine::STACK_GROUPS) continue;
            // (what fusing saves is per-launch overhead, which the per-phase launches amortise over their rounds:
            // measured +2.5 % at one round, +1.1 % at two (B = 32 guided clips per GPU), nothing at four)
            const double cost = (1.0 - 0.025 / chunks) * chunks * bn * (fl == 1 ? 1.0 / 0.93 : fl == 5 ? 1.04 : 1.0);
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
    e->last_mode = use_tail ? DR_MODE_FUSED_STACK_TAIL : (stack_from >= 0 ? DR_MODE_FUSED_STACK : DR_MODE_PER_PHASE);      // dr_launch_state
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
            e->unverified = true; e->fused_stream = st;
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
            if (dual) {
                a.NB = bmod; a.dual = bmod; a.nofold64 = (stack_ni == 2 && e->opt_blocked < 2);
                // (the single-chain 64-frame instance exists unsplit only, and it must be THE instance that runs - the tail
                // kernel's copy of this conv is what it has to agree with bit for bit: no split-K for this launch)
                if (a.nofold64) { a.ws = nullptr; a.ws_cnt = nullptr; }
            }
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
                ta.t4_ni = tuning().tail_t4;
                ta.g = e->g;
            }
        }
        ta.bar = e->tail_bar; ta.pbar = e->tail_pbar; ta.err = e->stack_err; ta.derr = e->stack_derr;
        // ticks 112..119 of dr_stack_status: the last tail launch of a chain that has a next step (all its parts run)
        ta.dbg = (e->stack_dbg_on && tail->next_t >= 0) ? e->stack_dbg + 112 : nullptr;
        HIPCHK(e, launch_tail(ta, st));
        e->tail_launches += 1;
        e->unverified = true; e->fused_stream = st;
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
int run_step(dr_engine* e, int sampler, float* x, const float* noise, int B, int T, int t, float w, uint64_t seed,
             int first_sample, hipStream_t st, float** result, ChainState* chain) {
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

void drop_graph(dr_engine* e) {
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    e->gkey = GraphKey{};
}

}  // namespace drh
