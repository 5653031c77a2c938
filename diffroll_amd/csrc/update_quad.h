// Philox4x32-10 + Box-Muller and the posterior update of one roll quad: shared by update_kernel (update.hip) and part T3
// of the tail kernel (tail.hip) - identical arithmetic.
#pragma once
#include "device_common.h"

namespace dr {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: z ~ N(0,1), keyed by (seed, global sample, step, element/4) so the
// noise of a sample does not depend on how the batch is sharded over GPUs.
// ---------------------------------------------------------------------------------------------
DR_DEVINL void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

DR_DEVINL void box_muller(uint32_t u0, uint32_t u1, float& z0, float& z1) {
    const float a = ((float)(u0 >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float bb = (float)(u1 >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float rad = sqrtf(-2.0f * logf(a));
    float sn, cs;
    sincosf(6.283185307179586f * bb, &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

// Classifier-free combine + x0-prediction posterior update of ONE float4 (4 consecutive elements, index i4) of the
// roll.  Same operation order as task/diffusion.py:953 and :957-967; contraction off so that no FMA is formed where
// the reference rounds twice.  Shared by update_kernel and the tail kernel (identical arithmetic).
DR_DEVINL float4 update_quad(const UpdateArgs& a, const long i4) {
#pragma clang fp contract(off)
    const float4 xc = reinterpret_cast<const float4*>(a.x0c)[i4];
    float x0[4] = {xc.x, xc.y, xc.z, xc.w};
    // per-call scalars: by value (eager launches) or from the device block (captured chain)
    const float gw = a.dyn ? a.dyn->w : a.w, g1pw = a.dyn ? a.dyn->onepw : a.onepw;
    const uint64_t seed = a.dyn ? a.dyn->seed : a.seed;
    const int first_sample = a.dyn ? a.dyn->first_sample : a.first_sample;
    if (a.x0u) {
        const float4 xu = reinterpret_cast<const float4*>(a.x0u)[i4];
        const float u[4] = {xu.x, xu.y, xu.z, xu.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) x0[e] = g1pw * x0[e] - gw * u[e];
    }
    const float c0 = a.coef[0], c1 = a.coef[1], c2 = a.coef[2], c3 = a.coef[3], c4 = a.coef[4];
    float o[4];
    // which updates draw noise at t > 0: x0 DDPM (0), eps ddpm (2), eps ddim2ddpm (4)
    const bool noisy = (a.mode == 0 || a.mode == 2 || a.mode == 4) && a.t > 0;
    float x[4] = {0.f, 0.f, 0.f, 0.f}, z[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.t > 0 || a.mode >= 2) {
        const float4 xv = reinterpret_cast<const float4*>(a.x)[i4];
        x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w;
    }
    if (noisy) {
        if (a.noise) {
            const float4 zv = reinterpret_cast<const float4*>(a.noise)[i4];
            z[0] = zv.x; z[1] = zv.y; z[2] = zv.z; z[3] = zv.w;
        } else {
            const long e0 = i4 * 4;
            const long smp = e0 / a.per_sample;
            const long within = (e0 - smp * a.per_sample) >> 2;
            uint32_t rnd[4];
            philox4x32_10((uint32_t)within, (uint32_t)(within >> 32), (uint32_t)a.t,
                          (uint32_t)(first_sample + smp), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
            box_muller(rnd[0], rnd[1], z[0], z[1]);
            box_muller(rnd[2], rnd[3], z[2], z[3]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float y = x0[e];   // network output: x0 prediction (modes 0/1) or epsilon (modes 2-4)
        if (a.mode <= 1) {
            // ddpm_x0 family :957-967 / ddim_x0 family :864-873 (c4 = sigma = 0, the 0*z term is dropped)
            if (a.t == 0) o[e] = y / c2;
            else {
                const float t1 = c0 * y;
                const float t2 = (c1 * (x[e] - c2 * y)) / c3;
                o[e] = (a.mode == 0) ? (t1 + t2) + c4 * z[e] : (t1 + t2);
            }
        } else if (a.mode == 2) {
            // ddpm :820-829: sqrt_recip_alphas_t * (x - betas_t * eps / sqrt_1m_acp_t) [+ sqrt(post_var_t) * z]
            const float m = c0 * (x[e] - (c1 * y) / c2);
            o[e] = (a.t == 0) ? m : m + c3 * z[e];
        } else {
            // ddim :885-890 / ddim2ddpm :902-909
            const float xe = (x[e] - c3 * y) / c2;
            if (a.t == 0) o[e] = xe;
            else if (a.mode == 3) o[e] = c0 * xe + c1 * y;
            else o[e] = (c0 * xe + c1 * y) + c4 * z[e];
        }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace dr
