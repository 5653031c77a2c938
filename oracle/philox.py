"""CPU replay of the engine's on-device noise (TEST INFRASTRUCTURE: only tests/ import this).

The reference draws z with torch.randn_like on whatever device x lives on (task/diffusion.py:967) - a stream no other
implementation can reproduce - so the engine's production noise is its own: Philox4x32-10 (Salmon et al., SC'11; the
generator behind curand / torch.cuda) keyed by (seed, GLOBAL sample index, step, element / 4) + Box-Muller, so that a
clip's noise does not depend on how the batch is sharded (diffroll_amd/csrc/update_quad.h).  This module restates that
stream in numpy so that a Philox-driven chain - what the command-line drivers run - can be held to the oracle.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counters / keys: uint32 arrays (broadcastable).  Returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def box_muller(u0, u1):
    """Two uint32 words -> two N(0, 1) float32 values, the device's expressions in float32."""
    a = ((u0 >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)     # (0, 1]
    b = (u1 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)                       # [0, 1)
    rad = np.sqrt(np.float32(-2.0) * np.log(a)).astype(np.float32)
    ang = (np.float32(6.283185307179586) * b).astype(np.float32)
    return (rad * np.cos(ang)).astype(np.float32), (rad * np.sin(ang)).astype(np.float32)


def step_noise(seed, first_sample, B, per_sample, t):
    """z of reverse step t for samples first_sample .. first_sample + B - 1: (B, per_sample) float32 (per_sample % 4 == 0)."""
    assert per_sample % 4 == 0
    within = np.arange(per_sample // 4, dtype=np.uint64)
    c0 = (within & np.uint64(0xFFFFFFFF)).astype(np.uint32)[None, :]
    c1 = (within >> np.uint64(32)).astype(np.uint32)[None, :]
    c3 = (np.uint32(first_sample) + np.arange(B, dtype=np.uint32))[:, None]
    r0, r1, r2, r3 = philox4x32_10(c0, c1, np.uint32(t), c3, np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    z0, z1 = box_muller(r0, r1)
    z2, z3 = box_muller(r2, r3)
    return np.stack([z0, z1, z2, z3], axis=-1).reshape(B, per_sample)


def chain_noise(seed, first_sample, steps, B, T):
    """The injected-noise tensor (steps, B, 1, T, 88) equivalent to a Philox chain with this seed: row t is the z of step t
    (row 0 is never used: task/diffusion.py:957-960 draws none at t = 0)."""
    import torch
    out = np.zeros((steps, B, 1, T, 88), dtype=np.float32)
    for t in range(1, steps):
        out[t] = step_noise(seed, first_sample, B, T * 88, t).reshape(B, 1, T, 88)
    return torch.from_numpy(out)
