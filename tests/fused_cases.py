"""Child process of tests/test_gpu_fused.py: runs every case of one block flavour with the per-phase kernels
pinned (by the tune.* options the parent passes in DR_TEST_TUNE, tools/tuning_env.py) to the SAME kernel flavours the fused kernel
is built from - 32x32 MFMA conv tiles of that width, no split-K, the direct-from-L2 1x1 - so that fused and
per-phase results must agree bit for bit.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle import diffroll_ref as R  # noqa: E402
from test_gpu_parity import make_model  # noqa: E402
from tools import tuning_env  # noqa: E402

tuning_env.install()        # the parent pins the kernel flavours of this process (DR_TEST_TUNE)

CASES = {
    1: [  # C, layers, k, B, T, sampler            (64-frame blocks)
        (64, 3, 9, 3, 40, "cfdg_ddpm_x0"),          # one M tile holding residual AND skip rows; dual first layer
        (64, 2, 3, 8, 65, "generation_ddpm_x0"),    # 8 groups (group-per-XCD mapping), 2 frame tiles per clip
        (128, 4, 9, 2, 125, "cfdg_ddpm_x0"),
        (128, 3, 15, 5, 129, "ddpm_x0"),            # halo 56 frames across the 3 tiles of a clip
        (192, 3, 9, 4, 200, "cfdg_ddpm_x0"),        # 3 M tiles: tile 1 straddles the residual / skip halves
        (192, 2, 9, 8, 1, "generation_ddpm_x0"),    # single frame
        (512, 2, 9, 8, 128, "generation_ddpm_x0"),  # full width, two 64-frame tiles per clip
        (512, 2, 9, 3, 300, "ddpm_x0"),             # full width, 5 tiles per clip (group of 40 blocks)
        (512, 3, 15, 16, 64, "cfdg_ddpm_x0"),       # 32 evaluations x 8 M tiles = 256 blocks
        (512, 2, 9, 40, 125, "ddpm_x0"),            # 40 evaluations x 16 blocks: three fused launches of 14 / 13 / 13 samples
        (512, 2, 9, 20, 125, "cfdg_ddpm_x0"),       # ... the middle chunk holds conditional AND unconditional samples
    ],
    2: [  # 128-frame blocks (chosen when 64-frame blocks would not fit the chip in one round)
        (512, 3, 9, 16, 125, "cfdg_ddpm_x0"),       # the bench geometry: 32 evaluations, one tile per clip
        (512, 2, 9, 8, 250, "cfdg_ddpm_x0"),        # 16 evaluations x 2 tiles x 8 M tiles = 256 blocks, halo exchange
        (512, 2, 15, 16, 200, "generation_ddpm_x0"),
        (384, 2, 9, 20, 129, "ddpm_x0"),            # 6 M tiles (residual / skip halves split inside no tile), ragged 2nd tile
        (512, 2, 9, 32, 125, "cfdg_ddpm_x0"),       # 64 evaluations: two fused launches (all conditional / all unconditional)
    ],
    5: [  # 160-frame blocks (the 640-frame geometries: 4 tiles per clip; blocked accumulation only)
        (512, 3, 9, 4, 640, "cfdg_ddpm_x0"),        # the reference's shipping geometry: 8 evaluations x 4 tiles x 8 M tiles = 256 blocks
        (512, 2, 15, 4, 640, "cfdg_ddpm_x0"),       # BASELINE config 5 per GPU: k = 15, halo 56 frames across the 4 tiles
        (512, 2, 9, 8, 320, "generation_ddpm_x0"),  # 2 tiles per clip, 8 groups of 16 blocks
        (128, 3, 15, 5, 161, "ddpm_x0"),            # ragged: the second tile holds ONE frame
        (192, 3, 9, 4, 200, "cfdg_ddpm_x0"),        # 3 M tiles: tile 1 straddles the residual / skip halves; 40-frame second tile
        (64, 2, 3, 8, 100, "generation_ddpm_x0"),   # one part-filled tile per clip (frames 100..159 do not exist)
        (384, 2, 9, 3, 500, "ddpm_x0"),             # 6 M tiles x 4 tiles; the last tile holds 20 frames
        (512, 2, 9, 16, 640, "generation_ddpm_x0"), # 16 evaluations x 32 blocks: two fused launches of 8 samples
    ],
}
# the split-bf16 flavours of the fused kernel (precision="bf16x3"; channel counts that are multiples of 128)
CASES_S3 = {
    1: [(128, 4, 9, 2, 125, "cfdg_ddpm_x0"), (128, 3, 15, 5, 129, "ddpm_x0"), (512, 2, 9, 8, 128, "generation_ddpm_x0"),
        (384, 2, 9, 3, 300, "ddpm_x0"), (512, 3, 15, 16, 64, "cfdg_ddpm_x0"), (512, 2, 9, 20, 125, "cfdg_ddpm_x0")],
    2: [(512, 3, 9, 16, 125, "cfdg_ddpm_x0"), (512, 2, 9, 8, 250, "cfdg_ddpm_x0"), (512, 2, 15, 16, 200, "generation_ddpm_x0"),
        (384, 2, 9, 20, 129, "ddpm_x0"), (128, 2, 9, 6, 65, "cfdg_ddpm_x0")],
}


def main():
    prec = "bf16x3" if len(sys.argv) > 2 and sys.argv[2] == "bf16x3" else "f32"
    ni = int(sys.argv[1])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = []
    for (C, layers, k, B, Tn, sampler) in (CASES_S3 if prec == "bf16x3" else CASES)[ni]:
        hp = dict(R.DEFAULT_HP)
        hp.update(residual_channels=C, residual_layers=layers, kernel_size=k, timesteps=6)
        p = R.synthetic_params(hp, seed=C + k)
        m = make_model(hp, p, sampler=sampler, w=0.5, precision=prec)
        g = torch.Generator().manual_seed(B * 1000 + Tn)
        wav = 0.1 * torch.randn(B, max(Tn * 512, 2048), generator=g)
        x = torch.randn(B, 1, Tn, 88, generator=g)
        z = torch.randn(B, 1, Tn, 88, generator=g)
        eng = m.engine
        eng.set_option("fused_stack", 0)
        ref = m.reverse_diffusion(x, wav, 3, noise=z)[0]
        rec = {"case": [C, layers, k, B, Tn, sampler], "runs": []}
        for xcd in (1, 0):
            eng.set_option("fused_stack", 2)
            eng.set_option("fused_stack_xcd", xcd)
            eng.stack_status()
            n0 = eng.stack_launches
            eng.profile_enable(True)
            got = m.reverse_diffusion(x, wav, 3, noise=z)[0]
            _, _, _, kname = eng.profile_read_ex()
            eng.profile_enable(False)
            flag, _ = eng.stack_status()
            rec["runs"].append({"xcd": xcd, "timed_out": flag, "launches": eng.stack_launches - n0,
                                "kernel": kname.split(" ")[0], "equal": bool(torch.equal(got, ref)),
                                "maxdiff": float((got - ref).abs().max())})
        sch = R.schedule(hp["beta_start"], hp["beta_end"], hp["timesteps"])
        with torch.no_grad():
            spec = None if sampler == "generation_ddpm_x0" else R.frontend(wav, hp, Tn)
            want = R.reverse_step(p, hp, sch, sampler, x, spec, 3, z, 0.5)
        rec["vs_oracle"] = float((ref.cpu() - want).abs().max())
        # the whole 6-step chain (captured graph): with the fused step every tail kernel also computes the next
        # step's input projection and the roll ping-pongs between two buffers - bit-identical to per-phase launches,
        # with the tail fusion on and off
        nz = torch.randn(hp["timesteps"], B, 1, Tn, 88, generator=g)
        eng.set_option("fused_stack", 0)
        chain_ref = m.sample(x, wav, noise=nz)[0]
        for tailopt in (1, 0):
            eng.set_option("fused_stack", 2)
            eng.set_option("fused_stack_xcd", 1)
            eng.set_option("fused_tail", tailopt)
            t0 = eng.tail_launches
            got = m.sample(x, wav, noise=nz)[0]
            flag, _ = eng.stack_status()
            rec["runs"].append({"xcd": 1, "chain": True, "tail": tailopt, "tail_launches": eng.tail_launches - t0,
                                "timed_out": flag, "launches": 1, "kernel": rec["runs"][0]["kernel"],
                                "equal": bool(torch.equal(got, chain_ref)), "maxdiff": float((got - chain_ref).abs().max())})
        eng.set_option("fused_tail", 1)
        out.append(rec)
        del m
    if ni == 1 and prec == "f32":
        # the remaining conditioner variants of the gate epilogue inside the fused kernel: the learned unconditional
        # spectrogram of condition='trainable_spec' (cond2, model/diffwave.py:656-658) and the spec == 0 branch of
        # cfdg_ddim_x0 (task/diffusion.py:1027-1055)
        from diffroll_amd import ClassifierFreeDiffRoll
        hp = dict(R.DEFAULT_HP)
        hp.update(residual_channels=128, residual_layers=3, kernel_size=9, timesteps=6)
        p = R.synthetic_params(hp, seed=9)
        for label, cond, sampler in (("trainable_spec", "trainable_spec", "cfdg_ddpm_x0"), ("zero_spec", "fixed", "cfdg_ddim_x0")):
            m = ClassifierFreeDiffRoll(
                residual_channels=128, unconditional=False, condition=cond, n_mels=229, norm_args=[0, 1, "imagewise"],
                residual_layers=3, kernel_size=9, dilation_base=2, dilation_bound=4,
                spec_args=dict(sample_rate=16000, n_fft=2048, hop_length=512, n_mels=229, f_min=0, f_max=8000,
                               center=True, normalized=True, pad_mode="reflect"),
                timesteps=6, training={"mode": "x_0"}, sampling={"type": sampler, "w": 0.5})
            sd = dict(p)
            if cond == "trainable_spec":
                sd["trainable_parameters"] = torch.rand(229, 641, generator=torch.Generator().manual_seed(2)) * 2 - 1
            m.load_state_dict(sd)
            g = torch.Generator().manual_seed(31)
            B, Tn = 4, 100
            wav = 0.1 * torch.randn(B, Tn * 512, generator=g)
            x = torch.randn(B, 1, Tn, 88, generator=g)
            z = torch.randn(B, 1, Tn, 88, generator=g)
            eng = m.engine
            eng.set_option("fused_stack", 0)
            ref = m.reverse_diffusion(x, wav, 3, noise=z)[0]
            eng.set_option("fused_stack", 2)
            eng.stack_status()
            n0 = eng.stack_launches
            got = m.reverse_diffusion(x, wav, 3, noise=z)[0]
            flag, _ = eng.stack_status()
            rec = {"case": [label], "vs_oracle": 0.0,
                   "runs": [{"xcd": 1, "timed_out": flag, "launches": eng.stack_launches - n0, "kernel": "stack_kernel<1>",
                             "equal": bool(torch.equal(got, ref)), "maxdiff": float((got - ref).abs().max())}]}
            # ... and as a whole chain: the tail kernel's next-step first-layer conv (T4) with the learned unconditional
            # conditioner / the spec == 0 bias, against per-phase launches (bitwise) and the oracle's loop
            nz = torch.randn(hp["timesteps"], B, 1, Tn, 88, generator=g)
            eng.set_option("fused_stack", 0)
            chain_ref = m.sample(x, wav, noise=nz)[0]
            eng.set_option("fused_stack", 2)
            t0 = eng.tail_launches
            chain = m.sample(x, wav, noise=nz)[0]
            flag, _ = eng.stack_status()
            hpc = dict(hp, condition=cond)
            with torch.no_grad():
                want = R.sample_chain(sd, hpc, sampler, x, wav, nz, w=0.5)
            rec["vs_oracle"] = float((chain.cpu() - want).abs().max())
            rec["runs"].append({"xcd": 1, "chain": True, "tail_launches": eng.tail_launches - t0, "timed_out": flag, "launches": 1,
                                "kernel": "stack_kernel<1>", "equal": bool(torch.equal(chain, chain_ref)),
                                "maxdiff": float((chain - chain_ref).abs().max())})
            out.append(rec)
    from diffroll_amd import _cabi
    v = _cabi.bounds_violations()          # checker builds (tools/checked_build.sh): this process's own record
    if v is not None:
        out.append({"case": ["DR_BOUNDS"], "vs_oracle": 0.0, "bounds": list(v),
                    "runs": [{"xcd": 1, "timed_out": 0, "launches": 1, "kernel": f"stack_kernel<{ni}>", "equal": v[3] == 0, "maxdiff": float(v[3])}]})
    print("FUSED_CASES " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
