"""Command-line drivers with the reference's command surface (sampling.py:22-73, infer.py:19-45).

    python sampling.py task=generation dataset.num_samples=8 dataloader.batch_size=4 model.args.kernel_size=9
    python sampling.py task=transcription dataset=Custom dataset.args.audio_path=my_audio dataset.args.audio_ext=wav \
        checkpoint_path=weights/Pretrain_MAESTRO-retrain_MAESTRO-k=9.ckpt
    python sampling.py task=inpainting task.inpainting_t=[500,650] dataset=Custom ...
    torchrun --nproc-per-node 8 sampling.py task=generation gpus=8 dataset.num_samples=128 dataloader.batch_size=128

Hydra is not a dependency: the same ``group=name`` / ``dotted.key=value`` override syntax is parsed here over
defaults that restate config/sampling.yaml, config/task/{generation,transcription,inpainting}.yaml,
config/model/ClassifierFreeDiffRoll.yaml and config/spec/mel.yaml.  Differences from the reference driver:
no Lightning Trainer / TensorBoard; rolls are written as ``rolls_batch<i>.npy`` plus ``raw_midi_<batch>_<i>.mid`` / ``clean_midi_e<batch>_<i>.mid``;
audio ingestion (utils/custom_dataset.py:55-91) reads .wav only (no mp3 codec in this image); resampling restates
torchaudio 0.11's windowed-sinc kernel (diffroll_amd/audio.py).
"""
from __future__ import annotations

import ast
import copy
import glob
import os
import sys
import time
import warnings
from typing import Any, Dict, List

import numpy as np
import torch

TASKS: Dict[str, Dict[str, Any]] = {
    # config/task/generation.yaml, inpainting.yaml (Custom variant: inpainting_t null), transcription.yaml
    "generation": dict(timesteps=200, beta_start=1e-4, beta_end=0.02, frame_threshold=0.5, generation_filter=0.02,
                       sampling=dict(type="generation_ddpm_x0"), inpainting_t=None, inpainting_f=None),
    "inpainting": dict(timesteps=200, beta_start=1e-4, beta_end=0.02, frame_threshold=0.5, generation_filter=0.02,
                       sampling=dict(type="inpainting_ddpm_x0", w=0.5), inpainting_t=[500, 650], inpainting_f=None),
    "transcription": dict(timesteps=200, beta_start=1e-4, beta_end=0.02, frame_threshold=0.5, generation_filter=0.02,
                          sampling=dict(type="cfdg_ddpm_x0", w=0.5), inpainting_t=None, inpainting_f=None),
}
DATASETS: Dict[str, Dict[str, Any]] = {
    "Sampling": dict(name="Sampling", num_samples=4, args=dict()),                     # noise only (generation)
    "Custom": dict(name="Custom", num_samples=4,                                         # config/dataset/Custom.yaml
                   args=dict(audio_path="my_audio", audio_ext="wav", max_segment_samples=327680, sample_rate=16000)),
    "Synthetic": dict(name="Synthetic", num_samples=4, args=dict(seed=0)),              # 0.1 * randn waveforms
}
DEFAULTS: Dict[str, Any] = dict(
    gpus=1, hop_length=512, sequence_length=327680, sampling_rate=16000, checkpoint_path=None, seed=0,
    output_dir="outputs", precision="f32",
    dataloader=dict(batch_size=4),
    model=dict(name="ClassifierFreeDiffRoll",
               args=dict(residual_channels=512, unconditional=False, condition="fixed", n_mels=229,
                         residual_layers=15, kernel_size=3, dilation_base=2, dilation_bound=4, spec_dropout=0.1,
                         norm_args=[0, 1, "imagewise"])),
    spec=dict(args=dict(sample_rate=16000, n_fft=2048, hop_length=512, n_mels=229, f_min=0, f_max=8000, center=True,
                        normalized=True, pad_mode="reflect")),
)


def _parse_value(text: str):
    if text.lower() in ("null", "none"):
        return None
    if text.lower() in ("true", "false"):
        return text.lower() == "true"
    try:
        return ast.literal_eval(text)
    except Exception:
        return text


def build_config(argv: List[str], default_task: str = "generation") -> Dict[str, Any]:
    cfg = copy.deepcopy(DEFAULTS)
    groups = {"task": default_task, "dataset": None}
    rest = []
    for arg in argv:
        if "=" not in arg:
            raise SystemExit(f"expected key=value, got '{arg}'")
        k, v = arg.split("=", 1)
        if k in groups:
            groups[k] = v
        else:
            rest.append((k, v))
    if groups["task"] not in TASKS:
        raise SystemExit(f"unknown task '{groups['task']}' (choose from {sorted(TASKS)})")
    cfg["task"] = copy.deepcopy(TASKS[groups["task"]])
    cfg["task"]["name"] = groups["task"]
    ds = groups["dataset"] or ("Sampling" if groups["task"] == "generation" else "Synthetic")
    if ds not in DATASETS:
        raise SystemExit(f"dataset '{ds}' is not available here (choose from {sorted(DATASETS)}); MAPS / MAESTRO "
                         "need the AudioLoader package and the datasets on disk")
    cfg["dataset"] = copy.deepcopy(DATASETS[ds])
    for k, v in rest:
        node = cfg
        parts = k.split(".")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = _parse_value(v)
    return cfg


def load_wav_folder(args: Dict[str, Any]) -> torch.Tensor:
    """utils/custom_dataset.py:55-91 over a folder: mono mix, torchaudio-0.11 windowed-sinc resampling, crop /
    zero-pad to max_segment_samples (diffroll_amd/audio.py)."""
    from .audio import ingest
    if str(args["audio_ext"]).lower() != "wav":
        raise SystemExit("only .wav can be decoded in this environment (no mp3/flac codec)")
    files = sorted(glob.glob(os.path.join(args["audio_path"], f"*.{args['audio_ext']}")))
    if not files:
        raise SystemExit(f"no *.{args['audio_ext']} files under {args['audio_path']}")
    return torch.stack([ingest(f, int(args["sample_rate"]), int(args["max_segment_samples"])) for f in files])


def make_model(cfg: Dict[str, Any], device):
    from . import ClassifierFreeDiffRoll
    task = cfg["task"]
    kwargs = dict(spec_args=cfg["spec"]["args"], timesteps=task["timesteps"], beta_start=task["beta_start"],
                  beta_end=task["beta_end"], frame_threshold=task["frame_threshold"],
                  generation_filter=task["generation_filter"], sampling=task["sampling"],
                  inpainting_t=task["inpainting_t"], inpainting_f=task["inpainting_f"], training={"mode": "x_0"})
    path = cfg["checkpoint_path"]
    if path:
        # a named checkpoint must exist (the reference fails in load_from_checkpoint, sampling.py:54): random
        # weights are only ever used when NO checkpoint is named
        if not os.path.exists(path):
            raise SystemExit(f"checkpoint_path '{path}' does not exist")
        # keyword overrides win over the checkpoint's hyper_parameters (sampling.py:54-65)
        m = ClassifierFreeDiffRoll.load_from_checkpoint(path, **{k: kwargs[k] for k in (
            "sampling", "frame_threshold", "generation_filter", "inpainting_t", "inpainting_f")})
    else:
        m = ClassifierFreeDiffRoll(**cfg["model"]["args"], **kwargs)
        torch.nn.init.normal_(m.output_projection.weight, 0.0, 0.02)
    m.precision = cfg.get("precision", "f32")
    return m.to(device)


def main(argv: List[str] = None, default_task: str = "generation") -> None:
    cfg = build_config(list(sys.argv[1:] if argv is None else argv), default_task)
    if not torch.cuda.is_available():
        raise SystemExit("sampling needs an MI355X: diffroll_amd has no CPU fallback")
    from . import launch
    gpus = int(cfg.get("gpus") or 1)
    if gpus > 1 and not launch.under_launcher():
        # `gpus=N` is the reference's one flag for N devices (Trainer(gpus=cfg.gpus), sampling.py:70): start the
        # N ranks here, one process per GPU
        if argv is not None:      # called as a function: there is no command line to re-execute
            raise SystemExit(f"gpus={gpus}: start the {gpus} ranks with torch.distributed.run (or run the driver script)")
        # (`python -m diffroll_amd.cli ...`: sys.argv[0] is this file, which cannot run as a plain script - relative
        # imports - so the ranks are started as the same module)
        main_mod = sys.modules.get("__main__")
        spec = getattr(main_mod, "__spec__", None)
        if spec is not None and spec.name and spec.name.startswith("diffroll_amd."):
            raise SystemExit(launch.spawn_ranks(gpus, spec.name, list(sys.argv[1:]), module=True))
        raise SystemExit(launch.spawn_ranks(gpus, os.path.abspath(sys.argv[0]), list(sys.argv[1:])))
    rank, world, local_rank = launch.rank_env()
    if launch.under_launcher():
        # Trainer(gpus=N) counts devices PER NODE (sampling.py:70): compare with the launcher's ranks on THIS node, and
        # only when gpus= was given - an explicitly launched job is authoritative over the default gpus=1 (ADVICE r3:
        # `torchrun --nproc-per-node 8 sampling.py ...` and multi-node jobs must not be rejected)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        explicit = any(a.split("=", 1)[0] == "gpus" for a in (sys.argv[1:] if argv is None else argv))
        if explicit and local_world != gpus:
            raise SystemExit(f"gpus={gpus} but the launcher started {local_world} rank(s) on this node (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = launch.init_process_group(device) if world > 1 else None
    from .distributed import sample_sharded

    S = int(cfg["dataset"]["num_samples"])
    hop = int(cfg["hop_length"])
    g = torch.Generator().manual_seed(int(cfg["seed"]))
    T = int(cfg["sequence_length"]) // hop                       # sampling.py:27 uses 640 = 327680 / 512
    name = cfg["dataset"]["name"]
    if name == "Custom":
        waveform = load_wav_folder(cfg["dataset"]["args"])
        S = min(S, waveform.shape[0]) if cfg["dataset"].get("num_samples") else waveform.shape[0]
        waveform = waveform[:S]
        T = waveform.shape[1] // hop
    elif name == "Synthetic":
        waveform = 0.1 * torch.randn(S, int(cfg["sequence_length"]), generator=g)
    else:                                                        # generation: waveform is ignored (sampling.py:45)
        waveform = torch.zeros(S, int(cfg["sequence_length"]))
    x = torch.randn(S, 1, T, 88, generator=g)                    # x_T drawn on the host like sampling.py:27
    bs = int(cfg["dataloader"]["batch_size"])
    if S < bs:
        warnings.warn(f"Batch size is larger than total number of audio clips. Forcing batch size to {S}")
        bs = S
    t_model = time.perf_counter()
    model = make_model(cfg, device)
    t_engine = time.perf_counter()
    eng = model.engine                       # dr_create + dr_commit (weight packing, uploads, hoisted tables)
    torch.cuda.synchronize()
    t_ready = time.perf_counter()
    os.makedirs(cfg["output_dir"], exist_ok=True)
    t0 = time.perf_counter()
    batch_s = []
    for bi, lo in enumerate(range(0, S, bs)):
        hi = min(lo + bs, S)
        tb = time.perf_counter()
        roll = sample_sharded(model, x[lo:hi], waveform[lo:hi], seed=int(cfg["seed"]) + bi)
        batch_s.append(time.perf_counter() - tb)
        if rank == 0:
            np.save(os.path.join(cfg["output_dir"], f"rolls_batch{bi}.npy"), roll.cpu().numpy())
            model.export_midi(roll, os.path.join(cfg["output_dir"], f"raw_midi_{bi}_"),       # names of predict_step
                              clean_prefix=os.path.join(cfg["output_dir"], f"clean_midi_e{bi}_"))
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.perf_counter() - t0
        print(f"{cfg['task']['name']}: {S} clips x {T} frames, {cfg['task']['timesteps']} steps, sampler "
              f"{cfg['task']['sampling']['type']}, {world} GPU(s): {dt:.2f} s ({S * T / dt:.1f} frames/s incl. load "
              f"and capture) -> {cfg['output_dir']}/")
        # the split a one-shot run pays once (sampling.py:53-73 is one process per batch of clips): see bench.py cold_start
        ct = eng.cold_times()
        steady = f", later batches {1e3 * min(batch_s[1:]):.1f} ms" if len(batch_s) > 1 else ""
        print(f"cold start: model (host) {t_engine - t_model:.2f} s, engine create + commit {t_ready - t_engine:.2f} s "
              f"(pack {ct[0]:.3f}, upload {ct[1]:.3f}, tables {ct[2]:.3f}), first batch {batch_s[0]:.3f} s "
              f"(graph capture + instantiate {ct[3]:.4f} s, {int(ct[4])} kernel nodes){steady}; "
              f"fused-kernel time-outs healed: {eng.fallbacks}")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
