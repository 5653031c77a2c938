"""One flag reaches N GPUs (the reference: ``Trainer(gpus=cfg.gpus)``, sampling.py:70, config/sampling.yaml:20-21).

The engine runs one process per GPU (``torch.distributed`` backend ``nccl`` = RCCL over xGMI).  A driver that is
started as a plain ``python <script> ... N`` re-executes itself under ``python -m torch.distributed.run`` with N
local ranks; a driver that already runs under a launcher (RANK / WORLD_SIZE in the environment) is left alone.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Tuple


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def under_launcher() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rank_env() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the launcher's environment (0, 1, 0 without one)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def visible_gpus() -> int:
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def share_gpu() -> bool:
    """Plumbing-test mode (bench.py --share-gpu / DR_BENCH_SHARE_GPU=1): every rank binds device 0 and the process
    group is gloo (RCCL cannot put two ranks on one device).  No hardware claim follows from such a run - it exists
    so that the N > 1 code path (spawn -> rendezvous -> barrier -> timed loop -> all_reduce -> gathers -> JSON line)
    executes before an 8-GPU node is available."""
    return os.environ.get("DR_BENCH_SHARE_GPU", "0") not in ("", "0")


def spawn_ranks(n: int, script: str, argv: List[str], port: Optional[int] = None, module: bool = False) -> int:
    """Run ``script argv`` (module=True: ``-m script argv``) as n local ranks under torch.distributed.run and return
    its exit code.  Fails with a device-count message (not a launcher hint) when fewer than n GPUs are visible
    (share_gpu(): one device is enough)."""
    have = visible_gpus()
    if have < (1 if share_gpu() else n):
        raise SystemExit(f"{os.path.basename(script)}: {n} GPUs requested but only {have} HIP device(s) visible "
                         f"on this node")
    env = dict(os.environ)
    # the host driver only supports dmabuf IPC: without this RCCL's intra-node transport fails in
    # hipIpcGetMemHandle
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DR_SELF_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    cmd += (["--module", script] if module else [script]) + list(argv)
    return subprocess.call(cmd, env=env)


def init_process_group(device, force_single: bool = False):
    """Join (or, with force_single, create a 1-rank) RCCL process group; returns the torch.distributed module or
    None when the process is a plain single-GPU run.  share_gpu(): a gloo group (collectives on host tensors)."""
    rank, world, _ = rank_env()
    if not under_launcher() and not force_single:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not under_launcher():
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if share_gpu():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return dist


def collective_device(dist, device):
    """Where the small tensors of host-side collectives (timings) live: the GPU under RCCL, the host under gloo."""
    import torch
    return device if dist is not None and dist.get_backend() == "nccl" else torch.device("cpu")


def dist_info(dist) -> dict:
    """What the JSON line says about the process group that actually ran."""
    import torch
    if dist is None:
        return {"ranks_seen": 1, "backend": None, "rccl_version": None, "launcher": "none (single process)"}
    try:
        v = torch.cuda.nccl.version()
        ver = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:       # noqa: BLE001 - report, never hide the run
        ver = f"unavailable ({type(e).__name__})"
    launcher = "torch.distributed.run (self-spawned by --gpus)" if os.environ.get("DR_SELF_SPAWNED") else (
        "torch.distributed.run" if under_launcher() else "single process, 1-rank group")
    info = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ver,
            "launcher": launcher}
    if share_gpu():
        info["share_gpu"] = True          # all ranks on device 0: a plumbing run, not a scaling point
    return info
