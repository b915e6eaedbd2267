"""One process per GPU without an external launcher.

`python bench.py --gpus N` (no torchrun) must still run N ranks.  The parent process here never imports torch, never
loads libsr_yolo2.so and never touches the GPU: it only starts N fresh children of the same command line with the
torch.distributed environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), relays rank 0's
stdout and returns the worst exit code.  No os.exec*: a process that has initialised the GPU must not be replaced,
and children are started before anything could initialise it.

The reference has no counterpart (its multi-GPU code is training-only pthreads inside one process,
src_yolo2/network_kernels.cu:279-376); this is the launcher side of DESIGN.md section 5.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # dmabuf IPC between the ranks' processes: the runtime switch is read by libhsa-runtime64 (the string is in the ROCm 7.2
    # library of this image), and this pool's host driver supports only dmabuf IPC -- the environment exports 0 already, here
    # and on the GPU boxes; inherited when set, never overridden (without it RCCL's hipIpcGetMemHandle fails between processes)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return env


def spawn_ranks(argv, world: int, timeout: float | None = None, env=None):
    """Run `argv` as ranks 0..world-1; returns (exit code, rank 0's stdout).  The other ranks' stdout is dropped,
    every rank's stderr goes to ours.  If one rank dies the others are terminated (they would wait in a collective)."""
    if world < 1:
        raise ValueError("world size %d" % world)
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, env),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = None if timeout is None else time.time() + timeout
    out0 = None
    rc = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [s for s in states if s not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(s == 0 for s in states):
                break
            if deadline is not None and time.time() > deadline:
                rc = 124
                break
            if states[0] is None:
                try:                                     # drain rank 0's pipe so it can never block on a full one
                    out0, _ = procs[0].communicate(timeout=0.5)
                except subprocess.TimeoutExpired:
                    pass
            else:
                time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    if out0 is None:
        try:
            out0, _ = procs[0].communicate(timeout=20)       # returns what earlier timed-out calls already collected
        except (subprocess.TimeoutExpired, ValueError):
            out0 = ""
    return rc, out0 or ""


def self_launch_if_needed(gpus: int, argv=None) -> None:
    """Called first thing in a benchmark's main(): with --gpus N > 1 and no WORLD_SIZE in the environment, become the
    launcher -- spawn N ranks of this same command, print rank 0's output, exit with their code."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if "torch" in sys.modules:
        sys.exit("launch: torch was imported before the ranks were started; the launcher must stay off the GPU")
    argv = [sys.executable] + list(sys.argv if argv is None else argv)
    rc, out = spawn_ranks(argv, gpus)
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(rc)
