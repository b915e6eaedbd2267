"""The N>1 path of bench.py on the CPU with gloo, world size 2: frame sharding (disjoint,
deterministic per-rank image seeds) and replication of a packed arena by ONE broadcast from
rank 0.  No GPU and no compute: this covers the distributed plumbing only."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from sr_object_detection_amd import synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    batch, size = 4, 16
    # rank 0 owns the "arena" (packed weights); the others receive it with one broadcast
    arena = torch.empty(1 << 16, dtype=torch.uint8)
    if rank == 0:
        arena.copy_(torch.from_numpy((synth.splitmix64(7, 1 << 16) & np.uint64(255)).astype(np.uint8)))
    else:
        arena.zero_()
    dist.broadcast(arena, src=0)
    # this rank's frames, as bench.py seeds them
    x = synth.image_batch(batch, 3, size, size, seed=0xC0FFEE + rank * batch)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    print(json.dumps({"rank": rank, "arena_sum": int(arena.to(torch.int64).sum()), "x_sum": float(x.astype(np.float64).sum()),
                      "first": float(x[0, 0, 0, 0]), "tmax": float(t)}))
    dist.destroy_process_group()
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_share_weights_and_split_frames(tmp_path):
    import json
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [json.loads(p.communicate(timeout=180)[0].strip().splitlines()[-1]) for p in procs]
    assert all(p.returncode == 0 for p in procs)
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["arena_sum"] == outs[1]["arena_sum"] > 0          # the broadcast replicated rank 0's bytes
    assert outs[0]["x_sum"] != outs[1]["x_sum"]                      # disjoint frame shards
    assert outs[0]["tmax"] == outs[1]["tmax"] == 2.0                 # MAX over ranks, as bench.py times
    # global image index rank*batch+i uses seed 0xC0FFEE + index: rank 1's first frame == image 4 of a single run
    from sr_object_detection_amd import synth
    whole = synth.image_batch(8, 3, 16, 16)
    assert outs[1]["first"] == float(whole[4, 0, 0, 0])


LAUNCH_WORKER = textwrap.dedent("""
    import os, sys, json
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    t = torch.tensor([10.0 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(7)
    dist.barrier()
    print("noise from rank %d" % rank)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "tmax": float(t)}))
    dist.destroy_process_group()
""")


def test_self_launcher_spawns_ranks_and_relays_rank0(tmp_path):
    """bench.py --gpus N without torchrun goes through sr_object_detection_amd.launch: N fresh children with the
    torch.distributed environment, rank 0's stdout relayed (only rank 0's), worst exit code returned"""
    import json
    from sr_object_detection_amd import launch
    script = tmp_path / "lw.py"
    script.write_text(LAUNCH_WORKER)
    rc, out = launch.spawn_ranks([sys.executable, str(script)], 2, timeout=240)
    assert rc == 0
    lines = [l for l in out.strip().splitlines() if not l.startswith("[Gloo]")]      # gloo's own chatter on stdout
    assert lines[0] == "noise from rank 0" and len(lines) == 2 and "rank 1" not in out
    assert json.loads(lines[-1]) == {"n_gpus": 2, "tmax": 20.0}
    # a dying rank takes the job down with its code instead of leaving the others in a collective
    rc, _ = launch.spawn_ranks([sys.executable, str(script), "--fail-rank", "1"], 2, timeout=240)
    assert rc != 0          # rank 1's own code (7) or the error its peer got from the broken collective, whichever ended first


def test_bench_parent_never_imports_torch_or_the_library(tmp_path):
    """the launching parent must stay off the GPU: with --gpus 2 and no WORLD_SIZE, bench.main() hands over to the
    launcher before torch / libsr_yolo2.so are imported"""
    probe = tmp_path / "probe.py"
    probe.write_text(textwrap.dedent("""
        import sys, os
        sys.path.insert(0, %r)
        sys.argv = ["bench.py", "--gpus", "2"]
        os.environ.pop("WORLD_SIZE", None)
        from sr_object_detection_amd import launch
        seen = {}
        def fake_spawn(argv, world, **kw):
            seen["argv"], seen["world"] = argv, world
            seen["torch"] = "torch" in sys.modules
            seen["lib"] = any("libsr_yolo2" in (getattr(m, "__file__", "") or "") for m in sys.modules.values())
            import sr_object_detection_amd.darknet as dk
            seen["lib_loaded"] = dk._lib is not None
            return 0, '{"n_gpus": 2}' + chr(10)
        launch.spawn_ranks = fake_spawn
        import bench
        try:
            bench.main()
        except SystemExit as e:
            assert e.code == 0
        assert seen["world"] == 2 and seen["argv"][1:] == ["bench.py", "--gpus", "2"], seen
        assert not seen["torch"] and not seen["lib_loaded"], seen
        print("OK")
    """) % ROOT)
    p = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout + p.stderr
