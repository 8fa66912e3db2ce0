"""One process per GPU: the self-launcher behind `bench.py --gpus N` and `scripts/configs_sharded.py --gpus N`.

A script that is asked for N > 1 GPUs and finds no launcher environment (WORLD_SIZE unset) re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`; a script that IS under
a launcher checks that the launcher's world size is the N it was asked for and refuses anything else -- a `--gpus 8` run must never
silently measure one GPU.  No torch import here: the re-exec happens before the HIP runtime is initialised."""
import os
import socket
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_world():
    """(world, rank, local_rank) of the launcher environment, or None when the process was started plainly."""
    if "WORLD_SIZE" not in os.environ:
        return None
    return int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def ensure_ranks(n_gpus, script, argv):
    """Returns (world, rank, local_rank) with world == n_gpus -- re-executing `script argv` under torch.distributed.run first when
    n_gpus > 1 and no launcher started this process (does not return in that case).  A launcher world of another size is an error."""
    if n_gpus < 1:
        raise SystemExit(f"--gpus {n_gpus}: need at least one GPU")
    lw = launcher_world()
    if lw is None:
        if n_gpus == 1:
            return 1, 0, 0
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
        sys.stdout.flush()
        sys.stderr.flush()
        os.execv(sys.executable, cmd)      # does not return
    world, rank, local = lw
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a figure for a different "
                         f"number of GPUs than asked for (start {n_gpus} ranks, or pass --gpus {world})")
    return world, rank, local
