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
    """(world, rank, local_rank) of the launcher environment, or None when the process was started plainly.  WORLD_SIZE=1 with neither RANK nor
    MASTER_ADDR is NOT a launcher (schedulers and container images pre-export it): such a process self-launches like a plain one."""
    if "WORLD_SIZE" not in os.environ:
        return None
    if int(os.environ["WORLD_SIZE"]) == 1 and "RANK" not in os.environ and "MASTER_ADDR" not in os.environ:
        return None
    return int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def ensure_ranks(n_gpus, script, argv):
    """Returns (world, rank, local_rank) -- re-executing `script argv` under torch.distributed.run first when n_gpus > 1 and no launcher
    started this process (does not return in that case).  n_gpus = None (no --gpus on the command line): the launcher's world size is adopted
    (`torchrun --nproc-per-node 8 bench.py` measures 8 ranks), 1 without a launcher.  An EXPLICIT n_gpus that disagrees with the launcher's
    world size is an error: a `--gpus 8` command must never silently measure another number of GPUs."""
    lw = launcher_world()
    if n_gpus is None:
        return lw if lw is not None else (1, 0, 0)
    if n_gpus < 1:
        raise SystemExit(f"--gpus {n_gpus}: need at least one GPU")
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


# ---------------------------------------------------------------------------------------------------------------- CPU / NUMA binding
def parse_cpulist(text):
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def format_cpulist(cpus):
    cpus = sorted(cpus)
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def gpu_numa_node(pci_bus_id=None, ordinal=0, sysfs="/sys"):
    """NUMA node of a GPU: by its PCI address ("0000:c1:00.0", as torch.cuda.get_device_properties reports it) or, without one, of the
    ordinal-th AMD display device in PCI order (/sys/class/drm/card*/device/numa_node).  None when the kernel does not say (-1, or no sysfs)."""
    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    if pci_bus_id:
        v = read(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node"))
        if v is not None:
            return int(v) if int(v) >= 0 else None
    drm = os.path.join(sysfs, "class", "drm")
    try:
        cards = [c for c in os.listdir(drm) if c.startswith("card") and c[4:].isdigit()]
    except OSError:
        return None
    amd = []
    for c in cards:
        dev = os.path.join(drm, c, "device")
        if (read(os.path.join(dev, "vendor")) or "").lower() == "0x1002":
            amd.append((os.path.basename(os.path.realpath(dev)), dev))
    amd.sort()
    if ordinal >= len(amd):
        return None
    v = read(os.path.join(amd[ordinal][1], "numa_node"))
    return int(v) if v is not None and int(v) >= 0 else None


def rank_cpus(local_rank, n_local, node_of_rank, allowed, node_cpus, core_of=None):
    """The CPU set of local rank `local_rank` of `n_local`: the ranks whose GPU sits on the same NUMA node split that node's allowed CPUs into
    equal contiguous DISJOINT chunks (in rank order, along whole physical cores when core_of = {cpu: core id} is given: SMT siblings stay with
    one rank); a rank whose node is unknown -- or whose node has fewer allowed CPUs than ranks -- takes its chunk of an even split of the allowed
    CPUs no bound rank uses.  Pure function (tests/test_host_cpu.py): node_of_rank[r] = NUMA node or None, allowed = this process's affinity set,
    node_cpus = {node: [cpus]}."""
    aset = set(allowed)
    order = (lambda c: (core_of.get(c, c), c)) if core_of else (lambda c: c)

    def node_pool(nd):
        return sorted((c for c in node_cpus.get(nd, []) if c in aset), key=order)

    def peers_of(nd):
        return [r for r in range(n_local) if node_of_rank[r] == nd]

    def bound(r):     # does rank r get a chunk of its own node?
        nd = node_of_rank[r]
        return nd is not None and len(node_pool(nd)) >= len(peers_of(nd))

    def chunk(pool, peers, r):
        k, n = peers.index(r), len(peers)
        q, rem = divmod(len(pool), n)
        lo = k * q + min(k, rem)
        return pool[lo:lo + q + (1 if k < rem else 0)] or pool

    if bound(local_rank):
        nd = node_of_rank[local_rank]
        return sorted(chunk(node_pool(nd), peers_of(nd), local_rank))
    loose = [r for r in range(n_local) if not bound(r)]
    taken = set()
    for r in range(n_local):
        if bound(r):
            taken |= set(chunk(node_pool(node_of_rank[r]), peers_of(node_of_rank[r]), r))
    pool = sorted((c for c in aset if c not in taken), key=order) or sorted(aset, key=order)
    return sorted(chunk(pool, loose, local_rank))


def bind_rank(local_rank, n_local, pci_bus_ids=None, sysfs="/sys"):
    """Pin this process to the CPUs next to ITS GPU (one process per GPU: eight Python ranks each spend ~0.3 ms of host time per step enqueuing
    kernels; unpinned on a two-socket host they migrate across sockets and share cores).  pci_bus_ids[r] = PCI address of local rank r's GPU
    (optional).  Returns the CPU list this rank now runs on; a no-op (current affinity returned) where sched_setaffinity does not exist."""
    if not hasattr(os, "sched_getaffinity"):
        return []
    allowed = sorted(os.sched_getaffinity(0))
    if n_local <= 1:
        return allowed
    nodes = [gpu_numa_node(pci_bus_ids[r] if pci_bus_ids else None, r, sysfs) for r in range(n_local)]
    node_cpus = {}
    for nd in set(n for n in nodes if n is not None):
        try:
            with open(os.path.join(sysfs, "devices", "system", "node", f"node{nd}", "cpulist")) as f:
                node_cpus[nd] = parse_cpulist(f.read())
        except OSError:
            node_cpus[nd] = []
    core_of = {}
    for c in allowed:      # SMT siblings -> one core id (the lowest sibling)
        try:
            with open(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list")) as f:
                core_of[c] = min(parse_cpulist(f.read()))
        except (OSError, ValueError):
            core_of[c] = c
    cpus = rank_cpus(local_rank, n_local, nodes, allowed, node_cpus, core_of)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return allowed
    return cpus
