"""
Multi-GPU: per-file data parallelism + ONE gather of the fixed-width metrics records.

IRs are independent, so the bundle shards embarrassingly: rank r analyses a contiguous block of FILES (both
channels of a file stay on one GPU) and no collective touches the data path.  The only exchange is a single
gather of (channels x METRICS_WIDTH) float64 records to rank 0 at the end -- over RCCL/xGMI when the ranks
hold GPUs (backend "nccl" is RCCL on ROCm), over gloo in the CPU tests.  Large arrays (spectrograms, EDCs)
are never gathered.  Because there are no cross-IR reductions the gathered records are bit-identical for any
world size.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None):
    """Initialise torch.distributed when WORLD_SIZE > 1.  Returns (rank, local_rank, world_size)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # IRA_DIST_BACKEND=gloo: rehearsal of the multi-rank host logic on a box with fewer GPUs than ranks (several
            # ranks then share a device; RCCL refuses that).  Never set in production runs.
            backend = os.environ.get("IRA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        # The gloo transport announces its connections on the process's STDOUT ("[Gloo] Rank 0 is connected to ..."), from
        # C++: it would precede the one JSON line a benchmark's rank 0 prints.  File descriptor 1 points at stderr while
        # the group is set up (Python's sys.stdout is flushed first and keeps its own descriptor number).
        saved = None
        if backend == "gloo":
            import sys
            try:
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
            except OSError:
                saved = None
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            if saved is not None:
                dist.barrier()                       # (the announcements are printed when the first collective connects)
        finally:
            if saved is not None:
                os.dup2(saved, 1)
                os.close(saved)
    return rank, local_rank, world


def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _core_order(cpus, sys_root: str = "/sys"):
    """The CPUs ordered so that the hardware threads of one core are neighbours (a slice then holds whole cores: two ranks
    never share a physical core through its SMT siblings)."""
    def first_sibling(c):
        try:
            return min(_parse_cpulist(open(os.path.join(sys_root, "devices", "system", "cpu", f"cpu{c}", "topology",
                                                         "thread_siblings_list")).read()))
        except (OSError, ValueError):
            return c
    return sorted(cpus, key=lambda c: (first_sibling(c), c))


def gpu_numa_cpus(sys_root: str = "/sys"):
    """[(pci address, numa node, [cpus])] for every AMD GPU of the host, in PCI bus order (the order HIP enumerates them in
    by default), read from sysfs: /sys/class/drm/card*/device/{vendor, numa_node, local_cpulist}.  Empty when the host
    shows none (containers without /sys/class/drm)."""
    import glob
    seen = {}
    for dev in glob.glob(os.path.join(sys_root, "class", "drm", "card[0-9]*", "device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip().lower() != "0x1002":
                continue
            addr = os.path.basename(os.path.realpath(dev))
            node = int(open(os.path.join(dev, "numa_node")).read().strip())
            cpus = _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
        except (OSError, ValueError):
            continue
        if cpus:
            seen[addr] = (addr, node, cpus)
    return [seen[k] for k in sorted(seen)]


def visible_gpus(gpus, env=None):
    """The sysfs GPU list as HIP will number it: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (comma-separated INDICES into the
    PCI-ordered list; schedulers and containers set them to restrict or reorder the devices a process sees) applied in that
    order -- ROCR filters first, HIP indexes what ROCR left.  Local rank r then maps to entry r.  A value that is not a plain
    index list (UUIDs: `GPU-...`) cannot be resolved from sysfs: returns [] and the caller falls back to the even split."""
    env = os.environ if env is None else env
    out = list(gpus)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        val = env.get(var)
        if val is None or val.strip() == "":
            continue
        try:
            idx = [int(v) for v in val.split(",") if v.strip() != ""]
        except ValueError:
            return []
        sel = []
        for i in idx:                       # the runtime stops at the first invalid index
            if i < 0 or i >= len(out):
                break
            sel.append(out[i])
        out = sel
    return out


def rank_cpu_affinity(local_rank: int, local_world: int, allowed=None, gpus=None):
    """
    The CPUs rank `local_rank` of `local_world` ranks on this host should run on (one process per
    GPU; eight ranks on one host share its PCIe root complexes and memory controllers): the cores of ITS GPU's NUMA node
    (sysfs), cut into disjoint, equal slices among the ranks whose GPUs sit on that node -- so that the pinned host
    batches a rank allocates are node-local and no two ranks' worker threads compete for a core.  When sysfs shows no
    GPU topology (or fewer GPUs than ranks) the allowed CPUs are dealt evenly in order.  The GPU list is the one HIP will
    enumerate: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES are applied to the sysfs list first (visible_gpus).
    Returns (cpus, how) with `how` a short description for the bench line (it names the GPU's PCI address, to be compared
    with the device's own pci_bus_id); single rank: (allowed, "unrestricted").
    """
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except AttributeError:
            allowed = list(range(os.cpu_count() or 1))
    allowed = sorted(allowed)
    if local_world <= 1:
        return allowed, "unrestricted (one rank)"
    gpus = visible_gpus(gpu_numa_cpus()) if gpus is None else gpus
    if len(gpus) >= local_world:
        # several ranks may rehearse on fewer GPUs (IRA_DIST_BACKEND=gloo): rank r uses GPU r mod #GPUs there too
        mine = gpus[local_rank % len(gpus)]
        peers = [r for r in range(local_world) if gpus[r % len(gpus)][1] == mine[1]]
        node_cpus = _core_order([c for c in mine[2] if c in set(allowed)])
        if len(node_cpus) >= len(peers):
            k, per = peers.index(local_rank), len(node_cpus) // len(peers)
            return sorted(node_cpus[k * per : (k + 1) * per]), f"numa node {mine[1]} of GPU {mine[0]}, slice {k + 1}/{len(peers)}"
    per = len(allowed) // local_world
    if per == 0:
        return allowed, "unrestricted (fewer CPUs than ranks)"
    allowed = _core_order(allowed)
    return sorted(allowed[local_rank * per : (local_rank + 1) * per]), f"even split of {len(allowed)} allowed CPUs (no GPU topology in sysfs)"


def apply_rank_cpu_affinity(local_rank: int, local_world: int):
    """Narrow this process to rank_cpu_affinity(...) -- call BEFORE importing torch / numpy thread pools / allocating pinned
    memory.  Returns (cpus, how, cpus allowed before)."""
    before = sorted(os.sched_getaffinity(0))
    cpus, how = rank_cpu_affinity(local_rank, local_world, before)
    if local_world > 1 and cpus and len(cpus) < len(before):
        os.sched_setaffinity(0, cpus)
    return cpus, how, before


def shard_files(num_files: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of file indices for `rank`: ceil(F/W) files per rank, last ranks may be short."""
    per = -(-int(num_files) // int(world)) if world > 0 else int(num_files)
    lo = min(num_files, rank * per)
    hi = min(num_files, lo + per)
    return lo, hi


def gather_metrics(local: np.ndarray, device=None, stream=None, equal_rows: bool = False) -> Optional[np.ndarray]:
    """
    Gather per-channel records (n_local, width) float64 from every rank to rank 0, in rank order (= file order
    under shard_files).  Returns the concatenated array on rank 0 and None elsewhere.  Single process: identity.
    `stream` (RCCL only): enqueue the collective relative to THIS stream instead of the current one, so that a
    gather of step k does not queue behind the kernels of step k+1 that are already on the compute stream.
    `equal_rows`: every rank holds the same number of rows (fixed shard sizes): the row-count exchange and its host
    synchronisation are skipped -- the whole exchange is ONE collective.
    """
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    if stream is not None and dist.get_backend() == "nccl":
        with torch.cuda.stream(stream):
            return _gather_metrics(local, device, equal_rows)
    return _gather_metrics(local, device, equal_rows)


def _gather_metrics(local: np.ndarray, device=None, equal_rows: bool = False) -> Optional[np.ndarray]:
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    dev = (device or torch.device("cuda", torch.cuda.current_device())) if on_gpu else torch.device("cpu")
    width = int(local.shape[1])
    if equal_rows:
        all_counts = [int(local.shape[0])] * world
    else:
        # 1) row counts (so ragged shards are handled), 2) one padded gather of the records
        counts = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        all_counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(all_counts, counts)
        all_counts = [int(c.item()) for c in all_counts]
    cap = max(all_counts) if all_counts else 0
    send = torch.zeros((cap, width), dtype=torch.float64, device=dev)
    if local.shape[0]:
        send[: local.shape[0]] = torch.from_numpy(local).to(dev)
    recv = [torch.empty((cap, width), dtype=torch.float64, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    return np.concatenate([r[:c].cpu().numpy() for r, c in zip(recv, all_counts)], axis=0)


def any_rank_true(flag: bool) -> bool:
    """Logical OR of a per-rank flag over all ranks (also the barrier of the bundle runner: every rank reaches it, the
    failing ones included).  Single process: the flag itself."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def balanced_assignment(sizes, world: int):
    """
    Ragged bundles: sort the files by size, largest first, and deal them round-robin, so that
    every rank gets the same number of files (+-1) and nearly the same number of samples.  Returns one ascending index
    array per rank (files keep their bundle order inside a rank); deterministic (stable sort), a partition of
    range(len(sizes)).  Equal sizes degenerate to an interleaved deal; callers keep shard_files' contiguous blocks then.
    """
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    return [np.sort(order[r::world]) for r in range(int(world))]


def restore_order(assignment) -> np.ndarray:
    """Row permutation that puts records gathered in rank order (assignment[0] rows, then assignment[1] ...) back into
    file order: gathered[restore_order(assignment)] is in bundle order."""
    flat = np.concatenate([np.asarray(a, dtype=np.int64) for a in assignment]) if len(assignment) else np.zeros(0, np.int64)
    return np.argsort(flat, kind="stable")


def barrier() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    on_gpu = dist.get_backend() == "nccl"
    dev = (device or torch.device("cuda", torch.cuda.current_device())) if on_gpu else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
