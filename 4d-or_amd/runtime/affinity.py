"""Host-thread placement for one-process-per-GPU jobs.

The steps on this path are host-sensitive: the one-scan scene-graph step is bound by ONE python thread enqueueing
~470 launches (110 vs 147 scans/s on two boxes of the same pool, profiles/r04_other_workloads.jsonl), and under
`torch.distributed.run` eight such threads plus RCCL's proxy threads share the host.  An MI355X node has its GPUs
spread over the sockets' NUMA nodes; a rank whose enqueueing thread runs on the far socket pays the cross-socket hop
on every doorbell write and every pinned-memory access.  `pin_to_gpu_numa` restricts the calling process to the cores
of the NUMA node its GPU hangs off (or, where the platform reports none — single-socket boxes, VMs: `numa_node` = -1 —
to an even slice of the visible cores per local rank, so that ranks at least do not migrate over each other).

Nothing here touches the device; everything is read from sysfs.  Reference behaviour replaced: none — the reference
trains on one GPU from one process (scene_graph_prediction/main.py:54-66) and leaves placement to the OS.
"""
import os
from typing import Iterable, List, Optional


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_pci_address(device_index: int) -> Optional[str]:
    """'dddd:bb:dd.f' of the visible device `device_index` (torch's device properties; None without a GPU)."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        p = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or dev is None:
            return None
        return f"{int(dom or 0):04x}:{int(bus):02x}:{int(dev):02x}.0"
    except Exception:            # placement is best effort: never fail a job over it
        return None


def numa_node_of_pci(address: str, sysfs: str = "/sys") -> int:
    """NUMA node of a PCI function (-1: the platform reports none)."""
    txt = _read(os.path.join(sysfs, "bus", "pci", "devices", address, "numa_node"))
    try:
        return int(txt) if txt is not None else -1
    except ValueError:
        return -1


def cpus_of_node(node: int, sysfs: str = "/sys") -> List[int]:
    txt = _read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist"))
    return parse_cpulist(txt) if txt else []


def even_slice(cpus: Iterable[int], local_rank: int, local_world: int) -> List[int]:
    """Contiguous share of `cpus` for one of `local_world` ranks (at least one core each)."""
    cpus = sorted(cpus)
    local_world = max(1, int(local_world))
    if not cpus:
        return []
    per = max(1, len(cpus) // local_world)
    lo = (int(local_rank) % local_world) * per
    if lo >= len(cpus):
        lo = (int(local_rank) % len(cpus))
        return cpus[lo:lo + 1]
    hi = len(cpus) if local_rank % local_world == local_world - 1 and per * local_world <= len(cpus) else lo + per
    return cpus[lo:hi]


def plan_affinity(allowed: Iterable[int], node_cpus: Iterable[int], local_rank: int, local_world: int,
                  ranks_on_node: Optional[int] = None, index_on_node: Optional[int] = None) -> List[int]:
    """The cores a rank should run on.  `node_cpus`: cores of its GPU's NUMA node (empty: unknown).  Ranks that share a
    NUMA node split its cores evenly when the caller knows how many they are (`ranks_on_node`, `index_on_node`)."""
    allowed = sorted(set(allowed))
    near = [c for c in node_cpus if c in set(allowed)]
    if near:
        if ranks_on_node and ranks_on_node > 1 and index_on_node is not None:
            return even_slice(near, index_on_node, ranks_on_node) or near
        return near
    return even_slice(allowed, local_rank, local_world) if local_world > 1 else allowed


#: a process is never pinned to fewer cores than this: the DataLoader workers and RCCL's proxy threads inherit
#: the mask, and an even slice of a small host (len(cpus) // local_world) can be a single core
MIN_CORES = 4


def pin_to_gpu_numa(local_rank: int = 0, local_world: int = 1, device_index: Optional[int] = None,
                    sysfs: str = "/sys", min_cores: int = MIN_CORES) -> dict:
    """Restrict this process to the cores next to its GPU.  Returns what was done (for the job's log / bench JSON):
    {"numa_node": n, "cpus": k, "pinned": bool, "why": "..."}.  PN2_PIN_NUMA=0 switches it off."""
    info = {"numa_node": -1, "cpus": 0, "pinned": False, "why": ""}
    if os.environ.get("PN2_PIN_NUMA") == "0":
        info["why"] = "PN2_PIN_NUMA=0"
        return info
    if not hasattr(os, "sched_setaffinity"):
        info["why"] = "no sched_setaffinity on this platform"
        return info
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except OSError as e:
        info["why"] = f"sched_getaffinity: {e}"
        return info
    dev = local_rank if device_index is None else device_index
    addr = gpu_pci_address(dev)
    node = numa_node_of_pci(addr, sysfs) if addr else -1
    info["numa_node"] = node
    node_cpus = cpus_of_node(node, sysfs) if node >= 0 else []
    ranks_on_node = index_on_node = None
    if node >= 0 and local_world > 1:
        # ranks of this job whose GPUs share the node (device i <-> local rank i under torch.distributed.run)
        peers = [r for r in range(local_world) if numa_node_of_pci(gpu_pci_address(r) or "", sysfs) == node]
        if local_rank in peers:
            ranks_on_node, index_on_node = len(peers), peers.index(local_rank)
    target = plan_affinity(allowed, node_cpus, local_rank, local_world, ranks_on_node, index_on_node)
    if target and target != allowed and len(target) < min(int(min_cores), len(allowed)):
        info["cpus"] = len(allowed)
        info["why"] = f"slice of {len(target)} core(s) is below the minimum of {int(min_cores)}: affinity left as it is"
        return info
    if not target or target == allowed:
        info["cpus"] = len(allowed)
        info["why"] = "one rank, no NUMA information: affinity left as it is" if not node_cpus else "already on the node's cores"
        return info
    try:
        os.sched_setaffinity(0, target)
    except OSError as e:
        info["why"] = f"sched_setaffinity: {e}"
        return info
    info.update(cpus=len(target), pinned=True,
                why=("cores of the GPU's NUMA node" if node_cpus else "even slice of the visible cores (no NUMA information)"))
    return info
