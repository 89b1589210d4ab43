"""Per-rank CPU placement for the one-process-per-GPU runs (`bench.py --gpus N`, `image_main.py` shards,
`image_fine_tune_attack.py`): N ranks x (main thread + clip-lane threads + reader / writer threads) otherwise float over both
sockets of the host and over each other's cores.

`pin_rank()` is called BEFORE the process touches the GPU (no HIP call, no exec): it restricts the process -- and with it every
thread it starts later, which inherit the mask -- to its share of the cores it was allowed to begin with.  Ranks are dealt to
NUMA nodes in order (ranks 0..N/2-1 on node 0, the rest on node 1 of a two-socket host: the usual attachment of an 8-GPU
node's devices) and the node's cores are cut evenly among its ranks.  Pure host logic; the reference has no counterpart
(it runs one process per GPU by hand, `run_image_guided.py:36-37`).
"""
import glob
import os
import re
from typing import Dict, List, Optional, Sequence


def _parse_cpulist(text: str) -> List[int]:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _core_id(cpu: int, sysfs: str = "/sys/devices/system/cpu") -> int:
    """Physical core of a logical CPU (SMT siblings share it): a node's list `0-63,128-191` would otherwise hand rank 0 the
    cores 0-31 and rank 2 their hyper-threads 128-159."""
    try:
        with open(os.path.join(sysfs, f"cpu{cpu}", "topology", "core_id")) as fh:
            return int(fh.read())
    except (OSError, ValueError):
        return cpu


def numa_nodes(allowed: Sequence[int], sysfs: str = "/sys/devices/system/node") -> List[List[int]]:
    """The allowed cores grouped by NUMA node (node order); one group holding everything when sysfs has no node info."""
    allowed_set = set(allowed)
    nodes = []
    for path in sorted(glob.glob(os.path.join(sysfs, "node[0-9]*", "cpulist")), key=lambda p: int(re.findall(r"node(\d+)", p)[-1])):
        try:
            with open(path) as fh:
                cpus = [c for c in _parse_cpulist(fh.read()) if c in allowed_set]
        except OSError:
            continue
        if cpus:
            nodes.append(sorted(cpus, key=lambda c: (_core_id(c), c)))      # hardware threads of one core next to each other
    covered = {c for n in nodes for c in n}
    if not nodes or covered != allowed_set:
        return [sorted(allowed_set)]
    return nodes


def plan(local_rank: int, local_world: int, nodes: Sequence[Sequence[int]]) -> List[int]:
    """Cores of `local_rank` among `local_world` ranks: ranks are dealt to the nodes in contiguous blocks, a node's cores are
    cut evenly (in order) among the ranks it received.  Every rank gets at least one core; with more ranks than cores the
    ranks of a node share it whole."""
    nn = len(nodes)
    if local_world <= 0 or not (0 <= local_rank < local_world):
        raise ValueError(f"local_rank {local_rank} outside 0..{local_world - 1}")
    if local_world < nn:                       # fewer ranks than nodes: a rank takes a contiguous run of whole nodes
        lo, hi = local_rank * nn // local_world, (local_rank + 1) * nn // local_world
        return [c for n in nodes[lo:hi] for c in n]
    node = local_rank * nn // local_world
    ranks_here = [r for r in range(local_world) if r * nn // local_world == node]
    k, cores = ranks_here.index(local_rank), list(nodes[node])
    if len(cores) < len(ranks_here):
        return cores
    lo, hi = k * len(cores) // len(ranks_here), (k + 1) * len(cores) // len(ranks_here)
    return cores[lo:hi]


#: the cores this process was allowed when the module was first imported: repeated calls plan from THESE, never from an already
#: narrowed mask (a second call would otherwise halve the first call's share)
_INITIAL = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []


def pin_rank(local_rank: Optional[int] = None, local_world: Optional[int] = None) -> Optional[Dict]:
    """Restrict this process to its share of the cores; returns {"cores": n, "first": c0, "last": c1, "nodes": k} or None when
    nothing was done (single rank, `I2V_PIN_CPUS=0`, a platform without sched_setaffinity)."""
    if os.environ.get("I2V_PIN_CPUS", "1") in ("0", ""):
        return None
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = _INITIAL or sorted(os.sched_getaffinity(0))
    nodes = numa_nodes(allowed)
    mine = plan(local_rank, local_world, nodes)
    if not mine:
        return None
    os.sched_setaffinity(0, mine)
    return {"cores": len(mine), "first": mine[0], "last": mine[-1], "nodes": len(nodes)}
