#!/usr/bin/env python3
"""bench.py's cpu_baseline leg, the timed part: the CPU oracle on one pair at ONE thread count, in a process of its own.

Why a process of its own: the team must be pinned before libgomp starts -- OMP_PROC_BIND / OMP_PLACES are read
once, and sched_setaffinity on a live process moves only the calling thread -- and bench.py's own threads (pose fetchers, the HIP
runtime's) must not share the team's cores.  The worker restricts itself to `threads` PHYSICAL cores (one hardware thread per core: the
NUMA node of the first allowed CPU first, then the neighbouring nodes; a team that fits one node is NUMA-local), binds the OpenMP team to them (OMP_PROC_BIND=close,
OMP_PLACES=cores), runs a warm-up and `runs` timed alignments and prints their wall times as JSON.

usage: bench_cpu_worker.py pair.npz spec.json      (spec: width height params{...} T_init|null threads runs placement)
placement "close": one NUMA node first; "spread": the physical cores dealt round-robin over the nodes (a kd-tree search is bound by memory
latency: both sockets' caches and channels can beat locality -- bench.py times both and quotes the faster steady one).
Test infrastructure: this file and tests/oracle_lib.py are the only callers of oracle/ outside tests/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def numa_local_cores(n, placement="close"):
    """n CPUs for a team of n threads: physical cores first (one hardware thread per core) -- those of the NUMA node of the first CPU
    this process may run on, then the other nodes' in node order (neighbouring nodes share a socket) --, SMT siblings only when the
    machine has fewer cores than n.  Returns (cpus, all on one node?)."""
    allowed = sorted(os.sched_getaffinity(0))
    nodes = []
    try:
        base = "/sys/devices/system/node"
        for d in sorted((x for x in os.listdir(base) if x.startswith("node") and x[4:].isdigit()), key=lambda x: int(x[4:])):
            cpus = set()
            for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                if not part:
                    continue
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            nodes.append(sorted(c for c in cpus if c in allowed))
    except OSError:
        pass
    if not any(nodes):
        nodes = [allowed]
    first = next(k for k, cs in enumerate(nodes) if allowed[0] in cs) if any(allowed[0] in cs for cs in nodes) else 0
    order = nodes[first:] + nodes[:first]
    seen, cores, rest, node_of = set(), [], [], {}
    for k, cs in enumerate(order):
        for c in cs:
            node_of[c] = k
            try:
                sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
            except OSError:
                sib = str(c)
            if sib in seen:
                rest.append(c)
            else:
                seen.add(sib)
                cores.append(c)
    if placement == "spread":           # physical cores dealt round-robin over the nodes: every socket's caches and memory channels
        per = [[c for c in cores if node_of[c] == k] for k in range(len(order))]
        cores = [per[k][j] for j in range(max(len(x) for x in per)) for k in range(len(per)) if j < len(per[k])]
    pick = (cores + rest)[:n]
    return pick, len({node_of[c] for c in pick}) == 1


def main():
    npz, spec_path = sys.argv[1], sys.argv[2]
    spec = json.load(open(spec_path))
    th = int(spec["threads"])
    placement = spec.get("placement", "close")
    cpus, local = numa_local_cores(th, placement)
    os.sched_setaffinity(0, cpus)
    os.environ["OMP_NUM_THREADS"] = str(th)
    os.environ["OMP_PROC_BIND"] = "close" if placement == "close" else "spread"
    os.environ["OMP_PLACES"] = "cores"
    os.environ.setdefault("OMP_WAIT_POLICY", "active")          # (a dedicated, pinned team: spinning is what a tuned CPU run would do)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import oracle_lib as O
    from slam3d_gx_amd import synth
    z = np.load(npz)
    intr = synth.Intrinsics.scaled(spec["width"], spec["height"]) if spec["height"] > 1 else synth.Intrinsics(width=spec["width"], height=1)
    p = O.params(intr, threads=th, **spec["params"])
    Ti = None if spec.get("T_init") is None else np.array(spec["T_init"], dtype=np.float64).reshape(4, 4)
    O.icp(z["s4"], z["t4"], p, T_init=Ti, trace=False)          # warm-up: thread team, page faults
    times = []
    for _ in range(int(spec["runs"])):
        t0 = time.perf_counter()
        O.icp(z["s4"], z["t4"], p, T_init=Ti, trace=False)
        times.append(time.perf_counter() - t0)
    print(json.dumps({"threads": th, "times_s": times, "cpus": cpus, "numa_local": bool(local), "placement": placement}))


if __name__ == "__main__":
    main()
