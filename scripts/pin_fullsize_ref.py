#!/usr/bin/env python3
"""Pin a FULL-SIZE bench workload to the compiled reference.

Runs on the GPU box (needs the GPU for the HIP side and the box's host cores + memory for the reference):

    python3 scripts/pin_fullsize_ref.py --workload bacteria15k -k 21 -w 200 --out gpurun_out/r5a/pin_bacteria15k.json

1. the workload's genomes are generated on the device (the generator bench.py uses) and written as plain FASTA files
   (sw_batch_write_fasta) to /dev/shm;
2. the REAL reference (oracle/_ref: /root/reference/cpp compiled by oracle/Makefile) reads those files:
   _build_native(paths, k, w, n_cpu, False) + _get_penalty_native(...)  (build.cpp:330-394, filter.cpp:15-137);
3. the HIP path builds the same workload from the device batch and exports its arrays;
4. kmers, nodes (all six fields, penalty bit for bit), edges and record_offsets are compared ELEMENT FOR ELEMENT, and the
   position-dependent checksums + counts are computed FROM THE REFERENCE'S ARRAYS with numpy (device.host_checksums, chunked)
   -- those are what goes into tests/golden/bench_checksums_ref.json; nothing the HIP path produced is written there.

If the host cannot hold the FASTA files + the reference's working set, the genome count is reduced to the largest that fits and
the JSON says so (the key then carries the reduced count and is not used by the full-size tests).
Checker only: imports oracle/ (allowed for tests / bench cpu_baseline class tooling), never part of the product path.
"""
from __future__ import annotations

import argparse
import json
import os
import resource
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def log(*a):
    print(f"[pin {time.strftime('%H:%M:%S')}]", *a, flush=True)


def meminfo():
    d = {}
    for line in open("/proc/meminfo"):
        k, v = line.split(":", 1)
        d[k] = int(v.split()[0]) * 1024
    return d


def cgroup_limit():
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            s = open(p).read().strip()
            if s != "max":
                v = int(s)
                if v < 1 << 60:
                    return v
        except OSError:
            pass
    return None


def cgroup_usage():
    for p in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
        try:
            return int(open(p).read().strip())
        except (OSError, ValueError):
            pass
    return None


def chunked_checksums(kmers, nodes, edges, step=1 << 24):
    """device.host_checksums over slices (its bases make the shares add up modulo 2^64): bounded temporaries."""
    from seqwin_amd.device import host_checksums
    a = b = c = 0
    e0, n0 = edges[:0], nodes[:0]
    k0 = kmers[:0]
    for s in range(0, len(kmers), step):
        a = (a + host_checksums(kmers[s:s + step], n0, e0, kmer_base=s)[0]) % 2**64
    for s in range(0, len(nodes), step):
        b = (b + host_checksums(k0, nodes[s:s + step], e0, node_base=s)[1]) % 2**64
    for s in range(0, len(edges), step):
        c = (c + host_checksums(k0, n0, edges[s:s + step], edge_base=s)[2]) % 2**64
    return a, b, c


def first_difference(x, y, step=1 << 24):
    import numpy as np
    if len(x) != len(y):
        return f"lengths {len(x)} != {len(y)}"
    for s in range(0, len(x), step):
        xa, ya = x[s:s + step], y[s:s + step]
        if not np.array_equal(xa, ya):
            i = int(np.flatnonzero(xa != ya)[0]) + s
            return f"first difference at {i}: {x[i]} != {y[i]}"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bacteria15k")
    ap.add_argument("-k", type=int, default=21)
    ap.add_argument("-w", type=int, default=200)
    ap.add_argument("--genomes", type=int, default=None)
    ap.add_argument("--n-cpu", type=int, default=0, help="reference threads (0: min(hardware threads, 128))")
    ap.add_argument("--out", required=True)
    ap.add_argument("--ref-bytes-per-occ", type=float, default=120.0,
                    help="estimate of the reference's peak RSS per minimizer occurrence, its output arrays included (sizing only; "
                         "measured on bacteria15k at 128 threads: 93)")
    ap.add_argument("--size-from", default=None,
                    help="JSON of an earlier (smaller) run of the same workload: its measured peak RSS per occurrence x 1.15 replaces the estimate")
    args = ap.parse_args()

    import numpy as np

    import oracle
    from bench import SEED, WORKLOADS, make_batch, write_fasta_fast
    from seqwin_amd.device import CHECKSUM_SCHEME, Batch, set_device

    ref = oracle.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref is missing: build it in the dev container (make -C oracle ref); it travels with gpurun")
    G, rpg, rl, anc, snp, _ = WORKLOADS[args.workload]
    G_full = G
    if args.genomes:
        G = args.genomes
    k, w = args.k, args.w
    cores = os.cpu_count() or 1
    n_cpu = args.n_cpu or min(cores, 128)
    mi = meminfo()
    lim, use = cgroup_limit(), cgroup_usage()
    avail = mi["MemAvailable"]
    if lim is not None and use is not None:
        avail = min(avail, lim - use)
    shm = os.statvfs("/dev/shm")
    shm_free = shm.f_bavail * shm.f_frsize
    host = {"hardware_threads": cores, "MemTotal_GB": round(mi["MemTotal"] / 1e9, 1), "MemAvailable_GB": round(mi["MemAvailable"] / 1e9, 1),
            "cgroup_limit_GB": None if lim is None else round(lim / 1e9, 1), "usable_GB": round(avail / 1e9, 1),
            "dev_shm_free_GB": round(shm_free / 1e9, 1)}
    try:
        host["cpu_model"] = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    log("host:", json.dumps(host))

    # sizing: FASTA text in /dev/shm (counts as memory) + the reference's working set + both sides' arrays
    bp_genome = rpg * rl if rpg else rl
    if not rpg:
        os.environ.setdefault("SEQWIN_AMD_WRITE_LOWER_PPM", "10000")   # ragged workloads: 1 % of the bases soft-masked in the FASTA text
    fasta_genome = bp_genome * (81.0 / 80.0) + rpg * 16
    occ_genome = bp_genome * 2.0 / (w + 1)
    ref_per_occ = args.ref_bytes_per_occ
    if anc >= G_full:   # iid genomes: every minimizer its own node and edge in the reference's maps
        ref_per_occ += 250.0
    if args.size_from:
        src, _, key = args.size_from.partition("#")       # file[#key]: e.g. tests/golden/bench_checksums_ref.json#random100k/k19/w200@2500
        pj = json.loads(Path(src).read_text())
        if key:
            pj = pj[key]
        ref_per_occ = 1.15 * pj["reference"]["max_rss_GB"] * 1e9 / pj["counts"]["kmers"]
        log(f"sizing from {args.size_from}: {ref_per_occ:.0f} B per occurrence (measured peak RSS x 1.15)")
    hip_per_occ = 72.0 if anc >= G_full else 24.0   # the HIP side's exported arrays (kmers 8 + nodes 40 + edges 24 per node / edge)
    per_genome = fasta_genome + occ_genome * (ref_per_occ + hip_per_occ)
    fit = int(min(avail * 0.88, avail - 24e9) / per_genome)
    fit_shm = int(shm_free * 0.95 / fasta_genome)
    G_run = max(1, min(G, fit, fit_shm))
    log(f"per genome ~{per_genome / 1e6:.1f} MB -> memory holds {fit} genomes, /dev/shm {fit_shm}; running {G_run} of {G}")

    set_device(0)
    t0 = time.perf_counter()
    batch = make_batch(WORKLOADS[args.workload], G_run, SEED)
    tmp = tempfile.mkdtemp(prefix="seqwin_pin_", dir="/dev/shm")
    result = {"workload": args.workload, "k": k, "w": w, "genomes": G_run, "genomes_of_workload": G_full, "host": host,
              "command": "python3 " + " ".join(sys.argv), "seed": SEED}
    try:
        paths, bp = write_fasta_fast(batch, G_run, tmp, min(64, cores))
        log(f"{G_run} FASTA files, {bp / 1e9:.2f} Gbp, written in {time.perf_counter() - t0:.1f} s to {tmp}")
        tar = np.arange(G_run) % 2 == 0        # as bench.py and tests/test_gpu_fullsize.py

        # the reference
        t1 = time.perf_counter()
        rk, rn, re_, ro, _ids = ref._build_native(paths, k, w, n_cpu, False)
        t2 = time.perf_counter()
        log(f"reference _build_native at n_cpu={n_cpu}: {t2 - t1:.1f} s; kmers {len(rk)}, nodes {len(rn)}, edges {len(re_)}; "
            f"max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB")
        ref._get_penalty_native(rk, rn, ro, tar, n_cpu)
        t3 = time.perf_counter()
        log(f"reference _get_penalty_native: {t3 - t2:.1f} s")
        del _ids
        shutil.rmtree(tmp, ignore_errors=True)
        result["reference"] = {"n_cpu": n_cpu, "build_wall_s": round(t2 - t1, 2), "get_penalty_wall_s": round(t3 - t2, 2),
                               "Gbp_per_s": round(bp / (t3 - t1) / 1e9, 3), "Gbp": round(bp / 1e9, 3),
                               "max_rss_GB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)}
        sums = chunked_checksums(rk, rn, re_)
        result["counts"] = {"kmers": int(len(rk)), "nodes": int(len(rn)), "edges": int(len(re_))}
        result["checksums"] = [f"{s:016x}" for s in sums]
        result["checksums_from"] = "the compiled reference's arrays (numpy, seqwin_amd.device.host_checksums)"
        result["checksum_scheme"] = CHECKSUM_SCHEME
        result["weight_sum"] = int(re_["weight"].sum(dtype=np.uint64))
        result["n_tar_sum"] = int(rn["n_tar"].sum(dtype=np.uint64))
        result["n_neg_sum"] = int(rn["n_neg"].sum(dtype=np.uint64))
        log("reference checksums:", result["checksums"], result["counts"])

        # the HIP path on the same genomes (device batch -> index -> host arrays)
        t4 = time.perf_counter()
        ix = batch.build_index(k, w, tar)
        dev_sums = ix.checksums()
        hk, hn, he = ix.export()
        ho = batch.record_offsets()
        t5 = time.perf_counter()
        log(f"HIP build + export: {t5 - t4:.1f} s; device checksums {[f'{s:016x}' for s in dev_sums]}")
        cmp_ = {}
        for name, x, y in (("kmers", hk, rk), ("nodes", hn, rn), ("edges", he, re_), ("record_offsets", ho, ro)):
            d = first_difference(x, y)
            cmp_[name] = d is None
            if d is not None:
                log(f"MISMATCH in {name}: {d}")
        # penalty bit for bit (array_equal on the structured dtype compares the f64 by value; compare the bit patterns too)
        cmp_["penalty_bits"] = bool(len(hn) == len(rn) and np.array_equal(hn["penalty"].view(np.uint64), rn["penalty"].view(np.uint64)))
        cmp_["device_checksums_equal_reference_checksums"] = [f"{s:016x}" for s in dev_sums] == result["checksums"]
        result["hip_vs_reference_elementwise"] = cmp_
        result["equal"] = all(cmp_.values())
        result["hip_wall_s"] = round(t5 - t4, 2)
        ix.close()
        log("element-wise:", json.dumps(cmp_))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        batch.close()
    result["total_wall_s"] = round(time.perf_counter() - t0, 1)
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps(result, indent=1, sort_keys=True) + "\n")
    log("written", out, "equal =", result.get("equal"))
    if not result.get("equal"):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
