#!/usr/bin/env python3
"""bench.py -- Gbp/s of the minimizer-index build on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over the synthetic genome set, with the 2-bit packed
contigs and record tables already resident in HBM: fused ntHash + window-minimizer sketch -> tuple
ordering -> nodes / kmers (radix sort + run-length) -> per-node target / non-target assembly counts
and penalty -> adjacency edges; for N > 1 additionally the exchange of tuples / node ranks / adjacency rows
between the per-GPU shards over RCCL (seqwin_amd/dist.py).

Default workload = BASELINE.json configs[2] / configs[3]: 15 000 genomes x 5 Mbp
(50 contigs each) from 30 ancestors with 1 % substitutions, k = 21, w = 200 -- 75 Gbp.  With --gpus N the SAME
15 000 genomes are sharded over the N GPUs with the reference's worker partition (build.cpp:350-356): strong
scaling; the checksums of the N slices must add up to the single-GPU checksums (shard-count invariance,
tests/smoke/test_graph.py:67-127 at full size).  `--scaling weak` gives every GPU its own genome set instead.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (genomes, records per genome, record length, ancestors, snp ppm, default scaling)
    "bacteria15k": (15_000, 50, 100_000, 30, 10_000, "strong"),   # configs[2] (1 GPU) / configs[3] (sharded), 75 Gbp
    "salmonella500": (512, 50, 96_000, 5, 10_000, "weak"),         # configs[1] stand-in, 2.46 Gbp per GPU
    "random": (2_000, 1, 5_000_000, 2_000, 0, "weak"),             # configs[4] slice: iid-uniform genomes, 10 Gbp per GPU
    "random100k": (12_500, 1, 5_000_000, 12_500, 0, "weak"),       # configs[4]: 100 000 x 5 Mbp over 8 GPUs (12 500 per GPU)
    "tiny": (16, 4, 50_000, 2, 10_000, "strong"),
    # r06 (VERDICT r5 missing #4): the shape of real draft assemblies -- per genome 20-300 contigs of 200 bp ... 1.5 Mbp (median ~17 kbp)
    # up to ~4.8 Mbp, scaffold gaps of 10-1000 N in one contig of ten (records per genome = 0: ragged; "record length" = genome bp)
    "ragged500": (512, 0, 4_800_000, 5, 10_000, "weak"),
}
SEED = 20260821
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
GOLDEN = ROOT / "tests" / "golden" / "bench_checksums.json"          # the HIP path's own earlier results (--write-golden)
GOLDEN_REF = ROOT / "tests" / "golden" / "bench_checksums_ref.json"  # counts + checksums of the COMPILED REFERENCE's arrays at full size
                                                                     # (scripts/pin_fullsize_ref.py on the GPU box; never written here)


def write_fasta_sample(batch, n, tmp):
    """First n assemblies of a device batch as plain FASTA files (decoded from HBM) -> (paths, bases)."""
    offs, ids = batch.records()
    paths, bp = [], 0
    for a in range(n):
        p = os.path.join(tmp, f"g{a}.fa")
        with open(p, "wb") as f:
            for r in range(int(offs[a]), int(offs[a + 1])):
                seq = batch.record(r)
                bp += len(seq)
                f.write(b">" + ids[a][r - int(offs[a])].encode() + b"\n" + seq + b"\n")
        paths.append(p)
    return paths, bp


def make_batch(workload_tuple, n_genomes, seed, first_genome=0):
    """The device batch of `n_genomes` genomes of a workload (uniform contigs, or -- records per genome 0 -- ragged ones)."""
    from seqwin_amd.device import Batch
    _, rpg, rl, anc, snp, _ = workload_tuple
    if rpg == 0:
        return Batch.synthetic_ragged(n_genomes, rl, n_ancestors=anc, snp_ppm=snp, seed=seed, first_genome=first_genome)
    return Batch.synthetic(n_genomes, rpg, rl, n_ancestors=anc, snp_ppm=snp, seed=seed, first_genome=first_genome)


def write_fasta_fast(batch, n, directory, n_cpu):
    """First n assemblies of a device batch as FASTA files, decoded by the library's host threads (sw_batch_write_fasta)
    -> (paths, bases).  (write_fasta_sample above is the per-record Python form the tests use on small samples.)"""
    from seqwin_amd._lib import c_u64, check, lib
    check(lib.sw_batch_write_fasta(batch._h, c_u64(0), c_u64(n), os.fsencode(directory), c_u64(n_cpu), c_u64(80)))
    offs = batch.record_offsets()
    info = batch.info()
    bp = info["total_bp"] if n == info["n_assemblies"] else None
    paths = [os.path.join(directory, f"g{a}.fa") for a in range(n)]
    if bp is None:    # (a prefix of a batch of equal-sized genomes; of a ragged one: what the files hold)
        if info["total_bp"] % info["n_assemblies"] == 0 and info["n_records"] % info["n_assemblies"] == 0:
            bp = info["total_bp"] // info["n_assemblies"] * n
        else:
            bp = 0
            for p in paths:
                with open(p, "rb") as f:
                    data = f.read()
                bp += len(data) - data.count(b"\n") - sum(len(ln) for ln in data.split(b"\n") if ln.startswith(b">"))
    return paths, bp


def sample_dir_and_size(n_wanted, bytes_per_genome):
    """Where the sample's FASTA files go (/dev/shm if it has room, else the temp dir) and how many genomes fit: the files plus
    the CPU reference's working set (~3 x the files) must stay well inside what the host has available."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    n = n_wanted
    if avail is not None:
        n = min(n, max(1, int(avail * 0.5 / (4.0 * bytes_per_genome))))
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(base)
            if st.f_bavail * st.f_frsize > 1.3 * n * bytes_per_genome and os.access(base, os.W_OK):
                return tempfile.mkdtemp(prefix="seqwin_cpu_", dir=base), n
        except OSError:
            continue
    n = max(1, min(n, 128))
    return tempfile.mkdtemp(prefix="seqwin_cpu_"), n


def graph_stats_of_last_build(stats):
    keys = ("ingest_upload_ms", "device_ms", "plan_ms", "sketch_ms", "order_ms", "nodes_ms", "edges_ms", "export_ms")
    return {k: round(v, 2) for k, v in zip(keys, stats)}


def e2e_build(paths, k, w, n_cpu, tar):
    """FASTA paths -> numpy arrays + get_penalty through the C ABI exactly as seqwin_amd._core does it, with the split of the
    wall time: sw_build (ingest + upload | device), sw_graph_export (download), sw_get_penalty."""
    import ctypes

    import numpy as np

    from seqwin_amd import _core
    from seqwin_amd._lib import c_u64, c_vp, check, lib
    arr = (ctypes.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    g = c_vp()
    thr0, c0 = cgroup_throttled_ms(), time.process_time()
    t0 = time.perf_counter()
    check(lib.sw_build(arr, ctypes.c_size_t(len(paths)), c_u64(k), c_u64(w), c_u64(n_cpu), ctypes.c_int(0), ctypes.byref(g)))
    t1, c1 = time.perf_counter(), time.process_time()
    try:
        sz = [c_u64() for _ in range(6)]
        check(lib.sw_graph_sizes(g, *[ctypes.byref(x) for x in sz]))
        nk, nn, ne, na, nb, _bp = (x.value for x in sz)
        kmers, nodes, edges = np.empty(nk, _core.KMER_DTYPE), np.empty(nn, _core.NODE_DTYPE), np.empty(ne, _core.EDGE_DTYPE)
        ro = np.empty(na + 1, np.uint32)
        blob = ctypes.create_string_buffer(max(nb, 1))
        check(lib.sw_graph_export(g, _core._ptr(kmers), _core._ptr(nodes), _core._ptr(edges), _core._ptr(ro), blob))
        t_ids = time.perf_counter()
        ids = _core._split_ids(blob.raw[:nb], ro)   # (ids_by_assembly, the fifth element of _build_native's tuple: inside the timed call)
        ids_ms = (time.perf_counter() - t_ids) * 1e3
        t2, c2 = time.perf_counter(), time.process_time()
        st = (ctypes.c_double * 8)()
        check(lib.sw_graph_stats(g, st))
    finally:
        lib.sw_graph_free(g)
    t3 = time.perf_counter()
    _core._get_penalty_native(kmers, nodes, ro, tar, n_cpu)
    t4 = time.perf_counter()
    c4, thr4 = time.process_time(), cgroup_throttled_ms()
    split = graph_stats_of_last_build(list(st))
    # CPU seconds of this process (all threads) over the call, and how long the container's CPU quota held its threads back
    split.update(cpu_s=round(c4 - c0, 3), cpu_s_build_export_penalty=[round(c1 - c0, 3), round(c2 - c1, 3), round(c4 - c2, 3)], quota_throttled_ms=None if thr0 is None or thr4 is None else round(thr4 - thr0, 1))
    split.update(sw_build_wall_ms=round((t1 - t0) * 1e3, 2), alloc_and_export_wall_ms=round((t2 - t1) * 1e3, 2),
                 ids_list_ms=round(ids_ms, 2), get_penalty_wall_ms=round((t4 - t3) * 1e3, 2), total_wall_ms=round((t4 - t0) * 1e3, 2),
                 output_MB=round((kmers.nbytes + nodes.nbytes + edges.nbytes) / 1e6, 1))
    return (kmers, nodes, edges, ro), t4 - t0, split


def cgroup_throttled_ms():
    """Total time the container's threads have been held back by its CPU quota so far (cgroup v2 cpu.stat throttled_usec), or None."""
    for path, key, scale in (("/sys/fs/cgroup/cpu.stat", "throttled_usec", 1e-3), ("/sys/fs/cgroup/cpu/cpu.stat", "throttled_time", 1e-6)):
        try:
            for line in open(path):
                f = line.split()
                if len(f) == 2 and f[0] == key:
                    return int(f[1]) * scale
        except Exception:
            continue
    return None


def cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None: a 256-thread host with a quota of N
    CPUs runs any number of threads at the speed of N."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 1)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 1)
    except Exception:
        return None


def _gzip_one(job):
    import zlib
    src, dst = job
    c = zlib.compressobj(6, zlib.DEFLATED, 31)          # wbits 31: a gzip member (zlib releases the GIL while it compresses)
    with open(src, "rb") as f, open(dst, "wb") as g:
        data = c.compress(f.read()) + c.flush()
        g.write(data)
    return len(data)


def e2e_gz_leg(paths, tmp, k, w, tar, n_cpu, n_files=512):
    """The first n_files genomes as .fa.gz (level 6, one member per file -- what NCBI ships) through sw_build + sw_graph_export +
    sw_get_penalty, once by the host route (zlib on n_cpu threads + the SIMD packer) and once by the device route (the host
    copies the compressed bytes, k_inflate / k_parse decode and pack on the GPU: csrc/ingest_dev.hip); the arrays of both must be
    equal, and equal to the plain-FASTA arrays of the same genomes restricted to them (checked through their own build)."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    m = min(n_files, len(paths))
    gz = [os.path.join(tmp, f"z{a}.fa.gz") for a in range(m)]
    t0 = time.perf_counter()
    try:
        with ThreadPoolExecutor(min(n_cpu, 32)) as pool:   # (threads, not processes: this process holds the GPU)
            sizes = list(pool.map(_gzip_one, list(zip(paths[:m], gz))))
        pack_s = time.perf_counter() - t0
        bp = 0
        for p in paths[:m]:
            bp += os.path.getsize(p)
        bp = int(bp * 80 / 81)                       # (text bytes -> bases, to within the header lines; reported as Mbp only)
        tar_m = np.asarray(tar[:m], np.bool_).copy()
        if tar_m.all() or not tar_m.any():
            tar_m[: m // 2] = True
            tar_m[m // 2:] = False
        plain, _, _ = e2e_build(paths[:m], k, w, n_cpu, tar_m)
        out = {"files": m, "compressed_MB": round(sum(sizes) / 1e6, 1), "level": 6, "n_cpu": n_cpu, "compress_s": round(pack_s, 1)}
        routes = {}
        for route, env in (("host", "0"), ("device", "1")):
            os.environ["SEQWIN_AMD_DEVICE_INFLATE"] = env
            try:
                got, wall, split = e2e_build(gz, k, w, n_cpu, tar_m)
            finally:
                os.environ.pop("SEQWIN_AMD_DEVICE_INFLATE", None)
            routes[route] = {"wall_s": round(wall, 3), "Gbp_per_s": round(bp / wall / 1e9, 2),
                             "ingest_upload_ms": split["ingest_upload_ms"], "device_ms": split["device_ms"],
                             "equal_to_plain_fasta": bool(all(np.array_equal(a, b) for a, b in zip(got, plain)))}
            del got
        out["routes"] = routes
        best = min(routes, key=lambda r: routes[r]["wall_s"])
        out.update(value=routes[best]["Gbp_per_s"], unit="Gbp/s", route=best, Mbp=round(bp / 1e6, 1))
        return out
    except Exception as e:   # the leg is reported, never fatal for the bench line
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        for p in gz:
            try:
                os.unlink(p)
            except OSError:
                pass


def cpu_baseline_and_parity(batch, k, w, n_genomes_sample, is_targets):
    """Time the reference CPU path (oracle/_ref, kind 'reference'; else the C restatement, kind 'port') on a bounded
    sample of the same workload -- FASTA files in /dev/shm -> final arrays incl. get_penalty -- and compare its
    arrays, element for element, with the HIP path's on the same files (checker only: nothing here is timed as `value`).
    The same files then go through the drop-in boundary (e2e: sw_build + sw_graph_export + sw_get_penalty) with the split of
    that wall time."""
    import numpy as np

    import oracle
    from seqwin_amd.device import Batch
    info = batch.info()
    cores = os.cpu_count() or 1
    per_genome = info["total_bp"] / max(1, info["n_assemblies"]) * 1.03
    tmp, n = sample_dir_and_size(min(n_genomes_sample, info["n_assemblies"]), per_genome)
    t_w = time.perf_counter()
    paths, bp = write_fasta_fast(batch, n, tmp, min(32, cores))
    write_s = time.perf_counter() - t_w
    tar = np.asarray(is_targets[:n], np.bool_).copy()
    if tar.all() or not tar.any():
        tar[: n // 2] = True
        tar[n // 2:] = False
    ref = oracle.load_ref()
    runs, all_cores = [], None
    try:
        if ref is not None:
            # the compiled reference at the README's thread count (8) -- `value` -- and at ALL the host's
            # hardware threads on the same files -- `all_cores`
            kind = "reference"
            for n_cpu in sorted({min(8, cores, n), min(cores, n)}, reverse=True):   # (all cores first: the 8-thread arrays are the ones compared)
                t0 = time.perf_counter()
                kmers, nodes, edges, ro, _ = ref._build_native(paths, k, w, n_cpu, False)
                ref._get_penalty_native(kmers, nodes, ro, tar, n_cpu)
                wall = time.perf_counter() - t0
                if n_cpu > 8:
                    all_cores = {"value": round(bp / wall / 1e9, 4), "unit": "Gbp/s", "cores": n_cpu, "wall_s": round(wall, 2),
                                 "cpu_quota_cores": cpu_quota()}
                else:
                    runs.append((wall, n_cpu))
        else:
            kind = "port"
            t0 = time.perf_counter()
            kmers, nodes, edges, ro, _ = oracle.build(paths, k, w)
            oracle.get_penalty(kmers, nodes, ro, tar)
            runs.append((time.perf_counter() - t0, 1))
        dt, used = min(runs)
        # T_e2e: the same files through the drop-in boundary.  Reported beside the baseline; never `value`.
        e2e_build(paths[:2], k, w, 2, tar[:2])   # (library warm-up: allocator, module load)
        e2e_runs, e2e_equal = [], True
        quota = cpu_quota()
        sweep = {min(c, cores) for c in (8, 32, 64, 128)} | ({max(1, min(cores, int(round(quota))))} if quota else set())
        if os.environ.get("SEQWIN_BENCH_E2E_NCPU"):   # e.g. "16,64,256" with SEQWIN_AMD_INGEST_WORKERS_MAX=256: the ingest's scaling table
            sweep = {min(int(c), cores) for c in os.environ["SEQWIN_BENCH_E2E_NCPU"].split(",")}
        # the first full call of the process pays once for what later calls reuse (page-locked buffers of the streaming ingest,
        # pool blocks, the download ring): timed and reported on its own, outside the sweep
        first_cpu = 32 if 32 in sweep else max(sweep)
        got, first_wall, first_split = e2e_build(paths, k, w, first_cpu, tar)
        del got
        for n_cpu in sorted(sweep):
            got, wall, split = e2e_build(paths, k, w, n_cpu, tar)
            e2e_runs.append((wall, n_cpu, split))
            e2e_equal = e2e_equal and bool(np.array_equal(got[0], kmers) and np.array_equal(got[1], nodes)
                                           and np.array_equal(got[2], edges) and np.array_equal(got[3], ro))
            del got
        e2e_dt, e2e_cpu, e2e_split = min(e2e_runs, key=lambda r: r[0])
        e2e = {"value": round(bp / e2e_dt / 1e9, 3), "unit": "Gbp/s", "n_cpu": e2e_cpu, "equal_to_cpu_baseline": e2e_equal,
               "genomes": n, "Mbp": round(bp / 1e6, 1), "split_ms": e2e_split,
               "by_n_cpu": {str(c): round(bp / t / 1e9, 2) for t, c, _ in e2e_runs}, "cpu_quota_cores": quota,
               "first_call": {"Gbp_per_s": round(bp / first_wall / 1e9, 2), "wall_s": round(first_wall, 3), "n_cpu": first_cpu,
                              "ingest_upload_ms": first_split["ingest_upload_ms"]},
               "sample": f"the same {n} FASTA files through sw_build + sw_graph_export + sw_get_penalty (ingest + PCIe + device + "
                         "download); wall " + ", ".join(f"{t:.3f} s at n_cpu={c}" for t, c, _ in e2e_runs),
               "vs_cpu_baseline": round(dt / e2e_dt, 1)}
        # the reference's DEFAULT input is .fna.gz (src/seqwin/config.py:158; gz branch of fasta_reader.cpp:109-203): the same
        # genomes as level-6 gzip members through the same boundary.  Few hundred files: enough to time, bounded to compress.
        e2e["gz"] = e2e_gz_leg(paths, tmp, k, w, tar, min(cores, 64))
        # the HIP path on the same files, through the same ingest as sw_build
        sb = Batch.from_fasta(paths, n_cpu=min(16, cores))
        six = sb.build_index(k, w, tar)
        gk, gn, ge = six.export()
        equal = bool(np.array_equal(gk, kmers) and np.array_equal(gn, nodes) and np.array_equal(ge, edges)
                     and np.array_equal(sb.record_offsets(), ro))
        six.close()
        sb.close()
    finally:
        for p in paths:
            try:
                os.unlink(p)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass
    base = {"value": round(bp / dt / 1e9, 4), "unit": "Gbp/s", "cores": used, "kind": kind,
            "sample": f"first {n} genomes of the workload ({bp / 1e6:.0f} Mbp) as plain FASTA in {os.path.dirname(tmp)} -> "
                      f"kmers/nodes/edges + get_penalty; wall " + ", ".join(f"{t:.2f} s at n_cpu={c}" for t, c in runs)
                      + f" (host has {cores} hardware threads; files written in {write_s:.1f} s)",
            "n_kmers": int(len(kmers)), "n_nodes": int(len(nodes)), "n_edges": int(len(edges))}
    if all_cores is not None:
        base["all_cores"] = all_cores
    parity = {"vs": kind, "sample_genomes": n, "equal": equal,
              "compared": "kmers, nodes (hash, start, stop, n_tar, n_neg, penalty bit-for-bit), edges, record_offsets"}
    return base, parity, e2e


def golden_checksums(workload, k, w):
    """(entry, "reference" | "self"): the reference-derived full-size values where they exist, else the HIP path's own."""
    from seqwin_amd.device import CHECKSUM_SCHEME
    key = f"{workload}/k{k}/w{w}"
    for path, src in ((GOLDEN_REF, "reference"), (GOLDEN, "self")):
        try:
            e = json.loads(path.read_text()).get(key)
        except Exception:
            e = None
        if e is not None and e.get("genomes", e.get("genomes_of_workload")) == e.get("genomes_of_workload") \
                and e.get("checksum_scheme") == CHECKSUM_SCHEME:   # (entries of an older checksum definition do not count)
            return e, src
    return None, None


def distinct_gpus(ranks):
    """How many different GPUs the ranks' records name.  One GPU = (host, UUID, PCI bus id) and, where a rank sees several devices, its
    device index: two ranks on ONE card agree in all of them; two cards whose runtime reports no or equal UUIDs still differ in bus id
    or index (a false "shared" would refuse a good node)."""
    return len({(r.get("host"), r.get("uuid") or None, r.get("pci_bus_id"), r.get("device") if (r.get("visible_devices") or 0) > 1 else None)
                for r in ranks})


def preflight(world, rank, local_rank, dev):
    """Before anything is timed at N > 1: the N ranks must drive N DISTINCT GPUs, every ordered pair of them
    must have peer access (a pair without it would be staged through host memory: a different, slower experiment), and the
    collectives must deliver messages of the size the exchanges really send (dist.check_collectives: one element more than a
    256 MiB round, verified element for element -- RCCL 2.26 at world size 1 silently dropped half of such a message, NOTES.md r03).
    Returns (info, reason): reason is None, or why this run must not be timed -- every rank gets the same answer.
    SEQWIN_BENCH_ALLOW_SHARED_GPU=1 / SEQWIN_BENCH_ALLOW_STAGED=1 lift the first two refusals (rehearsals on one card);
    SEQWIN_BENCH_PREFLIGHT_FORCE=distinct|peer|collectives makes that probe fail (tests)."""
    import torch
    import torch.distributed as dist

    from seqwin_amd import dist as swdist
    force = os.environ.get("SEQWIN_BENCH_PREFLIGHT_FORCE", "")
    me = {"rank": rank, "local_rank": local_rank, "host": os.uname().nodename, "pid": os.getpid()}
    try:
        me.update(device=torch.cuda.current_device(), visible_devices=torch.cuda.device_count())
        prop = torch.cuda.get_device_properties(me["device"])
        me.update(name=prop.name, uuid=str(getattr(prop, "uuid", "") or ""), pci_bus_id=getattr(prop, "pci_bus_id", None))
    except Exception as e:          # (what could be read stays in the line; the device index alone still tells two cards of a host apart)
        me["error"] = str(e)
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    distinct = distinct_gpus(ranks)
    if force == "distinct":
        distinct = 1
    info = {"ranks": ranks, "distinct_gpus": distinct}
    if world == 1:
        return info, None
    # peer access from this rank's device to every other rank's (where this process can see it)
    mine = []
    for r in ranks:
        if r.get("rank") == rank:
            continue
        d, ok = r.get("device"), None
        try:
            if r.get("host") == me.get("host") and isinstance(d, int) and d < torch.cuda.device_count() and d != torch.cuda.current_device():
                ok = bool(torch.cuda.can_device_access_peer(torch.cuda.current_device(), d))
        except Exception:
            ok = None
        mine.append(ok)
    if force == "peer":
        mine = [False] * len(mine)
    allp = [None] * world
    dist.all_gather_object(allp, mine)
    flat = [x for row in allp for x in row]
    info["peer_access"] = {"ordered_pairs": len(flat), "direct": sum(1 for x in flat if x is True), "not_direct": sum(1 for x in flat if x is False),
                           "unknown": sum(1 for x in flat if x is None)}
    reason = None
    if distinct != world and os.environ.get("SEQWIN_BENCH_ALLOW_SHARED_GPU") != "1":
        reason = (f"{world} ranks drive {distinct} distinct GPU(s): a scaling point needs one GPU per rank "
                  "(SEQWIN_BENCH_ALLOW_SHARED_GPU=1 times it anyway, as a rehearsal)")
    elif info["peer_access"]["not_direct"] and os.environ.get("SEQWIN_BENCH_ALLOW_STAGED") != "1":
        reason = (f"{info['peer_access']['not_direct']} of {len(flat)} ordered GPU pairs have no peer access: the exchanges would be staged "
                  "through host memory (SEQWIN_BENCH_ALLOW_STAGED=1 times that anyway)")
    else:
        try:
            if force == "collectives":
                raise RuntimeError("collective self-check failed: forced by SEQWIN_BENCH_PREFLIGHT_FORCE")
            swdist.check_collectives(dev)
            info["collectives_checked"] = f"per-peer messages of one element more than a round of {swdist._MSG_LIMIT >> 20} MiB" \
                if dist.get_backend() == "nccl" else "64 MiB messages (not RCCL)"
        except RuntimeError as e:
            reason = str(e)
    return info, reason


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bacteria15k", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="strong: the workload's genomes are sharded over the GPUs; weak: every GPU gets its own set")
    ap.add_argument("--genomes", type=int, default=None, help="override the workload's genome count")
    ap.add_argument("-k", "--kmerlen", type=int, default=21)
    ap.add_argument("-w", "--windowsize", type=int, default=200)
    ap.add_argument("--cpu-sample-genomes", type=int, default=2048,
                    help="genomes of the workload the CPU reference and the end-to-end leg run on (10 Gbp: ~25 s at 8 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--write-golden", action="store_true", help="record this run's N=1 checksums in tests/golden/")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: whatever a library prints there (RCCL's version banner when a communicator is
    # made, for one) goes to stderr -- file descriptor 1 is pointed at stderr for the run, the line is written to the real one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    force_coll = os.environ.get("SEQWIN_DIST_FORCE_COLLECTIVES") == "1"   # every collective issued at world size 1 (RCCL on one GPU)
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SEQWIN_BENCH_BACKEND", "nccl")   # "gloo": smoke-test the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from seqwin_amd import dist as swdist
    from seqwin_amd.device import Batch, set_device
    set_device(local_rank)

    G, rpg, rl, anc, snp, default_scaling = WORKLOADS[args.workload]
    if args.genomes:
        G = args.genomes
    scaling = args.scaling or default_scaling
    k, w = args.kmerlen, args.windowsize
    if scaling == "strong":
        # ONE job of G genomes; rank r holds the contiguous assembly range of "thread r" (build.cpp:350-356)
        G_total = G
        first, end = swdist.partition_assemblies(G_total, world)[rank]
        batch = make_batch(WORKLOADS[args.workload], end - first, SEED, first)
    else:
        # every rank holds its own G genomes (assemblies [rank*G, (rank+1)*G) of the job)
        G_total = G * world
        first, end = rank * G, (rank + 1) * G
        batch = make_batch(WORKLOADS[args.workload], G, SEED + rank)
    is_targets_global = np.arange(G_total) % 2 == 0
    my_targets = is_targets_global[first:end]
    bp_rank = batch.info()["total_bp"]
    total_bp = G_total * rpg * rl if rpg else bp_rank * world   # (ragged: one rank's bases; only N = 1 is quoted for it)
    if rpg:
        assert bp_rank == (end - first) * rpg * rl

    use_dist = world > 1 or force_coll or os.environ.get("SEQWIN_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: cost of the sharded path at N=1
    if use_dist:
        shard = swdist.Shard(batch, first_assembly=first, n_assemblies_total=G_total)
        engine = swdist.HipEngine("device" if world == 1 or dist.get_backend() == "nccl" else "host")

        def step():
            return swdist.build_sharded_index(shard, k, w, is_targets_global, engine=engine)
    else:
        def step():
            return batch.build_index(k, w, my_targets)

    pre_info, refused = preflight(world, rank, local_rank, engine.device if use_dist and world > 1 else torch.device("cuda", local_rank))
    if refused is not None:
        if rank == 0:
            sys.stderr.write(f"bench.py: NOT TIMED -- {refused}\n")
            os.write(real_stdout, (json.dumps({"metric": "Gbp/s minimizer-indexed", "value": None, "unit": "Gbp/s", "n_gpus": world,
                                               "refused": refused, "preflight": pre_info}) + "\n").encode())
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(3)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def release(ix):
        # give the previous result's HBM back to the pool before the next pass (as a pipeline would)
        if ix is not None:
            (ix.merged if hasattr(ix, "merged") else ix).close()

    # what a one-shot sw_build pays on the device side: launch plan (host + device, cached in the batch afterwards), first
    # hipMalloc of every pool block, and the build itself
    # (the first of the W warm-up steps; with --warmup 0 it is part of the timed region and not reported separately)
    ix, first_build_ms, first_plan_ms = None, None, None
    fence()
    for i in range(args.warmup):
        release(ix)
        t_first = time.perf_counter()
        ix = step()
        if i == 0:
            fence()
            first_build_ms = (time.perf_counter() - t_first) * 1e3
            first_plan_ms = ix.timings().get("plan_ms", 0.0)
    fence()
    stage = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        release(ix)
        ix = step()
        for key, v in ix.timings().items():
            if key.endswith("_ms") and key != "plan_ms":
                stage[key] = stage.get(key, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    red_dev = engine.device if use_dist and world > 1 else torch.device("cuda", local_rank)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-phase device time of the LAST step's sharded build, max over ranks (stream events: no synchronisation inside the step)
    dist_info = None
    if use_dist:
        ph = ix.phases_ms()
        names = sorted(ph)
        if world > 1:                                    # every rank walks the same phases (same route, same form)
            tph = torch.tensor([ph[n] for n in names], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tph, op=dist.ReduceOp.MAX)
            ph = dict(zip(names, tph.tolist()))
        dist_info = dict(ix.info, phases_ms_max_over_ranks={n: round(ph[n], 3) for n in ph},
                         collectives="issued" if (world > 1 or force_coll) else "skipped (one rank, SEQWIN_BENCH_FORCE_DIST)")

    # who ran: one entry per rank -- device index, UUID and PCI bus id of the GPU it drove -- gathered over the job's own process
    # group by the pre-flight, so that a SCALE record can show N ranks on N distinct GPUs over RCCL
    ranks_info = pre_info["ranks"]

    nk, nn, ne = ix.sizes()
    tm = ix.timings()
    counts = torch.tensor([nk, nn, ne], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(counts)
    n_occ_local = tm.get("n_occ_local", nk)
    # checksums of the whole (concatenated) result: at N > 1 the slices' shares add up modulo 2^64
    sums = ix.global_checksums() if use_dist else ix.checksums()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_bp / (dt / args.steps) / 1e9
        stage = {key: v / args.steps for key, v in stage.items()}
        # Dominant kernel: the fused sketch kernel (one launch per step per GPU) wherever the sketch stage is the longest
        # (w = 200); at small windows the index stages take over and the line says so.  Algorithmic bytes per launch
        # (DESIGN.md section 3.3): sketch = read the 2-bit input once + write one 16 B tuple per minimizer; nodes stage =
        # read the tuple + write the kmers entry + the nodes; edges stage = write + read one adjacency record + the edges.
        tot0 = counts.tolist()
        stage_bytes = {"sketch_ms": 0.25 * bp_rank + 16.0 * n_occ_local,
                       "nodes_ms": (24.0 * tot0[0] + 40.0 * tot0[1]) / world,
                       "edges_ms": (40.0 * tot0[0] + 24.0 * tot0[2]) / world}
        dominant = max(stage_bytes, key=lambda key: stage.get(key, 0.0))
        sk_bytes = stage_bytes[dominant]
        sk_ms = stage[dominant]
        achieved = sk_bytes / (sk_ms * 1e-3) / 1e9
        plan_L = 32 if w >= 32 else 16 if w >= 16 else 8 if w >= 8 else 4
        dom_kernel = {"sketch_ms": f"sketch_fast_kernel<{plan_L}, 256>" if (k <= 256 and w >= 4) else "sketch_generic_kernel",
                      "nodes_ms": "nodes stage: k_rs_pair_pass radix passes + k_nodes + unsort (no single dominant kernel)",
                      "edges_ms": "edges stage: k_rs_pass_p radix passes + run lengths (no single dominant kernel)"}[dominant]
        # whole path, SURVEY 8d: 0.25 N_bp + 40 N_occ + 40 N_adj + 40 N_node + 24 N_edge  (N_adj ~= N_occ)
        tot = counts.tolist()
        path_bytes = 0.25 * total_bp + 80.0 * tot[0] + 40.0 * tot[1] + 24.0 * tot[2]
        # HBM bytes per launch of the dominant kernel from the committed PMC profile of this same command
        # (profiles/traffic.json, written by scripts/summarize_profiles.py); null for other workloads
        traffic, traffic_src, valu_insts, valu_busy = None, None, None, (None, None)
        try:
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
            tj = tj.get("entries", {}).get(f"{args.workload}/k{k}/w{w}", tj if "entries" not in tj else {})   # one entry per (workload, k, w)
            import hashlib
            same_kernel = tj.get("sketch_hip_sha256") == hashlib.sha256(
                (ROOT / "seqwin_amd" / "csrc" / "sketch.hip").read_bytes()).hexdigest()   # stale profile -> traffic null
            if (tj.get("workload"), tj.get("k"), tj.get("w")) == (args.workload, k, w) and world == 1 and dominant == "sketch_ms" \
                    and not args.genomes and same_kernel:
                traffic, traffic_src = int(tj["hbm_bytes_per_launch"]), tj.get("source")
                valu_insts = tj.get("valu_wave_insts_per_launch")
                valu_busy = (tj.get("valu_insts_per_simd_cycle"), tj.get("valu_busy_source"))
        except Exception:
            pass
        out = {
            "metric": "Gbp/s minimizer-indexed", "value": round(value, 3), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {G_total} genomes x {rpg} contigs x {rl} bp ({anc} ancestors, "
                                    f"{snp / 1e4:g}% substitutions), on-device generator, seed {SEED}") if rpg else
                                   (f"{args.workload}: {G_total} ragged genomes of ~{rl} bp -- {batch.info()['n_records']} contigs of 200 bp ... 1.5 Mbp, "
                                    f"scaffold gaps of 10-1000 N in one contig of ten ({anc} ancestors, {snp / 1e4:g}% substitutions), on-device generator, seed {SEED}"),
                       "genomes": G_total, "genomes_per_gpu": end - first, "mean_bp": (rpg * rl) if rpg else total_bp // max(1, G_total), "k": k, "w": w,
                       "parallelism": f"assembly-sharded x{world} ({scaling} scaling)" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(sk_bytes), "avg_launch_ms": round(sk_ms, 4),
                         "dominant_stage": dominant,
                         "note": ("bound by the latency of a wave's own instruction stream and by integer VALU issue at w=200 "
                                  "(DESIGN.md section 3.1); measured HBM traffic is in profiles/") if dominant == "sketch_ms"
                                 else "the index stages (HBM-bound radix sorts) take over at small windows; `achieved` is the stage's algorithmic bytes over its time",
                         "path_algorithmic_bytes": int(path_bytes),
                         "path_achieved_GBs": round(path_bytes / world / (dt / args.steps) / 1e9, 2),
                         "path_frac": round(path_bytes / world / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                         "sketch_Gbp_per_s_per_gpu": round(bp_rank / (stage["sketch_ms"] * 1e-3) / 1e9, 2)},
            "stages_ms": {key: round(v, 4) for key, v in stage.items()},
            "plan_ms": round(first_plan_ms if first_plan_ms is not None else tm.get("plan_ms", 0.0), 3),
            "first_build_ms": round(first_build_ms, 3) if first_build_ms is not None else None,
            "counts": {"kmers": tot[0], "nodes": tot[1], "edges": tot[2]},
            "tiles": {key: int(tm.get(key, 0)) for key in ("n_tiles", "tiles_b256", "tiles_b64", "tiles_generic", "tiles_gap", "ovf_tiles")},
            "checksums": [f"{s:016x}" for s in sums],
        }
        try:   # how the radix passes rank on this device (the LDS-atomic form needs the device's self-check to pass)
            import ctypes

            from seqwin_amd._lib import check, lib
            rm = ctypes.c_int(-1)
            check(lib.sw_radix_rank_mode(ctypes.byref(rm)))
            out["radix_rank"] = {1: "lds_atomic (device self-check passed)", 0: "ballot"}.get(rm.value, str(rm.value))
        except Exception as e:   # (an older library)
            out["radix_rank"] = f"unknown ({e})"
        if dist_info is None:
            dist_info = {}
        dist_info.update(world=world, backend=(dist.get_backend() if dist.is_initialized() else None),
                         process_group_size=(dist.get_world_size() if dist.is_initialized() else 1),
                         distinct_gpus=pre_info["distinct_gpus"], peer_access=pre_info.get("peer_access"),
                         collectives_checked=pre_info.get("collectives_checked"), ranks=ranks_info)
        try:
            dist_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        out["dist"] = dist_info
        if valu_insts:
            # explanatory (not the mandated roofline): integer VALU issue, 256 CU x 4 SIMD x 32 lanes x 2.4 GHz peak
            lane_ops = valu_insts * 64.0
            out["roofline"]["valu"] = {"lane_ops_per_bp": round(lane_ops / bp_rank, 1),
                                       "achieved_Tlaneops_per_s": round(lane_ops / (stage["sketch_ms"] * 1e-3) / 1e12, 2),
                                       "peak_Tlaneops_per_s": 78.64, "frac": round(lane_ops / (stage["sketch_ms"] * 1e-3) / 78.64e12, 3),
                                       # MEASURED (r06): VALU wave-instructions per SIMD and kernel cycle (GRBM_GUI_ACTIVE / 8 XCDs); with the mix's
                                       # measured issue costs (2.05-4.9 cycles, mean ~3.6) the VALU port is occupied ~0.97 of the time.  gfx950 exposes
                                       # no VALU-cycle counter: SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU (profiles/*_pmc_valu_busy.txt)
                                       "insts_per_simd_cycle": valu_busy[0],
                                       "issue_interval_cycles": round(1.0 / valu_busy[0], 2) if valu_busy[0] else None,
                                       "busy_source": valu_busy[1],
                                       "source": "SQ_INSTS_VALU of the committed PMC profile (profiles/*_pmc_sketch.txt)",
                                       "note": "peak = 32 lanes/clk/SIMD (2 cycles per wave64 VOP2); three-operand VOP3, v_cndmask, "
                                               "v_cmp and 64-bit moves measure 4.2-4.9 cycles (scripts/micro/valu_kinds.hip), "
                                               "so this instruction mix cannot reach that peak"}
        parity = {}
        # (one rank holds the whole workload whatever the scaling mode is called; with more ranks only "strong" keeps the genomes)
        gold, gold_src = golden_checksums(args.workload, k, w) if (scaling == "strong" or world == 1) and not args.genomes else (None, None)
        if args.write_golden and world == 1 and not use_dist:
            # (only the file of the HIP path's own results; the reference-derived file is written by scripts/pin_fullsize_ref.py alone)
            allg = json.loads(GOLDEN.read_text()) if GOLDEN.exists() else {}
            from seqwin_amd.device import CHECKSUM_SCHEME
            allg[f"{args.workload}/k{k}/w{w}"] = {"checksums": out["checksums"], "counts": out["counts"], "checksum_scheme": CHECKSUM_SCHEME}
            GOLDEN.write_text(json.dumps(allg, indent=1, sort_keys=True) + "\n")
        if gold is not None:
            # full-size parity: counts + position-dependent checksums of kmers / nodes (all fields but the f64) / edges against the
            # values computed FROM THE COMPILED REFERENCE'S ARRAYS on the whole workload ("reference"; the pin run also compared
            # every array element for element, penalty bits included), or -- where no such run exists -- against an earlier run of
            # this library ("self").  At N > 1 the same line is shard-count invariance (tests/smoke/test_graph.py:67-127).
            parity["n1_checksums_equal"] = gold["checksums"] == out["checksums"] and gold["counts"] == out["counts"]
            parity["full_size_vs"] = gold_src
            if gold_src == "reference":
                parity["full_size_reference"] = {kk: gold[kk] for kk in ("genomes", "reference", "command") if kk in gold}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], ref_par, out["e2e"] = cpu_baseline_and_parity(batch, k, w, args.cpu_sample_genomes, my_targets)
            parity.update(ref_par)
        if parity:
            out["parity"] = parity
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
