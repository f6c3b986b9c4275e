#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the dev container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py

The reference extension (cpp/src/bindings/python_bindings.cpp, compiled by oracle/Makefile into
oracle/_ref/) is imported and called on
  (i)   the reference's own smoke FASTA files (copied verbatim as data into tests/golden/smoke/)
        at several (k, w), followed by _get_penalty_native with is_targets = [T, T, F, F];
  (ii)  seeded synthetic assemblies written to tests/golden/synth/ (N runs, lowercase, U, IUPAC,
        empty records, records shorter than k+w-1, multi-record assemblies, duplicated contigs,
        CRLF / blank lines, one gzip member);
  (iii) w = 1 builds, which expose out_hash of EVERY valid k-mer (known answers for the hash itself)
        for k covering all k%4 remainders and both rotate periods;
  (iv)  _get_penalty_native / _filter_kmers_native on the synthetic cases of the reference's
        tests/smoke/test_graph.py:190-219, 248-304.
Only inputs and outputs are stored; no reference source text is copied.
"""
from __future__ import annotations

import gzip
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

SMOKE = [HERE / "smoke/targets/target-1.fasta", HERE / "smoke/targets/target-2.fasta",
         HERE / "smoke/non-targets/non-target-1.fasta", HERE / "smoke/non-targets/non-target-2.fasta"]
SMOKE_KW = [(17, 10), (21, 200), (7, 10), (15, 50), (31, 50)]
HASH_K = [3, 4, 15, 16, 17, 18, 19, 21, 31, 32, 33, 34, 47, 64]
# round 3: the k and w the GPU tests use against the oracle beyond the fast kernel's k <= 256 / the tiles' w <= 4096
HASH_K_WIDE = [100, 255, 256, 257, 300]
LONG_KW = [(64, 1000), (100, 4097), (255, 200), (256, 1000), (257, 20000), (300, 1000)]


def synth_sets(rng: np.random.Generator) -> dict[str, list[Path]]:
    out = HERE / "synth"
    out.mkdir(exist_ok=True)
    B = np.array(list("ACGT"))

    def rand(n):
        return "".join(B[rng.integers(0, 4, n)])

    def mutate(s, rate):
        a = np.array(list(s))
        idx = rng.random(len(a)) < rate
        a[idx] = B[rng.integers(0, 4, int(idx.sum()))]
        return "".join(a)

    def wrap(s, width=70, eol="\n"):
        return eol.join(s[i:i + width] for i in range(0, len(s), width)) + eol if s else ""

    sets: dict[str, list[Path]] = {}

    # pan: 6 assemblies from one ancestor, 3 contigs each, 1% SNPs  -> shared nodes / edges with weight > 1
    anc = [rand(3000), rand(2500), rand(1200)]
    paths = []
    for a in range(6):
        p = out / f"pan_{a}.fa"
        p.write_text("".join(f">pan{a}_c{c} synthetic\n" + wrap(mutate(s, 0.01)) for c, s in enumerate(anc)))
        paths.append(p)
    sets["pan"] = paths

    # edge cases
    e0 = out / "edge_0.fa"  # N runs, lowercase, U, IUPAC, stretches shorter than k
    s = rand(900)
    s = s[:100] + "N" * 37 + s[100:300].lower() + "RYKM" + s[300:320] + "N" + s[320:330] + "N" * 3 + s[330:600].replace("T", "U") + "n" + s[600:]
    e0.write_text(">e0 with gaps\n" + wrap(s, 60))
    e1 = out / "edge_1.fa"  # empty record, short records, CRLF, blank lines, duplicated contig (self edges, repeated edges)
    d = rand(700)
    e1.write_text(">empty\n>short\nACGTACGTAC\n\n>dupA\r\n" + wrap(d, 80, "\r\n") + "  \n>dupB extra words\n" + wrap(d, 50) +
                  ">tandem\n" + wrap("ACGTTGCA" * 120, 64) + ">polyA\n" + wrap("A" * 600, 100))
    e2 = out / "edge_2.fa.gz"  # gzip member, leading N's, trailing N's
    e2_text = ">gz1\n" + wrap("NNNNNNNN" + rand(800) + "NNNN", 61) + ">gz2\n" + wrap(mutate(d, 0.02), 70)
    if not e2.exists() or gzip.open(e2, "rt").read() != e2_text:   # (a gzip member carries its time stamp: not rewritten when unchanged)
        with gzip.open(e2, "wt") as f:
            f.write(e2_text)
    e3 = out / "edge_3.fa"  # no records at all
    e3.write_text("")
    sets["edge"] = [e0, e1, e2, e3]

    # long: contigs long enough for windows of thousands of k-mers (own generator: the sets above stay as they are)
    rl = np.random.default_rng(20261004)

    def randl(n):
        return "".join(B[rl.integers(0, 4, n)])

    base = randl(52_000)
    paths = []
    for a in range(3):
        sq = list(base)
        idx = rl.random(len(sq)) < 0.01
        for i in np.nonzero(idx)[0]:
            sq[i] = B[rl.integers(0, 4)]
        sq = "".join(sq)
        if a == 1:   # N runs: one long, two closer than k = 300 to each other
            sq = sq[:9000] + "N" * 700 + sq[9000:30000] + "N" + sq[30000:30200] + "NN" + sq[30200:]
        p = out / f"long_{a}.fa"
        p.write_text(f">long{a}_c0\n" + wrap(sq, 80) + f">long{a}_c1 short\n" + wrap(randl(1500), 80))
        paths.append(p)
    sets["long"] = paths
    return sets


def run_build(ref, paths, k, w):
    kmers, nodes, edges, offs, ids = ref._build_native([str(p) for p in paths], k, w, 1, False)
    return dict(kmers=kmers, nodes=nodes, edges=edges, record_offsets=offs), [list(t) for t in ids]


def main() -> None:
    ref = oracle.load_ref()
    if ref is None:
        sys.exit("oracle/_ref is not built: run `make -C oracle ref` in the dev container")
    rng = np.random.default_rng(20260821)
    sets = synth_sets(rng)
    manifest = {"cases": []}
    out = HERE / "vectors"
    out.mkdir(exist_ok=True)

    def add(name, paths, k, w, is_targets=None):
        arrays, ids = run_build(ref, paths, k, w)
        if is_targets is not None and len(arrays["nodes"]):
            scored = arrays["nodes"].copy()
            ref._get_penalty_native(arrays["kmers"], scored, arrays["record_offsets"],
                                    np.asarray(is_targets, np.bool_), 1)
            arrays["nodes_scored"] = scored
        f = out / f"{name}.npz"
        if f.exists():   # (an existing vector is checked, not rewritten: the archives are not byte-reproducible)
            old = np.load(f)
            assert sorted(old.files) == sorted(arrays) and all(np.array_equal(old[k_], arrays[k_]) for k_ in arrays), name
        else:
            np.savez_compressed(f, **arrays)
        manifest["cases"].append(dict(name=name, paths=[str(p.relative_to(HERE)) for p in paths], k=k, w=w,
                                      is_targets=is_targets, ids=ids,
                                      n_kmers=int(len(arrays["kmers"])), n_nodes=int(len(arrays["nodes"])),
                                      n_edges=int(len(arrays["edges"]))))

    for k, w in SMOKE_KW:
        add(f"smoke_k{k}_w{w}", SMOKE, k, w, [True, True, False, False])
    for k, w in [(21, 200), (15, 20), (31, 64), (19, 33)]:
        add(f"pan_k{k}_w{w}", sets["pan"], k, w, [True, True, True, False, False, False])
    for k, w in [(21, 200), (11, 5), (17, 10), (5, 1), (33, 40)]:
        add(f"edge_k{k}_w{w}", sets["edge"], k, w, [True, False, True, False])
    for k in HASH_K + HASH_K_WIDE:  # w = 1: every valid k-mer is its own minimizer
        add(f"hash_k{k}", [sets["edge"][0], SMOKE[0]], k, 1)
    for k, w in LONG_KW:
        add(f"long_k{k}_w{w}", sets["long"], k, w, [True, False, True])

    # (iv) operator-level known answers, inputs as in the reference's test_graph.py
    K, N = oracle.KMER_DTYPE, oracle.NODE_DTYPE
    kmers = np.array([(0, 0), (1, 0), (2, 1), (3, 2), (4, 4), (5, 2), (6, 3), (7, 5), (8, 6), (9, 4)], dtype=K)
    nodes = np.array([(10, 0, 5, 0, 0, 0.0), (20, 5, 7, 0, 0, 0.0), (30, 7, 9, 0, 0, 0.0), (40, 9, 10, 0, 0, 0.0),
                      (50, 10, 10, 9, 9, 9.0), (60, 5, 9, 0, 0, 0.0)], dtype=N)
    offs = np.array([0, 2, 4, 5, 7], dtype=np.uint32)
    tar = np.array([True, False, True, False])
    scored = nodes.copy()
    ref._get_penalty_native(kmers, scored, offs, tar, 1)
    fk = np.array([(10, 0), (11, 0), (20, 1), (30, 2), (31, 2), (32, 2)], dtype=K)
    fn = np.array([(10, 0, 2, 1, 0, 0.1), (20, 2, 3, 1, 0, 0.2), (30, 3, 6, 1, 1, 0.3)], dtype=N)
    fk2, fn2 = ref._filter_kmers_native(fk, fn, [30, 10])
    np.savez_compressed(out / "operators.npz", pen_kmers=kmers, pen_nodes=nodes, pen_offsets=offs, pen_targets=tar,
                        pen_scored=scored, flt_kmers=fk, flt_nodes=fn, flt_used=np.array([30, 10], np.uint64),
                        flt_kmers_out=fk2, flt_nodes_out=fn2)
    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1))
    print(f"wrote {len(manifest['cases'])} cases")


if __name__ == "__main__":
    main()
