"""CPU: the C-ABI library loads, exports every symbol include/seqwin_hip.h declares, keeps the
reference's wire formats, validates arguments before touching a device, refuses to run without one,
and its host FASTA reader / 2-bit packer agrees with the oracle's reader byte for byte."""
import ctypes
import gzip
import random
import re
from pathlib import Path

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, ROOT
from seqwin_amd import EDGE_DTYPE, KMER_DTYPE, NODE_DTYPE, KmerGraph, _core, _get_penalty, _filter_kmers
from seqwin_amd._lib import c_u64, c_vp, check, lib

HEADER = (ROOT / "include" / "seqwin_hip.h").read_text()
NO_GPU = lib.sw_device_count() == 0


def test_every_declared_symbol_is_exported():
    names = sorted(set(re.findall(r"\b(sw_[a-z_0-9]+)\s*\(", HEADER)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/seqwin_hip.h but not exported"
    assert b"gfx950" in lib.sw_version()


def test_ctypes_prototypes_follow_the_header():
    """seqwin_amd/_abi.py (restype + argtypes of every entry point, applied once by _lib._load) is generated from the header:
    the committed table must be what scripts/gen_abi.py makes of include/seqwin_hip.h today, cover exactly the exported
    symbols, and be in force -- a bare Python int converts to uint64_t (no hand-written c_uint64 wrap to forget, VERDICT r4
    weak #9), a wrong argument count raises instead of corrupting the call -- and importing the package does not import torch."""
    import subprocess
    import sys

    from seqwin_amd._abi import PROTOTYPES
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_abi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    declared = set(re.findall(r"\b(sw_[a-z_0-9]+)\s*\(", HEADER))
    assert set(PROTOTYPES) == declared
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        assert fn.restype == restype and list(fn.argtypes) == argtypes, name
    with pytest.raises(TypeError):
        lib.sw_set_device()                      # too few arguments
    with pytest.raises(ctypes.ArgumentError):
        lib.sw_index_sizes(1.5, None, None, None)   # a float is no handle
    # (2^40 as a bare int: converted to uint64_t, so the library sees k = 2^40 and answers with its own ValueError for k > 65535)
    arr = (ctypes.c_char_p * 1)()
    g = c_vp()
    assert lib.sw_build(arr, 0, 1 << 40, 10, 1, 0, ctypes.byref(g)) == 2 and b"65535" in lib.sw_last_error()
    r = subprocess.run([sys.executable, "-c", "import sys; import seqwin_amd; from seqwin_amd import _lib; "
                        "print('torch' in sys.modules, _lib.HIP_RUNTIME)"], capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0 and r.stdout.split()[0] == "False", (r.stdout, r.stderr)


def test_dtype_layouts():
    # reference tests/smoke/test_graph.py:45-64
    assert KMER_DTYPE.itemsize == 8 and KMER_DTYPE.names == ("pos", "record_idx")
    assert NODE_DTYPE.itemsize == 40 and NODE_DTYPE.names == ("hash", "start", "stop", "n_tar", "n_neg", "penalty")
    assert NODE_DTYPE["start"] == np.dtype(np.uintp) and NODE_DTYPE["n_tar"] == np.dtype(np.uint32)
    assert EDGE_DTYPE.itemsize == 24 and [EDGE_DTYPE.fields[f][1] for f in ("first", "second", "weight")] == [0, 8, 16]


def test_argument_validation_happens_without_a_device(smoke_paths):
    with pytest.raises(ValueError):
        KmerGraph(smoke_paths, kmerlen=2, windowsize=10)
    with pytest.raises(ValueError):
        KmerGraph(smoke_paths, kmerlen=21, windowsize=0)
    with pytest.raises(TypeError):
        KmerGraph(smoke_paths, kmerlen=7, windowsize=10, is_targets=[True, False])   # test_graph.py:130-141
    with pytest.raises(TypeError):
        _core._build_native("not-a-list", 7, 10)
    with pytest.raises(TypeError):
        _core._build_native([], 7.5, 10)
    kmers = np.zeros(3, KMER_DTYPE); nodes = np.zeros(1, NODE_DTYPE); nodes["stop"] = 3
    offs = np.array([0, 1, 2], np.uint32)
    ro = nodes.copy(); ro.flags.writeable = False
    with pytest.raises(ValueError, match="writable"):
        _get_penalty(kmers, ro, offs, [True, False])
    with pytest.raises(TypeError):
        _get_penalty(kmers, nodes, offs.astype(np.uint64), [True, False])          # test_graph.py:318-321
    with pytest.raises(TypeError):
        _get_penalty(kmers.astype([("pos", "<u8"), ("record_idx", "<u8")]), nodes, offs, [True, False])
    with pytest.raises(ValueError):
        _get_penalty(kmers, nodes, offs[:-1], [True, False])
    with pytest.raises(ValueError):
        _get_penalty(kmers, nodes, np.array([1, 2, 3], np.uint32), [True, False])
    with pytest.raises(ValueError):
        _get_penalty(kmers, nodes, np.array([0, 3, 2], np.uint32), [True, False])
    with pytest.raises(ValueError):
        _get_penalty(kmers, nodes, offs, [True, True])
    with pytest.raises(ValueError):
        _get_penalty(kmers, nodes, offs, [False, False])
    with pytest.raises(ValueError):
        _get_penalty(kmers, nodes, offs, np.array([[True, False, True]]))    # shape[0] != len(offsets) - 1


def test_used_hashes_reach_the_library_as_the_pybind_caster_fills_them():
    """_core._hashes_array (the std::vector<uint64_t> of python_bindings.cpp:137-150): the one-go NumPy conversion gives what the
    per-element check gives -- for sets of np.uint64 (what kmers.py:312 passes), Python ints up to 2^64 - 1, mixed integer types,
    bools, empty iterables, generators -- and floats, negatives, too large values and non-integers raise what the per-element check
    raises (the caster's errors)."""
    def per_element(items):
        return np.fromiter((_core._size_t(h, "used_hashes") for h in items), dtype=np.uint64, count=len(items))

    rng = np.random.default_rng(9)
    good = [
        frozenset(np.uint64(x) for x in rng.integers(0, 2**63, 500, dtype=np.uint64) * 2 + 1),
        [int(x) for x in rng.integers(0, 2**62, 300)],
        [2**64 - 1, 0, 5],
        [np.uint32(7), np.int64(9), 11],
        (np.uint64(1),),
        [True, False, 3],
        [],
        set(),
    ]
    for items in good:
        lst = list(items)
        got = _core._hashes_array(iter(lst))      # (an iterator: consumed once)
        assert got.dtype == np.uint64 and got.flags.c_contiguous and np.array_equal(got, per_element(lst))
    for bad in ([1.5, 2], [-1, 2], [2**64, 1], ["7"], [None], [np.float32(2)], [[1, 2], [3, 4]], [np.int64(-5)]):
        with pytest.raises(Exception) as want:
            per_element(bad)
        with pytest.raises(type(want.value)):
            _core._hashes_array(bad)
    with pytest.raises(TypeError):
        _core._hashes_array(5)


def test_ids_by_assembly_from_the_exported_blob():
    """_core._split_ids: the fifth element of _build_native's tuple (python_bindings.cpp:73-80: a list with one tuple of str per
    assembly) from the NUL-terminated id blob and the record offsets sw_graph_export fills -- against the plain per-id loop, with
    assemblies without records, no assembly at all, ids that are not ASCII; an id that is not UTF-8 raises UnicodeDecodeError as
    pybind11's str cast does; the cyclic GC is left as it was found."""
    import gc

    def plain(blob, offs):
        names = blob.split(b"\0")[:-1] if blob else []
        return [tuple(s.decode("utf-8") for s in names[int(offs[a]):int(offs[a + 1])]) for a in range(len(offs) - 1)]

    rng = random.Random(5)
    for case in range(40):
        n_asm = rng.choice([0, 1, 2, 7, 300])
        counts = [rng.choice([0, 0, 1, 2, 5, 60]) for _ in range(n_asm)]
        ids = [("".join(rng.choice("abcXYZ_.|0189é糖") for _ in range(rng.randint(1, 12)))).encode("utf-8") for _ in range(sum(counts))]
        blob = b"".join(i + b"\0" for i in ids)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
        got = _core._split_ids(blob, offs)
        assert got == plain(blob, offs) and len(got) == n_asm
        assert all(type(t) is tuple and all(type(s) is str for s in t) for t in got)
    assert _core._split_ids(b"", np.zeros(1, np.uint32)) == []
    for enabled in (True, False):
        (gc.enable if enabled else gc.disable)()
        try:
            _core._split_ids(b"a\0b\0", np.array([0, 1, 2], np.uint32))
            assert gc.isenabled() == enabled
            with pytest.raises(UnicodeDecodeError):
                _core._split_ids(b"ok\0\xff\xfe\0", np.array([0, 2], np.uint32))
            assert gc.isenabled() == enabled
        finally:
            gc.enable()


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(smoke_paths):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KmerGraph(smoke_paths, kmerlen=21, windowsize=200)
    kmers = np.zeros(1, KMER_DTYPE); nodes = np.zeros(1, NODE_DTYPE); nodes["stop"] = 1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _get_penalty(kmers, nodes, np.array([0, 1, 1], np.uint32), [True, False])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _filter_kmers(kmers, nodes, {1})


def _host_ingest(paths, n_cpu=1):
    arr = (ctypes.c_char_p * max(len(paths), 1))(*[str(p).encode() for p in paths])
    hb = c_vp()
    check(lib.sw_host_ingest(arr, ctypes.c_size_t(len(paths)), c_u64(n_cpu), ctypes.byref(hb)))
    try:
        v = [c_u64() for _ in range(5)]
        check(lib.sw_hostbatch_info(hb, *[ctypes.byref(x) for x in v]))
        na, nr, bp, nb, _ = (x.value for x in v)
        offs = np.empty(na + 1, np.uint32); lens = np.empty(nr, np.uint32); blob = ctypes.create_string_buffer(max(nb, 1))
        check(lib.sw_hostbatch_tables(hb, offs.ctypes.data_as(c_vp), blob, lens.ctypes.data_as(c_vp)))
        ids = blob.raw[:nb].split(b"\0")[:-1] if nb else []
        seqs = []
        for r in range(nr):
            buf = ctypes.create_string_buffer(max(int(lens[r]), 1)); n = c_u64()
            check(lib.sw_hostbatch_record(hb, c_u64(r), buf, c_u64(int(lens[r])), ctypes.byref(n)))
            seqs.append(buf.raw[:n.value])
        return offs, [i.decode() for i in ids], seqs, bp
    finally:
        lib.sw_hostbatch_free(hb)


def _canon(seq: bytes) -> bytes:
    t = bytearray(b"N" * 256)
    for a, b in zip(b"ACGTUacgtu", b"ACGTTACGTT"):
        t[a] = b
    return seq.translate(bytes(t))


def _line_length_file(tmp_path):
    """Sequence lines of every length 1 .. 200 (whole 32- and 64-byte steps, masked tails, the split of a step into two
    32-base blocks), every 7th line with invalid bases at both ends and in the middle, every 11th with a blank inside."""
    rng = random.Random(5)
    out = [b">lens"]
    for n in list(range(1, 201)) + [255, 256, 257, 1000]:
        line = bytearray(rng.choice(b"ACGTacgtU") for _ in range(n))
        if n % 7 == 0:
            line[0] = ord("N"); line[-1] = ord("n"); line[n // 2] = ord("R")
        if n % 11 == 0:
            line[n // 3] = ord(" ")
        out.append(bytes(line))
        if n % 13 == 0:
            out.append(b">r%d x" % n)
    p = tmp_path / "line_lengths.fa"
    p.write_bytes(b"\n".join(out) + b"\n")
    return p


def test_host_ingest_matches_oracle_reader(tmp_path):
    files = sorted((GOLDEN / "synth").glob("*")) + sorted((GOLDEN / "smoke").glob("*/*.fasta")) + [_line_length_file(tmp_path)]
    tricky = tmp_path / "tricky.fa"
    tricky.write_bytes(b"\n  \n>id1 desc more\r\nACGT acgt\tNN\r\n\r\n>id2\n>id3\tx\nAC\x0bGT\n  GG  \n>id4\nACGTNRYKMacgtnU-*\nTTTT")
    gz = tmp_path / "t.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(b">g1\nACGTACGTAC\nGGGG\n>g2\nNNNN\n")
    files += [tricky, gz]
    # fixed-width lines of 64 ... 129 columns: every 64-byte step of the chunk packer holds no or exactly one line end, at every
    # position of the step as the file goes on (r06: one line end is taken out by a byte permutation, the step goes to the packer as
    # one 64-base chunk); 1 % invalid bases and lower case around them, so that chunks with gaps take the 32-base blocks
    rng = random.Random(17)
    for width in (64, 65, 66, 79, 80, 81, 96, 100, 127, 128, 129):
        seq = bytearray(rng.choice(b"ACGT") for _ in range(40_000 + width))
        for _ in range(len(seq) // 100):
            seq[rng.randrange(len(seq))] = rng.choice(b"NnRacgtu")
        pth = tmp_path / f"w{width}.fa"
        pth.write_bytes(b">w%d\n" % width + b"\n".join(bytes(seq[i:i + width]) for i in range(0, len(seq), width)) + b"\n>tail\nACGT")
        files.append(pth)
    for n_cpu in (1, 3):
        offs, ids, seqs, bp = _host_ingest(files, n_cpu)
        exp_ids, exp_seqs, exp_offs = [], [], [0]
        for f in files:
            recs = oracle.read_fasta(f)
            exp_ids += [r[0] for r in recs]; exp_seqs += [_canon(r[1]) for r in recs]
            exp_offs.append(len(exp_ids))
        assert ids == exp_ids and offs.tolist() == exp_offs
        assert seqs == exp_seqs
        assert bp == sum(len(s) for s in exp_seqs)


def test_host_ingest_gz_files_of_every_kind(tmp_path):
    """The .gz route of the host ingest (fast_inflate.hpp; zlib's gzread loop for whatever it declines) over a set of files that are
    ALL gzip-named: the oracle's reader's tables for files of very different sizes side by side, several members, stored and fixed
    blocks, empty text, a file that is not gzip at all (zlib's transparent read) -- and a wrong CRC, a truncated file and a flipped
    bit at every place of the list: what zlib's gzread loop yields (what could be inflated, fasta_reader.cpp:134-150) or a clean
    error, the neighbours untouched.  (r06 ran this set through a two-files-per-worker decoder as well; it is gone, NOTES.md.)"""
    import zlib
    rng = np.random.default_rng(41)
    files, texts = [], []

    def text(n_rec, n):
        out = []
        for r in range(n_rec):
            seq = rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), n, p=[.24, .24, .24, .24, .005, .005, .005, .005, .02]).tobytes()
            out.append(b">rec%d x\n" % r + b"\n".join(seq[i:i + 80] for i in range(0, n, 80)) + b"\n")
        return b"".join(out)

    def add(name, payload, raw=None):
        pth = tmp_path / name
        pth.write_bytes(raw if raw is not None else gzip.compress(payload, 6))
        files.append(pth); texts.append(payload)

    add("a0.fa.gz", text(3, 200_000))
    add("a1.fa.gz", text(1, 50))                               # (one side ends long before the other)
    add("a2.fa.gz", b"")                                       # empty text
    add("a3.fa.gz", text(2, 70_000), raw=gzip.compress(text(0, 0) + b">m1\nACGT\n", 1) + gzip.compress(b">m2\nGGCC\n", 9))
    texts[-1] = b">m1\nACGT\n>m2\nGGCC\n"
    t = text(2, 30_000)
    co = zlib.compressobj(0, zlib.DEFLATED, 31); add("a4.fa.gz", t, raw=co.compress(t) + co.flush())      # stored blocks
    t = text(1, 3_000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED); add("a5.fa.gz", t, raw=co.compress(t) + co.flush())
    t = text(1, 90_000)
    add("a6.fa.gz", t, raw=t)                                  # not gzip: read as it is, like gzread does
    add("a7.fa.gz", text(4, 120_000))
    add("a8.fa.gz", text(1, 1_000_000))                        # (odd count: the last file goes alone)
    exp_ids, exp_seqs, exp_offs = [], [], [0]
    for f in files:
        recs = oracle.read_fasta(f)
        exp_ids += [r[0] for r in recs]; exp_seqs += [_canon(r[1]) for r in recs]
        exp_offs.append(len(exp_ids))
    for n_cpu in (1, 3):
        offs, ids, seqs, bp = _host_ingest(files, n_cpu)
        assert ids == exp_ids and offs.tolist() == exp_offs
        assert seqs == exp_seqs
    # damaged files at every place of the list
    good = files[0]
    blob = bytearray(gzip.compress(text(2, 40_000), 6))
    crc = tmp_path / "crc.fa.gz"; bad = bytearray(blob); bad[-6] ^= 0x40; crc.write_bytes(bad)
    cut = tmp_path / "cut.fa.gz"; cut.write_bytes(blob[:len(blob) // 2])
    flip = tmp_path / "flip.fa.gz"; bad = bytearray(blob); bad[len(bad) // 3] ^= 0x10; flip.write_bytes(bad)
    for damaged in (crc, cut, flip):
        for order in ([damaged, good, good], [good, damaged, good], [good, good, damaged]):
            try:
                want = [_canon(r[1]) for r in oracle.read_fasta(damaged)]
            except (RuntimeError, ValueError):
                want = None
            k = order.index(damaged)
            try:
                offs, ids, seqs, bp = _host_ingest(order, 2)
            except (RuntimeError, ValueError):
                assert want is None or damaged is not cut    # (a truncated file is read as far as it goes, by both)
                continue
            good_seqs = exp_seqs[exp_offs[0]:exp_offs[1]]
            for j, f in enumerate(order):
                mine = seqs[offs[j]:offs[j + 1]]
                if f is good:
                    assert mine == good_seqs
                elif want is not None:
                    assert mine == want
            if damaged is cut:
                assert want is not None


def test_host_ingest_errors(tmp_path):
    with pytest.raises(RuntimeError, match="Unable to open FASTA"):
        _host_ingest([tmp_path / "missing.fa"])
    with pytest.raises(RuntimeError, match="Unable to open gzip FASTA"):
        _host_ingest([tmp_path / "missing.fa.gz"])
    bad = tmp_path / "bad.fa"; bad.write_text("ACGT\n>r\nACGT\n")
    with pytest.raises(RuntimeError, match="sequence encountered before header"):
        _host_ingest([bad])
    ctl = tmp_path / "ctl.fa"; ctl.write_bytes(b">r\nACGT\x01ACGT\n")
    with pytest.raises(ValueError, match="control byte"):
        _host_ingest([ctl])
    assert _host_ingest([])[0].tolist() == [0]
    empty = tmp_path / "empty.fa"; empty.write_text("")
    assert _host_ingest([empty])[0].tolist() == [0, 0]


def _byte_soup_files(tmp_path, seed=7, n_files=12):
    """Adversarial FASTA: every byte value except the refused control bytes, blanks inside lines, CRLF, lines of every
    length around the packer's 32-byte step, records shorter than a step, '>' inside sequence lines."""
    rng = np.random.default_rng(seed)
    refused = {1, 3, 4, 5, 7}
    any_byte = np.array([b for b in range(1, 256) if b not in refused and b != 10], np.uint8)   # (the oracle wrapper returns NUL-terminated strings)
    files = []
    for f in range(n_files):
        parts = []
        for r in range(int(rng.integers(1, 6))):
            parts.append(b">rec%d some description\n" % r)
            n = int(rng.choice([0, 1, 31, 32, 33, 63, 64, 65, 500, 5000, 40000]))
            style = rng.random()
            if style < 0.5:
                seq = rng.choice(np.frombuffer(b"ACGTacgtUuNn", np.uint8), n)
            elif style < 0.8:
                seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
                k = max(1, n // 200)
                if n:
                    seq[rng.integers(0, n, k) % n] = rng.choice(any_byte, k)
            else:
                seq = rng.choice(any_byte, n)
            width = int(rng.choice([1, 7, 31, 32, 33, 60, 64, 80, 100, 100000]))
            eol = [b"\n", b"\r\n", b" \n", b"\t\r\n"][int(rng.integers(0, 4))]
            raw = seq.tobytes()
            for i in range(0, len(raw), width):
                line = raw[i:i + width]
                if line[:1] == b">":          # would start a new record: keep it a sequence line
                    line = b" " + line
                parts.append(line + eol)
            if rng.random() < 0.3:
                parts.append(b"\n   \n")
        p = tmp_path / f"rnd{seed}_{f}.fa"
        p.write_bytes(b"".join(parts))
        files.append(p)
    return files


def test_host_ingest_random_bytes_match_oracle_reader(tmp_path):
    """The SIMD packer (32 bytes per step, low-nibble classification) against the oracle's reader on adversarial content."""
    files = _byte_soup_files(tmp_path)
    for n_cpu in (1, 4):
        offs, ids, seqs, bp = _host_ingest(files, n_cpu)
        exp_ids, exp_seqs, exp_offs = [], [], [0]
        for f in files:
            recs = oracle.read_fasta(f)
            exp_ids += [r[0] for r in recs]; exp_seqs += [_canon(r[1]) for r in recs]
            exp_offs.append(len(exp_ids))
        assert ids == exp_ids and offs.tolist() == exp_offs
        assert seqs == exp_seqs
        assert bp == sum(len(s) for s in exp_seqs)


def test_host_ingest_across_line_ends_matches_oracle_reader(tmp_path):
    """The 64-bytes-per-step packer that squeezes line ends out in the register (r05, AVX-512 VBMI2 hosts; elsewhere this runs the
    line packer) against the oracle's reader: mostly clean sequence text in lines of every width, so that its steps start at
    every offset inside a line, with the bytes that end a run of steps -- '>', blanks, CR, tabs -- and the ones that do not --
    lower case, N, IUPAC, bytes >= 0x80 -- sprinkled in, blank lines, headers right behind a full step, no final newline."""
    rng = np.random.default_rng(23)
    rare = np.frombuffer(b">> \r\tNnacgtuURYK-*\x7f\x80\xff\x0b", np.uint8)
    files = []
    for f in range(160):
        parts = []
        for r in range(int(rng.integers(1, 5))):
            parts.append(b">r%d d\n" % r if rng.random() < 0.8 else b">r%d\r\n" % r)
            n = int(rng.choice([0, 63, 64, 65, 127, 128, 129, 300, 1000, 3000]))
            seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
            if n and rng.random() < 0.7:
                k = int(rng.integers(1, 2 + n // 100))
                seq[rng.integers(0, n, k)] = rng.choice(rare, k)
            width = int(rng.integers(1, 150)) if rng.random() < 0.7 else int(rng.choice([60, 64, 70, 80, 128]))
            raw = seq.tobytes()
            for i in range(0, len(raw), width):
                line = raw[i:i + width]
                if line[:1] == b">":
                    line = b"A" + line
                parts.append(line + (b"\n" if rng.random() < 0.97 else b"\n\n" if rng.random() < 0.5 else b"\r\n"))
        blob = b"".join(parts)
        if rng.random() < 0.3:
            blob = blob.rstrip(b"\r\n")
        p = tmp_path / f"chunk_{f}.fa"
        p.write_bytes(blob)
        files.append(p)
    # several read blocks (256 KiB each) with the block ends at every offset inside a line, one line longer than a block
    big = [b">big\n"]
    for width in (79, 80, 61, 300_000, 1):
        raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), 300_007 if width > 1 else 500, p=[.24, .24, .24, .24, .04]).tobytes()
        big += [raw[i:i + width] + b"\n" for i in range(0, len(raw), width)] + [b">next %d\n" % width]
    p = tmp_path / "blocks.fa"
    p.write_bytes(b"".join(big))
    files.append(p)
    offs, ids, seqs, bp = _host_ingest(files, 3)
    exp_ids, exp_seqs, exp_offs = [], [], [0]
    for f in files:
        recs = oracle.read_fasta(f)
        exp_ids += [r[0] for r in recs]; exp_seqs += [_canon(r[1]) for r in recs]
        exp_offs.append(len(exp_ids))
    assert ids == exp_ids and offs.tolist() == exp_offs
    assert seqs == exp_seqs


def test_host_ingest_under_sanitizers(tmp_path):
    """host_ingest.cpp (hand-written SIMD over hostile bytes, threads, zlib streams) rebuilt with AddressSanitizer + UBSan
    (`make -C seqwin_amd/csrc asan`) and run on the byte soup, gzip members, truncated gzip, blank / header-only files and
    refused inputs: no sanitizer report, and the same records as the regular library.  CPU only."""
    import subprocess
    if subprocess.run(["make", "-C", str(ROOT / "seqwin_amd" / "csrc"), "asan"], capture_output=True).returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    exe = ROOT / "seqwin_amd" / "csrc" / "build" / "ingest_san"
    files = _byte_soup_files(tmp_path, seed=11, n_files=8)
    gz = tmp_path / "two_members.fa.gz"
    gz.write_bytes(gzip.compress(b">m1\nACGTNACGT\nAC") + gzip.compress(b"GT\n>m2 x\n" + b"ACGT" * 9000 + b"\n"))
    hdr = tmp_path / "headers_only.fa"; hdr.write_bytes(b">a\n>b\r\n>c")
    blank = tmp_path / "blank.fa"; blank.write_bytes(b"\n\n   \n")
    # a truncated gzip stream yields what could be inflated, as with the reference's gzread loop (fasta_reader.cpp:134-150)
    trunc = tmp_path / "trunc.fa.gz"
    trunc.write_bytes(gzip.compress(b">t\n" + b"ACGTTGCA" * 5000 + b"\n>u\nACGTACGTACGTACGTACGTAAAA\n")[:-10])
    assert [(i, len(q)) for i, q in oracle.read_fasta(trunc)] == [("t", 40000), ("u", 21)]
    files += [gz, hdr, blank, trunc] + sorted((GOLDEN / "synth").glob("edge_*"))
    env = dict(__import__("os").environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    files.append(_line_length_file(tmp_path))
    # the packer a host offers by default (64 bytes per step with AVX-512 BW, else 32 with AVX2), the 32-byte one, the byte loop
    # (r05: by default, with AVX-512 VBMI2, 64 bytes per step across line ends; SEQWIN_AMD_LINE_PACKER keeps the line-by-line form)
    # SEQWIN_AMD_READ_BLOCK_KB=1: plain files go through the parser in 1 KiB blocks (default 256 KiB: most of these files are one block)
    for n_cpu, scalar in ((1, ""), (3, ""), (3, "SEQWIN_AMD_READ_BLOCK_KB"), (2, "SEQWIN_AMD_LINE_PACKER"), (2, "SEQWIN_AMD_NO_AVX512"),
                          (2, "SEQWIN_AMD_SCALAR_INGEST")):
        e = dict(env, **({scalar: "1"} if scalar else {}))
        dump = tmp_path / f"dump_{n_cpu}_{scalar}.bin"
        out = subprocess.run([str(exe), str(n_cpu), str(dump)] + [str(f) for f in files], capture_output=True, text=True, env=e)
        assert out.returncode == 0 and "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
        offs, ids, seqs, bp = _host_ingest(files, n_cpu)
        blob = b"".join(i.encode() + b"\0" for i in ids)
        lens = np.array([len(x) for x in seqs], np.uint32)
        assert dump.read_bytes() == offs.astype(np.uint32).tobytes() + blob + lens.tobytes() + b"".join(seqs)
        assert f"{len(files)} assemblies {len(seqs)} records {bp} bp" in out.stdout
    # the streaming form sw_build uses (r05: word buffers from an arena, chunks to a sink in order, the parsers inside their window of
    # the sink) with a host-memory sink, under ASan / UBSan and -- where the runtime is there -- ThreadSanitizer: the same dump
    ref = tmp_path / "dump_1_.bin"
    for n_cpu, extra in ((4, {}), (3, {"SEQWIN_AMD_INGEST_WINDOW": "1"}), (2, {"SEQWIN_AMD_READ_BLOCK_KB": "1"})):
        dump = tmp_path / f"dump_sink_{n_cpu}.bin"
        out = subprocess.run([str(exe), str(n_cpu), str(dump)] + [str(f) for f in files], capture_output=True, text=True,
                             env=dict(env, INGEST_SAN_SINK="1", **extra))
        assert out.returncode == 0 and "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
        assert dump.read_bytes() == ref.read_bytes()
    if subprocess.run(["make", "-C", str(ROOT / "seqwin_amd" / "csrc"), "tsan"], capture_output=True).returncode == 0:
        texe = ROOT / "seqwin_amd" / "csrc" / "build" / "ingest_tsan"
        for n_cpu, extra in ((6, {}), (5, {"SEQWIN_AMD_INGEST_WINDOW": "2"})):
            dump = tmp_path / f"dump_tsan_{n_cpu}.bin"
            out = subprocess.run([str(texe), str(n_cpu), str(dump)] + [str(f) for f in files], capture_output=True, text=True,
                                 env=dict(__import__("os").environ, INGEST_SAN_SINK="1", TSAN_OPTIONS="halt_on_error=0", **extra))
            assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
            assert dump.read_bytes() == ref.read_bytes()
    # every input a .gz file (levels 1-9), with the sink and without; good, several-member, truncated and bit-flipped files side by side
    gzs = []
    for j, f in enumerate(files[:9]):
        g = tmp_path / f"allgz_{j}.fa.gz"
        blob = gzip.compress(f.read_bytes(), 1 + j % 9) if f.suffix != ".gz" else f.read_bytes()
        if j == 3:
            blob = blob[:len(blob) * 2 // 3]
        if j == 6:
            blob = blob[:len(blob) // 2] + bytes([blob[len(blob) // 2] ^ 4]) + blob[len(blob) // 2 + 1:]
        g.write_bytes(blob)
        gzs.append(g)
    gzs = gzs + [gz, trunc]
    gz_dumps = []
    for n_cpu, extra in ((1, {}), (3, {}), (4, {"INGEST_SAN_SINK": "1"}), (3, {"INGEST_SAN_SINK": "1", "SEQWIN_AMD_INGEST_WINDOW": "1"})):
        dump = tmp_path / f"dump_allgz_{n_cpu}_{len(extra)}.bin"
        out = subprocess.run([str(exe), str(n_cpu), str(dump)] + [str(f) for f in gzs], capture_output=True, text=True,
                             env=dict(env, **extra))
        if out.returncode == 3:     # (the bit flip reached the text as a refused control byte: nothing to compare, but no report either)
            assert "refused" in out.stderr and "Sanitizer" not in out.stderr, out.stderr[-2000:]
            continue
        assert out.returncode == 0 and "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
        gz_dumps.append(dump.read_bytes())
    if gz_dumps:
        offs, ids, seqs, bp = _host_ingest(gzs, 2)
        want = offs.astype(np.uint32).tobytes() + b"".join(i.encode() + b"\0" for i in ids) + np.array([len(x) for x in seqs], np.uint32).tobytes() + b"".join(seqs)
        assert all(d == want for d in gz_dumps)
        if (ROOT / "seqwin_amd" / "csrc" / "build" / "ingest_tsan").exists():
            dump = tmp_path / "dump_allgz_tsan.bin"
            out = subprocess.run([str(ROOT / "seqwin_amd" / "csrc" / "build" / "ingest_tsan"), "6", str(dump)] + [str(f) for f in gzs], capture_output=True,
                                 text=True, env=dict(__import__("os").environ, INGEST_SAN_SINK="1", TSAN_OPTIONS="halt_on_error=0"))
            assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
            assert dump.read_bytes() == want
    # refused / broken inputs must fail cleanly under the sanitizers too (exit code 3, no report)
    ctl = tmp_path / "ctl.fa"; ctl.write_bytes(b">r\nACGT\x01ACGT\n")
    nohdr = tmp_path / "nohdr.fa"; nohdr.write_bytes(b"ACGT\n>r\nAC\n")
    notgz = tmp_path / "notgz.fa.gz"; notgz.write_bytes(b"\x1f\x8b\x08\x00" + bytes(range(200)))
    for bad in (ctl, nohdr, notgz, tmp_path / "missing.fa"):
        out = subprocess.run([str(exe), "2", str(tmp_path / "x.bin"), str(files[0]), str(bad)], capture_output=True, text=True, env=env)
        assert out.returncode == 3 and "refused" in out.stderr and "Sanitizer" not in out.stderr, (bad.name, out.returncode, out.stderr[-2000:])


def test_device_gz_decoder_and_parser_under_sanitizers(tmp_path):
    """The device gzip ingest's DEFLATE decoder and FASTA parser / packer (seqwin_amd/csrc/gz_dev.hpp -- the code the kernels
    of ingest_dev.hip run one file per lane) compiled for the HOST with AddressSanitizer + UBSan
    (`make -C seqwin_amd/csrc gzsan`), on buffers laid out and sized as in HBM: byte soup at every compression level and
    strategy, multi-block streams with flush points, all gzip header fields -> the records, ids and bases of the host reader
    (replaces the gzip branch of fasta_reader.cpp:109-203 and the parse of :41-95); truncated, bit-flipped, padded and
    mis-sized streams, several members, control bytes, sequence before a header -> declined (the library then takes the
    host route), never a sanitizer report, never an accepted stream whose text differs from zlib's.  CPU only."""
    import subprocess
    import zlib
    if subprocess.run(["make", "-C", str(ROOT / "seqwin_amd" / "csrc"), "gzsan"], capture_output=True).returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    exe = ROOT / "seqwin_amd" / "csrc" / "build" / "gz_dev_san"
    env = dict(__import__("os").environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    rng = __import__("random").Random(23)

    def member(text: bytes, i: int) -> bytes:
        level, strat = rng.choice(range(10)), rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
        co = zlib.compressobj(level, zlib.DEFLATED, -rng.choice([9, 12, 15]), rng.choice([1, 8, 9]), strat)
        body, at = b"", 0
        while at < len(text):                      # several blocks, empty stored blocks at the flush points
            step = rng.choice([1, 100, 5000, 10**6])
            body += co.compress(text[at:at + step])
            at += step
            if rng.random() < 0.3:
                body += co.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        body += co.flush()
        flg = [0, 8, 2 | 4 | 8 | 16][i % 3]
        head = bytes([0x1F, 0x8B, 8, flg, 0, 0, 0, 0, 0, 255])
        if flg & 4:
            head += (5).to_bytes(2, "little") + b"extra"
        if flg & 8:
            head += b"name.fa\0"
        if flg & 16:
            head += b"a comment\0"
        if flg & 2:
            head += (zlib.crc32(head) & 0xFFFF).to_bytes(2, "little")
        return head + body + zlib.crc32(text).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little")

    plain = _byte_soup_files(tmp_path, seed=29, n_files=20)
    extra = [b"", b">only_header", b">a\n>b\r\n>c", b"\n\n   \n", b">r\n" + b"ACGT" * 70000 + b"\n", b">hp\n" + b"A" * 100000 + b"\n>at\n" + b"AT" * 30000,
             b">x desc\r\n" + bytes(rng.choice(b"ACGTN") for _ in range(3000)) * 40 + b"\r\n"]
    for j, t in enumerate(extra):
        q = tmp_path / f"extra{j}.fa"
        q.write_bytes(t)
        plain.append(q)
    gz = []
    for i, q in enumerate(plain):
        g = tmp_path / (q.name + ".gz")
        g.write_bytes(member(q.read_bytes(), i))
        assert gzip.decompress(g.read_bytes()) == q.read_bytes()
        gz.append(g)
    dump = tmp_path / "dump.bin"
    out = subprocess.run([str(exe), str(dump)] + [str(g) for g in gz], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    offs, ids, seqs, bp = _host_ingest(plain, 2)
    blob = b"".join(i.encode() + b"\0" for i in ids)
    lens = np.array([len(x) for x in seqs], np.uint32)
    assert dump.read_bytes() == offs.astype(np.uint32).tobytes() + blob + lens.tobytes() + b"".join(seqs)
    assert f"{len(gz)} assemblies {len(seqs)} records {bp} bp" in out.stdout

    # what must be declined (exit code 4) without a report
    good = gz[0].read_bytes()
    declined = {
        "two_members": gzip.compress(b">m1\nACGT\n") + gzip.compress(b">m2\nAC\n"),
        "control": member(b">r\nACGT\x01ACGT\n", 0),
        "nohdr": member(b"ACGT\n>r\nAC\n", 0),
        "notgz": b"\x1f\x8b\x08\x00" + bytes(range(200)),
        "reserved_flag": bytes([0x1F, 0x8B, 8, 0x20]) + good[4:],
        "bad_crc": good[:-8] + bytes([good[-8] ^ 1]) + good[-7:],
        "isize_short": good[:-4] + (max(1, int.from_bytes(good[-4:], "little")) - 1).to_bytes(4, "little"),
        "isize_long": good[:-4] + (int.from_bytes(good[-4:], "little") + 3).to_bytes(4, "little"),
    }
    for name, raw in declined.items():
        q = tmp_path / f"{name}.fa.gz"
        q.write_bytes(raw)
        out = subprocess.run([str(exe), str(tmp_path / "x.bin"), str(gz[1]), str(q)], capture_output=True, text=True, env=env)
        assert out.returncode == 4 and "declined" in out.stderr and "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, \
            (name, out.returncode, out.stderr[-2000:])
    # mutated streams: declined, or accepted with exactly the text zlib gives for them
    bases = [g.read_bytes() for g in gz if g.stat().st_size > 40][:10]
    for it in range(250):
        raw = bytearray(rng.choice(bases))
        hl, kind = 10, rng.randrange(3)             # (mutations stay behind the fixed part of the header)
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                raw[rng.randrange(hl, len(raw) - 8)] ^= 1 << rng.randrange(8)
        elif kind == 1:
            cut = rng.randrange(hl, len(raw) - 8)
            raw = raw[:cut] + raw[-8:]
        else:
            at = rng.randrange(hl, len(raw) - 8)
            raw[at:at] = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 20)))
        q = tmp_path / "mut.fa.gz"
        q.write_bytes(bytes(raw))
        out = subprocess.run([str(exe), str(tmp_path / "m.bin"), str(q)], capture_output=True, text=True, env=env)
        assert out.returncode in (0, 4) and "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, (it, out.returncode, out.stderr[-2000:])
        if out.returncode == 0:                     # (the CRC-32 matched: the mutation hit bytes that do not reach the text)
            try:
                text = gzip.decompress(bytes(raw))
            except Exception:
                text = None
            assert text is not None, it
            ref = tmp_path / "mut_ref.fa"
            ref.write_bytes(text)
            o2, i2, s2, _ = _host_ingest([ref], 1)
            b2 = b"".join(i.encode() + b"\0" for i in i2)
            assert (tmp_path / "m.bin").read_bytes() == o2.astype(np.uint32).tobytes() + b2 + np.array([len(x) for x in s2], np.uint32).tobytes() + b"".join(s2)


def test_host_inflate_against_zlib_under_sanitizers(tmp_path):
    """seqwin_amd/csrc/fast_inflate.hpp -- the whole-buffer gzip / DEFLATE decoder the host ingest uses for .gz files since r05
    (the gzip branch of fasta_reader.cpp:109-203; zlib's gzread loop stays as the route for everything unusual) -- against zlib,
    compiled with AddressSanitizer + UBSan (`make -C seqwin_amd/csrc finfsan`, tests/tools/finf_san_driver.cpp).  Contract: what
    it ACCEPTS is byte for byte what zlib delivers, every well-formed member made by zlib's deflate (levels 0 / 1 / 6 / 9, all
    strategies, several members, empty members, capacities exact to the byte) is accepted, and on flipped, truncated, extended
    or overwritten streams it never touches memory outside its buffers; its CRC-32 (PCLMULQDQ folding where the CPU has it)
    equals zlib's on random spans.  Then the library itself: the same .gz files through the ingest with the fast decoder and with
    SEQWIN_AMD_ZLIB_INFLATE=1 give the same batch, including members with header fields the fast decoder leaves to zlib."""
    import os
    import subprocess
    import sys
    import zlib
    if subprocess.run(["make", "-C", str(ROOT / "seqwin_amd" / "csrc"), "finfsan"], capture_output=True).returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    exe = ROOT / "seqwin_amd" / "csrc" / "build" / "finf_san"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (7, 8):
        r = subprocess.run([str(exe), str(seed), "1500"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        assert "cases:" in r.stdout and "crc32: 400 spans equal" in r.stdout
    # the library: fast route == zlib route on the same files (one thread per route, in child processes: the switch is read once)
    rng = random.Random(5)
    files = []
    for i in range(12):
        n = rng.choice([0, 1, 50, 4000, 200000])
        seq = "".join(rng.choice("ACGTNacgtRY") for _ in range(n))
        text = "".join(f">r{i}_{j} d\n" + "\n".join(seq[a:a + 70] for a in range(j, len(seq), 70 * 3)) + "\n" for j in range(3)).encode()
        if i % 4 == 3:      # two members
            data = gzip.compress(text[: len(text) // 2], 6) + gzip.compress(text[len(text) // 2:], 1)
        elif i % 4 == 2:    # FNAME + FHCRC: the header CRC is zlib's to check
            head = bytes([0x1F, 0x8B, 8, 2 | 8, 0, 0, 0, 0, 0, 255]) + b"x.fa\0"
            head += (zlib.crc32(head) & 0xFFFF).to_bytes(2, "little")
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            data = head + co.compress(text) + co.flush() + zlib.crc32(text).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little")
        else:
            data = gzip.compress(text, rng.choice([0, 1, 6, 9]))
        p = tmp_path / f"f{i}.fa.gz"
        p.write_bytes(data)
        files.append(str(p))
    # a stream that expands ~1000 x (the first reservation trusts ISIZE only up to 8 x the compressed size: found by doubling), and one
    # whose ISIZE field lies upwards (four bytes anyone can write: the member is then zlib's to refuse)
    big = (">poly\n" + "A" * 3_000_000 + "\n").encode()
    (tmp_path / "poly.fa.gz").write_bytes(gzip.compress(big, 9))
    files.append(str(tmp_path / "poly.fa.gz"))
    code = ("import sys, ctypes, hashlib; sys.path.insert(0, %r); from seqwin_amd._lib import lib, check, c_vp, c_u64\n"
            "paths = [p.encode() for p in sys.argv[1:]]; arr = (ctypes.c_char_p * len(paths))(*paths); h = c_vp()\n"
            "check(lib.sw_host_ingest(arr, len(paths), 2, ctypes.byref(h)))\n"
            "v = [c_u64() for _ in range(5)]; check(lib.sw_hostbatch_info(h, *[ctypes.byref(x) for x in v]))\n"
            "d = hashlib.sha256()\n"
            "for r in range(v[1].value):\n"
            "    n = c_u64(); check(lib.sw_hostbatch_record(h, r, None, 0, ctypes.byref(n)))\n"
            "    b = ctypes.create_string_buffer(max(n.value, 1)); check(lib.sw_hostbatch_record(h, r, b, n.value, ctypes.byref(n))); d.update(b.raw[:n.value] + b'|')\n"
            "print([x.value for x in v], d.hexdigest())\n") % str(ROOT)
    outs = []
    for extra in ({}, {"SEQWIN_AMD_ZLIB_INFLATE": "1"}):
        r = subprocess.run([sys.executable, "-c", code] + files, capture_output=True, text=True, env=dict(os.environ, **extra), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and outs[0].strip()


def test_half_word_rotate_formulas(tmp_path):
    """The sketch kernel writes srol / sror (hashing_internals.hpp:29-35, 69-74) and their 4-fold forms as funnel shifts
    and bit-field inserts on 32-bit halves (sketch.hip: srol1, sror1, srol4, sror4); tests/tools/rot_check.cpp restates
    exactly those formulas on the host and compares them with the reference definitions on 2 M values + all single bits."""
    import subprocess
    exe = tmp_path / "rot_check"
    subprocess.check_call(["g++", "-O2", str(ROOT / "tests" / "tools" / "rot_check.cpp"), "-o", str(exe)])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "bad = 0"


def test_multi_device_host_side_under_thread_sanitizer(tmp_path):
    """The HOST side of a multi-device build -- csrc/multi.hip's worker threads, rendezvous and peer pulls, the caching pool's
    hand-over of blocks between threads and streams (csrc/api.hip), spare events per device, the staged route, the streaming
    ingest's sinks -- under ThreadSanitizer on a mock HIP runtime (tests/tools/hip_mock: streams as FIFO queues each drained by
    its own thread, events, peer copies as memcpy, several DISTINCT mock devices; every .hip file compiled --cuda-host-only).
    This is the code an 8-GPU node runs first (the reference's counterpart: the worker threads and
    merge_thread_graphs of build.cpp:342-367).  Two programs:
      tsan_build  sw_build / sw_graph_export through the whole library over 1-8 mock devices (kernels are no-ops: empty graphs);
      tsan_multi  multi.hip's choreography over a fake engine whose "kernels" read and write every buffer on the stream's own
                  thread and check stamps -- non-zero sizes, both hash routes, direct / staged copies, injected engine failures.
    No ThreadSanitizer report, no stale or foreign bytes, no wrong result.  CPU only; sanitizers never run on the GPU box."""
    import os
    import subprocess
    mk = subprocess.run(["make", "-C", str(ROOT / "tests" / "tools" / "hip_mock"), "-j6"], capture_output=True, text=True)
    if mk.returncode != 0:
        if "sanitizer" in mk.stderr.lower() or "tsan" in mk.stderr.lower():
            pytest.skip("no ThreadSanitizer runtime for this toolchain")
        raise AssertionError(mk.stderr[-3000:])
    out_dir = ROOT / "seqwin_amd" / "csrc" / "build" / "hip_mock"
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    for k in ("SEQWIN_DEVICES", "SEQWIN_MULTI_NO_P2P", "SEQWIN_DIST_HASH_ROUTE"):
        env.pop(k, None)
    r = subprocess.run([str(out_dir / "tsan_build"), str(tmp_path), "14"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout[-1500:], r.stderr[-4000:])
    assert "two threads on two devices: rc 0 0" in r.stdout
    for extra, jobs in (({"HIP_MOCK_DEVICES": "8"}, 250), ({"HIP_MOCK_DEVICES": "8", "MOCK_FAIL": "25"}, 250),
                        ({"HIP_MOCK_DEVICES": "3", "SEQWIN_MULTI_NO_P2P": "1"}, 150), ({"HIP_MOCK_DEVICES": "4", "HIP_MOCK_NO_PEER": "1", "MOCK_FAIL": "10"}, 150),
                        ({"HIP_MOCK_DEVICES": "1", "MOCK_FAIL": "40"}, 150)):
        r = subprocess.run([str(out_dir / "tsan_multi"), str(jobs), "5"], capture_output=True, text=True, env=dict(env, **extra), timeout=600)
        assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr and "BAD DATA" not in r.stderr, (extra, r.stdout[-500:], r.stderr[-4000:])
        assert f"{jobs} jobs:" in r.stdout and " 0 wrong results, 0 stamp mismatches" in r.stdout, r.stdout
        if "MOCK_FAIL" not in extra:
            assert f"{jobs} built, 0 failed" in r.stdout, r.stdout
    # ... and the harness notices what it is there to notice: the choreography without the synchronisation in front of a rendezvous,
    # and the pool's hand-over between threads as it was until r04 (only the new owner's main stream waits: the hazard behind r05's one
    # unexplained GPU memory fault -- the upload stream writes a fresh block while the previous owner's kernels still do), must FAIL
    mk = subprocess.run(["make", "-C", str(ROOT / "tests" / "tools" / "hip_mock"), "-j6", "broken"], capture_output=True, text=True)
    assert mk.returncode == 0, mk.stderr[-3000:]
    for exe, extra, jobs in (("tsan_multi_nosync", {"HIP_MOCK_DEVICES": "2"}, 120), ("tsan_multi_r04pool", {"HIP_MOCK_DEVICES": "1"}, 1200)):
        r = subprocess.run([str(out_dir / exe), str(jobs), "3"], capture_output=True, text=True, env=dict(env, **extra), timeout=600)
        assert r.returncode != 0 and ("ThreadSanitizer: data race" in r.stderr or "BAD DATA" in r.stderr), (exe, r.stdout[-500:], r.stderr[-1500:])


def test_release_library_carries_no_test_hooks():
    """The switches that force size-dependent paths, inject faults or lower bounds are live in the TEST library only
    (csrc/common.hpp: SW_TEST_GETENV; `make test`): the release library reads them as unset -- their names are not even in the
    binary -- so a deployment is left with the "use" switches of DESIGN.md section 8a.  Both libraries are built from the same
    sources; this suite runs on the test library (tests/conftest.py), tests/test_release_library.py on the release one."""
    import os
    import subprocess
    rel, tst = ROOT / "seqwin_amd" / "libseqwin_hip.so", ROOT / "seqwin_amd" / "libseqwin_hip_test.so"
    assert rel.exists() and tst.exists()
    hooks = ["SORT_KEYBITS", "FAULT_INJECT", "SLOT_CAP", "WINDOW_SPLIT", "RADIX_RANK", "EDGE_SKIP_PASSES", "UNSORT_DIRECT", "NO_PACKED_EDGES",
             "SCALAR_INGEST", "LINE_PACKER", "READ_BLOCK_KB", "EXPORT_WHOLE", "PLAIN_DOWNLOAD", "PINNED_SLAB_MB", "INGEST_WINDOW"]
    use = ["SEQWIN_DEVICES", "SEQWIN_AMD_HBM_BUDGET_GB", "SEQWIN_AMD_LOWMEM_CHUNK_MBP", "SEQWIN_AMD_NO_RESIDENT", "SEQWIN_AMD_DEVICE_INFLATE",
           "SEQWIN_AMD_POOL_DEBUG"]
    blob_rel, blob_tst = rel.read_bytes(), tst.read_bytes()
    for h in hooks:
        name = b'SEQWIN_AMD_' + h.encode() + b'\0'
        assert name not in blob_rel, f"{h} is readable in the release library"
        assert name in blob_tst, f"{h} is missing from the test library"
    for u in use:
        assert u.encode() + b'\0' in blob_rel, u
    assert os.path.samefile(os.environ["SEQWIN_AMD_LIB"], tst) or os.environ.get("SEQWIN_AMD_RELEASE_LIB") == "1"
    # the suite really runs on the test library, a fresh interpreter without the suite's environment on the release one
    r = subprocess.run([__import__("sys").executable, "-c", "from seqwin_amd._lib import LIB_PATH; print(LIB_PATH)"], capture_output=True, text=True,
                       cwd=str(ROOT), env={k: v for k, v in os.environ.items() if k not in ("SEQWIN_AMD_LIB",)})
    assert r.returncode == 0 and r.stdout.strip().endswith("libseqwin_hip.so"), (r.stdout, r.stderr)
