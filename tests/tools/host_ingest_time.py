"""Where the host ingest's time goes (no GPU work): sw_host_ingest over N plain FASTA files of 5 Mbp in /dev/shm at several thread
counts and file-access modes, beside bare read() of the same files by as many threads.  usage: host_ingest_time.py [files] [n_cpu ...]"""
import ctypes, os, shutil, sys, tempfile, threading, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
os.environ.setdefault("SEQWIN_AMD_NO_TORCH", "1")
from seqwin_amd._lib import check, lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CPUS = [int(x) for x in sys.argv[2:]] or [1, 8, 16, 32, 128]
tmp = tempfile.mkdtemp(prefix="ingt_", dir="/dev/shm")
try:
    rng = np.random.default_rng(1)
    seeds = []
    for a in range(8):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=5_000_000)
        buf = np.full(62_500 * 81, 10, dtype=np.uint8)
        buf.reshape(62_500, 81)[:, :80] = seq.reshape(62_500, 80)
        p = f"{tmp}/s{a}.fa"
        with open(p, "wb") as f:
            f.write(b">rec0 x\n" + buf.tobytes())
        seeds.append(p)
    paths = []
    for a in range(N):
        p = f"{tmp}/g{a}.fa"
        shutil.copyfile(seeds[a % 8], p)
        paths.append(p)
    tot = sum(os.path.getsize(p) for p in paths)
    arr = (ctypes.c_char_p * N)(*[p.encode() for p in paths])
    print(f"{N} files, {tot / 1e9:.2f} GB", flush=True)

    def ingest(ncpu):
        hb = ctypes.c_void_p()
        t0 = time.perf_counter()
        check(lib.sw_host_ingest(arr, N, ncpu, ctypes.byref(hb)))
        dt = time.perf_counter() - t0
        lib.sw_hostbatch_free(hb)
        return dt

    def bare_read(ncpu):
        def work(i):
            buf = bytearray(6_000_000)
            for p in paths[i::ncpu]:
                fd = os.open(p, os.O_RDONLY)
                os.readv(fd, [buf])
                os.close(fd)
        th = [threading.Thread(target=work, args=(i,)) for i in range(ncpu)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return time.perf_counter() - t0

    for ncpu in CPUS:
        row = [f"n_cpu={ncpu:4d}"]
        for mode in ("0", "1", "2"):
            os.environ["SEQWIN_AMD_MMAP"] = mode
            dt = min(ingest(ncpu) for _ in range(3))
            row.append(f"mmap={mode}: {dt * 1e3:7.1f} ms {tot / dt / 1e9:6.2f} GB/s")
        os.environ.pop("SEQWIN_AMD_MMAP")
        dt = min(bare_read(ncpu) for _ in range(2))
        row.append(f"bare read(): {dt * 1e3:7.1f} ms {tot / dt / 1e9:6.2f} GB/s")
        print("   ".join(row), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
