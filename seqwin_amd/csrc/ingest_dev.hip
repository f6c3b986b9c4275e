// ingest_dev.hip -- gzip-FASTA ingest on the device: inflate + FASTA parse + 2-bit pack, one file per lane.
//
// Replaces, for inputs made of many .gz files (NCBI's default, src/seqwin/config.py:158), the host side of
// seqwin::internal::read_fasta's gzip branch (reference cpp/src/utils/fasta_reader.cpp:109-203: gzopen / gzread in 64 KiB
// pieces, one worker per file) and the parse of :41-95.  The host only moves the compressed bytes: zlib's inflate runs at
// ~0.3 GB/s of text per core, so 15 000 genomes (75 GB of text) take ~15 s on 16 cores; here every file is one lane of a
// wave -- DEFLATE is serial per stream, but the streams are independent -- with its Huffman tables in LDS (2.3 KB per lane,
// one wave per CU), and the text never leaves HBM.
//
// Three kernels:
//   k_inflate   RFC 1951 (stored / fixed / dynamic blocks) into text[text_off[f] .. + ISIZE); a file qualifies when it is a
//               single gzip member (RFC 1952) whose stream ends exactly at its trailer and inflates to ISIZE bytes.
//   k_parse<0>  the reader's rules (host_ingest.cpp parse_assembly, which restates fasta_reader.cpp:41-95) as a byte-serial
//               state machine: counts records, packed words, valid runs, id bytes; CRC-32 of the text (gzread verifies it).
//   k_parse<1>  the same walk again, writing the packed words and the record / run tables at their final places.
// Anything irregular -- a file that does not qualify, a CRC mismatch, a control byte, sequence before a header -- makes
// device_gz_ingest() return false without raising: the caller then takes the host path, which reports the error the way
// the reference does.  The result is bit-identical to the host path's (tests/test_gpu_parity.py: test_device_gz_ingest*).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "device.hpp"
#include "gz_dev.hpp"

namespace sw {
namespace {

using namespace gz;

struct InflateArgs {
    const uint8_t *comp;
    const uint64_t *data_start, *data_end;   // deflate data of file f: comp[data_start[f], data_end[f])
    uint8_t *text;
    const uint64_t *text_off;                // [n_files + 1]; file f may hold text_off[f + 1] - text_off[f] >= its ISIZE bytes
    const uint32_t *isize;
    uint32_t n_files;
    uint32_t *status;
    unsigned long long *prof;                // (SEQWIN_AMD_DEBUG_TIMING) file 0: clocks in table set-up / decoding, blocks, lookups, matches
};

__global__ __launch_bounds__(64) void k_inflate(const InflateArgs A)
{
    __shared__ LaneTables T[64];
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= A.n_files) return;
    A.status[f] = inflate_one(T[threadIdx.x], A.comp, A.data_start[f], A.data_end[f], A.text, A.text_off[f], A.isize[f],
                              A.prof && f == 0 ? A.prof : nullptr);
}

struct ParseArgs {
    const uint8_t *text;
    const uint64_t *text_off;       // [n_files + 1]
    const uint32_t *isize;
    uint32_t n_files;
    ParseCounts *counts;            // <0>: written
    // <1>: where file f's output starts
    const uint64_t *word_base, *id_base;
    const uint32_t *rec_base_idx, *run_base;
    ParseDst dst;                   // the arrays (bases unset)
};

template <bool WRITE> __global__ __launch_bounds__(64) void k_parse(const ParseArgs A)
{
    __shared__ uint8_t cls[256];
    __shared__ uint32_t crc_tab[4][256];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) {
        cls[i] = char_class(i);
        if (!WRITE) crc_tab[0][i] = crc_entry(i);
    }
    __syncthreads();
    if (!WRITE) {
        for (uint32_t i = threadIdx.x; i < 256; i += 64)
            for (int s = 1; s < 4; ++s) crc_tab[s][i] = (crc_tab[s - 1][i] >> 8) ^ crc_tab[0][crc_tab[s - 1][i] & 0xFFu];
        __syncthreads();
    }
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= A.n_files) return;
    ParseDst D = A.dst;
    if (WRITE) {
        D.word_base = A.word_base[f];
        D.id_base = A.id_base[f];
        D.rec_base_idx = A.rec_base_idx[f];
        D.run_base = A.run_base[f];
    }
    ParseCounts pc;
    parse_one<WRITE>(cls, crc_tab, A.text + A.text_off[f], A.isize[f], D, pc);
    if (!WRITE) A.counts[f] = pc;
}

std::atomic<uint64_t> g_device_gz_batches{0};

bool ends_with_gz(const char *s)
{
    const size_t n = strlen(s);
    return n >= 3 && !strcmp(s + n - 3, ".gz");
}

struct Pinned {   // one pinned staging buffer per worker, kept for the life of the process (allocating costs ~0.25 ms per MiB)
    static constexpr size_t BYTES = 4u << 20;
    char *p = nullptr;
    hipStream_t st = nullptr;
};
Pinned &pinned_slot(int device, size_t i)   // (the calling thread has made `device` current: the stream belongs to it)
{
    static std::mutex mu;
    static std::map<std::pair<int, size_t>, Pinned *> *slots = new std::map<std::pair<int, size_t>, Pinned *>;   // leaked on purpose
    std::lock_guard<std::mutex> lock(mu);
    Pinned *&s = (*slots)[{device, i}];
    if (!s) {
        s = new Pinned;
        SW_HIP(hipHostMalloc((void **)&s->p, Pinned::BYTES, hipHostMallocDefault));
        SW_HIP(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    }
    return *s;
}

template <class T> void to_host(std::vector<T> &dst, const DevArray<T> &src, size_t n)
{
    dst.resize(n);
    if (n) SW_HIP(hipMemcpy(dst.data(), src.p, n * sizeof(T), hipMemcpyDeviceToHost));
}

}  // namespace

// All paths end in ".gz" and there are enough of them (SEQWIN_AMD_DEVICE_INFLATE=1: any number; =0: never): inflate, parse
// and pack on the device.  true: b.host / b.d_packed / b.packed_words are filled as ingest_to_device's host route fills
// them (the caller uploads the record tables); false: nothing was changed -- take the host route.
bool device_gz_ingest(const char *const *paths, size_t n_paths, uint64_t n_cpu, sw_batch &b)
{
    const char *mode = getenv("SEQWIN_AMD_DEVICE_INFLATE");
    if (mode && !strcmp(mode, "0")) return false;
    const bool forced = mode && !strcmp(mode, "1");
    // A lane inflates ~1.7 MB of text per second whatever the number of files (64 independent decoders run in lockstep, one wave per
    // CU: ~7 500 clocks per symbol, r03), a host thread ~370 MB/s: the device wins from ~320 files per host thread on
    // (measured: 8 192 files of 1 Mbp, 16 threads: 817 ms against 1 404 ms; 1 024 files: 760 against 178 ms).
    // r05: the host side got faster (fast_inflate.hpp: ~0.65 GB/s of text per usable CPU, inflate + parse + pack, bench.py e2e.gz) and
    // is counted in the CPUs the process may really use, not in the threads it was told to start (a 16-CPU quota on a 256-thread
    // host); the rule is now made from the files' sizes below: device time ~ text of the LARGEST file / 1.27 MB/s (512 files of
    // 5 Mbp: 3.98 s), host time ~ all text / (0.65 GB/s x CPUs).  15 000 genomes of 5 Mbp on 16 CPUs: 4 s against 7 s -> device;
    // 512: 4 s against 0.25 s -> host.
    // End of r06: the host decoder's super table (fast_inflate.hpp) halved the host's CPU time: 1 024 .fa.gz genomes (5.2 GB of text)
    // -> numpy on the GPU box cost 6.7-6.9 CPU-seconds before and 3.4-4.2 after (profiles/r06_e2e_gz_decoder_ab.txt), i.e. 1.2-1.5 GB/s
    // of text per CPU-second for the whole call; the rule takes 1.0 -- and the device has to win by a quarter of its own estimate,
    // since its figure at full size is an extrapolation (15 000 genomes of 5 Mbp: 4.0 s, never measured end to end): 15 000 genomes
    // on 16 CPUs, 4.0 x 1.25 against 4.8 s -> host; on 8 CPUs, against 9.6 s -> device.
    if (n_paths == 0 || n_paths >= 0xFFFFFFFFull) return false;
    for (size_t i = 0; i < n_paths; ++i)
        if (!ends_with_gz(paths[i])) return false;
    const bool timing = getenv("SEQWIN_AMD_DEBUG_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto decline = [&](const char *why, size_t file, unsigned long long v) {
        if (timing) fprintf(stderr, "[seqwin_amd] device gz ingest declined: %s (file %zu: %s, %llu)\n", why, file, paths[file], v);
        return false;
    };

    // -- sizes, then the compressed bytes to the device (every file at a 16-byte boundary) ---------------------------
    std::vector<uint64_t> fsize(n_paths), coff(n_paths + 1, 0);
    for (size_t i = 0; i < n_paths; ++i) {
        struct stat st;
        if (stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) return decline("not a regular file of 18 bytes or more", i, 0);
        fsize[i] = (uint64_t)st.st_size;
        coff[i + 1] = coff[i] + ((fsize[i] + 15) & ~15ull);
    }
    if (!forced) {
        uint64_t largest = 0, total = 0;
        for (size_t i = 0; i < n_paths; ++i) {
            largest = std::max(largest, fsize[i]);
            total += fsize[i];
        }
        const double ratio = 3.3;   // text per compressed byte of level-6 FASTA
        const double cpus = (double)std::min<uint64_t>(std::max<uint64_t>(1, n_cpu), usable_cpus());
        const double t_dev = (double)largest * ratio / 1.27e6, t_host = (double)total * ratio / (1.0e9 * cpus);
        if (t_dev * 1.25 >= t_host) return false;   // (the host route; nothing has been changed)
    }
    size_t free_b = 0, total_b = 0;
    SW_HIP(hipMemGetInfo(&free_b, &total_b));
    if (coff[n_paths] * 8 > free_b) {   // (r06: blocks the pool has cached for earlier builds are not "free" to hipMemGetInfo: give them back, ask again)
        dev_pool_trim();
        SW_HIP(hipMemGetInfo(&free_b, &total_b));
    }
    if (coff[n_paths] * 8 > free_b) return decline("not enough free HBM", 0, free_b);          // (text ~4x the compressed bytes, + packed words + tables: the host route streams)
    DevArray<uint8_t> d_comp(coff[n_paths] + 16);
    std::vector<uint64_t> dstart(n_paths), dend(n_paths);
    std::vector<uint32_t> isize(n_paths), crc_want(n_paths);
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    std::atomic<size_t> bad_file{0};
    const size_t n_workers = std::max<size_t>(1, std::min<size_t>({(size_t)std::max<uint64_t>(1, n_cpu), n_paths, 32}));
    std::mutex err_mu;
    std::exception_ptr device_error;                   // a HIP failure in a reader is an error of the call, not a reason to decline
    struct Fd {                                        // closes on every way out of the loop body
        int fd;
        ~Fd() { if (fd >= 0) close(fd); }
    };
    auto reader = [&](size_t w) {
        try {
            SW_HIP(hipSetDevice(b.device));            // (a new thread starts on device 0)
            Pinned &pin = pinned_slot(b.device, w);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_paths || !ok.load()) break;
                const Fd file{open(paths[i], O_RDONLY)};
                const int fd = file.fd;
                if (fd < 0) { bad_file.store(i); ok.store(false); break; }
                uint64_t done = 0;
                bool first = true, bad = false;
                uint8_t tail[8] = {0};
                while (done < fsize[i]) {
                    SW_HIP(hipStreamSynchronize(pin.st));              // the slot's previous copy has left the host
                    const size_t want = (size_t)std::min<uint64_t>(Pinned::BYTES, fsize[i] - done);
                    size_t got = 0;
                    while (got < want) {
                        const ssize_t r = read(fd, pin.p + got, want - got);
                        if (r <= 0) break;
                        got += (size_t)r;
                    }
                    if (got != want) { bad = true; break; }
                    if (first) {
                        dstart[i] = gzip_header_len((const uint8_t *)pin.p, std::min<uint64_t>(got, fsize[i]));
                        if (dstart[i] == 0) { bad = true; break; }
                        first = false;
                    }
                    for (uint64_t k = 0; k < 8; ++k) {                 // the trailer: the file's last eight bytes
                        const uint64_t at = fsize[i] - 8 + k;
                        if (at >= done && at < done + got) tail[k] = (uint8_t)pin.p[at - done];
                    }
                    SW_HIP(hipMemcpyAsync(d_comp.p + coff[i] + done, pin.p, got, hipMemcpyHostToDevice, pin.st));
                    done += got;
                }
                if (bad || dstart[i] + 8 > fsize[i]) { bad_file.store(i); ok.store(false); break; }
                crc_want[i] = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
                isize[i] = (uint32_t)tail[4] | ((uint32_t)tail[5] << 8) | ((uint32_t)tail[6] << 16) | ((uint32_t)tail[7] << 24);
                dend[i] = coff[i] + fsize[i] - 8;
                dstart[i] += coff[i];
            }
            SW_HIP(hipStreamSynchronize(pin.st));
        } catch (...) {                                // SW_HIP raised: hipHostMalloc / hipMemcpyAsync / a device fault
            std::lock_guard<std::mutex> g(err_mu);
            if (!device_error) device_error = std::current_exception();
            ok.store(false);
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t w = 0; w < n_workers; ++w) th.emplace_back(reader, w);
        for (auto &t : th) t.join();
    }
    if (device_error) std::rethrow_exception(device_error);
    if (!ok.load()) return decline("unreadable, or not a plain gzip header", bad_file.load(), 0);
    const auto t1 = std::chrono::steady_clock::now();

    // -- inflate -------------------------------------------------------------------------------------------------------
    // (The three kernels, the memsets and the synchronous copies below run on the NULL stream: the batch is not handed out before
    //  the hipDeviceSynchronize at the end, and every DevArray released on the way is fenced by the pool on its next owner's stream.)
    std::vector<uint64_t> toff(n_paths + 1, 0);
    for (size_t i = 0; i < n_paths; ++i) toff[i + 1] = toff[i] + (((uint64_t)isize[i] + 15) & ~15ull);
    SW_HIP(hipMemGetInfo(&free_b, &total_b));
    if (toff[n_paths] + toff[n_paths] / 2 > free_b) {
        dev_pool_trim();
        SW_HIP(hipMemGetInfo(&free_b, &total_b));
    }
    if (toff[n_paths] + toff[n_paths] / 2 > free_b) return decline("not enough free HBM for the text", 0, toff[n_paths]);
    const uint32_t nf = (uint32_t)n_paths;
    DevArray<uint8_t> d_text(toff[n_paths] + 16);
    // Lane j of the kernels takes file perm[j], the files in descending size: a wave lasts as long as its largest file, so
    // files of like size share a wave (genomes of one job differ by a few x in size)
    std::vector<uint32_t> perm(nf);
    for (uint32_t i = 0; i < nf; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t c) { return isize[a] > isize[c]; });
    auto permuted = [&](const auto &v) {
        typename std::decay<decltype(v)>::type out(nf);
        for (uint32_t j = 0; j < nf; ++j) out[j] = v[perm[j]];
        return out;
    };
    DevArray<uint64_t> d_dstart(nf), d_dend(nf), d_toff(nf + 1);
    DevArray<uint32_t> d_isize(nf), d_status(nf);
    SW_HIP(hipMemcpy(d_dstart.p, permuted(dstart).data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_dend.p, permuted(dend).data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_toff.p, permuted(toff).data(), nf * 8ull, hipMemcpyHostToDevice));      // (the kernels read text_off[f] only)
    SW_HIP(hipMemcpy(d_isize.p, permuted(isize).data(), nf * 4ull, hipMemcpyHostToDevice));
    const unsigned blocks = (nf + 63) / 64;
    DevArray<unsigned long long> d_prof(8);
    SW_HIP(hipMemset(d_prof.p, 0, 64));
    InflateArgs ia{d_comp.p, d_dstart.p, d_dend.p, d_text.p, d_toff.p, d_isize.p, nf, d_status.p, timing ? d_prof.p : nullptr};
    hipLaunchKernelGGL(k_inflate, dim3(blocks), dim3(64), 0, nullptr, ia);
    SW_HIP(hipGetLastError());
    if (timing) {
        unsigned long long hp[8];
        SW_HIP(hipMemcpy(hp, d_prof.p, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[seqwin_amd] inflate, file 0: %llu bytes, %llu blocks, %llu lookups, %llu matches; clocks: table set-up %llu, decoding %llu, "
                        "of which %llu in %llu window loads from HBM\n", hp[5], hp[2], hp[3], hp[4], hp[0], hp[1], hp[6], hp[7]);
    }
    std::vector<uint32_t> status;
    to_host(status, d_status, nf);
    for (uint32_t j = 0; j < nf; ++j)
        if (status[j] != ST_OK) return decline("inflate status", perm[j], status[j]);   // corrupt, or more than one member: the host route decides
    d_comp.release();
    const auto t2 = std::chrono::steady_clock::now();

    // -- parse: count, place, write ------------------------------------------------------------------------------------
    DevArray<ParseCounts> d_counts(nf);
    ParseArgs pa{};
    pa.text = d_text.p;
    pa.text_off = d_toff.p;
    pa.isize = d_isize.p;
    pa.n_files = nf;
    pa.counts = d_counts.p;
    hipLaunchKernelGGL(k_parse<false>, dim3(blocks), dim3(64), 0, nullptr, pa);
    SW_HIP(hipGetLastError());
    std::vector<ParseCounts> counts_by_lane, counts(nf);
    to_host(counts_by_lane, d_counts, nf);
    for (uint32_t j = 0; j < nf; ++j) counts[perm[j]] = counts_by_lane[j];
    std::vector<uint64_t> word_base(nf + 1, 0), id_base(nf + 1, 0);
    std::vector<uint32_t> rec_idx(nf + 1, 0), run_base(nf + 1, 0);
    uint64_t total_bp = 0;
    for (uint32_t i = 0; i < nf; ++i) {
        if (counts[i].err) return decline("parse error flags", i, counts[i].err);
        if (counts[i].crc != crc_want[i]) return decline("CRC-32 mismatch", i, counts[i].crc);
        word_base[i + 1] = word_base[i] + counts[i].n_words;
        id_base[i + 1] = id_base[i] + counts[i].n_id;
        const uint64_t r = (uint64_t)rec_idx[i] + counts[i].n_rec, u = (uint64_t)run_base[i] + counts[i].n_runs;
        if (r > UINT32_MAX || u > UINT32_MAX) return decline("more than 2^32-1 records or runs", i, r);    // (the host route raises: build.cpp:136-140)
        rec_idx[i + 1] = (uint32_t)r;
        run_base[i + 1] = (uint32_t)u;
        total_bp += counts[i].total_bp;
    }
    const uint64_t n_words = word_base[nf], n_rec = rec_idx[nf], n_runs = run_base[nf], n_id = id_base[nf];
    DevArray<uint32_t> d_packed(n_words * 2 + 8);
    DevArray<uint64_t> d_wb(nf), d_ib(nf), d_rec_base(n_rec);
    DevArray<uint32_t> d_ri(nf), d_ub(nf), d_rec_len(n_rec), d_rec_run_off(n_rec), d_run_pos(n_runs), d_run_len(n_runs);
    DevArray<char> d_ids(n_id);
    SW_HIP(hipMemcpy(d_wb.p, permuted(word_base).data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_ib.p, permuted(id_base).data(), nf * 8ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_ri.p, permuted(rec_idx).data(), nf * 4ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemcpy(d_ub.p, permuted(run_base).data(), nf * 4ull, hipMemcpyHostToDevice));
    SW_HIP(hipMemsetAsync(d_packed.p + 2 * n_words, 0, 8 * 4, nullptr));   // the read slack
    pa.word_base = d_wb.p;
    pa.id_base = d_ib.p;
    pa.rec_base_idx = d_ri.p;
    pa.run_base = d_ub.p;
    pa.dst.words = reinterpret_cast<uint64_t *>(d_packed.p);
    pa.dst.rec_len = d_rec_len.p;
    pa.dst.rec_run_off = d_rec_run_off.p;
    pa.dst.run_pos = d_run_pos.p;
    pa.dst.run_len = d_run_len.p;
    pa.dst.rec_base = d_rec_base.p;
    pa.dst.ids = d_ids.p;
    hipLaunchKernelGGL(k_parse<true>, dim3(blocks), dim3(64), 0, nullptr, pa);
    SW_HIP(hipGetLastError());

    // -- the tables the planner reads, back on the host ----------------------------------------------------------------
    HostBatch h;
    h.n_assemblies = n_paths;
    h.total_bp = total_bp;
    h.record_offsets.assign(rec_idx.begin(), rec_idx.end());
    to_host(h.rec_len, d_rec_len, n_rec);
    to_host(h.rec_base, d_rec_base, n_rec);
    to_host(h.rec_run_off, d_rec_run_off, n_rec);
    h.rec_run_off.push_back((uint32_t)n_runs);
    to_host(h.run_pos, d_run_pos, n_runs);
    to_host(h.run_len, d_run_len, n_runs);
    std::vector<char> ids;
    to_host(ids, d_ids, n_id);
    h.ids_blob.assign(ids.data(), ids.size());
    h.chunks.resize(n_paths);
    h.chunk_word0.assign(word_base.begin(), word_base.end());
    SW_HIP(hipDeviceSynchronize());
    b.host = std::move(h);
    b.d_packed = std::move(d_packed);
    b.packed_words = n_words * 2 + 8;
    g_device_gz_batches.fetch_add(1);
    if (timing)
        fprintf(stderr, "[seqwin_amd] device gz ingest: %zu files, read + upload %.1f ms, inflate %.1f ms, parse + tables %.1f ms\n", n_paths,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    return true;
}

}  // namespace sw

extern "C" uint64_t sw_device_gz_batches(void) { return sw::g_device_gz_batches.load(); }
