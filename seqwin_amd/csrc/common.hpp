// common.hpp -- shared declarations of libseqwin_hip.so (host side).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/seqwin_hip.h"

namespace sw {

// ---- error plumbing: C++ exceptions inside, int codes at the C ABI -------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void raise(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

void set_last_error(const char *msg);

template <class F> int guarded(F &&f)
{
    try {
        f();
        return SW_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return SW_ERR_RUNTIME;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return SW_ERR_RUNTIME;
    }
}

// ---- host-side result of FASTA ingest (k-independent) ---------------------------------------
// Bases are packed 2 bits each (A0 C1 G2 T/U3; invalid bases are stored as 0 and described by
// the run table), 16 per uint32 word, base i of the stream in bits [2*(i%16), 2*(i%16)+2) of word
// i/16.  Every record starts on a 32-base boundary.
struct HostBatch {
    uint64_t n_assemblies = 0;
    uint64_t total_bp = 0;
    std::vector<uint32_t> record_offsets;   // [n_assemblies + 1]
    std::string ids_blob;                   // NUL-terminated ids in record order
    std::vector<uint32_t> rec_len;          // [R]
    std::vector<uint64_t> rec_base;         // [R] offset of the record's first base in the packed stream
    std::vector<uint32_t> rec_run_off;      // [R + 1] index of the record's first valid run
    std::vector<uint32_t> run_pos, run_len; // maximal runs of valid bases, in (record, pos) order
    std::vector<uint32_t> packed;           // 2-bit stream
};

// host_ingest.cpp
void ingest_fasta(const char *const *paths, size_t n_paths, uint64_t n_cpu, HostBatch &out);
void check_kw(uint64_t k, uint64_t w);

}  // namespace sw
