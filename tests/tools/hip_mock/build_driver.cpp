// build_driver.cpp -- sw_build through the WHOLE host side of libseqwin_hip (ingest threads, upload ring, pool, plan, launches,
// multi-device workers) under ThreadSanitizer, on the mock HIP runtime (hip_mock.cpp): kernels are no-ops, so every count read
// back is zero and the graphs come out empty -- what is exercised is who touches what from which thread and stream.
// usage: tsan_build <tmpdir> [rounds]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/seqwin_hip.h"
#include "hip_mock.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static std::vector<std::string> write_files(const std::string &dir, int n, int tag)
{
    std::vector<std::string> paths;
    for (int a = 0; a < n; ++a) {
        const std::string p = dir + "/t" + std::to_string(tag) + "_" + std::to_string(a) + ".fa";
        FILE *f = fopen(p.c_str(), "w");
        if (!f) { perror("fopen"); exit(2); }
        const int recs = 1 + (int)(rnd() % 3);
        for (int r = 0; r < recs; ++r) {
            fprintf(f, ">r%d_%d\n", a, r);
            const int len = 200 + (int)(rnd() % 20000);
            for (int i = 0; i < len; ++i) {
                fputc("ACGT"[rnd() & 3], f);
                if (i % 80 == 79) fputc('\n', f);
            }
            fputc('\n', f);
        }
        fclose(f);
        paths.push_back(p);
    }
    return paths;
}

static int one_build(const std::vector<std::string> &paths, uint64_t k, uint64_t w, uint64_t n_cpu, bool low_memory)
{
    std::vector<const char *> cp;
    for (auto &p : paths) cp.push_back(p.c_str());
    sw_graph *g = nullptr;
    int rc = sw_build(cp.data(), cp.size(), k, w, n_cpu, low_memory ? 1 : 0, &g);
    if (rc != SW_OK) {
        fprintf(stderr, "sw_build rc=%d: %s\n", rc, sw_last_error());
        return rc;
    }
    uint64_t nk = 0, nn = 0, ne = 0, na = 0, nb = 0, bp = 0;
    rc = sw_graph_sizes(g, &nk, &nn, &ne, &na, &nb, &bp);
    if (rc == SW_OK) {
        std::vector<sw_kmer> kmers(nk + 1);
        std::vector<sw_node> nodes(nn + 1);
        std::vector<sw_edge> edges(ne + 1);
        std::vector<uint32_t> ro(na + 1);
        std::string blob(nb + 1, '\0');
        rc = sw_graph_export(g, kmers.data(), nodes.data(), edges.data(), ro.data(), &blob[0]);
        if (rc != SW_OK) fprintf(stderr, "sw_graph_export rc=%d: %s\n", rc, sw_last_error());
        if (na != paths.size()) { fprintf(stderr, "n_assemblies %llu != %zu\n", (unsigned long long)na, paths.size()); rc = 99; }
    }
    sw_graph_free(g);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const int rounds = argc > 2 ? atoi(argv[2]) : 6;
    static const char *const device_lists[] = {"", "0,1", "0,0", "1,0,1", "0,1,2,3", "3,3,2,2,1,1,0,0", "all"};
    int bad = 0;
    for (int r = 0; r < rounds && !bad; ++r) {
        const char *devs = device_lists[r % 7];
        if (*devs) setenv("SEQWIN_DEVICES", devs, 1);
        else unsetenv("SEQWIN_DEVICES");
        if (r % 3 == 2) setenv("SEQWIN_MULTI_NO_P2P", "1", 1);
        else unsetenv("SEQWIN_MULTI_NO_P2P");
        if (r % 7 == 0 && r % 5 != 4) {   // the single-device rounds: ingest and sketch overlapped (a second host thread drives the device)
            setenv("SEQWIN_AMD_PIPELINE", "1", 1);
            setenv("SEQWIN_AMD_PIPELINE_CHUNK_MBP", "0", 1);
        } else {
            unsetenv("SEQWIN_AMD_PIPELINE");
        }
        const auto paths = write_files(dir, 3 + (int)(rnd() % 9), r);
        const int rc = one_build(paths, 15 + rnd() % 10, 10 + rnd() % 200, 1 + rnd() % 8, r % 5 == 4);
        printf("round %d devices=[%s]%s files=%zu rc=%d\n", r, devs, r % 3 == 2 ? " no-p2p" : "", paths.size(), rc);
        if (rc) bad = 1;
        for (auto &p : paths) remove(p.c_str());
    }
    // two host threads building at once, each on its own device (one process per GPU is the model, but nothing forbids this)
    if (!bad) {
        unsetenv("SEQWIN_DEVICES");
        int rcs[2] = {0, 0};
        const auto pa = write_files(dir, 5, 1000), pb = write_files(dir, 6, 1001);
        std::thread ta([&] { sw_set_device(0); for (int i = 0; i < 3; ++i) rcs[0] |= one_build(pa, 21, 200, 2, false); });
        std::thread tb([&] { sw_set_device(1); for (int i = 0; i < 3; ++i) rcs[1] |= one_build(pb, 17, 50, 2, false); });
        ta.join();
        tb.join();
        printf("two threads on two devices: rc %d %d\n", rcs[0], rcs[1]);
        bad = rcs[0] | rcs[1];
        for (auto &p : pa) remove(p.c_str());
        for (auto &p : pb) remove(p.c_str());
    }
    sw_release_resident();
    sw_pool_trim();
    uint64_t st[4];
    hip_mock_stats(st);
    printf("mock runtime: %llu kernel launches, %llu copies, %llu peer copies, %llu hipFree\n", (unsigned long long)st[0], (unsigned long long)st[1],
           (unsigned long long)st[2], (unsigned long long)st[3]);
    return bad;
}
