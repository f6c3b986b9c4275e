/*
 * seqwin_hip.h -- C ABI of libseqwin_hip.so, the MI355X (gfx950) implementation of Seqwin's
 * minimizer-index hot path.
 *
 * The entry points are exactly what the reference's native boundary for this path binds
 * (the pybind11 module `seqwin.graph._core`, cpp/src/bindings/python_bindings.cpp:43-169, over
 * cpp/include/seqwin/{build,filter,graph}.hpp); each declaration cites the reference interface it
 * replaces.  Plain pointers and sizes only -- no C++/torch types cross this boundary.
 *
 * Conventions
 *   - Every function returning `int` returns SW_OK or an SW_ERR_* code; the message is available
 *     from sw_last_error() (thread-local).  Code -> Python exception mapping mirrors pybind11's
 *     translation of the reference's C++ exceptions: SW_ERR_RUNTIME <- std::runtime_error /
 *     std::logic_error -> RuntimeError; SW_ERR_VALUE <- std::invalid_argument -> ValueError.
 *   - There is NO CPU fallback: if no HIP device is usable the calls fail with SW_ERR_DEVICE.
 *   - Wire formats are the reference's PODs, byte for byte (cpp/include/seqwin/graph.hpp:15-53):
 *     sw_kmer 8 B, sw_node 40 B, sw_edge 24 B.
 */
#ifndef SEQWIN_HIP_H
#define SEQWIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SW_OK 0
#define SW_ERR_RUNTIME 1 /* std::runtime_error / std::logic_error in the reference */
#define SW_ERR_VALUE 2   /* std::invalid_argument in the reference */
#define SW_ERR_DEVICE 3  /* no usable HIP device / HIP runtime error (no reference analogue) */

/* seqwin::Kmer, cpp/include/seqwin/graph.hpp:15-20 */
typedef struct sw_kmer {
    uint32_t pos;
    uint32_t record_idx;
} sw_kmer;

/* seqwin::Node, cpp/include/seqwin/graph.hpp:28-41 */
typedef struct sw_node {
    uint64_t hash;
    uint64_t start;
    uint64_t stop;
    uint32_t n_tar;
    uint32_t n_neg;
    double penalty;
} sw_node;

/* seqwin::Edge, cpp/include/seqwin/graph.hpp:46-53 */
typedef struct sw_edge {
    uint64_t first;
    uint64_t second;
    uint64_t weight;
} sw_edge;

typedef struct sw_graph sw_graph; /* opaque result of sw_build (host-resident arrays) */
typedef struct sw_batch sw_batch; /* opaque device-resident batch of 2-bit packed assemblies */
typedef struct sw_index sw_index; /* opaque device-resident index built from a batch */

/* ---- diagnostics -------------------------------------------------------------------------- */
const char *sw_last_error(void);
const char *sw_version(void);
/* Log sink (replaces log_python, cpp/src/utils/logging.cpp:9-29, which logs to Python's root logger from native code):
 * fn(level, message) with level in {"debug", "info", "warning", "error"} is called on the thread that called into the
 * library; NULL (the default) drops the messages.  sw_build reports one "info" line per call. */
typedef void (*sw_log_fn)(const char *level, const char *message);
void sw_set_log_callback(sw_log_fn fn);
/* Number of visible HIP devices (0 when none / runtime unusable). Does not initialise a context. */
int sw_device_count(void);
/* Select the HIP device used by the calling thread for all later calls (default 0). */
int sw_set_device(int device);

/* ---- drop-in entry points (host arrays in, host arrays out) ------------------------------- */

/*
 * Replaces `_build_native(assembly_paths, kmerlen, windowsize, n_cpu, low_memory)`
 * (python_bindings.cpp:50-90) = seqwin::build (cpp/include/seqwin/build.hpp:22-28,
 * cpp/src/seqwin/build.cpp:330-394).  FASTA/gz files are read and 2-bit packed on `n_cpu` host
 * threads, sketched and indexed on the GPU, and the result is copied back into *out.
 * `low_memory` (build.cpp:264-325: the reference recomputes the sketches in a second pass to keep the peak down) streams the
 * assemblies through HBM in consecutive chunks of SEQWIN_AMD_LOWMEM_CHUNK_MBP Mbp (default 4096): only 24 B per
 * minimizer of a finished chunk stay resident, and the index is built once from the concatenated tuple stream -- results are
 * identical either way (reference tests/smoke/test_graph.py:222-245).  The same route is taken without the flag when the
 * files exceed SEQWIN_AMD_HBM_BUDGET_GB (if set).
 * Any windowsize >= 1 is taken (minimizer.cpp:53-90 has no upper limit either; a window longer than a record's k-mers
 * gives no minimizer for it, :56-58).
 * Errors: k < 3, k > 65535, w < 1 -> SW_ERR_VALUE; unreadable file, FASTA
 * without header, > 2^32-1 records or bases per record -> SW_ERR_RUNTIME (build.cpp:136-147,337-339;
 * fasta_reader.cpp:69-71,99-102).
 */
int sw_build(const char *const *assembly_paths, size_t n_assemblies, uint64_t kmerlen, uint64_t windowsize,
             uint64_t n_cpu, int low_memory, sw_graph **out);

/* Sizes of the arrays held by a graph (the shapes `_build_native` returns). `ids_bytes` is the size of
 * the NUL-separated record-id blob (one id per FASTA record, in global record order). */
int sw_graph_sizes(const sw_graph *g, uint64_t *n_kmers, uint64_t *n_nodes, uint64_t *n_edges,
                   uint64_t *n_assemblies, uint64_t *ids_bytes, uint64_t *total_bp);

/* Copy the graph into caller-owned buffers (numpy arrays allocated by the Python glue):
 * kmers[n_kmers], nodes[n_nodes] (n_tar = n_neg = 0, penalty = 0.0), edges[n_edges],
 * record_offsets[n_assemblies + 1], ids_blob[ids_bytes].  Replaces array_to_numpy
 * (python_bindings.cpp:21-39, 69-83) without foreign-owned memory outliving the call.  (Large results cross PCIe through a
 * ring of pinned slots, nodes and edges in a packed form that host threads expand into these buffers: same bytes.) */
int sw_graph_export(const sw_graph *g, sw_kmer *kmers, sw_node *nodes, sw_edge *edges,
                    uint32_t *record_offsets, char *ids_blob);
/* Where the time of the sw_build that made g (and of its sw_graph_export) went, in ms: out[0] ingest + upload (host parse and
 * 2-bit packing with the pipelined copy to HBM), [1] the device part (wall), [2] launch plan, [3] sketch kernel, [4] tuple
 * order, [5] nodes stage, [6] edges stage, [7] sw_graph_export (0 before it ran). */
int sw_graph_stats(const sw_graph *g, double *out8);

/* Frees the graph handle.  The device index of an EXPORTED graph stays resident (one per process, replaced by the next
 * sw_build): sw_get_penalty / sw_filter_kmers work on it instead of uploading the caller's arrays again when those are
 * still the exported ones (same sizes and position-dependent checksums, verified on the host).  sw_release_resident()
 * gives that HBM back at once; SEQWIN_AMD_NO_RESIDENT=1 disables the mechanism. */
void sw_graph_free(sw_graph *g);
void sw_release_resident(void);
/* Device memory comes from a caching pool (steady-state builds do no hipMalloc).  sw_pool_trim() hands every cached, unused
 * block back to the driver -- for a host program that shares the GPU's memory with another allocator (torch's, in the
 * multi-GPU choreography); the pool trims itself when one of its own hipMalloc calls fails.  HOST memory the library keeps for
 * the life of the process: the page-locked rings of the up- and download (32 + 64 MiB per device) and the page-locked blocks the
 * streaming ingest packs into (grown on demand to at most SEQWIN_AMD_PINNED_POOL_MB, default 1024; 0 disables them). */
void sw_pool_trim(void);
/* out[3] = { 1 if SEQWIN_AMD_POOL_DEBUG is on (every release waits for the device and poisons the block, every reuse waits and
 * checks the poison: a soak mode for the multi-device path), blocks found written after their release, host-side hand-overs of
 * a block between threads / streams so far (0 in single-stream builds) } */
void sw_pool_debug_stats(uint64_t *out);
/* out[3] = { occurrences of the resident index (0: none), sw_get_penalty calls served from it, sw_filter_kmers calls served from it } */
void sw_resident_stats(uint64_t *out);

/*
 * Replaces `_get_penalty_native(kmers, nodes, record_offsets, is_targets, n_cpu)`
 * (python_bindings.cpp:92-135) = seqwin::get_penalty (cpp/include/seqwin/filter.hpp:11-20,
 * cpp/src/seqwin/filter.cpp:15-137).  Mutates nodes[].{n_tar,n_neg,penalty} in place.
 * Unlike the reference, the kmers length is passed and node ranges are checked against it.
 * All validation failures are SW_ERR_VALUE with the reference's messages (filter.cpp:33-60,103-123).
 */
int sw_get_penalty(const sw_kmer *kmers, uint64_t n_kmers, sw_node *nodes, uint64_t n_nodes,
                   const uint32_t *record_offsets, uint64_t n_record_offsets, const uint8_t *is_targets,
                   uint64_t n_assemblies, uint64_t n_cpu);

/*
 * Replaces `_filter_kmers_native(kmers, nodes, used_hashes)` (python_bindings.cpp:137-168) =
 * seqwin::filter_kmers (cpp/src/seqwin/filter.cpp:139-201).  Two-phase: call with kmers_out ==
 * nodes_out == NULL to obtain the output sizes, then again with buffers of those sizes; the second call of the same
 * thread with the same arguments only copies the result of the first out (one upload, one compute per pair).
 */
int sw_filter_kmers(const sw_kmer *kmers, uint64_t n_kmers, const sw_node *nodes, uint64_t n_nodes,
                    const uint64_t *used_hashes, uint64_t n_used, sw_kmer *kmers_out, sw_node *nodes_out,
                    uint64_t *n_kmers_out, uint64_t *n_nodes_out);

/* ---- host ingest only (no GPU needed; used by the CPU test-suite to check the FASTA reader / packer) */
typedef struct sw_hostbatch sw_hostbatch;
/* Read + 2-bit pack FASTA / .gz files exactly as sw_build / sw_batch_from_fasta do
 * (replaces read_fasta, cpp/src/utils/fasta_reader.cpp:207-213), without uploading. */
int sw_host_ingest(const char *const *assembly_paths, size_t n_assemblies, uint64_t n_cpu, sw_hostbatch **out);
int sw_hostbatch_info(const sw_hostbatch *hb, uint64_t *n_assemblies, uint64_t *n_records, uint64_t *total_bp,
                      uint64_t *ids_bytes, uint64_t *n_runs);
/* record_offsets[n_assemblies + 1], ids blob, rec_len[n_records] (any may be NULL) */
int sw_hostbatch_tables(const sw_hostbatch *hb, uint32_t *record_offsets, char *ids_blob, uint32_t *rec_len);
/* Decode record `record_idx` back to ASCII (A/C/G/T; 'N' for every invalid base). */
int sw_hostbatch_record(const sw_hostbatch *hb, uint64_t record_idx, char *seq_out, uint64_t cap, uint64_t *len_out);
void sw_hostbatch_free(sw_hostbatch *hb);

/* ---- device-resident pipeline (what sw_build is made of; used by bench.py and multi-GPU) --- */

#define SW_MAX_WINDOW 4096u   /* largest window the tile kernels take directly */
#define SW_WINDOW_SPLIT 2048u /* windows above this are sketched with w' = 1024 and selected from that superset (sketch.hip: get_plan) */

/* Host ingest: read + pack FASTA files (fasta_reader.cpp:207-213 semantics) on n_cpu host threads and upload them
 * as one device-resident batch (assembly i of the batch = assembly_paths[i]). */
int sw_batch_from_fasta(const char *const *assembly_paths, size_t n_assemblies, uint64_t n_cpu, sw_batch **out);
/* Inputs that are all single-member .gz files and number 320 per host thread or more (SEQWIN_AMD_DEVICE_INFLATE=1: any number, =0: never)
 * are inflated, parsed and packed ON THE DEVICE (the gzip branch of fasta_reader.cpp:109-203 and the parse of :41-95, one
 * file per lane; csrc/ingest_dev.hip) -- same batch, bit for bit; anything irregular in a file sends the whole call through
 * the host route, which reports errors the way the reference does.  Number of batches this process built that way: */
uint64_t sw_device_gz_batches(void);

/* Synthetic batch generated ON DEVICE: n_genomes assemblies x records_per_genome records of
 * `record_len` bases each, derived from `n_ancestors` iid-uniform ancestors with per-base
 * substitution probability snp_ppm / 1e6 (counter-based RNG, `seed`).  No host ingest. */
int sw_batch_synthetic(uint64_t n_genomes, uint64_t records_per_genome, uint64_t record_len,
                       uint64_t n_ancestors, uint64_t snp_ppm, uint64_t seed, sw_batch **out);
/* Genomes [first_genome, first_genome + n_genomes) of the same job: bases and record ids are those the unsharded
 * batch holds for these genomes (one rank's contiguous assembly range, cpp/src/seqwin/build.cpp:350-356). */
int sw_batch_synthetic_shard(uint64_t n_genomes, uint64_t records_per_genome, uint64_t record_len,
                             uint64_t n_ancestors, uint64_t snp_ppm, uint64_t seed, uint64_t first_genome,
                             sw_batch **out);

/* Ragged draft assemblies for the measurements (r06): genome g of the job has 20 ... 300 contigs whose lengths follow a bell over
 * 200 bp ... 1.5 Mbp (200 * 2^x, x a sum of four uniforms; median ~17 kbp) until ~genome_bp bases are reached, cut from ancestor
 * g % n_ancestors with snp_ppm substitutions; one contig in ten carries one to three scaffold gaps of 10 ... 1000 N.  The shape of
 * real inputs (tests/targets.txt assemblies): short-record tiles, gap tiles and the generic kernel's list mode at scale. */
int sw_batch_synthetic_ragged(uint64_t n_genomes, uint64_t genome_bp, uint64_t n_ancestors, uint64_t snp_ppm, uint64_t seed, uint64_t first_genome,
                              sw_batch **out);
/* Copy the ASCII sequence of record `record_idx` (A/C/G/T, 'N' for invalid bases) back to the host;
 * used by tests to hand the same input to the oracle. *len_out receives the record length. */
int sw_batch_record(const sw_batch *b, uint64_t record_idx, char *seq_out, uint64_t cap, uint64_t *len_out);
/* Assemblies [first_assembly, first_assembly + n_assemblies) of a batch as plain FASTA files <dir>/g<global index>.fa (ids as
 * ingested, sequence lines of line_width bases, 0 = one line per record; bases outside the valid runs are written as N),
 * decoded from HBM by n_cpu host threads.  Tooling for the benchmarks and tests that feed a device-generated batch to the
 * FASTA boundary and to the CPU reference. */
int sw_batch_write_fasta(const sw_batch *b, uint64_t first_assembly, uint64_t n_assemblies, const char *dir, uint64_t n_cpu,
                         uint64_t line_width);

int sw_batch_info(const sw_batch *b, uint64_t *n_assemblies, uint64_t *n_records, uint64_t *total_bp,
                  uint64_t *device_bytes);
/* record_offsets[n_assemblies + 1] and the NUL-separated id blob of a batch. */
int sw_batch_records(const sw_batch *b, uint32_t *record_offsets, char *ids_blob, uint64_t ids_cap,
                     uint64_t *ids_bytes);
void sw_batch_free(sw_batch *b);

/* Timings of the last sw_index_build on this index, in milliseconds (HIP events on the build stream). */
typedef struct sw_timings {
    double total_ms;
    double sketch_ms;    /* the fused ntHash + window-minimum kernel (dominant kernel) */
    double order_ms;     /* tile-order compaction of the tuple stream */
    double nodes_ms;     /* radix sort by hash + run-length -> nodes / kmers / ranks */
    double counts_ms;    /* per-node target / non-target assembly counts + penalty (side stream, overlaps edges_ms) */
    double edges_ms;     /* adjacency pairs -> sort -> weights */
    uint64_t sketch_launches;
    uint64_t n_tiles;
    uint64_t total_bp;
    uint64_t n_windows;
    uint64_t ovf_tiles;  /* fast-class tiles done by the generic kernel's list pass: tiles crossing invalid bases + tiles with more suffix records than published */
    double plan_ms;      /* host + upload time the (batch, k, w) launch plan took when it was built (tile tables, roll LUTs);
                            cached in the batch afterwards, so it is outside total_ms except for the first build */
    uint64_t plan_cached; /* 1: this build found the plan in the batch's cache */
    /* r06: the plan's tiles by class -- fast kernel with 256-thread / 64-thread workgroups, generic kernel (records that are mostly
     * gaps, w < 4, k > 256), and the fast-class tiles that cross an invalid-base gap (done by the generic kernel's list mode) */
    uint64_t tiles_b256, tiles_b64, tiles_generic, tiles_gap;
} sw_timings;

/*
 * Build the index of a resident batch: sketch (ntHash + windowed minimizers, nthash_kmer.hpp /
 * minimizer.cpp:53-90) -> nodes/kmers (build.cpp:153-253, build_internals.cpp:159-251) ->
 * edges (build.cpp:177-189, build_internals.cpp:253-291) -> if is_targets != NULL, per-node
 * counts and penalty (filter.cpp:62-136).  Everything stays in HBM; `stream` is a hipStream_t
 * (0 = default stream).  The call is asynchronous only up to internal size read-backs.
 */
int sw_index_build(const sw_batch *b, uint64_t kmerlen, uint64_t windowsize, const uint8_t *is_targets,
                   uint64_t n_assemblies, void *stream, sw_index **out);

int sw_index_sizes(const sw_index *ix, uint64_t *n_kmers, uint64_t *n_nodes, uint64_t *n_edges);
int sw_index_timings(const sw_index *ix, sw_timings *t);
/* D2H copies of the final arrays (any pointer may be NULL to skip that array). */
int sw_index_export(const sw_index *ix, sw_kmer *kmers, sw_node *nodes, sw_edge *edges);
/* 64-bit checksums of the three arrays (sum over elements of a mix of the element and its index, so order matters),
 * computed on device; seqwin_amd.device.host_checksums is the numpy restatement. */
int sw_index_checksums(const sw_index *ix, uint64_t *kmers_sum, uint64_t *nodes_sum, uint64_t *edges_sum);
/* The same for one rank's slice of a sharded index: element i of the slice counts as element base + i of the
 * concatenated arrays, so the sums of all slices add up (mod 2^64) to the checksums of the unsharded index --
 * shard-count invariance (reference tests/smoke/test_graph.py:67-127) checked without gathering.  sums[3]. */
int sw_index_checksums_at(const sw_index *ix, uint64_t kmer_base, uint64_t node_base, uint64_t edge_base, uint64_t *sums);
/* Self-check of a resident index on the device -- the size-independent properties of the reference's output, for sets
 * too large to compare on the host.  out[10]: numbers of violations [0] nodes strictly ascending by hash
 * (build_internals.cpp:220), [1] node ranges partition the occurrences (:203-218), [2] (record_idx, pos) strictly ascending
 * inside a node (build.cpp:232-240), [3] edges strictly ascending by (first, second) (build_internals.cpp:261),
 * [4] first <= second, [5] 1 <= weight <= n_assemblies (build.cpp:177-189), [6] edge endpoints are node hashes,
 * [7] (scored != 0) 1 <= n_tar + n_neg <= min(node size, n_assemblies) (filter.cpp:62-136); then [8] the sum of the
 * edge weights and [9] reserved.  All of [0..7] are 0 for a correct index. */
int sw_index_verify(const sw_index *ix, uint64_t n_assemblies, int scored, uint64_t *out);
/* The sketch stage alone: (out_hash, pos, record_idx) of every minimizer in (record_idx, pos) order.
 * Two-phase like sw_filter_kmers: pass NULL buffers to get *n_out. */
int sw_sketch(const sw_batch *b, uint64_t kmerlen, uint64_t windowsize, void *stream, uint64_t *out_hash,
              sw_kmer *kmers, uint64_t cap, uint64_t *n_out);
void sw_index_free(sw_index *ix);

/* ---- next rows of the path, device-resident: what kmers.filter_graph does with the arrays ------
 * sums[3] = { sum n_tar, sum n_tar^2, sum n_tar * n_neg } over the nodes: the exact integer sums behind the
 * minimizer-sketch penalty threshold (src/seqwin/kmers.py:426-429). */
int sw_index_threshold_sums(const sw_index *ix, uint64_t *sums);
/* kmers._filter_edges_and_nodes (src/seqwin/kmers.py:132-173) without leaving HBM: *out holds the edges with
 * weight > edge_weight_th (already floored, np.uintp(th)) and the nodes that are an endpoint of one; no kmers. */
int sw_index_filter_graph(const sw_index *ix, uint64_t edge_weight_th, sw_index **out);
/* seqwin::filter_kmers (filter.cpp:139-201) on a resident index: *out holds the nodes of `nodes_from` whose hash is
 * in used_hashes (host array) and their kmers (taken from `ix`), ranges re-based; no edges. */
int sw_index_filter_kmers(const sw_index *ix, const sw_index *nodes_from, const uint64_t *used_hashes, uint64_t n_used,
                          sw_index **out);

/* ---- multi-GPU merge (one process per GPU; the exchange itself is done by the host side with
 *      torch.distributed / RCCL on the device buffers below).  Together these replace
 *      merge_thread_graphs (cpp/src/seqwin/build_internals.cpp:295-392) across GPUs. -------------- */

/* Device addresses of the index arrays (sw_kmer[n_kmers], sw_node[n_nodes], sw_edge[n_edges]). */
int sw_index_device_ptrs(const sw_index *ix, void **kmers, void **nodes, void **edges);
/* Write one row per occurrence, in the index's (hash, record_idx, pos) order, into the caller's DEVICE
 * buffer rows[n_kmers][2] (u64): {node hash, pos | (record_idx + rec_offset) << 32}
 * (record re-basing: build_internals.cpp:334-355, 243-246). */
int sw_index_occ_rows(const sw_index *ix, uint64_t rec_offset, void *rows_dev, void *stream);
/* Copy the edges, as rows[n_edges][3] (u64) = {first, second, weight}, into the caller's DEVICE buffer. */
int sw_index_edge_rows(const sw_index *ix, void *rows_dev, void *stream);
/* For n_bounds ascending hash bounds: occ_split[j] = number of occurrences whose node hash < node_bounds[j];
 * edge_split[j] = number of edges whose `first` < edge_bounds[j] (host outputs). */
int sw_index_splits(const sw_index *ix, const uint64_t *node_bounds, const uint64_t *edge_bounds, uint64_t n_bounds,
                    uint64_t *occ_split, uint64_t *edge_split, void *stream);
/* Build one rank's slice of the merged graph from exchanged rows (DEVICE buffers):
 * occ_rows[n_occ][2] concatenated in source-rank order, edge_rows[n_edge_rows][3] = {first, second, weight}.
 * Node ranges are offset by kmer_base (number of occurrences owned by lower ranks).  record_offsets /
 * is_targets are the GLOBAL host arrays (is_targets may be NULL to skip the counts). */
int sw_index_merge(const void *occ_rows_dev, uint64_t n_occ, const void *edge_rows_dev, uint64_t n_edge_rows,
                   uint64_t kmer_base, const uint32_t *record_offsets, const uint8_t *is_targets,
                   uint64_t n_assemblies, void *stream, sw_index **out);

/* ---- multi-GPU, tuple-exchange form (what bench.py --gpus N times; seqwin_amd/dist.py drives it) ---------
 * Instead of building a partial graph per GPU and merging (above), every GPU sketches its shard and the TUPLES are
 * exchanged by hash range, so each occurrence is sorted exactly once, by its owner:
 *   sw_occ_sketch -> sw_occ_partition -> [all_to_all rows] -> sw_slice_build (nodes / kmers / counts + rank of every
 *   received row) -> [all_to_all ranks back] -> sw_occ_adjacency -> [all_to_all adjacency rows] -> sw_slice_edges. */
typedef struct sw_occ sw_occ; /* device-resident (out_hash, pos|record) stream of one shard in (record_idx, pos) order */
int sw_occ_sketch(const sw_batch *b, uint64_t kmerlen, uint64_t windowsize, void *stream, sw_occ **out);
int sw_occ_size(const sw_occ *o, uint64_t *n, double *sketch_ms);
/* A shard's tuple stream straight from its FASTA files, the files streamed through HBM in chunks of ~chunk_bp bases (0: one
 * chunk) -- low_memory inside a multi-device build (build.cpp:264-325 keeps the peak down by a second pass; here only 16 B per
 * minimizer of a finished chunk stay resident).  *batch_out holds the shard's record tables only (no packed bases). */
int sw_occ_sketch_paths(const char *const *assembly_paths, size_t n_assemblies, uint64_t kmerlen, uint64_t windowsize, uint64_t n_cpu,
                        uint64_t chunk_bp, void *stream, sw_batch **batch_out, sw_occ **occ_out);
void sw_occ_free(sw_occ *o);
/* Stable partition by owner = number of ascending bounds <= out_hash.  DEVICE outputs: rows[n][2] =
 * {out_hash, pos | (record_idx + rec_offset) << 32} grouped by owner, perm[n] (u32: original index of row j; may be
 * NULL -- the handle remembers the partition and sw_occ_adjacency walks it again instead of scattering through perm).
 * HOST output: counts[n_bounds + 1]. */
int sw_occ_partition(const sw_occ *o, const uint64_t *bounds, uint64_t n_bounds, uint64_t rec_offset, void *rows_dev,
                     void *perm_dev, uint64_t *counts, void *stream);
/* Owner: nodes / kmers (+ counts and penalty if is_targets) of its hash range from received rows (concatenated in
 * source-rank order = global (record_idx, pos) order).  ranks_dev[n] (u32, DEVICE) receives the slice-local node rank
 * of every received row.  Node ranges are offset by kmer_base.  The index has no edges yet. */
int sw_slice_build(const void *rows_dev, uint64_t n, uint64_t kmer_base, const uint32_t *record_offsets,
                   const uint8_t *is_targets, uint64_t n_assemblies, void *ranks_dev, void *stream, sw_index **out);
/* Copy the node hashes (u64[n_nodes]) of an index into a DEVICE buffer. */
int sw_index_node_hashes(const sw_index *ix, void *dst_dev, void *stream);
/* Source: adjacency rows of consecutive minimizers of a record, from the GLOBAL node rank of every partitioned row
 * (rank_by_row_dev, u32[n]; perm_dev is only read when the handle was not partitioned by sw_occ_partition and may be NULL
 * otherwise); grouped by edge owner = number of ascending rank_bounds <= rank_lo.
 * asm_bits == 0: DEVICE rows[<= n-1][2] = {(rank_lo << n_bits) | rank_hi, global assembly};
 * asm_bits  > 0 (requires 2 n_bits + asm_bits <= 64): DEVICE rows[<= n-1] = one packed key
 *               (((rank_lo << n_bits) | rank_hi) << asm_bits) | global assembly  -- half the exchange volume.
 * HOST output counts[n_bounds + 1]. */
int sw_occ_adjacency(const sw_occ *o, const void *perm_dev, const void *rank_by_row_dev, uint64_t n_bits,
                     uint64_t asm_bits, uint64_t asm_base, const uint64_t *rank_bounds, uint64_t n_bounds, void *rows_dev,
                     uint64_t *counts, void *stream);
/* Owner: edges of its rank range from received adjacency rows (source-rank order; same format as above), hashes
 * looked up in the job-wide rank -> hash table (DEVICE u64[total nodes]).  Attaches the edges to `ix`. */
int sw_slice_edges(sw_index *ix, const void *adj_rows_dev, uint64_t m, uint64_t n_bits, uint64_t asm_bits,
                   const void *rank_hash_dev, void *stream);

/* ---- the sort primitive of the index stage, on its own (tests, timing) ------------------------------------------------
 * Stable LSD radix sort of n DEVICE u64 keys by bits [begin_bit, end_bit) -- lsd_radix_sort_key
 * (cpp/src/seqwin/build_internals.cpp:76-144) on the device: the hand-written onesweep of csrc/radix.hip (rocPRIM's under
 * SEQWIN_AMD_SORT=rocprim).  keys_dev / alt_dev are a double buffer of n keys each; *sorted_in_alt tells which one holds the
 * result; *ms (may be NULL) the device time. */
int sw_sort_keys64(void *keys_dev, void *alt_dev, uint64_t n, uint64_t begin_bit, uint64_t end_bit, void *stream, int *sorted_in_alt,
                   double *ms);
/* The pair form -- lsd_radix_sort (cpp/src/seqwin/build_internals.cpp:76-110) as the node sort uses it: n DEVICE u32 keys, each
 * with a 16-byte payload, stably by key bits [0, end_bit), end_bit in {8, 16, 24, 32} (csrc/radix.hip's pair passes; rocPRIM's
 * under SEQWIN_AMD_SORT=rocprim or SEQWIN_AMD_PAIR_SORT=rocprim).  Double buffers of n keys / n payloads each. */
/* How the radix passes rank keys inside a wave on the current device: *mode = 1 by one LDS atomic per key -- the device passed
 * the start-up check that the lanes of one LDS atomic are served in lane order (what keeps those passes stable) --, 0 by ballots
 * (the check failed, or SEQWIN_AMD_RADIX_RANK=ballot).  Runs the check if it has not run yet. */
int sw_radix_rank_mode(int *mode);
/* Always-on order guards (r05): the kernels that stream over the sorted occurrences (k_nodes) and the sorted edge keys
 * (k_rle_keys) count every place where (hash, stream index) resp. the key does not ascend -- the stability contract of
 * lsd_radix_sort, cpp/src/seqwin/build_internals.cpp:76-144.  A build that sees such a place logs a WARNING, switches the device
 * to ballot / rocPRIM ranking for the rest of the process and sorts again (bit-identical result).  These counters say how often
 * that happened in this process (0 on a healthy device; SEQWIN_AMD_FAULT_INJECT=rank makes the suite take the path). */
int sw_order_guard_trips(uint64_t *node_sort, uint64_t *edge_sort);
int sw_sort_pairs32(void *keys_dev, void *keys_alt_dev, void *vals_dev, void *vals_alt_dev, uint64_t n, uint64_t end_bit, void *stream,
                    int *sorted_in_alt, double *ms);

/* ---- pairs form of the adjacency exchange (what dist.py uses whenever every slice marked its ranks) ---------------------
 * The weight of an edge is the number of its adjacency records minus the records that repeat the pair inside one
 * assembly, and only records touching an occurrence whose node occurs more than once in its assembly can do that.  A slice
 * build with a record table and fewer than 2^31 nodes IN THE SLICE marks those occurrences in bit 31 of the slice-local
 * ranks it returns (sw_index_ranks_marked); the sources then send ONE 64-bit key per adjacency record plus the few candidate
 * records with their assembly, and the owners count.
 * Ranks travel slice-LOCAL (32 bits); a source re-bases them itself: global rank = node_base[owner of the tuple] + local
 * rank, which may need more than 32 bits (100 000 random genomes: 5e9 distinct minimizers; the reference indexes nodes with
 * size_t, cpp/include/seqwin/graph.hpp:28-41, cpp/src/seqwin/build_internals.cpp:159-223).  An edge belongs to the owner of
 * its rank_lo's range [lo_base, next lo_base) (rank_bounds), so its key is
 *     (rank_lo - lo_base) << hi_bits | rank_hi,      hi_bits = bits of the total node count, lo_bits = bits of the widest range,
 * lo_bits + hi_bits <= 64 (else SW_ERR_RUNTIME: more GPUs are needed). */
int sw_index_ranks_marked(const sw_index *ix, int *marked);
/* Source: keys_dev[<= n-1] (u64, DEVICE) of every adjacency record, grouped by edge owner (counts[n_bounds + 1]); the candidate
 * records stay in the handle as rows {key, global assembly} grouped by owner (cand_counts[n_bounds + 1]) until
 * sw_occ_candidates copies them (sum of cand_counts rows of 2 u64) to a DEVICE buffer.
 * rank_by_row_dev: slice-local ranks with the repeat mark, in the order of the rows sw_occ_partition wrote (what the reverse
 * all-to-all delivers); node_base[n_owners + 1]: prefix of the node counts of the tuple partition's owners (HOST);
 * rank_bounds[n_bounds]: ascending splitters of the GLOBAL rank space (HOST); key_bits[2] receives {lo_bits, hi_bits}. */
int sw_occ_adjacency_pairs(const sw_occ *o, const void *rank_by_row_dev, const uint64_t *node_base, uint64_t n_owners, uint64_t asm_base,
                           const uint64_t *rank_bounds, uint64_t n_bounds, void *keys_dev, uint64_t *counts, uint64_t *cand_counts,
                           uint64_t *key_bits, void *stream);
int sw_occ_candidates(const sw_occ *o, void *rows_dev, void *stream);
/* Owner: edges of its rank range from the received keys and candidate rows (source-rank order).  keys_dev is sorted in place
 * (its contents are not preserved).  lo_base: first global rank of this owner's range.  rank_hash_dev (DEVICE u64[n_owners * pad]): the job-wide rank -> hash table as
 * all_gather_into_tensor leaves it -- slice owner o's node hashes (sw_index_node_hashes) at [o * pad, o * pad + its count). */
int sw_slice_edges_pairs(sw_index *ix, void *keys_dev, uint64_t m, const void *cand_rows_dev, uint64_t n_cand, uint64_t lo_bits,
                         uint64_t hi_bits, uint64_t lo_base, uint64_t asm_bits, const void *rank_hash_dev, const uint64_t *node_base,
                         uint64_t n_owners, uint64_t pad, void *stream);
/* The same without the job-wide table (rank_hash_dev == NULL above: 8 B per node of the WHOLE job on every GPU -- 40 GB at
 * 5e9 nodes): the edges then hold global ranks and their hashes are asked from the node owners, 12 B per distinct endpoint.
 *   1. edge owner: sw_index_edge_hash_requests -> *n_requests distinct endpoint ranks, as owner-LOCAL ranks (u32) in ascending
 *      global order, counts[n_owners] of them for each node owner; sw_index_edge_hash_request_rows copies them out (DEVICE);
 *   2. the requests travel to the node owners (all-to-all by counts); a node owner answers the ranks it received with
 *      sw_index_node_hash_lookup: hashes_dev[i] = hash of its node local_ranks_dev[i];
 *   3. the replies travel back in request order; sw_index_edge_hash_attach(ix, replies_dev, *n_requests) writes them into the
 *      edges.  Until then sw_index_export / _checksums of this index see ranks in edges.first / .second. */
int sw_index_edge_hash_requests(sw_index *ix, const uint64_t *node_base, uint64_t n_owners, uint64_t *counts, uint64_t *n_requests,
                                void *stream);
int sw_index_edge_hash_request_rows(const sw_index *ix, void *local_ranks_dev, void *stream);
int sw_index_node_hash_lookup(const sw_index *ix, const void *local_ranks_dev, uint64_t n, void *hashes_dev, void *stream);
int sw_index_edge_hash_attach(sw_index *ix, const void *replies_dev, uint64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SEQWIN_HIP_H */
