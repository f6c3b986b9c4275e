// multi_mock_engine.h -- force-included (-include) when csrc/multi.hip is compiled for tests/tools/hip_mock/multi_choreography.cpp:
// the engine entry points the workers call, and the two functions multi.hip defines, get other names, so that the harness can
// supply a fake engine (same prototypes: include/seqwin_hip.h is read through these macros as well) next to the real library.
#pragma once
#define sw_batch_from_fasta mock_batch_from_fasta
#define sw_batch_info mock_batch_info
#define sw_batch_records mock_batch_records
#define sw_batch_free mock_batch_free
#define sw_occ_sketch mock_occ_sketch
#define sw_occ_size mock_occ_size
#define sw_occ_sketch_paths mock_occ_sketch_paths
#define sw_occ_free mock_occ_free
#define sw_occ_partition mock_occ_partition
#define sw_slice_build mock_slice_build
#define sw_index_node_hashes mock_index_node_hashes
#define sw_occ_adjacency_pairs mock_occ_adjacency_pairs
#define sw_occ_candidates mock_occ_candidates
#define sw_slice_edges_pairs mock_slice_edges_pairs
#define sw_index_edge_hash_requests mock_index_edge_hash_requests
#define sw_index_edge_hash_request_rows mock_index_edge_hash_request_rows
#define sw_index_node_hash_lookup mock_index_node_hash_lookup
#define sw_index_edge_hash_attach mock_index_edge_hash_attach
#define build_multi_device build_multi_device_mock
#define devices_from_env devices_from_env_mock
