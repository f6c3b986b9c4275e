// hip_mock.h -- what a harness may ask of the mock runtime beyond the HIP API (tests/tools/hip_mock/hip_mock.cpp)
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
extern "C" {
// run fn(arg) on the stream's thread, in queue order: a stand-in for a kernel that touches the buffers `arg` names
void hip_mock_enqueue(hipStream_t stream, void (*fn)(void *), void *arg);
// {kernel launches, copies, peer copies, hipFree calls} so far
void hip_mock_stats(uint64_t *out4);
}
