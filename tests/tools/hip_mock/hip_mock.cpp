// hip_mock.cpp -- a HOST-ONLY stand-in for libamdhip64: the 40 runtime entry points libseqwin_hip.so imports, no GPU.
//
// Why: ThreadSanitizer cannot run on the GPU box, and the host side of a multi-device build (csrc/multi.hip: worker threads,
// rendezvous, peer pulls; csrc/api.hip: the caching pool's hand-over of blocks between threads and streams, spare events per
// device, the staged route) is exactly the code a first run on an 8-GPU node executes for the first time.
// The library's .hip files are compiled with `hipcc --cuda-host-only -fsanitize=thread` and linked against THIS file instead of
// the HIP runtime (tests/tools/hip_mock/Makefile).  What the mock models:
//   * devices: HIP_MOCK_DEVICES of them (default 4), a current device per thread, "device memory" = zeroed host memory;
//   * streams as FIFO queues, each drained by its own thread: an asynchronous copy / memset / kernel launch is an entry of the
//     queue, executed by the stream's thread in order -- so two streams really run concurrently, and a block handed from one to
//     another without an event or a host synchronisation IS a data race TSan sees;
//   * the legacy NULL stream of a device: its own queue; the blocking calls (hipMemcpy, hipMemset) run on it and wait for it
//     alone -- streams made with hipStreamNonBlocking (every stream of the library) do not synchronise with it;
//   * events: record = a marker in the stream's queue, synchronize / query / stream-wait on the last recorded marker; an event
//     recorded on a stream of another device fails with hipErrorInvalidHandle, as the real runtime does;
//   * kernels are NO-OPS (their queue entry orders what follows; results read back are zeros), except functions a harness
//     enqueues itself with hip_mock_enqueue() -- tests/tools/hip_mock/multi_choreography.cpp touches the buffers that way;
//   * peer copies are memcpy on the pulling stream's thread; hipDeviceCanAccessPeer answers 0 under HIP_MOCK_NO_PEER=1;
//   * hipFree waits for every stream of the block's device, poisons the block and frees it (use after free -> TSan report).
// Test infrastructure only: nothing under seqwin_amd/ knows about it.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "hip_mock.h"

namespace {

int n_devices()
{
    static const int n = [] {
        const char *e = getenv("HIP_MOCK_DEVICES");
        const int v = e ? atoi(e) : 4;
        return v > 0 && v <= 64 ? v : 4;
    }();
    return n;
}

thread_local int t_device = 0;
thread_local hipError_t t_last_error = hipSuccess;
hipError_t fail(hipError_t e)
{
    t_last_error = e;
    return e;
}

}  // namespace

// ---- streams --------------------------------------------------------------------------------------------------------
struct ihipStream_t {
    int device = 0;
    unsigned flags = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::function<void()>> q;
    uint64_t enq = 0, done = 0;   // entries ever enqueued / completed
    bool stop = false;
    std::thread th;
    ihipStream_t(int dev, unsigned fl) : device(dev), flags(fl)
    {
        th = std::thread([this] { run(); });
    }
    void run()
    {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv_work.wait(lock, [&] { return stop || !q.empty(); });
            if (q.empty()) return;
            std::function<void()> f = std::move(q.front());
            q.pop_front();
            lock.unlock();
            f();
            lock.lock();
            ++done;
            cv_idle.notify_all();
        }
    }
    void push(std::function<void()> f)
    {
        std::lock_guard<std::mutex> lock(mu);
        q.push_back(std::move(f));
        ++enq;
        cv_work.notify_one();
    }
    void drain()   // everything enqueued so far has run
    {
        std::unique_lock<std::mutex> lock(mu);
        const uint64_t target = enq;
        cv_idle.wait(lock, [&] { return done >= target; });
    }
};

namespace {

struct Runtime {
    std::mutex mu;
    std::vector<ihipStream_t *> null_stream;            // per device, made on first use
    std::set<ihipStream_t *> streams;                   // every live stream (the NULL streams too)
    std::map<void *, std::pair<int, size_t>> blocks;    // device memory: ptr -> (device, bytes)
    std::vector<size_t> used;                           // per device
    std::set<void *> host_blocks;
    std::set<std::pair<int, int>> peer;                 // (device, peer) enabled
    std::atomic<uint64_t> launches{0}, copies{0}, peer_copies{0}, frees{0};
    Runtime() : null_stream(n_devices(), nullptr), used(n_devices(), 0) {}
};
Runtime &rt()
{
    static Runtime *r = new Runtime;   // leaked: stream threads outlive static destruction
    return *r;
}

ihipStream_t *resolve(hipStream_t s)   // nullptr -> the calling thread's device's NULL stream
{
    if (s) return s;
    Runtime &r = rt();
    std::lock_guard<std::mutex> lock(r.mu);
    ihipStream_t *&ns = r.null_stream[t_device];
    if (!ns) {
        ns = new ihipStream_t(t_device, 0);
        r.streams.insert(ns);
    }
    return ns;
}

std::vector<ihipStream_t *> streams_of(int dev)
{
    Runtime &r = rt();
    std::lock_guard<std::mutex> lock(r.mu);
    std::vector<ihipStream_t *> v;
    for (ihipStream_t *s : r.streams)
        if (s->device == dev) v.push_back(s);
    return v;
}

constexpr size_t TOTAL_MEM = 288ull << 30;

}  // namespace

// ---- events ---------------------------------------------------------------------------------------------------------
// Every hipEventRecord makes a record of its own; the event points at its latest one.  A wait / synchronize / query refers to the
// record that was the latest WHEN IT WAS ISSUED, as in the real runtime -- an event that goes back to a pool and is recorded again
// on another stream must not release an earlier waiter.
struct EventRecord {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    void complete()
    {
        std::lock_guard<std::mutex> lock(mu);
        done = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return done; });
    }
    bool is_done()
    {
        std::lock_guard<std::mutex> lock(mu);
        return done;
    }
};
struct ihipEvent_t {
    int device = 0;
    std::mutex mu;
    std::shared_ptr<EventRecord> last;   // null: never recorded (complete)
    std::shared_ptr<EventRecord> latest()
    {
        std::lock_guard<std::mutex> lock(mu);
        return last;
    }
};

extern "C" {

void hip_mock_enqueue(hipStream_t stream, void (*fn)(void *), void *arg)
{
    resolve(stream)->push([fn, arg] { fn(arg); });
}
void hip_mock_stats(uint64_t *out4)
{
    out4[0] = rt().launches.load();
    out4[1] = rt().copies.load();
    out4[2] = rt().peer_copies.load();
    out4[3] = rt().frees.load();
}

// ---- devices ----
hipError_t hipGetDeviceCount(int *count) { *count = n_devices(); return hipSuccess; }
hipError_t hipGetDevice(int *dev) { *dev = t_device; return hipSuccess; }
hipError_t hipSetDevice(int dev)
{
    if (dev < 0 || dev >= n_devices()) return fail(hipErrorInvalidDevice);
    t_device = dev;
    return hipSuccess;
}
hipError_t hipGetLastError(void)
{
    const hipError_t e = t_last_error;
    t_last_error = hipSuccess;
    return e;
}
const char *hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorInvalidHandle: return "invalid resource handle";
    case hipErrorNotReady: return "device not ready";
    case hipErrorPeerAccessAlreadyEnabled: return "peer access is already enabled";
    default: return "mock error";
    }
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int dev)
{
    if (dev < 0 || dev >= n_devices()) return fail(hipErrorInvalidDevice);
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hip_mock device %d", dev);
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->totalGlobalMem = TOTAL_MEM;
    p->sharedMemPerBlock = 160 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->regsPerBlock = 65536;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->maxThreadsDim[0] = p->maxThreadsDim[1] = p->maxThreadsDim[2] = 1024;
    p->maxGridSize[0] = p->maxGridSize[1] = p->maxGridSize[2] = 2147483647;
    p->clockRate = 2400000;
    p->multiProcessorCount = 256;
    p->maxThreadsPerMultiProcessor = 2048;
    p->l2CacheSize = 4 << 20;
    p->major = 9;
    p->minor = 5;
    p->pciBusID = 0x10 + dev;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t attr, int dev)
{
    if (dev < 0 || dev >= n_devices()) return fail(hipErrorInvalidDevice);
    switch (attr) {
    case hipDeviceAttributeWarpSize: *v = 64; break;
    case hipDeviceAttributeMultiprocessorCount: *v = 256; break;
    case hipDeviceAttributeMaxThreadsPerBlock: *v = 1024; break;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: *v = 160 * 1024; break;
    case hipDeviceAttributeMaxSharedMemoryPerMultiprocessor: *v = 160 * 1024; break;
    case hipDeviceAttributeMaxThreadsPerMultiProcessor: *v = 2048; break;
    case hipDeviceAttributeClockRate: *v = 2400000; break;
    case hipDeviceAttributeL2CacheSize: *v = 4 << 20; break;
    case hipDeviceAttributeMaxGridDimX: *v = 2147483647; break;
    case hipDeviceAttributeMaxBlockDimX: *v = 1024; break;
    default: *v = 0; break;
    }
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer)
{
    if (dev < 0 || dev >= n_devices() || peer < 0 || peer >= n_devices()) return fail(hipErrorInvalidDevice);
    const char *e = getenv("HIP_MOCK_NO_PEER");
    *can = (dev != peer && !(e && atoi(e))) ? 1 : 0;
    return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned)
{
    if (peer < 0 || peer >= n_devices() || peer == t_device) return fail(hipErrorInvalidDevice);
    Runtime &r = rt();
    std::lock_guard<std::mutex> lock(r.mu);
    if (!r.peer.insert(std::make_pair(t_device, peer)).second) return fail(hipErrorPeerAccessAlreadyEnabled);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void)
{
    for (ihipStream_t *s : streams_of(t_device)) s->drain();
    return hipSuccess;
}

// ---- memory ----
hipError_t hipMalloc(void **p, size_t bytes)
{
    Runtime &r = rt();
    {
        std::lock_guard<std::mutex> lock(r.mu);
        if (r.used[t_device] + bytes > TOTAL_MEM) return fail(hipErrorOutOfMemory);
    }
    void *q = nullptr;
    if (posix_memalign(&q, 256, bytes ? bytes : 1) != 0) return fail(hipErrorOutOfMemory);
    memset(q, 0, bytes);
    std::lock_guard<std::mutex> lock(r.mu);
    r.blocks[q] = std::make_pair(t_device, bytes);
    r.used[t_device] += bytes;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    Runtime &r = rt();
    int dev;
    size_t bytes;
    {
        std::lock_guard<std::mutex> lock(r.mu);
        auto it = r.blocks.find(p);
        if (it == r.blocks.end()) return fail(hipErrorInvalidValue);
        dev = it->second.first;
        bytes = it->second.second;
    }
    for (ihipStream_t *s : streams_of(dev)) s->drain();   // (the runtime's hipFree waits for the block's device)
    {
        std::lock_guard<std::mutex> lock(r.mu);
        r.blocks.erase(p);
        r.used[dev] -= bytes;
    }
    memset(p, 0xDD, bytes);
    free(p);
    ++r.frees;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b)
{
    Runtime &r = rt();
    std::lock_guard<std::mutex> lock(r.mu);
    *total_b = TOTAL_MEM;
    *free_b = TOTAL_MEM - r.used[t_device];
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned)
{
    void *q = nullptr;
    if (posix_memalign(&q, 4096, bytes ? bytes : 1) != 0) return fail(hipErrorOutOfMemory);
    std::lock_guard<std::mutex> lock(rt().mu);
    rt().host_blocks.insert(q);
    *p = q;
    return hipSuccess;
}
hipError_t hipHostFree(void *p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lock(rt().mu);
        if (!rt().host_blocks.erase(p)) return fail(hipErrorInvalidValue);
    }
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s)
{
    ++rt().copies;
    resolve(s)->push([=] { if (n) memmove(dst, src, n); });
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t s)
{
    if (dst_dev < 0 || dst_dev >= n_devices() || src_dev < 0 || src_dev >= n_devices()) return fail(hipErrorInvalidDevice);
    ++rt().peer_copies;
    resolve(s)->push([=] { if (n) memmove(dst, src, n); });
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
    ++rt().copies;
    ihipStream_t *ns = resolve(nullptr);
    ns->push([=] { if (n) memmove(dst, src, n); });
    ns->drain();
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s)
{
    resolve(s)->push([=] { if (n) memset(dst, v, n); });
    return hipSuccess;
}
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t s)
{
    resolve(s)->push([=] {
        uint32_t *p = (uint32_t *)dst;
        for (size_t i = 0; i < count; ++i) p[i] = (uint32_t)v;
    });
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t n)
{
    ihipStream_t *ns = resolve(nullptr);
    ns->push([=] { if (n) memset(dst, v, n); });
    ns->drain();
    return hipSuccess;
}

// ---- streams ----
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags)
{
    ihipStream_t *st = new ihipStream_t(t_device, flags);
    std::lock_guard<std::mutex> lock(rt().mu);
    rt().streams.insert(st);
    *s = st;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    resolve(s)->drain();
    return hipSuccess;
}
int hipGetStreamDeviceId(hipStream_t s) { return resolve(s)->device; }

// ---- events ----
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
    ihipEvent_t *ev = new ihipEvent_t;
    ev->device = t_device;
    *e = ev;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    if (!e) return fail(hipErrorInvalidHandle);
    delete e;   // (queued markers and waiters hold their record by shared_ptr)
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    if (!e) return fail(hipErrorInvalidHandle);
    ihipStream_t *st = resolve(s);
    if (st->device != e->device) return fail(hipErrorInvalidHandle);   // (as the real runtime: ADVICE r4)
    auto rec = std::make_shared<EventRecord>();
    {
        std::lock_guard<std::mutex> lock(e->mu);
        e->last = rec;
    }
    st->push([rec] { rec->complete(); });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    if (!e) return fail(hipErrorInvalidHandle);
    if (auto rec = e->latest()) rec->wait();
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e)
{
    if (!e) return fail(hipErrorInvalidHandle);
    auto rec = e->latest();
    return (!rec || rec->is_done()) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    if (!a || !b) return fail(hipErrorInvalidHandle);
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    if (!e) return fail(hipErrorInvalidHandle);
    if (auto rec = e->latest()) resolve(s)->push([rec] { rec->wait(); });
    return hipSuccess;
}

// ---- kernels: registration is a no-op, a launch is an empty entry of the stream's queue ----
void **__hipRegisterFatBinary(const void *)
{
    static void *handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void **, void *, const char *, size_t, unsigned) {}

namespace {
struct CallConfig {
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
};
thread_local std::vector<CallConfig> t_call_stack;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    t_call_stack.push_back(CallConfig{grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    if (t_call_stack.empty()) return fail(hipErrorInvalidValue);
    const CallConfig c = t_call_stack.back();
    t_call_stack.pop_back();
    *grid = c.grid;
    *block = c.block;
    *shmem = c.shmem;
    *stream = c.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t, hipStream_t s)
{
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || (size_t)block.x * block.y * block.z > 1024) return fail(hipErrorInvalidValue);
    ++rt().launches;
    resolve(s)->push([] {});
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t)
{
    *n = 2;
    return hipSuccess;
}

}  // extern "C"
