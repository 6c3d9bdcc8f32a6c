// TEST-ONLY stand-in for the HIP runtime and for the kernel launchers behind csrc/vdyn_capi.hip, so that the
// library's own host layer -- argument checks, staging offsets, scratch growth, the pipelined upload, the peer-exchange
// bookkeeping -- runs on a CPU under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_layer_sanitizers.py).
// Not a CPU path of the product and not a parity artefact: no vehicle dynamics is computed here and oracle/ is never
// touched.  A "kernel" below READS every byte its launch arguments promise as input and WRITES every byte they promise
// as output (ASan sees any promise the host layer's buffers do not keep), with content that is an exact function of
// the inputs, so the Python driver can also check that every region arrived where it should.
//
// Execution model: DEFERRED and adversarial.  Nothing enqueued on a stream runs at the call; operations sit in
// per-stream FIFO queues and run only when the host waits (hipStreamSynchronize, hipEventSynchronize, hipFree, ...),
// one at a time, the stream picked at random (seed: HIPSTUB_SEED) among those whose head operation is not blocked on
// an event.  A missing dependency between streams, or a host buffer reused before the copy that reads it has run, shows
// up as wrong data (or as a deadlock report) under some seed instead of passing because the stub happened to be
// synchronous.  "Device" memory is plain malloc memory of exactly the requested size.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../python-motionplanning_amd/csrc/vdyn_internal.hpp"

namespace {

struct Event {
    unsigned long long recorded = 0, completed = 0;
};
struct Op {
    enum Kind { RUN, RECORD, WAIT } kind;
    std::function<void()> fn;
    Event *ev = nullptr;
    unsigned long long gen = 0;
};
struct Stream {
    std::deque<Op> q;
};

std::recursive_mutex g_mu;             // the runtime is called from one thread at a time per handle; tests may use two handles
std::set<Stream *> g_streams;
std::set<Event *> g_events;
Stream g_null_stream;
std::map<void *, size_t> g_dev, g_pinned;
unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
int g_fail_malloc_after = -1;          // HIPSTUB_FAIL_MALLOC_AFTER: the n-th hipMalloc from now fails (allocation-failure paths)
bool g_init = false;
volatile unsigned long long g_sink = 0;

void init()
{
    if (g_init) return;
    g_init = true;
    g_streams.insert(&g_null_stream);
    if (const char *s = std::getenv("HIPSTUB_SEED")) g_rng ^= 0x2545F4914F6CDD1Dull * (std::strtoull(s, nullptr, 10) + 1);
    if (const char *s = std::getenv("HIPSTUB_FAIL_MALLOC_AFTER")) g_fail_malloc_after = std::atoi(s);
}
unsigned long long rnd()
{
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return g_rng;
}
Stream *sp(hipStream_t s) { return s ? reinterpret_cast<Stream *>(s) : &g_null_stream; }

[[noreturn]] void die(const char *msg)
{
    std::fprintf(stderr, "hipstub: %s\n", msg);
    std::abort();
}

bool runnable(Stream *s)
{
    if (s->q.empty()) return false;
    const Op &o = s->q.front();
    return o.kind != Op::WAIT || o.ev->completed >= o.gen;
}
// run ONE operation of a randomly chosen runnable stream; false when nothing can run
bool step()
{
    std::vector<Stream *> ready;
    for (Stream *s : g_streams)
        if (runnable(s)) ready.push_back(s);
    if (ready.empty()) return false;
    Stream *s = ready[rnd() % ready.size()];
    Op o = std::move(s->q.front());
    s->q.pop_front();
    if (o.kind == Op::RUN) o.fn();
    else if (o.kind == Op::RECORD) o.ev->completed = std::max(o.ev->completed, o.gen);
    return true;
}
template <typename Done>
void run_until(Done done, const char *what)
{
    while (!done())
        if (!step()) {
            std::fprintf(stderr, "hipstub: deadlock while waiting for %s (an event waited on is never recorded)\n", what);
            std::abort();
        }
}
void drain_all()
{
    run_until([] {
        for (Stream *s : g_streams)
            if (!s->q.empty()) return false;
        return true;
    }, "the device to go idle");
}
void enqueue(hipStream_t s, std::function<void()> fn)
{
    Op o;
    o.kind = Op::RUN;
    o.fn = std::move(fn);
    sp(s)->q.push_back(std::move(o));
}

bool inside(const std::map<void *, size_t> &m, const void *p, size_t bytes)
{
    auto it = m.upper_bound(const_cast<void *>(p));
    if (it == m.begin()) return false;
    --it;
    const char *b = static_cast<const char *>(it->first);
    return static_cast<const char *>(p) >= b && static_cast<const char *>(p) + bytes <= b + it->second;
}

// ---- what a "kernel" does with its arguments ---------------------------------------------------------------------
template <typename T>
double read_all(const T *p, size_t count)          // every element is read (ASan), the sum comes back (exact for small integers)
{
    double s = 0;
    if (p)
        for (size_t i = 0; i < count; ++i) s += (double)p[i];
    g_sink += (unsigned long long)(long long)s;
    return s;
}
template <typename T>
void write_all(T *p, size_t count, double base)    // element i <- base + (i mod 251)
{
    if (p)
        for (size_t i = 0; i < count; ++i) p[i] = (T)(base + (double)(i % 251));
}

}  // namespace

// ------------------------------------------------------------------------------------------------ HIP runtime ----
extern "C" {

hipError_t hipGetDeviceCount(int *count)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    const char *e = std::getenv("HIPSTUB_DEVICES");
    *count = e ? std::atoi(e) : 1;
    return *count > 0 ? hipSuccess : hipErrorNoDevice;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int)
{
    std::memset(prop, 0, sizeof(*prop));
    const char *e = std::getenv("HIPSTUB_ARCH");
    std::snprintf(prop->gcnArchName, sizeof(prop->gcnArchName), "%s", e ? e : "gfx950:sramecc+:xnack-");
    std::snprintf(prop->name, sizeof(prop->name), "hipstub");
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "hipstub error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *stream, unsigned)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    Stream *s = new Stream;
    g_streams.insert(s);
    *stream = reinterpret_cast<hipStream_t>(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    Stream *s = sp(stream);
    if (!g_streams.count(s)) die("hipStreamSynchronize on a destroyed / unknown stream");
    run_until([s] { return s->q.empty(); }, "a stream");
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Stream *s = sp(stream);
    if (!g_streams.count(s) || s == &g_null_stream) die("hipStreamDestroy on a destroyed / unknown stream");
    run_until([s] { return s->q.empty(); }, "a stream being destroyed");
    g_streams.erase(s);
    delete s;
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Event *e = new Event;
    g_events.insert(e);
    *event = reinterpret_cast<hipEvent_t>(e);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t event)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Event *e = reinterpret_cast<Event *>(event);
    if (!g_events.count(e)) die("hipEventDestroy on a destroyed / unknown event");
    // a stream may still hold a wait on / a record of it: let those run first, as the real runtime keeps the event alive
    drain_all();
    g_events.erase(e);
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Event *e = reinterpret_cast<Event *>(event);
    if (!g_events.count(e) || !g_streams.count(sp(stream))) die("hipEventRecord: unknown event or stream");
    Op o;
    o.kind = Op::RECORD;
    o.ev = e;
    o.gen = ++e->recorded;
    sp(stream)->q.push_back(std::move(o));
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Event *e = reinterpret_cast<Event *>(event);
    if (!g_events.count(e) || !g_streams.count(sp(stream))) die("hipStreamWaitEvent: unknown event or stream");
    if (e->recorded == 0) return hipSuccess;        // never recorded: no-op, as in the real runtime
    Op o;
    o.kind = Op::WAIT;
    o.ev = e;
    o.gen = e->recorded;                            // the record most recently ENQUEUED at this call
    sp(stream)->q.push_back(std::move(o));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t event)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    Event *e = reinterpret_cast<Event *>(event);
    if (!g_events.count(e)) die("hipEventSynchronize: unknown event");
    const unsigned long long target = e->recorded;
    run_until([e, target] { return e->completed >= target; }, "an event");
    return hipSuccess;
}

hipError_t hipMalloc(void **ptr, size_t bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    if (g_fail_malloc_after == 0) {
        g_fail_malloc_after = -1;
        *ptr = nullptr;
        return hipErrorOutOfMemory;
    }
    if (g_fail_malloc_after > 0) --g_fail_malloc_after;
    void *p = std::malloc(bytes ? bytes : 1);
    if (!p) return hipErrorOutOfMemory;
    std::memset(p, 0xA5, bytes);                     // uninitialised device memory is garbage, not zeros
    g_dev[p] = bytes;
    *ptr = p;
    return hipSuccess;
}
hipError_t hipFree(void *ptr)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!ptr) return hipSuccess;
    if (!g_dev.count(ptr)) die("hipFree of a pointer hipMalloc did not return (or freed twice)");
    drain_all();                                     // hipFree waits for the device
    g_dev.erase(ptr);
    std::free(ptr);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **ptr, size_t bytes, unsigned)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    void *p = std::malloc(bytes ? bytes : 1);
    if (!p) return hipErrorOutOfMemory;
    std::memset(p, 0x5A, bytes);
    g_pinned[p] = bytes;
    *ptr = p;
    return hipSuccess;
}
hipError_t hipHostFree(void *ptr)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!ptr) return hipSuccess;
    if (!g_pinned.count(ptr)) die("hipHostFree of a pointer hipHostMalloc did not return (or freed twice)");
    drain_all();
    g_pinned.erase(ptr);
    std::free(ptr);
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned)
{
    *dev = host;
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    if (!g_streams.count(sp(stream))) die("hipMemcpyAsync: unknown stream");
    if (bytes == 0) return hipSuccess;
    if (kind == hipMemcpyHostToDevice && !inside(g_dev, dst, bytes) && !inside(g_pinned, dst, bytes)) die("H2D copy: destination is not device memory of that size");
    if (kind == hipMemcpyDeviceToHost && !inside(g_dev, src, bytes) && !inside(g_pinned, src, bytes)) die("D2H copy: source is not device memory of that size");
    // (device addresses of mapped host memory count as device memory: the small-call staging of vdyn_capi.hip uses them)
    if ((kind == hipMemcpyDeviceToDevice || kind == hipMemcpyDeviceToDeviceNoCU) &&
        ((!inside(g_dev, src, bytes) && !inside(g_pinned, src, bytes)) || (!inside(g_dev, dst, bytes) && !inside(g_pinned, dst, bytes))))
        die("D2D copy: source or destination is not device memory of that size");
    if (kind == hipMemcpyHostToDevice && !inside(g_pinned, src, bytes)) {
        // pageable source: the runtime has consumed it when the call returns (the caller may free it at once)
        std::vector<char> tmp(static_cast<const char *>(src), static_cast<const char *>(src) + bytes);
        enqueue(stream, [dst, tmp = std::move(tmp)] { std::memcpy(dst, tmp.data(), tmp.size()); });
        return hipSuccess;
    }
    if (kind == hipMemcpyDeviceToHost && !inside(g_pinned, dst, bytes)) {
        // pageable destination: the call returns when the data has arrived
        Stream *s = sp(stream);
        run_until([s] { return s->q.empty(); }, "a stream (copy into pageable memory)");
        std::memcpy(dst, src, bytes);
        return hipSuccess;
    }
    enqueue(stream, [dst, src, bytes] { std::memmove(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!inside(g_dev, dst, bytes)) die("hipMemsetAsync: not device memory of that size");
    enqueue(stream, [dst, value, bytes] { std::memset(dst, value, bytes); });
    return hipSuccess;
}
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *handle, void *dev_ptr)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g_dev.count(dev_ptr)) return hipErrorInvalidValue;
    std::memset(handle, 0, sizeof(*handle));
    std::memcpy(handle->reserved, &dev_ptr, sizeof(dev_ptr));
    return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void **dev_ptr, hipIpcMemHandle_t handle, unsigned)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    void *p = nullptr;
    std::memcpy(&p, handle.reserved, sizeof(p));
    if (!g_dev.count(p)) return hipErrorInvalidValue;
    *dev_ptr = p;                                    // one process: the "peer" mapping is the buffer itself
    return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void *dev_ptr)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return g_dev.count(dev_ptr) ? hipSuccess : hipErrorInvalidValue;
}

// test hooks (ctypes)
void hipstub_fail_malloc_after(int n)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    g_fail_malloc_after = n;
}
long long hipstub_live_allocations(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return (long long)(g_dev.size() + g_pinned.size());
}
long long hipstub_live_streams_and_events(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return (long long)(g_streams.size() - (g_streams.count(&g_null_stream) ? 1 : 0) + g_events.size());
}

}  // extern "C"

// ------------------------------------------------------------------- level 2: the REAL launchers' host halves ----
// -DHIPSTUB_REAL_LAUNCHERS (tests/hipstub/build_launchers.sh): this file, csrc/vdyn_capi.hip and BOTH kernel translation
// units are compiled by hipcc --cuda-host-only (the real HIP headers, no device code) under the sanitizers, so the
// launchers' host code runs too -- make_dev_params, the long-double tire fits and their cache, the fleet / candidate
// table builders, chunk and LDS sizing, the choice of kernel instance.  A launch then arrives here as hipLaunchKernel:
// its geometry is checked (block 1..1024 threads, grid >= 1, dynamic LDS within the function's limit and the CU's
// 160 KB) and it is counted per kernel name; nothing is computed, so the driver skips its content checks.
#ifdef HIPSTUB_REAL_LAUNCHERS
namespace {
// (filled by the translation units' module constructors, which may run before this file's statics are constructed)
std::map<const void *, std::string> &kernel_names()
{
    static auto *m = new std::map<const void *, std::string>;
    return *m;
}
#define g_kernel_names kernel_names()
std::map<const void *, int> g_max_dyn_lds;
std::map<std::string, long long> g_launches;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
std::vector<CallCfg> g_cfg;
}  // namespace
extern "C" {
void **__hipRegisterFatBinary(const void *)
{
    static void *handle = nullptr;
    return &handle;
}
void __hipRegisterFunction(void **, const void *host_fn, char *, const char *device_name, unsigned, void *, void *, void *,
                           void *, int *)
{
    g_kernel_names[host_fn] = device_name;
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_cfg.push_back({grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g_cfg.empty()) die("__hipPopCallConfiguration without a push");
    *grid = g_cfg.back().grid;
    *block = g_cfg.back().block;
    *shmem = g_cfg.back().shmem;
    *stream = g_cfg.back().stream;
    g_cfg.pop_back();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *fn, hipFuncAttribute attr, int value)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g_kernel_names.count(fn)) die("hipFuncSetAttribute on a function that was never registered");
    if (attr == hipFuncAttributeMaxDynamicSharedMemorySize) {
        if (value < 0 || value > 160 * 1024) die("hipFuncSetAttribute: dynamic LDS beyond the CU's 160 KB");
        g_max_dyn_lds[fn] = value;
    }
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    auto it = g_kernel_names.find(fn);
    if (it == g_kernel_names.end()) die("hipLaunchKernel: a function that was never registered");
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    if (threads == 0 || threads > 1024) { std::fprintf(stderr, "hipstub: %s: block of %llu threads\n", it->second.c_str(), threads); std::abort(); }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || grid.x > 0x7fffffffu) { std::fprintf(stderr, "hipstub: %s: grid %u x %u x %u\n", it->second.c_str(), grid.x, grid.y, grid.z); std::abort(); }
    const size_t limit = g_max_dyn_lds.count(fn) ? (size_t)g_max_dyn_lds[fn] : (size_t)64 * 1024;
    if (shmem > limit) { std::fprintf(stderr, "hipstub: %s: %zu B of dynamic LDS, limit %zu\n", it->second.c_str(), shmem, limit); std::abort(); }
    if (!args) die("hipLaunchKernel: null argument array");
    if (!g_streams.count(sp(stream))) die("hipLaunchKernel: unknown stream");
    ++g_launches[it->second];
    enqueue(stream, [] {});
    return hipSuccess;
}
long long hipstub_launch_count(const char *substring)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    long long n = 0;
    for (auto &kv : g_launches)
        if (!substring || kv.first.find(substring) != std::string::npos) n += kv.second;
    return n;
}
long long hipstub_distinct_kernels(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return (long long)g_launches.size();
}
}  // extern "C"
#else
// --------------------------------------------------------------------------------------------- the launchers -----
// Stand-ins for csrc/vdyn_kernels.hip's launch_* (the interface vdyn_capi.hip calls).  The rollout family is an exact
// toy integrator with the REAL kernels' composition property -- rollout(a) then rollout(b) == rollout(a + b), trajectory
// rows included -- so that the pipelined upload can be checked against a closed form:
//     per step t, rollout r:  u = sum_j (j + 1) ctrl(t, j, r);   state[i][r] += (i + 1) u   (rows 12..21: += u)
// (exact in fp32 / fp64 for the small integers the driver feeds it).  Everything else answers "sum of all inputs + i".
namespace vdyn {

namespace {
template <typename T>
hipError_t rollout_like(const RolloutArgs<T> &a, hipStream_t st, double extra_per_rollout_from_ids)
{
    (void)extra_per_rollout_from_ids;
    if (a.n <= 0 || !a.state0 || !a.terminal || (a.H > 0 && !a.ctrl)) return hipErrorInvalidValue;
    double mu = 0;
    if (a.mu4) mu = a.mu4[0] + a.mu4[1] + a.mu4[2] + a.mu4[3];     // host pointer: read at the launch, like make_dev_params
    RolloutArgs<T> b = a;
    b.mu4 = nullptr;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    init();
    if (!g_streams.count(sp(st))) die("launch on an unknown stream");
    enqueue(st, [b, mu] {
        const size_t n = (size_t)b.n;
        const int rows = b.state_rows;
        std::vector<double> s((size_t)rows * n);
        for (size_t i = 0; i < s.size(); ++i) s[i] = (double)b.state0[i];
        if (b.layout == VDYN_CTRL_SHARED) read_all(b.path_id, n);
        if (b.vehicle_id) read_all(b.vehicle_id, n);
        if (b.fleet_tab) read_all(b.fleet_tab, (size_t)fleet_table_len<T>(b.V));
        size_t row = 0;
        std::vector<double> u(n, 0.0);
        for (int t = 0; t < b.H; ++t) {
            for (size_t r = 0; r < n; ++r) {
                double acc = 0;
                for (int j = 0; j < b.k; ++j) {
                    double c;
                    if (b.layout == VDYN_CTRL_PER_ROLLOUT) c = (double)b.ctrl[((size_t)t * b.k + j) * n + r];
                    else {
                        const int p = std::min(std::max(b.path_id[r], 0), b.P - 1);
                        c = (double)b.ctrl[((size_t)p * b.H + t) * b.k + j];
                    }
                    acc += (j + 1) * c;
                }
                u[r] = acc + mu + (b.vehicle_id ? (double)b.vehicle_id[r] : 0.0);
                for (int i = 0; i < rows; ++i) s[(size_t)i * n + r] += (i < 12 ? (i + 1) : 1) * u[r];
            }
            if (b.traj && b.traj_stride > 0 && (t + 1) % b.traj_stride == 0) {
                for (size_t i = 0; i < 12 * n; ++i) b.traj[row * 12 * n + i] = (T)s[i];
                ++row;
            }
        }
        for (size_t i = 0; i < s.size(); ++i) b.terminal[i] = (T)s[i];
        if (b.state_dot && b.H > 0)
            for (size_t i = 0; i < 10 * n; ++i) b.state_dot[i] = (T)((double)(i / n + 1) * u[i % n]);
        if (b.outputs && b.H > 0)
            for (size_t i = 0; i < 18 * n; ++i) b.outputs[i] = (T)((double)(i / n + 101) * u[i % n]);
    });
    return hipSuccess;
}
}  // namespace

template <typename T>
hipError_t launch_rollout(const VdynParams &, const RolloutArgs<T> &a, hipStream_t st) { return rollout_like<T>(a, st, 0); }
template <typename T>
hipError_t launch_rollout_fleet(const RolloutArgs<T> &a, bool, hipStream_t st)
{
    if (!a.fleet_tab || !a.vehicle_id || a.V < 1) return hipErrorInvalidValue;
    return rollout_like<T>(a, st, 1);
}
template <typename T>
int fleet_table_len(int V) { return 37 * V + 3; }
template <typename T>
void build_fleet_table(const VdynParams *classes, int V, const double *mu4, T *out, bool *all_small)
{
    double s = mu4 ? mu4[0] + mu4[1] + mu4[2] + mu4[3] : 0;
    for (int v = 0; v < V; ++v) s += classes[v].m;
    write_all(out, (size_t)fleet_table_len<T>(V), 0.0);
    g_sink += (unsigned long long)s;
    *all_small = true;
}

template <typename T>
hipError_t launch_rollout_spiral(const VdynParams &, int64_t n, int H, const T *state0, const T *spiral, double, double,
                                 double torque, double, const double *mu4, T *terminal, T *traj, int traj_stride,
                                 hipStream_t st)
{
    if (n <= 0 || !state0 || !spiral || !terminal) return hipErrorInvalidValue;
    const double mu = mu4 ? mu4[0] + mu4[1] + mu4[2] + mu4[3] : 0;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    enqueue(st, [=] {
        const double base = read_all(state0, 12 * (size_t)n) + read_all(spiral, 3 * (size_t)n) + mu + torque;
        write_all(terminal, 12 * (size_t)n, base);
        if (traj && traj_stride > 0) write_all(traj, (size_t)(H / traj_stride) * 12 * (size_t)n, base + 1);
    });
    return hipSuccess;
}

template <typename T>
hipError_t launch_nonfinite_lanes(int rows, int64_t n, const T *x, int *status, unsigned long long *count, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    enqueue(st, [=] {
        const double base = read_all(x, (size_t)rows * (size_t)n);
        write_all(status, (size_t)n, 0.0);
        if (count) *count += (unsigned long long)((long long)base & 0xffff);
    });
    return hipSuccess;
}
bool tire_fit_coefficients(double C, float *coef)
{
    for (int i = 0; i < 9; ++i) coef[i] = (float)(C + i);
    return C >= 0 && C < 2.9;
}
bool tire_fit_coefficients64(double C, double *coef)
{
    for (int i = 0; i < 17; ++i) coef[i] = C + i;
    return C >= 0 && C < 2.9;
}
template <typename T>
hipError_t launch_fastmath_eval(int, int64_t n, const T *x, double c, T *out0, T *out1, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    enqueue(st, [=] {
        const double base = read_all(x, (size_t)n) + c;
        write_all(out0, (size_t)n, base);
        write_all(out1, (size_t)n, base + 1);
    });
    return hipSuccess;
}
template <typename T>
hipError_t launch_planar_model(const VdynParams &, int64_t n, const T *state, const T *ctrl12, const T *acc_prev,
                               T *state_dot, T *aux, T *outputs, T *acc, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    enqueue(st, [=] {
        const size_t N = (size_t)n;
        const double base = read_all(state, 10 * N) + read_all(ctrl12, 12 * N) + read_all(acc_prev, 2 * N);
        write_all(state_dot, 10 * N, base);
        write_all(aux, 4 * N, base + 1);
        write_all(outputs, 18 * N, base + 2);
        write_all(acc, 2 * N, base + 3);
    });
    return hipSuccess;
}
template <typename T>
size_t mpc_scratch_bytes(int E, int C, int H) { return sizeof(T) * ((size_t)H * C * 4 + (size_t)E * 16) + 64; }
template <typename T>
hipError_t launch_mpc_argmin(const VdynParams &, int E, int C, int H, const T *ego, const T *cand, const T *goal, double,
                             double w_delta, T *best_cost, int *best_idx, T *cost_all, void *scratch, hipStream_t st)
{
    if (!scratch) return hipErrorInvalidValue;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    enqueue(st, [=] {
        std::memset(scratch, 0x11, mpc_scratch_bytes<T>(E, C, H));      // the kernel owns all of it
        const double base = read_all(ego, 12 * (size_t)E) + read_all(cand, 2 * (size_t)H * C) + read_all(goal, 2 * (size_t)E) + w_delta;
        write_all(best_cost, (size_t)E, base);
        write_all(best_idx, (size_t)E, 0.0);
        write_all(cost_all, (size_t)E * C, base + 1);
    });
    return hipSuccess;
}
template <typename T>
size_t closed_loop_aux_bytes(int P, int Wmax, bool update_only) { return update_only ? 0 : sizeof(T) * ((size_t)P * Wmax + 40 * (size_t)P); }
namespace {
template <typename T>
void closed_loop_body(const ClosedLoopArgs<T> &a, bool update_only)
{
    const size_t n = (size_t)a.n;
    if (a.aux) std::memset(a.aux, 0x22, closed_loop_aux_bytes<T>(a.P, a.Wmax, update_only));
    if (update_only) {
        const double base = read_all(a.state0, 12 * n) + read_all(a.cstate0, 6 * n) + read_all(a.wp, (size_t)a.P * a.Wmax * 2) +
                            read_all(a.wcount, (size_t)a.P) + read_all(a.path_id, n);
        write_all(a.cstate, 6 * n, base + 1);
        write_all(a.ctrl_out, 3 * n, base + 2);
        return;
    }
    // a toy closed loop with the REAL kernel's chaining property -- launches of H1 then H2 sub-steps with the phase carried
    // (phase, phase + H1) and the state / controller state fed back ARE one launch of H1 + H2, log rows included:
    //   per sub-step t, vehicle r:  u = (phase + t) + sum(wp[path r]);  state[i] += (i + 1) u;
    //   controller state[j] += (j + 1) u on the sub-steps the controllers fire, (phase + t) % ctrl_every == 0;
    //   log[t][c] = state[c % 12] + c,  datalog[t][c] = state[c % 12] + 100 + c   (after the sub-step), except the LAST column
    //   of each: the u of the launch's most recent controller update, -1 before its first -- like the real log's target
    //   index, NOT carried from launch to launch, so only cuts at multiples of ctrl_every reproduce the single launch
    read_all(a.wcount, (size_t)a.P);
    std::vector<double> wsum((size_t)a.P);
    for (int p = 0; p < a.P; ++p) wsum[p] = read_all(a.wp + (size_t)p * a.Wmax * 2, (size_t)a.Wmax * 2);
    std::vector<double> s(12 * n), cs(6 * n);
    for (size_t i = 0; i < 12 * n; ++i) s[i] = (double)a.state0[i];
    for (size_t i = 0; i < 6 * n; ++i) cs[i] = (double)a.cstate0[i];
    std::vector<double> last(n, -1.0);
    for (int t = 0; t < a.H; ++t) {
        const bool fire = (a.phase + t) % a.ctrl_every == 0;
        for (size_t r = 0; r < n; ++r) {
            const int p = std::min(std::max(a.path_id[r], 0), a.P - 1);
            const double u = (double)(a.phase + t) + wsum[p];
            for (int i = 0; i < 12; ++i) s[(size_t)i * n + r] += (i + 1) * u;
            if (fire) {
                for (int j = 0; j < 6; ++j) cs[(size_t)j * n + r] += (j + 1) * u;
                last[r] = u;
            }
        }
        if (a.log)
            for (int c = 0; c < 16; ++c)
                for (size_t r = 0; r < n; ++r) a.log[((size_t)t * 16 + c) * n + r] = (T)(c == 15 ? last[r] : s[(size_t)(c % 12) * n + r] + c);
        if (a.datalog)
            for (int c = 0; c < 45; ++c)
                for (size_t r = 0; r < n; ++r)
                    a.datalog[((size_t)t * 45 + c) * n + r] = (T)(c == 44 ? last[r] : s[(size_t)(c % 12) * n + r] + 100 + c);
    }
    for (size_t i = 0; i < 12 * n; ++i) a.terminal[i] = (T)s[i];
    for (size_t i = 0; i < 6 * n; ++i) a.cstate[i] = (T)cs[i];
}
}  // namespace
template <typename T>
hipError_t launch_closed_loop(const VdynParams &, const VdynCtrlGains &, const ClosedLoopArgs<T> &a, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const ClosedLoopArgs<T> b = a;
    enqueue(st, [b] { closed_loop_body<T>(b, false); });
    return hipSuccess;
}
template <typename T>
hipError_t launch_controller_update(const VdynCtrlGains &, const ClosedLoopArgs<T> &a, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const ClosedLoopArgs<T> b = a;
    enqueue(st, [b] { closed_loop_body<T>(b, true); });
    return hipSuccess;
}
template <typename T>
hipError_t launch_select_best_path(const SelectArgs<T> &a, hipStream_t st)
{
    double host = 0;
    for (int i = 0; i < a.nc; ++i) host += a.offsets[i] + a.radii[i];                  // host arrays: read at the launch
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    SelectArgs<T> b = a;
    b.offsets = b.radii = nullptr;
    enqueue(st, [b, host] {
        double base = host + b.weight;
        for (int e = 0; e < b.E; ++e)
            for (int p = 0; p < b.P; ++p)
                for (int j = 0; j < b.L; ++j) {
                    const int64_t o = e * b.ego_stride + p * b.path_stride + j * b.point_stride;
                    base += (double)b.x[o] + (double)b.y[o] + (double)b.yaw[o];
                }
        for (int e = 0; e < (b.obst_ego_stride ? b.E : 1); ++e) base += read_all(b.obst + e * b.obst_ego_stride, 2 * (size_t)b.M);
        base += read_all(b.goal, 2 * (size_t)b.E) + read_all(b.collision_in, (size_t)b.E * b.P) + read_all(b.validity, (size_t)b.E * b.P);
        write_all(b.collision_free, (size_t)b.E * b.P, 0.0);
        write_all(b.best_idx, (size_t)b.E, 0.0);
        write_all(b.best_score, (size_t)b.E, base);
    });
    return hipSuccess;
}
template <typename T>
hipError_t launch_plan_lattice(const LatticeArgs<T> &a, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const LatticeArgs<T> b = a;
    enqueue(st, [b] {
        const size_t E = (size_t)b.E, EP = E * (size_t)b.P;
        const double base = read_all(b.px, (size_t)b.nwp) + read_all(b.py, (size_t)b.nwp) + read_all(b.ego, 3 * E) +
                            read_all(b.params_in, 3 * EP) + b.goal_v + b.lookahead + b.path_offset;
        write_all(b.closest_idx, E, 0.0);
        write_all(b.goal_idx, E, 1.0);
        write_all(b.closest_len, E, base);
        write_all(b.goal_set, 4 * EP, base + 1);
        write_all(b.params, 3 * EP, base + 2);
        write_all(b.paths, 3 * 49 * EP, base + 3);
        write_all(b.validity, EP, 1.0);
        write_all(b.cost, EP, base + 4);
    });
    return hipSuccess;
}
template <typename T>
hipError_t launch_interpolate_waypoints(int E, int P, int L, const T *paths, const int *best_idx, double res, int Wmax,
                                        T *wp_out, int *wcount, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    enqueue(st, [=] {
        const double base = read_all(paths, (size_t)E * P * 3 * L) + res;
        // IN / OUT: an ego without a path (best_idx < 0) keeps its table and count
        for (int e = 0; e < E; ++e) {
            if (best_idx[e] < 0) {
                read_all(wp_out + (size_t)e * Wmax * 2, (size_t)Wmax * 2);
                g_sink += (unsigned long long)wcount[e];
                continue;
            }
            write_all(wp_out + (size_t)e * Wmax * 2, (size_t)Wmax * 2, base + e);
            wcount[e] = Wmax;
        }
    });
    return hipSuccess;
}

#define HIPSTUB_INSTANTIATE(T)                                                                                              \
    template hipError_t launch_rollout<T>(const VdynParams &, const RolloutArgs<T> &, hipStream_t);                         \
    template hipError_t launch_rollout_fleet<T>(const RolloutArgs<T> &, bool, hipStream_t);                                 \
    template int fleet_table_len<T>(int);                                                                                   \
    template void build_fleet_table<T>(const VdynParams *, int, const double *, T *, bool *);                               \
    template hipError_t launch_rollout_spiral<T>(const VdynParams &, int64_t, int, const T *, const T *, double, double,    \
                                                 double, double, const double *, T *, T *, int, hipStream_t);               \
    template hipError_t launch_nonfinite_lanes<T>(int, int64_t, const T *, int *, unsigned long long *, hipStream_t);       \
    template hipError_t launch_fastmath_eval<T>(int, int64_t, const T *, double, T *, T *, hipStream_t);                    \
    template hipError_t launch_planar_model<T>(const VdynParams &, int64_t, const T *, const T *, const T *, T *, T *, T *, \
                                               T *, hipStream_t);                                                           \
    template size_t mpc_scratch_bytes<T>(int, int, int);                                                                    \
    template hipError_t launch_mpc_argmin<T>(const VdynParams &, int, int, int, const T *, const T *, const T *, double,    \
                                             double, T *, int *, T *, void *, hipStream_t);                                 \
    template size_t closed_loop_aux_bytes<T>(int, int, bool);                                                               \
    template hipError_t launch_closed_loop<T>(const VdynParams &, const VdynCtrlGains &, const ClosedLoopArgs<T> &,         \
                                              hipStream_t);                                                                 \
    template hipError_t launch_controller_update<T>(const VdynCtrlGains &, const ClosedLoopArgs<T> &, hipStream_t);         \
    template hipError_t launch_select_best_path<T>(const SelectArgs<T> &, hipStream_t);                                     \
    template hipError_t launch_plan_lattice<T>(const LatticeArgs<T> &, hipStream_t);                                        \
    template hipError_t launch_interpolate_waypoints<T>(int, int, int, const T *, const int *, double, int, T *, int *,     \
                                                        hipStream_t);
HIPSTUB_INSTANTIATE(float)
HIPSTUB_INSTANTIATE(double)

}  // namespace vdyn
#endif  // HIPSTUB_REAL_LAUNCHERS
