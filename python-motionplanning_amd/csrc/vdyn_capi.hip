// vdyn_capi.hip -- the C ABI of include/vdyn.h: handle, argument checks, the
// `_dev` entry points (enqueue on the caller's stream) and the `_host` entry
// points (stage through pinned + device scratch owned by the handle).
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include "vdyn_internal.hpp"

// One slice of a staging copy.  The destination is written once and next read by a DMA engine (or, on the way out, by
// the caller much later): non-temporal stores keep it out of the caches and spare the read-for-ownership of every
// destination line, which an ordinary memcpy of a 512 KB slice pays (glibc switches to streaming stores only for much
// larger copies).  VDYN_COPY_NT=0 selects plain memcpy (A/B on the GPU box: tools/host_abi_threads.py).
static void stream_copy(char *dst, const char *src, size_t n)
{
#if defined(__x86_64__)
    static const bool nt = [] { const char *e = std::getenv("VDYN_COPY_NT"); return !(e && e[0] == '0'); }();
    if (nt && n >= 4096) {
        const size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
        if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
        const size_t blocks = n / 64;
        for (size_t i = 0; i < blocks; ++i) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src)),
                          b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 16)),
                          c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 32)),
                          d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 48));
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst), a);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 16), b);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 32), c);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 48), d);
            src += 64;
            dst += 64;
        }
        _mm_sfence();
        n -= blocks * 64;
    }
#endif
    if (n) std::memcpy(dst, src, n);
}

// Host copies of the _host entry points (caller's pageable memory <-> the handle's pinned staging buffer).  One thread
// moves 8-12 GB/s; the PCIe link behind it moves several times that, so a large staging copy -- 105 MB of per-rollout
// controls at BASELINE configs[2] -- is spread over a few worker threads that do nothing else (no HIP call is ever made
// from them).  The calling thread takes slices too; copies under kParallelBytes stay a plain memcpy.
class CopyPool {
public:
    static constexpr size_t kParallelBytes = 1u << 20, kSliceBytes = 512u << 10;

    // CPUs this process may run on: the affinity mask, capped by the cgroup quota (a one-GPU share of a big host)
    static int usable_cpus()
    {
        int n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string quota;
        long long period = 0;
        if (f >> quota >> period && quota != "max" && period > 0) {
            const long long q = std::atoll(quota.c_str());
            if (q > 0) n = std::min<long long>(n, std::max<long long>(1, q / period));
        }
        return n;
    }
    static int default_threads()
    {
        if (const char *e = std::getenv("VDYN_COPY_THREADS")) {
            const int v = std::atoi(e);
            if (v >= 1 && v <= 64) return v;
        }
        return std::min(8, std::max(1, usable_cpus() / 2));
    }

    explicit CopyPool(int threads) : nthreads_(std::max(1, threads))
    {
        for (int i = 1; i < nthreads_; ++i) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int threads() const { return nthreads_; }

    // dst <- src, blocking.  One job at a time (the handle is single-threaded by contract).
    void copy(void *dst, const void *src, size_t bytes)
    {
        if (bytes == 0) return;
        if (bytes < kParallelBytes || nthreads_ == 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        const size_t slices = (bytes + kSliceBytes - 1) / kSliceBytes;
        {
            // the job's fields change only while no worker is inside work() (a worker woken late for a job that is
            // already complete still reads them once): workers enter and leave work() under this mutex's count
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [this] { return active_ == 0; });
            dst_ = static_cast<char *>(dst);
            src_ = static_cast<const char *>(src);
            bytes_ = bytes;
            slices_ = slices;
            next_.store(0, std::memory_order_relaxed);
            left_ = slices;
            ++generation_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return left_ == 0; });
    }

private:
    void work()
    {
        size_t mine = 0;
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= slices_) break;
            const size_t off = i * kSliceBytes;
            stream_copy(dst_ + off, src_ + off, std::min(kSliceBytes, bytes_ - off));
            ++mine;
        }
        if (mine) {
            std::lock_guard<std::mutex> lk(m_);
            left_ -= mine;
            if (left_ == 0) done_.notify_all();
        }
    }
    void run()
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                ++active_;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) done_.notify_all();
            }
        }
    }
    const int nthreads_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_ = false;
    unsigned long long generation_ = 0;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0, slices_ = 0, left_ = 0;
    int active_ = 0;            // workers inside work()
    std::atomic<size_t> next_{0};
};

struct VdynHandle {
    int device = 0;
    VdynParams p{};
    std::string err;
    hipStream_t stream = nullptr;   // used by the _host entry points
    void *d_scratch = nullptr;
    size_t d_bytes = 0;
    void *h_pinned = nullptr;
    size_t h_bytes = 0;
    int lanes_per_rollout = 1;      // VDYN_OPT_LANES_PER_ROLLOUT
    int state_rows = 12;            // VDYN_OPT_STATE_ROWS
    void *d_fleet = nullptr;        // device copy of the per-class constants of the last fleet call
    size_t d_fleet_bytes = 0;
    void *d_aux = nullptr;          // controllers' auxiliary waypoint tables (segment lengths, bounding circles)
    size_t d_aux_bytes = 0;
    void *d_count = nullptr;        // one counter (vdyn_nonfinite_lanes_*)
    void *d_cand = nullptr;         // MPC candidate table with the steering angles' (sin, cos) per entry
    size_t d_cand_bytes = 0;
    // peer exchange (vdyn_xchg_*): copies run on these streams, beside the compute stream -- one per destination
    // (round robin beyond kCopyStreams), so that the copies to different peers can use different SDMA engines and
    // xGMI links at the same time instead of queueing behind each other
    static constexpr int kCopyStreams = 8;
    hipStream_t copy_stream[kCopyStreams] = {};
    hipEvent_t ev_ready = nullptr, ev_done[kCopyStreams] = {};
    int push_streams = 0;           // copy streams in use (0: no push issued since the last vdyn_xchg_wait)
    void *h_mapped = nullptr;       // small host-coherent buffer the GPU reads / writes in place
    void *d_mapped = nullptr;       // its device address
    // _host entry points: worker threads of the staging copies (created with the first large copy), the upload stream
    // of the pipelined rollout and its events (rollout_host_pipelined)
    CopyPool *pool = nullptr;
    hipStream_t stage_stream = nullptr, down_stream = nullptr;
    hipEvent_t ev_h2d[2] = {}, ev_kernel[2] = {}, ev_d2h[2] = {};

    void host_copy(void *dst, const void *src, size_t bytes)
    {
        if (bytes == 0) return;         // (an optional buffer the caller left out: null pointer, zero bytes)
        if (bytes >= CopyPool::kParallelBytes && !pool) pool = new (std::nothrow) CopyPool(CopyPool::default_threads());
        if (pool) pool->copy(dst, src, bytes);
        else std::memcpy(dst, src, bytes);
    }

    int fail(int code, const std::string &msg)
    {
        err = msg;
        return code;
    }
    int fail_hip(const char *what, hipError_t e)
    {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return VDYN_ERR_HIP;
    }
};

static std::string g_create_error;

#define VDYN_HIP(h, expr)                                        \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return (h)->fail_hip(#expr, e_);   \
    } while (0)

static bool params_ok(const VdynParams *p, std::string *why)
{
    const double v[] = {p->m, p->a, p->b, p->Izz, p->Jw, p->hg, p->T, p->wL, p->wR, p->rw, p->g};
    for (double x : v)
        if (!std::isfinite(x)) { *why = "VdynParams: non-finite field"; return false; }
    if (!(p->m > 0) || !(p->Izz > 0) || !(p->Jw > 0)) { *why = "VdynParams: m, Izz, Jw must be > 0"; return false; }
    if (!(p->a + p->b > 0) || !(p->wL + p->wR > 0)) { *why = "VdynParams: a+b and wL+wR must be > 0"; return false; }
    for (int i = 0; i < 4; ++i)
        if (!std::isfinite(p->B[i]) || !std::isfinite(p->C[i])) { *why = "VdynParams: non-finite Pacejka B/C"; return false; }
    return true;
}

extern "C" {

int vdyn_abi_version(void) { return VDYN_ABI_VERSION; }

#ifndef VDYN_BUILD_ID
#define VDYN_BUILD_ID "unknown"
#endif
const char *vdyn_build_id(void) { return VDYN_BUILD_ID; }

int vdyn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void vdyn_params_default(VdynParams *p)
{
    if (!p) return;
    // VehicleParameters.__init__ defaults, vehicle_model.py:18-61
    const double mf = 987.89, mr = 869.93, mus = 50, L = 2.906, ab_ratio = 0.85, T = 1.536;
    const double hg = 0.55419, Jw = 1, kf = 26290, rr = 0.329;
    p->m = mf + mr;
    p->b = L / (1 + ab_ratio);
    p->a = L - p->b;
    p->Izz = 0.5 * p->m * p->a * p->b;
    p->Jw = Jw;
    p->hg = hg;
    p->T = T;
    p->wL = T / 2;
    p->wR = T / 2;
    p->rw = rr - (mf / 2 + mus) / kf;
    p->g = 9.81;
    for (int i = 0; i < 4; ++i) {
        p->B[i] = 20.6357;
        p->C[i] = 1.5047;
    }
}

void vdyn_ctrl_gains_default(VdynCtrlGains *g)
{
    if (!g) return;
    g->k = 100.0;                                     // drive.py:71
    g->k_soft = 1.0;                                  // drive.py:72
    g->max_steer = 30.0 * 3.141592653589793 / 180.0;  // drive.py:56 np.deg2rad(30)
    g->lookahead = 5.0;                               // stanley_controller.py:46
    g->deadband = 0.01;                               // stanley_controller.py:47
    g->kp = 1000.0;                                   // drive.py:83
    g->ki = 100.0;                                    // drive.py:84
    g->kd = 0.0;                                      // drive.py:85
    g->filter_gain = 1e-5 / (2 * 0.001);              // drive.py:137
}

int vdyn_create(const VdynParams *p, int device, VdynHandle **out)
{
    if (!p || !out) { g_create_error = "vdyn_create: null argument"; return VDYN_ERR_ARG; }
    *out = nullptr;
    std::string why;
    if (!params_ok(p, &why)) { g_create_error = why; return VDYN_ERR_ARG; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("vdyn_create: no HIP device (") + hipGetErrorString(e) + ")";
        return VDYN_ERR_NODEV;
    }
    if (device < 0 || device >= ndev) { g_create_error = "vdyn_create: device ordinal out of range"; return VDYN_ERR_ARG; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return VDYN_ERR_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("vdyn_create: device is ") + prop.gcnArchName +
                         ", this library carries gfx950 code only";
        return VDYN_ERR_NODEV;
    }
    VdynHandle *h = new (std::nothrow) VdynHandle;
    if (!h) { g_create_error = "vdyn_create: out of host memory"; return VDYN_ERR_OOM; }
    h->device = device;
    h->p = *p;
    if ((e = hipSetDevice(device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("vdyn_create: ") + hipGetErrorString(e);
        delete h;
        return VDYN_ERR_HIP;
    }
    *out = h;
    return VDYN_OK;
}

int vdyn_set_params(VdynHandle *h, const VdynParams *p)
{
    if (!h || !p) return VDYN_ERR_ARG;
    std::string why;
    if (!params_ok(p, &why)) return h->fail(VDYN_ERR_ARG, why);
    h->p = *p;
    return VDYN_OK;
}

void vdyn_destroy(VdynHandle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    if (h->h_mapped) (void)hipHostFree(h->h_mapped);
    if (h->d_fleet) (void)hipFree(h->d_fleet);
    if (h->d_aux) (void)hipFree(h->d_aux);
    if (h->d_cand) (void)hipFree(h->d_cand);
    if (h->d_count) (void)hipFree(h->d_count);
    for (int i = 0; i < VdynHandle::kCopyStreams; ++i) {
        if (h->copy_stream[i]) { (void)hipStreamSynchronize(h->copy_stream[i]); (void)hipStreamDestroy(h->copy_stream[i]); }
        if (h->ev_done[i]) (void)hipEventDestroy(h->ev_done[i]);
    }
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    if (h->stage_stream) { (void)hipStreamSynchronize(h->stage_stream); (void)hipStreamDestroy(h->stage_stream); }
    if (h->down_stream) { (void)hipStreamSynchronize(h->down_stream); (void)hipStreamDestroy(h->down_stream); }
    for (int i = 0; i < 2; ++i) {
        if (h->ev_h2d[i]) (void)hipEventDestroy(h->ev_h2d[i]);
        if (h->ev_kernel[i]) (void)hipEventDestroy(h->ev_kernel[i]);
        if (h->ev_d2h[i]) (void)hipEventDestroy(h->ev_d2h[i]);
    }
    delete h->pool;
    delete h;
}

const char *vdyn_last_error(const VdynHandle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int vdyn_set_option(VdynHandle *h, int option, int value)
{
    if (!h) return VDYN_ERR_ARG;
    if (option == VDYN_OPT_LANES_PER_ROLLOUT && (value == 0 || value == 1 || value == 4)) {
        h->lanes_per_rollout = value;
        return VDYN_OK;
    }
    if (option == VDYN_OPT_STATE_ROWS && (value == 12 || value == 22)) {
        h->state_rows = value;
        return VDYN_OK;
    }
    return h->fail(VDYN_ERR_ARG, "vdyn_set_option: unknown option or value");
}

int vdyn_stream_synchronize(VdynHandle *h, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    return VDYN_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(VdynIpcHandle), "VdynIpcHandle too small for hipIpcMemHandle_t");

int vdyn_xchg_alloc(VdynHandle *h, uint64_t bytes, void **dev_ptr, VdynIpcHandle *out)
{
    if (!h) return VDYN_ERR_ARG;
    if (!dev_ptr || !out || bytes == 0) return h->fail(VDYN_ERR_ARG, "xchg_alloc: null argument or zero size");
    VDYN_HIP(h, hipSetDevice(h->device));
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return h->fail(VDYN_ERR_OOM, "xchg_alloc: hipMalloc failed"); }
    hipIpcMemHandle_t ipc;
    hipError_t e = hipIpcGetMemHandle(&ipc, p);
    if (e != hipSuccess) { (void)hipFree(p); return h->fail_hip("hipIpcGetMemHandle", e); }
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->bytes, &ipc, sizeof(ipc));
    *dev_ptr = p;
    return VDYN_OK;
}

int vdyn_xchg_free(VdynHandle *h, void *dev_ptr)
{
    if (!h) return VDYN_ERR_ARG;
    if (!dev_ptr) return VDYN_OK;
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, hipFree(dev_ptr));
    return VDYN_OK;
}

int vdyn_xchg_open(VdynHandle *h, const VdynIpcHandle *peer, void **peer_ptr)
{
    if (!h) return VDYN_ERR_ARG;
    if (!peer || !peer_ptr) return h->fail(VDYN_ERR_ARG, "xchg_open: null argument");
    VDYN_HIP(h, hipSetDevice(h->device));
    hipIpcMemHandle_t ipc;
    std::memcpy(&ipc, peer->bytes, sizeof(ipc));
    VDYN_HIP(h, hipIpcOpenMemHandle(peer_ptr, ipc, hipIpcMemLazyEnablePeerAccess));
    return VDYN_OK;
}

int vdyn_xchg_close(VdynHandle *h, void *peer_ptr)
{
    if (!h) return VDYN_ERR_ARG;
    if (!peer_ptr) return VDYN_OK;
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, hipIpcCloseMemHandle(peer_ptr));
    return VDYN_OK;
}

int vdyn_xchg_push(VdynHandle *h, void *const *dst, int32_t n_dst, uint64_t dst_offset, const void *src,
                   uint64_t bytes, void *after_stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (n_dst < 0 || n_dst > 64) return h->fail(VDYN_ERR_ARG, "xchg_push: 0..64 destinations");
    if (n_dst == 0 || bytes == 0) return VDYN_OK;
    if (!dst || !src) return h->fail(VDYN_ERR_ARG, "xchg_push: null pointer");
    for (int i = 0; i < n_dst; ++i)
        if (!dst[i]) return h->fail(VDYN_ERR_ARG, "xchg_push: null destination");
    VDYN_HIP(h, hipSetDevice(h->device));
    const int ns = std::min<int>(n_dst, VdynHandle::kCopyStreams);
    if (!h->ev_ready) VDYN_HIP(h, hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    for (int i = 0; i < ns; ++i) {
        if (!h->copy_stream[i]) VDYN_HIP(h, hipStreamCreateWithFlags(&h->copy_stream[i], hipStreamNonBlocking));
        if (!h->ev_done[i]) VDYN_HIP(h, hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
    }
    // Pushes follow each other on the DEVICE: push k + 1's copies sit behind push k's on the same in-order copy
    // streams (a destination always maps to the same stream, so the later block lands last), and what the caller
    // has to protect -- the source block, until the copies have read it -- is a device-side wait too
    // (vdyn_xchg_fence).  The host never blocks here, and a push is one event record plus, per destination, one
    // stream-wait and one copy: with seven peers the host spends less time queueing a step than the GPU running it
    // (completion events are recorded by the fence, once per several pushes, not here).
    hipError_t e = hipEventRecord(h->ev_ready, (hipStream_t)after_stream);           // the block is complete ...
    for (int i = 0; i < ns && e == hipSuccess; ++i)
        e = hipStreamWaitEvent(h->copy_stream[i], h->ev_ready, 0);                   // ... before any copy reads it
    int queued = 0;
    for (int i = 0; i < n_dst && e == hipSuccess; ++i) {
        // plain device-to-device copies: to a PEER device the runtime takes the DMA engines over xGMI by itself.
        // Asking for them explicitly (hipMemcpyDeviceToDeviceNoCU) was measured and dropped: on ROCm 7.2 that call blocks
        // the HOST until the copy has run (0.22 ms per step instead of 0.03: profiles/README.md, round 4 log)
        e = hipMemcpyAsync(static_cast<char *>(dst[i]) + dst_offset, src, bytes, hipMemcpyDeviceToDevice,
                           h->copy_stream[i % ns]);
        if (e == hipSuccess) ++queued;
    }
    if (e != hipSuccess) {
        // some copies may be queued and still read `src` / write the peers' slots: let them finish before the caller
        // frees or closes anything on this error (best effort: the streams' own errors are not the one reported)
        if (queued > 0)
            for (int i = 0; i < ns; ++i) (void)hipStreamSynchronize(h->copy_stream[i]);
        return h->fail_hip("xchg_push", e);
    }
    h->push_streams = std::max(h->push_streams, ns);
    return VDYN_OK;
}

int vdyn_xchg_fence(VdynHandle *h, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (h->push_streams == 0) return VDYN_OK;                   // nothing pushed since the last vdyn_xchg_wait
    VDYN_HIP(h, hipSetDevice(h->device));
    for (int i = 0; i < h->push_streams; ++i) {
        VDYN_HIP(h, hipEventRecord(h->ev_done[i], h->copy_stream[i]));              // behind every copy queued so far
        VDYN_HIP(h, hipStreamWaitEvent((hipStream_t)stream, h->ev_done[i], 0));
    }
    return VDYN_OK;
}

int vdyn_xchg_wait(VdynHandle *h)
{
    if (!h) return VDYN_ERR_ARG;
    if (h->push_streams == 0) return VDYN_OK;
    VDYN_HIP(h, hipSetDevice(h->device));
    for (int i = 0; i < h->push_streams; ++i) VDYN_HIP(h, hipStreamSynchronize(h->copy_stream[i]));   // in order: every push
    h->push_streams = 0;
    return VDYN_OK;
}

}  // extern "C"

// ------------------------------------------------------------- _dev implementations --

namespace {

template <typename T>
int planar_model_dev(VdynHandle *h, int64_t n, const T *state, const T *ctrl12, const T *acc_prev,
                     T *state_dot, T *aux, T *outputs, T *acc, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (n < 0) return h->fail(VDYN_ERR_ARG, "planar_model: n < 0");
    if (n == 0) return VDYN_OK;
    if (!state || !ctrl12 || !acc_prev || !state_dot || !acc)
        return h->fail(VDYN_ERR_ARG, "planar_model: null state / ctrl12 / acc_prev / state_dot / acc");
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, vdyn::launch_planar_model<T>(h->p, n, state, ctrl12, acc_prev, state_dot, aux, outputs,
                                             acc, (hipStream_t)stream));
    return VDYN_OK;
}

// The lane kernels write trajectory / log / DataLog rows through RowWriter (vdyn_kernels.hip): wave-uniform 64-bit row
// base + 32-bit column and lane offsets.  A row of `cols` columns must therefore span at most 2^31 bytes
// (fp32 trajectories: n <= 44.7 M rollouts; fp32 DataLog: n <= 11.9 M vehicles) -- split larger batches.
template <typename T>
static bool row_writer_fits(int64_t n, int cols) { return (int64_t)cols * n * (int64_t)sizeof(T) <= ((int64_t)1 << 31); }

template <typename T>
int rollout_dev(VdynHandle *h, const vdyn::RolloutArgs<T> &a, void *stream, const char *who)
{
    if (!h) return VDYN_ERR_ARG;
    const std::string w(who);
    if (a.n < 0 || a.H < 0) return h->fail(VDYN_ERR_ARG, w + ": n < 0 or H < 0");
    if (a.k != 2 && a.k != 12) return h->fail(VDYN_ERR_ARG, w + ": k must be 2 or 12");
    if (a.layout != VDYN_CTRL_PER_ROLLOUT && a.layout != VDYN_CTRL_SHARED)
        return h->fail(VDYN_ERR_ARG, w + ": unknown control layout");
    if (!std::isfinite(a.dt)) return h->fail(VDYN_ERR_ARG, w + ": dt is not finite");
    if (a.n == 0) return VDYN_OK;
    if (!a.state0 || !a.terminal) return h->fail(VDYN_ERR_ARG, w + ": null state buffer");
    if (a.H > 0 && !a.ctrl) return h->fail(VDYN_ERR_ARG, w + ": null ctrl");
    if (a.layout == VDYN_CTRL_SHARED && (!a.path_id || a.P <= 0))
        return h->fail(VDYN_ERR_ARG, w + ": shared controls need path_id and P > 0");
    if (a.traj && a.traj_stride <= 0) return h->fail(VDYN_ERR_ARG, w + ": traj needs traj_stride > 0");
    if (a.traj && !row_writer_fits<T>(a.n, 12))
        return h->fail(VDYN_ERR_ARG, w + ": a trajectory row (12 n values) must not exceed 2^31 bytes; split the batch");
    if ((a.state_dot || a.outputs) && a.layout != VDYN_CTRL_PER_ROLLOUT)
        return h->fail(VDYN_ERR_ARG, w + ": diagnostics (state_dot / outputs) come with per-rollout controls");
    VDYN_HIP(h, hipSetDevice(h->device));
    vdyn::RolloutArgs<T> b = a;
    b.state_rows = h->state_rows;
    if (b.state_rows != 12) {
        if (sizeof(T) != 4) return h->fail(VDYN_ERR_ARG, w + ": VDYN_OPT_STATE_ROWS = 22 is an fp32 option");
        if (a.state_dot || a.outputs) return h->fail(VDYN_ERR_ARG, w + ": no diagnostics with VDYN_OPT_STATE_ROWS = 22");
    }
    // 0 = automatic: wheel-parallel while it is measurably faster (tools/sweep_lanes.py at the
    // sustained clock, against the lane kernel, round 2's builds: fp32 1.42x up to 16384 rollouts, 0.84x at
    // 24576; fp64 1.92x up to 16384, 1.17x at 32768, 0.80x at 40960)
    const int64_t auto_max = sizeof(T) == 4 ? 16384 : 32768;
    b.lanes_per_rollout = h->lanes_per_rollout == 0 ? (a.n <= auto_max ? 4 : 1) : h->lanes_per_rollout;
    if (b.state_rows != 12) b.lanes_per_rollout = 1;         // the compensated sum lives in the lane kernel
    const hipError_t launched = vdyn::launch_rollout<T>(h->p, b, (hipStream_t)stream);
    if (launched == hipErrorInvalidValue && b.state_rows != 12)
        return h->fail(VDYN_ERR_ARG, w + ": VDYN_OPT_STATE_ROWS = 22 needs the fitted tire chain (B >= 0, validated fits)");
    VDYN_HIP(h, launched);
    return VDYN_OK;
}

template <typename T>
int rollout_spiral_dev(VdynHandle *h, int64_t n, int32_t H, const T *state0, const T *spiral, double wheelbase,
                       double max_steer, double torque, double dt, const double *mu4, T *terminal, T *traj,
                       int32_t traj_stride, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (n < 0 || H < 0) return h->fail(VDYN_ERR_ARG, "rollout_spiral: n < 0 or H < 0");
    if (!std::isfinite(dt) || !std::isfinite(wheelbase) || !std::isfinite(torque) || std::isnan(max_steer) || max_steer < 0)
        return h->fail(VDYN_ERR_ARG, "rollout_spiral: dt / wheelbase / torque must be finite, max_steer >= 0");
    if (n == 0) return VDYN_OK;
    if (!state0 || !spiral || !terminal) return h->fail(VDYN_ERR_ARG, "rollout_spiral: null buffer");
    if (traj && traj_stride <= 0) return h->fail(VDYN_ERR_ARG, "rollout_spiral: traj needs traj_stride > 0");
    if (traj && !row_writer_fits<T>(n, 12))
        return h->fail(VDYN_ERR_ARG, "rollout_spiral: a trajectory row (12 n values) must not exceed 2^31 bytes; split the batch");
    const double half_pi = 1.5707963267948966;
    const double tan_max = max_steer >= half_pi ? INFINITY : std::tan(max_steer);
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, vdyn::launch_rollout_spiral<T>(h->p, n, H, state0, spiral, wheelbase, tan_max, torque, dt, mu4,
                                               terminal, traj, traj_stride, (hipStream_t)stream));
    return VDYN_OK;
}

template <typename T>
int nonfinite_dev(VdynHandle *h, int32_t rows, int64_t n, const T *x, int32_t *status, int64_t *count, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (rows <= 0 || n < 0) return h->fail(VDYN_ERR_ARG, "nonfinite_lanes: need rows > 0, n >= 0");
    if (count) *count = 0;
    if (n == 0) return VDYN_OK;
    if (!x || !status) return h->fail(VDYN_ERR_ARG, "nonfinite_lanes: null buffer");
    VDYN_HIP(h, hipSetDevice(h->device));
    unsigned long long *dcount = nullptr;
    if (count) {
        if (!h->d_count) {
            if (hipMalloc(&h->d_count, sizeof(unsigned long long)) != hipSuccess) return h->fail(VDYN_ERR_OOM, "nonfinite_lanes: allocation failed");
        }
        dcount = static_cast<unsigned long long *>(h->d_count);
        VDYN_HIP(h, hipMemsetAsync(dcount, 0, sizeof(unsigned long long), (hipStream_t)stream));
    }
    VDYN_HIP(h, vdyn::launch_nonfinite_lanes<T>(rows, n, x, status, dcount, (hipStream_t)stream));
    if (count) {
        unsigned long long c = 0;
        VDYN_HIP(h, hipMemcpyAsync(&c, dcount, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
        VDYN_HIP(h, hipStreamSynchronize((hipStream_t)stream));
        *count = (int64_t)c;
    }
    return VDYN_OK;
}

template <typename T>
int fastmath_dev(VdynHandle *h, int32_t fn, int64_t n, const T *x, double c, T *out0, T *out1, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (fn < 0 || fn > (sizeof(T) == 4 ? 8 : 5) || n < 0) return h->fail(VDYN_ERR_ARG, "fastmath_eval: unknown function or n < 0");
    if (n == 0) return VDYN_OK;
    if (!x || !out0) return h->fail(VDYN_ERR_ARG, "fastmath_eval: null buffer");
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, vdyn::launch_fastmath_eval<T>(fn, n, x, c, out0, out1, (hipStream_t)stream));
    return VDYN_OK;
}

// Fleet rollout: build the per-class table on the host, put it on the device (a small pageable
// copy, enqueued on the caller's stream in front of the kernel), launch.
template <typename T>
int rollout_fleet_dev(VdynHandle *h, vdyn::RolloutArgs<T> a, const VdynParams *classes, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (a.n < 0 || a.H < 0 || (a.k != 2 && a.k != 12) ||
        (a.layout != VDYN_CTRL_PER_ROLLOUT && a.layout != VDYN_CTRL_SHARED) || !std::isfinite(a.dt))
        return rollout_dev<T>(h, a, stream, "rollout_fleet");
    if (a.V < 1 || a.V > 256 || !classes) return h->fail(VDYN_ERR_ARG, "rollout_fleet: need 1 <= V <= 256 classes");
    std::string why;
    for (int v = 0; v < a.V; ++v)
        if (!params_ok(&classes[v], &why)) return h->fail(VDYN_ERR_ARG, "rollout_fleet: class " + std::to_string(v) + ": " + why);
    if (a.n == 0) return VDYN_OK;
    if (!a.state0 || !a.terminal || !a.vehicle_id || (a.H > 0 && !a.ctrl) ||
        (a.layout == VDYN_CTRL_SHARED && (!a.path_id || a.P <= 0)) || (a.traj && a.traj_stride <= 0))
        return h->fail(VDYN_ERR_ARG, "rollout_fleet: null buffer or bad traj_stride");
    if (a.traj && !row_writer_fits<T>(a.n, 12))
        return h->fail(VDYN_ERR_ARG, "rollout_fleet: a trajectory row (12 n values) must not exceed 2^31 bytes; split the batch");
    VDYN_HIP(h, hipSetDevice(h->device));
    const size_t bytes = sizeof(T) * (size_t)vdyn::fleet_table_len<T>(a.V);
    if (bytes > h->d_fleet_bytes) {
        if (h->d_fleet) { (void)hipFree(h->d_fleet); h->d_fleet = nullptr; h->d_fleet_bytes = 0; }
        if (hipMalloc(&h->d_fleet, bytes) != hipSuccess) return h->fail(VDYN_ERR_OOM, "fleet table allocation failed");
        h->d_fleet_bytes = bytes;
    }
    std::vector<T> tab(bytes / sizeof(T));
    bool all_small = true;
    vdyn::build_fleet_table<T>(classes, a.V, a.mu4, tab.data(), &all_small);
    // pageable source: the runtime has consumed `tab` when this call returns
    VDYN_HIP(h, hipMemcpyAsync(h->d_fleet, tab.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    a.fleet_tab = static_cast<const T *>(h->d_fleet);
    VDYN_HIP(h, vdyn::launch_rollout_fleet<T>(a, all_small, (hipStream_t)stream));
    return VDYN_OK;
}

template <typename T>
int mpc_dev(VdynHandle *h, int E, int C, int H, const T *ego, const T *cand, const T *goal, double dt,
            double w_delta, T *best_cost, int32_t *best_idx, T *cost_all, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (E < 0 || C <= 0 || H < 0) return h->fail(VDYN_ERR_ARG, "mpc_argmin: need E >= 0, C > 0, H >= 0");
    if (!std::isfinite(dt)) return h->fail(VDYN_ERR_ARG, "mpc_argmin: dt is not finite");
    if (E == 0) return VDYN_OK;
    if (!ego || !goal || !best_cost || !best_idx || (H > 0 && !cand))
        return h->fail(VDYN_ERR_ARG, "mpc_argmin: null buffer");
    VDYN_HIP(h, hipSetDevice(h->device));
    // candidate table with (sin, cos) of the steering angles + per-chunk partial minima: handle-owned scratch, grown on demand
    // (hipFree waits for launches still reading the old one)
    const size_t c4 = vdyn::mpc_scratch_bytes<T>(E, C, H);
    if (c4 > h->d_cand_bytes) {
        if (h->d_cand) { (void)hipFree(h->d_cand); h->d_cand = nullptr; h->d_cand_bytes = 0; }
        if (hipMalloc(&h->d_cand, c4) != hipSuccess) {
            (void)hipGetLastError();
            h->d_cand = nullptr;
            return h->fail(VDYN_ERR_OOM, "mpc_argmin: candidate table allocation failed");
        }
        h->d_cand_bytes = c4;
    }
    VDYN_HIP(h, vdyn::launch_mpc_argmin<T>(h->p, E, C, H, ego, cand, goal, dt, w_delta, best_cost,
                                           best_idx, cost_all, h->d_cand, (hipStream_t)stream));
    return VDYN_OK;
}

// the handle's device scratch and pinned staging buffer, grown on demand (contents are not kept)
static int ensure_scratch(VdynHandle *h, size_t dev_bytes, size_t pinned_bytes)
{
    if (dev_bytes > h->d_bytes) {
        if (h->d_scratch) { (void)hipFree(h->d_scratch); h->d_scratch = nullptr; h->d_bytes = 0; }
        if (hipMalloc(&h->d_scratch, dev_bytes) != hipSuccess) {
            (void)hipGetLastError();
            h->d_scratch = nullptr;
            return h->fail(VDYN_ERR_OOM, "device scratch allocation failed");
        }
        h->d_bytes = dev_bytes;
    }
    if (pinned_bytes > h->h_bytes) {
        if (h->h_pinned) { (void)hipHostFree(h->h_pinned); h->h_pinned = nullptr; h->h_bytes = 0; }
        if (hipHostMalloc(&h->h_pinned, pinned_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h->h_pinned = nullptr;
            return h->fail(VDYN_ERR_OOM, "pinned staging allocation failed");
        }
        h->h_bytes = pinned_bytes;
    }
    return VDYN_OK;
}

// ---- host staging: [inputs | outputs] packed in one pinned buffer and one device
// buffer, so a call costs one H2D copy, the kernel, one D2H copy.
class Stage {
public:
    explicit Stage(VdynHandle *h) : h_(h) {}
    // returns the byte offset of the region inside the staging buffers
    size_t in(const void *src, size_t bytes) { return add(ins_, const_cast<void *>(src), bytes); }
    size_t out(void *dst, size_t bytes) { return add(outs_, dst, bytes); }

    // Calls this small (the single-vehicle drop-ins of drive.py:141-143) skip both DMA copies:
    // the kernel reads its inputs from, and writes its outputs to, a host-coherent mapped
    // buffer across PCIe, so a call costs one launch and one stream synchronise.
    // Only for calls of a few KB: every word crosses PCIe on its own, so anything a lane
    // loops over (waypoint tables, control horizons) must go through HBM instead.
    static constexpr size_t kMappedBytes = 4 * 1024;

    int upload()
    {
        in_end_ = 0;
        size_t off = 0;
        for (auto &r : ins_) { r.off = off; off += round(r.bytes); }
        in_end_ = off;
        for (auto &r : outs_) { r.off = off; off += round(r.bytes); }
        total_ = off;
        VDYN_HIP(h_, hipSetDevice(h_->device));
        mapped_ = total_ <= kMappedBytes;
        if (mapped_) {
            if (!h_->h_mapped) {
                if (hipHostMalloc(&h_->h_mapped, kMappedBytes, hipHostMallocMapped) != hipSuccess)
                    return h_->fail(VDYN_ERR_OOM, "mapped staging allocation failed");
                VDYN_HIP(h_, hipHostGetDevicePointer(&h_->d_mapped, h_->h_mapped, 0));
            }
            host_ = (char *)h_->h_mapped;
            dev_ = (char *)h_->d_mapped;
            for (auto &r : ins_) h_->host_copy(host_ + r.off, r.ptr, r.bytes);
            return VDYN_OK;
        }
        if (int rc = ensure_scratch(h_, total_, total_)) return rc;
        host_ = (char *)h_->h_pinned;
        dev_ = (char *)h_->d_scratch;
        for (auto &r : ins_) h_->host_copy(host_ + r.off, r.ptr, r.bytes);
        if (in_end_ > 0)
            VDYN_HIP(h_, hipMemcpyAsync(dev_, host_, in_end_, hipMemcpyHostToDevice, h_->stream));
        return VDYN_OK;
    }
    template <typename T> T *dev(size_t idx, bool is_out) const
    {
        const auto &r = (is_out ? outs_ : ins_)[idx];
        return r.ptr ? reinterpret_cast<T *>(dev_ + r.off) : nullptr;
    }
    int download()
    {
        if (!mapped_ && total_ > in_end_)
            VDYN_HIP(h_, hipMemcpyAsync(host_ + in_end_, dev_ + in_end_, total_ - in_end_, hipMemcpyDeviceToHost,
                                        h_->stream));
        VDYN_HIP(h_, hipStreamSynchronize(h_->stream));
        for (auto &r : outs_)
            if (r.ptr) h_->host_copy(r.ptr, host_ + r.off, r.bytes);
        return VDYN_OK;
    }

private:
    struct Region { void *ptr; size_t bytes; size_t off; };
    static size_t round(size_t b) { return (b + 255) & ~(size_t)255; }
    size_t add(std::vector<Region> &v, void *p, size_t bytes)
    {
        v.push_back({p, p ? bytes : 0, 0});
        return v.size() - 1;
    }
    VdynHandle *h_;
    std::vector<Region> ins_, outs_;
    size_t in_end_ = 0, total_ = 0;
    bool mapped_ = false;
    char *host_ = nullptr, *dev_ = nullptr;
};

template <typename T>
int rollout_fleet_host(VdynHandle *h, vdyn::RolloutArgs<T> a, const VdynParams *classes)
{
    if (!h) return VDYN_ERR_ARG;
    if (a.n <= 0 || a.H < 0 || (a.k != 2 && a.k != 12) || !a.state0 || !a.terminal || !a.vehicle_id ||
        (a.H > 0 && !a.ctrl) || (a.layout != VDYN_CTRL_PER_ROLLOUT && a.layout != VDYN_CTRL_SHARED) ||
        (a.layout == VDYN_CTRL_SHARED && (!a.path_id || a.P <= 0)) || (a.traj && a.traj_stride <= 0))
        return a.n == 0 ? VDYN_OK : rollout_fleet_dev<T>(h, a, classes, h->stream);
    const size_t e = sizeof(T) * (size_t)a.n;
    const size_t ctrl_bytes = a.layout == VDYN_CTRL_PER_ROLLOUT ? (size_t)a.H * a.k * e
                                                                : sizeof(T) * (size_t)a.P * a.H * a.k;
    Stage s(h);
    const size_t i0 = s.in(a.state0, 12 * e), i1 = s.in(a.ctrl, ctrl_bytes),
                 i2 = s.in(a.path_id, a.layout == VDYN_CTRL_SHARED ? sizeof(int32_t) * (size_t)a.n : 0),
                 i3 = s.in(a.vehicle_id, sizeof(int32_t) * (size_t)a.n);
    const size_t o0 = s.out(a.terminal, 12 * e),
                 o1 = s.out(a.traj, a.traj ? (size_t)(a.H / a.traj_stride) * 12 * e : 0);
    int rc = s.upload();
    if (rc) return rc;
    a.state0 = s.dev<T>(i0, false); a.ctrl = s.dev<T>(i1, false); a.path_id = s.dev<int>(i2, false);
    a.vehicle_id = s.dev<int>(i3, false);
    a.terminal = s.dev<T>(o0, true); a.traj = s.dev<T>(o1, true);
    rc = rollout_fleet_dev<T>(h, a, classes, h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int planar_model_host(VdynHandle *h, int64_t n, const T *state, const T *ctrl12, const T *acc_prev,
                      T *state_dot, T *aux, T *outputs, T *acc)
{
    if (!h) return VDYN_ERR_ARG;
    if (n < 0) return h->fail(VDYN_ERR_ARG, "planar_model: n < 0");
    if (n == 0) return VDYN_OK;
    if (!state || !ctrl12 || !acc_prev || !state_dot || !acc)
        return h->fail(VDYN_ERR_ARG, "planar_model: null state / ctrl12 / acc_prev / state_dot / acc");
    const size_t e = sizeof(T) * (size_t)n;
    Stage s(h);
    const size_t i0 = s.in(state, 10 * e), i1 = s.in(ctrl12, 12 * e), i2 = s.in(acc_prev, 2 * e);
    const size_t o0 = s.out(state_dot, 10 * e), o1 = s.out(aux, 4 * e), o2 = s.out(outputs, 18 * e),
                 o3 = s.out(acc, 2 * e);
    int rc = s.upload();
    if (rc) return rc;
    rc = planar_model_dev<T>(h, n, s.dev<T>(i0, false), s.dev<T>(i1, false), s.dev<T>(i2, false),
                             s.dev<T>(o0, true), s.dev<T>(o1, true), s.dev<T>(o2, true), s.dev<T>(o3, true),
                             h->stream);
    if (rc) return rc;
    return s.download();
}

// (Shared-table rollouts come through here too when their TRAJECTORY is large: the chunk's table is gathered on the host.)
// Per-rollout controls [H][k][n] from host memory are the one _host input that is LARGE (105 MB at BASELINE configs[2],
// against 3 MB of states): staged whole -- memcpy into pinned memory, one H2D copy, the kernel -- the call costs the sum
// of the three.  Here the horizon is cut into chunks of whole steps (a chunk of [H][k][n] is one contiguous slab) that
// flow through two pinned and two device buffers: while the worker threads copy chunk c + 1 into pinned memory, chunk c
// crosses PCIe on the upload stream and the kernel integrates chunk c - 1 on the compute stream, handing the state on
// device to device (two state buffers, alternating).  What makes this exact is the kernels' split-horizon identity:
// rollout(a) then rollout(b) from its terminal state IS rollout(a + b), bit for bit, trajectory rows included when the
// cut is a multiple of traj_stride (tests/test_gpu_parity.py holds every kernel to it), so the host ABI still returns
// what the device ABI returns for the same inputs.
//   pinned: [state0 | terminal | ctrl 0 | ctrl 1 | traj]      device: [state A | state B | ctrl 0 | ctrl 1 | traj]
static constexpr size_t kPipelineMinBytes = 8u << 20;      // below this the whole-buffer staging is as fast

static int ensure_pipeline(VdynHandle *h)
{
    VDYN_HIP(h, hipSetDevice(h->device));
    if (!h->stage_stream) VDYN_HIP(h, hipStreamCreateWithFlags(&h->stage_stream, hipStreamNonBlocking));
    if (!h->down_stream) VDYN_HIP(h, hipStreamCreateWithFlags(&h->down_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!h->ev_h2d[i]) VDYN_HIP(h, hipEventCreateWithFlags(&h->ev_h2d[i], hipEventDisableTiming));
        if (!h->ev_kernel[i]) VDYN_HIP(h, hipEventCreateWithFlags(&h->ev_kernel[i], hipEventDisableTiming));
        if (!h->ev_d2h[i]) VDYN_HIP(h, hipEventCreateWithFlags(&h->ev_d2h[i], hipEventDisableTiming));
    }
    return VDYN_OK;
}

// byte offsets of the regions of a staging buffer (the same layout in pinned and in device memory), 256-byte aligned
struct Layout {
    size_t total = 0;
    size_t add(size_t bytes)
    {
        const size_t off = total;
        total += (bytes + 255) & ~(size_t)255;
        return off;
    }
};

// The way OUT of a chunked run: rows a kernel chunk wrote into device buffer b cross PCIe on the download stream into
// pinned buffer b while the next chunk is integrated, and the worker threads copy pinned buffer b ^ 1 -- the chunk before
// -- into the caller's memory meanwhile.  Two buffers each: chunk c + 2's kernel waits (on the device) for chunk c's
// download, and the download of chunk c + 2 is queued only after the host has emptied pinned buffer b (program order).
// Staging memory is two chunks, not the whole log: 65536 vehicles x 100 sub-steps of DataLog are 1.2 GB.
struct Downloader {
    struct Seg { size_t off, dev_bytes; void *dst; };   // region of the chunk buffer -> the caller's memory
    VdynHandle *h;
    char *dev[2], *pin[2];
    std::vector<Seg> pending[2];
    bool busy[2] = {false, false};
    Downloader(VdynHandle *h_, char *dev0, char *dev1, char *pin0, char *pin1) : h(h_), dev{dev0, dev1}, pin{pin0, pin1} {}

    // before chunk c's kernel is launched: its device buffer must have been downloaded (chunk c - 2).  (Today the host
    // has already waited for that download -- flush() of chunk c - 2 ran during chunk c - 1 -- so this wait never
    // blocks; it is here so that the buffer discipline does not hang on how the host happens to be ordered.)
    int before_kernel(int c)
    {
        if (c >= 2) VDYN_HIP(h, hipStreamWaitEvent(h->stream, h->ev_d2h[c & 1], 0));
        return VDYN_OK;
    }
    // after chunk c's kernel (its ev_kernel[b] recorded): queue the download of `segs`, then empty the other buffer
    int after_kernel(int c, const std::vector<Seg> &segs)
    {
        const int b = c & 1;
        VDYN_HIP(h, hipStreamWaitEvent(h->down_stream, h->ev_kernel[b], 0));
        for (const Seg &sg : segs)
            if (sg.dev_bytes) VDYN_HIP(h, hipMemcpyAsync(pin[b] + sg.off, dev[b] + sg.off, sg.dev_bytes, hipMemcpyDeviceToHost, h->down_stream));
        VDYN_HIP(h, hipEventRecord(h->ev_d2h[b], h->down_stream));
        pending[b] = segs;
        busy[b] = true;
        return flush(b ^ 1);
    }
    int flush(int b)
    {
        if (!busy[b]) return VDYN_OK;
        VDYN_HIP(h, hipEventSynchronize(h->ev_d2h[b]));
        for (const Seg &sg : pending[b]) h->host_copy(sg.dst, pin[b] + sg.off, sg.dev_bytes);
        busy[b] = false;
        return VDYN_OK;
    }
    int finish(int chunks) { if (int rc = flush(chunks & 1)) return rc; return flush((chunks & 1) ^ 1); }
};

// (inside the pipelined functions: a failed HIP call drains the streams, then reports)
#define VDYN_PIPE(expr)                                                     \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) return drain(h->fail_hip(#expr, e_));         \
    } while (0)

template <typename T>
int closed_loop_dev(VdynHandle *h, const VdynCtrlGains *g, const vdyn::ClosedLoopArgs<T> &a, bool update_only,
                    void *stream);

template <typename T>
int rollout_host_pipelined(VdynHandle *h, vdyn::RolloutArgs<T> a, const char *who)
{
    const size_t e = sizeof(T) * (size_t)a.n, rows = (size_t)h->state_rows;
    // a step's controls: k rows of n values per rollout, or -- shared table [P][H][k] -- k values per path: a chunk's
    // table [P][hn][k] is gathered from the caller's, path by path (it is small: what streams then is the trajectory)
    const bool shared = a.layout == VDYN_CTRL_SHARED;
    const size_t step_bytes = shared ? sizeof(T) * (size_t)a.P * a.k : (size_t)a.k * e;
    // what a step moves: its controls up and, with a trajectory, 12 rows down every traj_stride steps
    const size_t step_moves = step_bytes + (a.traj ? 12 * e / (size_t)a.traj_stride : 0);
    // chunk: about a twelfth of the horizon's bytes, 2 .. 16 MB, whole steps, a multiple of traj_stride.  The first two
    // chunks are a quarter and a half of that: nothing overlaps the first chunk's host copy and upload (the pipeline's
    // fill), so it is kept short; measured on one box, 105 MB: 2.47 ms with eight equal chunks of 13 MB, 2.42 ramped
    const size_t want = std::min<size_t>(std::max<size_t>((size_t)a.H * step_moves / 12, 2u << 20), 16u << 20);
    int64_t hc = std::max<int64_t>(1, (int64_t)(want / step_moves));
    const int64_t unit = a.traj ? a.traj_stride : 1;
    hc = std::max<int64_t>(unit, hc / unit * unit);
    hc = std::min<int64_t>(hc, a.H);
    auto ramp = [&](int c) {                                    // steps of chunk c
        const int64_t part = c == 0 ? hc / 4 : c == 1 ? hc / 2 : hc;
        return std::max<int64_t>(unit, part / unit * unit);
    };
    const size_t chunk_bytes = (size_t)hc * step_bytes;
    const size_t traj_chunk = a.traj ? (size_t)(hc / a.traj_stride) * 12 * e : 0;
    Layout L;
    const size_t o_state[2] = {L.add(rows * e), L.add(rows * e)};       // pinned: state0 | terminal; device: state A | state B
    const size_t o_ctrl[2] = {L.add(chunk_bytes), L.add(chunk_bytes)};
    const size_t o_traj[2] = {L.add(traj_chunk), L.add(traj_chunk)};
    const size_t pid_bytes = shared ? sizeof(int32_t) * (size_t)a.n : 0, o_pid = L.add(pid_bytes);
    if (int rc = ensure_scratch(h, L.total, L.total)) return rc;
    if (int rc = ensure_pipeline(h)) return rc;
    char *pin = static_cast<char *>(h->h_pinned), *dev = static_cast<char *>(h->d_scratch);
    T *dev_state[2] = {reinterpret_cast<T *>(dev + o_state[0]), reinterpret_cast<T *>(dev + o_state[1])};
    Downloader down(h, dev + o_traj[0], dev + o_traj[1], pin + o_traj[0], pin + o_traj[1]);

    // On any failure past this point copies may be in flight from / into the staging buffers: drain the streams
    // before the error goes back (the buffers belong to the handle and the next call reuses them).
    auto drain = [&](int rc) {
        (void)hipStreamSynchronize(h->stage_stream);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->down_stream);
        return rc;
    };

    h->host_copy(pin + o_state[0], a.state0, rows * e);
    VDYN_PIPE(hipMemcpyAsync(dev_state[0], pin + o_state[0], rows * e, hipMemcpyHostToDevice, h->stage_stream));
    if (shared) {
        h->host_copy(pin + o_pid, a.path_id, pid_bytes);
        VDYN_PIPE(hipMemcpyAsync(dev + o_pid, pin + o_pid, pid_bytes, hipMemcpyHostToDevice, h->stage_stream));
    }
    const char *src = reinterpret_cast<const char *>(a.ctrl);
    int64_t t0 = 0, traj_row = 0;
    int c = 0;
    for (int64_t hn = 0; t0 < a.H; t0 += hn, ++c) {
        const int b = c & 1;
        hn = std::min<int64_t>(ramp(c), a.H - t0);
        const size_t bytes = (size_t)hn * step_bytes;
        // pinned ctrl[b] was the source of chunk c - 2's upload: that copy must have run before the host overwrites it
        if (c >= 2) VDYN_PIPE(hipEventSynchronize(h->ev_h2d[b]));
        if (shared) {
            const size_t path_chunk = (size_t)hn * a.k * sizeof(T), path_full = (size_t)a.H * a.k * sizeof(T);
            for (int p = 0; p < a.P; ++p)
                std::memcpy(pin + o_ctrl[b] + (size_t)p * path_chunk, src + (size_t)p * path_full + (size_t)t0 * a.k * sizeof(T), path_chunk);
        } else {
            h->host_copy(pin + o_ctrl[b], src + (size_t)t0 * step_bytes, bytes);
        }
        // device ctrl[b] is what chunk c - 2's kernel reads: the upload stream waits for that kernel, on the device
        if (c >= 2) VDYN_PIPE(hipStreamWaitEvent(h->stage_stream, h->ev_kernel[b], 0));
        VDYN_PIPE(hipMemcpyAsync(dev + o_ctrl[b], pin + o_ctrl[b], bytes, hipMemcpyHostToDevice, h->stage_stream));
        VDYN_PIPE(hipEventRecord(h->ev_h2d[b], h->stage_stream));
        // (the upload stream is in order: chunk 0's event also covers the state upload queued before it)
        VDYN_PIPE(hipStreamWaitEvent(h->stream, h->ev_h2d[b], 0));
        if (a.traj)
            if (int rc = down.before_kernel(c)) return drain(rc);
        vdyn::RolloutArgs<T> ac = a;
        ac.H = (int)hn;
        ac.state0 = dev_state[b];
        ac.terminal = dev_state[b ^ 1];
        ac.ctrl = reinterpret_cast<const T *>(dev + o_ctrl[b]);
        ac.path_id = shared ? reinterpret_cast<const int *>(dev + o_pid) : nullptr;
        ac.traj = a.traj ? reinterpret_cast<T *>(dev + o_traj[b]) : nullptr;
        if (int rc = rollout_dev<T>(h, ac, h->stream, who)) return drain(rc);
        VDYN_PIPE(hipEventRecord(h->ev_kernel[b], h->stream));
        if (a.traj) {
            // this chunk's trajectory rows go down while the next chunk is integrated; the chunk before is copied out
            const size_t nrows = (size_t)(hn / a.traj_stride);
            if (int rc = down.after_kernel(c, {{0, nrows * 12 * e, reinterpret_cast<char *>(a.traj) + (size_t)traj_row * 12 * e}}))
                return drain(rc);
            traj_row += (int64_t)nrows;
        }
    }
    VDYN_PIPE(hipMemcpyAsync(pin + o_state[1], dev_state[c & 1], rows * e, hipMemcpyDeviceToHost, h->stream));
    if (a.traj)
        if (int rc = down.finish(c)) return drain(rc);
    VDYN_PIPE(hipStreamSynchronize(h->stream));
    h->host_copy(a.terminal, pin + o_state[1], rows * e);
    return VDYN_OK;
}

// The closed loop's logs are the host ABI's other large transfer, downwards: the 45-column DataLog of 65536 vehicles x 100
// sub-steps is 1.2 GB for a 0.2 ms kernel.  The horizon is cut at multiples of ctrl_every (launches chained there repeat
// the single launch's log rows bit for bit: include/vdyn.h; tests/test_gpu_parity.py holds the kernels to it) and the
// chunks' rows stream out through the Downloader while the next chunk runs; state and controller state go from launch to
// launch on the device.
template <typename T>
int closed_loop_host_pipelined(VdynHandle *h, const VdynCtrlGains *g, vdyn::ClosedLoopArgs<T> a, int64_t hc)
{
    const size_t e = sizeof(T) * (size_t)a.n;
    const size_t wp_bytes = sizeof(T) * (size_t)a.P * a.Wmax * 2, wc_bytes = sizeof(int32_t) * (size_t)a.P,
                 pid_bytes = sizeof(int32_t) * (size_t)a.n;
    Layout L;                                                   // inputs first: one upload covers them
    const size_t o_s0 = L.add(12 * e), o_c0 = L.add(6 * e), o_wp = L.add(wp_bytes), o_wc = L.add(wc_bytes), o_pid = L.add(pid_bytes);
    const size_t in_end = L.total;
    const size_t o_s1 = L.add(12 * e), o_c1 = L.add(6 * e);
    const size_t log_chunk = a.log ? (size_t)hc * 16 * e : 0, dl_chunk = a.datalog ? (size_t)hc * 45 * e : 0;
    const size_t o_rows[2] = {L.add(log_chunk + 256 + dl_chunk), L.add(log_chunk + 256 + dl_chunk)};
    const size_t dl_off = (log_chunk + 255) & ~(size_t)255;     // DataLog rows behind the log rows inside a chunk buffer
    if (int rc = ensure_scratch(h, L.total, L.total)) return rc;
    if (int rc = ensure_pipeline(h)) return rc;
    char *pin = static_cast<char *>(h->h_pinned), *dev = static_cast<char *>(h->d_scratch);
    auto drain = [&](int rc) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->down_stream);
        return rc;
    };
    h->host_copy(pin + o_s0, a.state0, 12 * e);
    h->host_copy(pin + o_c0, a.cstate0, 6 * e);
    h->host_copy(pin + o_wp, a.wp, wp_bytes);
    h->host_copy(pin + o_wc, a.wcount, wc_bytes);
    h->host_copy(pin + o_pid, a.path_id, pid_bytes);
    VDYN_PIPE(hipMemcpyAsync(dev, pin, in_end, hipMemcpyHostToDevice, h->stream));
    T *dstate[2] = {reinterpret_cast<T *>(dev + o_s0), reinterpret_cast<T *>(dev + o_s1)};
    T *dcs[2] = {reinterpret_cast<T *>(dev + o_c0), reinterpret_cast<T *>(dev + o_c1)};
    Downloader down(h, dev + o_rows[0], dev + o_rows[1], pin + o_rows[0], pin + o_rows[1]);
    int64_t t0 = 0;
    int c = 0;
    for (int64_t hn = 0; t0 < a.H; t0 += hn, ++c) {
        const int b = c & 1;
        // the first chunk ends where (phase + t) is next a multiple of ctrl_every, every later one spans whole periods
        hn = c == 0 ? hc - (a.phase % a.ctrl_every + a.ctrl_every) % a.ctrl_every : hc;
        if (hn <= 0) hn = hc;
        hn = std::min<int64_t>(hn, a.H - t0);
        if (int rc = down.before_kernel(c)) return drain(rc);
        vdyn::ClosedLoopArgs<T> ac = a;
        ac.H = (int)hn;
        ac.phase = a.phase + (int)t0;
        ac.state0 = dstate[b];
        ac.cstate0 = dcs[b];
        ac.terminal = dstate[b ^ 1];
        ac.cstate = dcs[b ^ 1];
        ac.wp = reinterpret_cast<const T *>(dev + o_wp);
        ac.wcount = reinterpret_cast<const int *>(dev + o_wc);
        ac.path_id = reinterpret_cast<const int *>(dev + o_pid);
        ac.log = a.log ? reinterpret_cast<T *>(dev + o_rows[b]) : nullptr;
        ac.datalog = a.datalog ? reinterpret_cast<T *>(dev + o_rows[b] + dl_off) : nullptr;
        if (int rc = closed_loop_dev<T>(h, g, ac, false, h->stream)) return drain(rc);
        VDYN_PIPE(hipEventRecord(h->ev_kernel[b], h->stream));
        std::vector<Downloader::Seg> segs;
        if (a.log) segs.push_back({0, (size_t)hn * 16 * e, reinterpret_cast<char *>(a.log) + (size_t)t0 * 16 * e});
        if (a.datalog) segs.push_back({dl_off, (size_t)hn * 45 * e, reinterpret_cast<char *>(a.datalog) + (size_t)t0 * 45 * e});
        if (int rc = down.after_kernel(c, segs)) return drain(rc);
    }
    // terminal and controller state: into the pinned regions of the inputs (uploaded long ago)
    VDYN_PIPE(hipMemcpyAsync(pin + o_s0, dstate[c & 1], 12 * e, hipMemcpyDeviceToHost, h->stream));
    VDYN_PIPE(hipMemcpyAsync(pin + o_c0, dcs[c & 1], 6 * e, hipMemcpyDeviceToHost, h->stream));
    if (int rc = down.finish(c)) return drain(rc);
    VDYN_PIPE(hipStreamSynchronize(h->stream));
#undef VDYN_PIPE
    h->host_copy(a.terminal, pin + o_s0, 12 * e);
    h->host_copy(a.cstate, pin + o_c0, 6 * e);
    return VDYN_OK;
}

template <typename T>
int rollout_host(VdynHandle *h, vdyn::RolloutArgs<T> a, const char *who)
{
    if (!h) return VDYN_ERR_ARG;
    if (a.n < 0 || a.H < 0 || (a.k != 2 && a.k != 12) ||
        (a.layout != VDYN_CTRL_PER_ROLLOUT && a.layout != VDYN_CTRL_SHARED) ||
        (a.layout == VDYN_CTRL_SHARED && a.P <= 0) || (a.traj && a.traj_stride <= 0))
        return rollout_dev<T>(h, a, h->stream, who);  // reports the precise argument error
    if (a.n == 0) return VDYN_OK;
    if (!a.state0 || !a.terminal || (a.H > 0 && !a.ctrl) || (a.layout == VDYN_CTRL_SHARED && !a.path_id))
        return h->fail(VDYN_ERR_ARG, std::string(who) + ": null buffer");
    const size_t e = sizeof(T) * (size_t)a.n;
    const size_t ctrl_bytes = a.layout == VDYN_CTRL_PER_ROLLOUT
                                  ? (size_t)a.H * a.k * e
                                  : sizeof(T) * (size_t)a.P * a.H * a.k;
    // large controls on the way in (per-rollout layout) or a large trajectory on the way out (either layout): in chunks
    const bool big_in = a.layout == VDYN_CTRL_PER_ROLLOUT && ctrl_bytes >= kPipelineMinBytes;
    const bool big_out = a.traj && (size_t)(a.H / a.traj_stride) * 12 * e >= kPipelineMinBytes;
    if ((big_in || big_out) && a.H >= 2 && !a.state_dot && !a.outputs && (!a.traj || a.traj_stride <= a.H / 2) &&
        std::isfinite(a.dt) && !(h->state_rows != 12 && sizeof(T) != 4) && !(a.traj && !row_writer_fits<T>(a.n, 12)))
        return rollout_host_pipelined<T>(h, a, who);
    Stage s(h);
    const size_t rows = (size_t)h->state_rows;
    const size_t i0 = s.in(a.state0, rows * e), i1 = s.in(a.ctrl, ctrl_bytes),
                 i2 = s.in(a.path_id, a.layout == VDYN_CTRL_SHARED ? sizeof(int32_t) * (size_t)a.n : 0);
    const size_t o0 = s.out(a.terminal, rows * e),
                 o1 = s.out(a.traj, a.traj ? (size_t)(a.H / a.traj_stride) * 12 * e : 0),
                 o2 = s.out(a.state_dot, 10 * e), o3 = s.out(a.outputs, 18 * e);
    int rc = s.upload();
    if (rc) return rc;
    a.state0 = s.dev<T>(i0, false);
    a.ctrl = s.dev<T>(i1, false);
    a.path_id = s.dev<int>(i2, false);
    a.terminal = s.dev<T>(o0, true);
    a.traj = s.dev<T>(o1, true);
    a.state_dot = s.dev<T>(o2, true);
    a.outputs = s.dev<T>(o3, true);
    rc = rollout_dev<T>(h, a, h->stream, who);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int rollout_spiral_host(VdynHandle *h, int64_t n, int32_t H, const T *state0, const T *spiral, double wheelbase,
                        double max_steer, double torque, double dt, const double *mu4, T *terminal, T *traj,
                        int32_t traj_stride)
{
    if (!h) return VDYN_ERR_ARG;
    if (n <= 0 || H < 0 || !state0 || !spiral || !terminal || (traj && traj_stride <= 0))
        return rollout_spiral_dev<T>(h, n, H, state0, spiral, wheelbase, max_steer, torque, dt, mu4, terminal, traj,
                                     traj_stride, h->stream);      // n == 0, or reports the precise argument error
    const size_t e = sizeof(T) * (size_t)n;
    Stage s(h);
    const size_t i0 = s.in(state0, 12 * e), i1 = s.in(spiral, 3 * e);
    const size_t o0 = s.out(terminal, 12 * e), o1 = s.out(traj, traj ? (size_t)(H / traj_stride) * 12 * e : 0);
    int rc = s.upload();
    if (rc) return rc;
    rc = rollout_spiral_dev<T>(h, n, H, s.dev<T>(i0, false), s.dev<T>(i1, false), wheelbase, max_steer, torque, dt, mu4,
                               s.dev<T>(o0, true), s.dev<T>(o1, true), traj_stride, h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int nonfinite_host(VdynHandle *h, int32_t rows, int64_t n, const T *x, int32_t *status, int64_t *count)
{
    if (!h) return VDYN_ERR_ARG;
    if (rows <= 0 || n <= 0 || !x || !status) return nonfinite_dev<T>(h, rows, n, x, status, count, h->stream);
    Stage s(h);
    const size_t i0 = s.in(x, sizeof(T) * (size_t)rows * (size_t)n);
    const size_t o0 = s.out(status, sizeof(int32_t) * (size_t)n);
    int rc = s.upload();
    if (rc) return rc;
    rc = nonfinite_dev<T>(h, rows, n, s.dev<T>(i0, false), s.dev<int32_t>(o0, true), count, h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int fastmath_host(VdynHandle *h, int32_t fn, int64_t n, const T *x, double c, T *out0, T *out1)
{
    if (!h) return VDYN_ERR_ARG;
    if (n <= 0 || !x || !out0) return fastmath_dev<T>(h, fn, n, x, c, out0, out1, h->stream);
    Stage s(h);
    const size_t i0 = s.in(x, sizeof(T) * (size_t)n);
    const size_t o0 = s.out(out0, sizeof(T) * (size_t)n), o1 = s.out(out1, sizeof(T) * (size_t)n);
    int rc = s.upload();
    if (rc) return rc;
    rc = fastmath_dev<T>(h, fn, n, s.dev<T>(i0, false), c, s.dev<T>(o0, true), s.dev<T>(o1, true), h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int mpc_host(VdynHandle *h, int E, int C, int H, const T *ego, const T *cand, const T *goal, double dt,
             double w_delta, T *best_cost, int32_t *best_idx, T *cost_all)
{
    if (!h) return VDYN_ERR_ARG;
    if (E < 0 || C <= 0 || H < 0) return h->fail(VDYN_ERR_ARG, "mpc_argmin: need E >= 0, C > 0, H >= 0");
    if (E == 0) return VDYN_OK;
    if (!ego || !goal || !best_cost || !best_idx || (H > 0 && !cand))
        return h->fail(VDYN_ERR_ARG, "mpc_argmin: null buffer");
    Stage s(h);
    const size_t i0 = s.in(ego, sizeof(T) * 12 * (size_t)E), i1 = s.in(cand, sizeof(T) * 2 * (size_t)H * C),
                 i2 = s.in(goal, sizeof(T) * 2 * (size_t)E);
    const size_t o0 = s.out(best_cost, sizeof(T) * (size_t)E), o1 = s.out(best_idx, sizeof(int32_t) * (size_t)E),
                 o2 = s.out(cost_all, sizeof(T) * (size_t)E * C);
    int rc = s.upload();
    if (rc) return rc;
    rc = mpc_dev<T>(h, E, C, H, s.dev<T>(i0, false), s.dev<T>(i1, false), s.dev<T>(i2, false), dt, w_delta,
                    s.dev<T>(o0, true), s.dev<int32_t>(o1, true), s.dev<T>(o2, true), h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int closed_loop_check(VdynHandle *h, const VdynCtrlGains *g, const vdyn::ClosedLoopArgs<T> &a, bool update_only,
                      const char *who)
{
    const std::string w(who);
    if (!g) return h->fail(VDYN_ERR_ARG, w + ": null gains");
    if (a.n < 0 || a.H < 0) return h->fail(VDYN_ERR_ARG, w + ": n < 0 or H < 0");
    if (a.ctrl_every <= 0 || a.phase < 0) return h->fail(VDYN_ERR_ARG, w + ": need ctrl_every > 0, phase >= 0");
    if (a.P <= 0 || a.Wmax <= 0) return h->fail(VDYN_ERR_ARG, w + ": need P > 0 and Wmax > 0");
    if (!std::isfinite(a.dt)) return h->fail(VDYN_ERR_ARG, w + ": dt is not finite");
    if (a.n == 0) return VDYN_OK;
    if (!a.state0 || !a.cstate0 || !a.wp || !a.wcount || !a.path_id || !a.cstate)
        return h->fail(VDYN_ERR_ARG, w + ": null buffer");
    if (update_only ? !a.ctrl_out : !a.terminal) return h->fail(VDYN_ERR_ARG, w + ": null output buffer");
    if ((a.datalog && !row_writer_fits<T>(a.n, 45)) || (a.log && !row_writer_fits<T>(a.n, 16)))
        return h->fail(VDYN_ERR_ARG, w + ": a log row (16 n values) / DataLog row (45 n values) must not exceed 2^31 bytes; "
                                         "split the batch");
    return VDYN_OK;
}

template <typename T>
int closed_loop_dev(VdynHandle *h, const VdynCtrlGains *g, const vdyn::ClosedLoopArgs<T> &a, bool update_only,
                    void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    const char *who = update_only ? "controller_update" : "closed_loop";
    int rc = closed_loop_check<T>(h, g, a, update_only, who);
    if (rc || a.n == 0) return rc;
    VDYN_HIP(h, hipSetDevice(h->device));
    // tables too large for LDS: segment lengths + bounding circles in a handle-owned device buffer
    // (grown on demand; hipFree waits for the launches still reading the old one)
    vdyn::ClosedLoopArgs<T> b = a;
    const size_t aux_bytes = vdyn::closed_loop_aux_bytes<T>(a.P, a.Wmax, update_only);
    if (aux_bytes > h->d_aux_bytes) {
        if (h->d_aux) { (void)hipFree(h->d_aux); h->d_aux = nullptr; h->d_aux_bytes = 0; }
        if (hipMalloc(&h->d_aux, aux_bytes) != hipSuccess) {
            (void)hipGetLastError();
            h->d_aux = nullptr;
            return h->fail(VDYN_ERR_OOM, std::string(who) + ": auxiliary waypoint tables do not fit");
        } else {
            h->d_aux_bytes = aux_bytes;
        }
    }
    b.aux = aux_bytes > 0 && h->d_aux_bytes >= aux_bytes ? static_cast<T *>(h->d_aux) : nullptr;
    if (update_only) VDYN_HIP(h, vdyn::launch_controller_update<T>(*g, b, (hipStream_t)stream));
    else VDYN_HIP(h, vdyn::launch_closed_loop<T>(h->p, *g, b, (hipStream_t)stream));
    return VDYN_OK;
}

template <typename T>
int closed_loop_host(VdynHandle *h, const VdynCtrlGains *g, vdyn::ClosedLoopArgs<T> a, bool update_only)
{
    if (!h) return VDYN_ERR_ARG;
    const char *who = update_only ? "controller_update" : "closed_loop";
    int rc = closed_loop_check<T>(h, g, a, update_only, who);
    if (rc || a.n == 0) return rc;
    const size_t e = sizeof(T) * (size_t)a.n;
    const size_t row_bytes = (a.log ? 16 * e : 0) + (a.datalog ? 45 * e : 0);
    if (!update_only && row_bytes > 0 && a.ctrl_every >= 1 && (size_t)a.H * row_bytes >= kPipelineMinBytes) {
        // chunks of whole controller periods, 2 .. 16 MB of log rows each; at least two of them
        const size_t want = std::min<size_t>(std::max<size_t>((size_t)a.H * row_bytes / 12, 2u << 20), 16u << 20);
        const int64_t ce = a.ctrl_every;
        const int64_t hc = std::max<int64_t>(ce, (int64_t)(want / row_bytes) / ce * ce);
        if (hc < a.H) return closed_loop_host_pipelined<T>(h, g, a, hc);
    }
    Stage s(h);
    const size_t i0 = s.in(a.state0, 12 * e), i1 = s.in(a.cstate0, 6 * e),
                 i2 = s.in(a.wp, sizeof(T) * (size_t)a.P * a.Wmax * 2),
                 i3 = s.in(a.wcount, sizeof(int32_t) * (size_t)a.P), i4 = s.in(a.path_id, sizeof(int32_t) * (size_t)a.n);
    const size_t o0 = s.out(a.terminal, 12 * e), o1 = s.out(a.cstate, 6 * e),
                 o2 = s.out(a.log, a.log ? (size_t)a.H * 16 * e : 0), o3 = s.out(a.ctrl_out, 3 * e),
                 o4 = s.out(a.datalog, a.datalog ? (size_t)a.H * 45 * e : 0);
    rc = s.upload();
    if (rc) return rc;
    a.state0 = s.dev<T>(i0, false);
    a.cstate0 = s.dev<T>(i1, false);
    a.wp = s.dev<T>(i2, false);
    a.wcount = s.dev<int>(i3, false);
    a.path_id = s.dev<int>(i4, false);
    a.terminal = s.dev<T>(o0, true);
    a.cstate = s.dev<T>(o1, true);
    a.log = s.dev<T>(o2, true);
    a.ctrl_out = s.dev<T>(o3, true);
    a.datalog = s.dev<T>(o4, true);
    rc = closed_loop_dev<T>(h, g, a, update_only, h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
vdyn::ClosedLoopArgs<T> cl_args(int64_t n, int32_t H, int32_t ctrl_every, int32_t phase, const T *state0,
                                const T *cstate_in, const T *wp, int32_t Wmax, const int32_t *wcount,
                                const int32_t *path_id, int32_t P, double dt, T *terminal, T *cstate_out, T *log,
                                T *ctrl_out, T *datalog = nullptr)
{
    vdyn::ClosedLoopArgs<T> a;
    a.n = n; a.H = H; a.ctrl_every = ctrl_every; a.phase = phase; a.state0 = state0; a.cstate0 = cstate_in;
    a.wp = wp; a.Wmax = Wmax; a.wcount = wcount; a.path_id = path_id; a.P = P; a.dt = dt;
    a.terminal = terminal; a.cstate = cstate_out; a.log = log; a.ctrl_out = ctrl_out; a.datalog = datalog;
    return a;
}

template <typename T>
int select_dev(VdynHandle *h, const vdyn::SelectArgs<T> &a, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (a.E < 0 || a.P <= 0 || a.L <= 0 || a.M < 0) return h->fail(VDYN_ERR_ARG, "select_best_path: bad E / P / L / M");
    if (a.P > 64) return h->fail(VDYN_ERR_ARG, "select_best_path: at most 64 paths per ego");
    if (a.nc <= 0 || a.nc > 8 || !a.offsets || !a.radii)
        return h->fail(VDYN_ERR_ARG, "select_best_path: need 1..8 circles (host offsets / radii)");
    if (a.E == 0) return VDYN_OK;
    if (!a.x || !a.y || !a.yaw || !a.goal || !a.collision_free || !a.best_idx || !a.best_score || (a.M > 0 && !a.obst))
        return h->fail(VDYN_ERR_ARG, "select_best_path: null buffer");
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, vdyn::launch_select_best_path<T>(a, (hipStream_t)stream));
    return VDYN_OK;
}

template <typename T>
int select_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, const T *paths, const T *obst, int32_t M,
                int32_t obst_per_ego, const double *offsets, const double *radii, int32_t nc, const T *goal,
                double weight, const int32_t *collision_in, const int32_t *validity, int32_t *collision_free,
                int32_t *best_idx, T *best_score)
{
    if (!h) return VDYN_ERR_ARG;
    vdyn::SelectArgs<T> a;
    a.E = E; a.P = P; a.L = L; a.M = M; a.offsets = offsets; a.radii = radii; a.nc = nc; a.weight = weight;
    a.ego_stride = (int64_t)P * 3 * L; a.path_stride = 3 * (int64_t)L; a.point_stride = 1;
    a.obst_ego_stride = obst_per_ego ? 2 * (int64_t)M : 0;
    if (E <= 0 || P <= 0 || L <= 0 || M < 0 || !paths || !goal || !collision_free || !best_idx || !best_score ||
        (M > 0 && !obst)) {
        if (E == 0) return VDYN_OK;
        return h->fail(VDYN_ERR_ARG, "select_best_path: bad size or null buffer");
    }
    Stage s(h);
    const size_t i0 = s.in(paths, sizeof(T) * (size_t)E * P * 3 * L),
                 i1 = s.in(obst, sizeof(T) * 2 * (size_t)M * (obst_per_ego ? E : 1)),
                 i2 = s.in(goal, sizeof(T) * 2 * (size_t)E),
                 i3 = s.in(collision_in, sizeof(int32_t) * (size_t)E * P),
                 i4 = s.in(validity, sizeof(int32_t) * (size_t)E * P);
    const size_t o0 = s.out(collision_free, sizeof(int32_t) * (size_t)E * P),
                 o1 = s.out(best_idx, sizeof(int32_t) * (size_t)E), o2 = s.out(best_score, sizeof(T) * (size_t)E);
    int rc = s.upload();
    if (rc) return rc;
    const T *base = s.dev<T>(i0, false);
    a.x = base; a.y = base + L; a.yaw = base + 2 * (int64_t)L;
    a.obst = s.dev<T>(i1, false);
    a.goal = s.dev<T>(i2, false);
    a.collision_in = s.dev<int>(i3, false);
    a.validity = s.dev<int>(i4, false);
    a.collision_free = s.dev<int>(o0, true);
    a.best_idx = s.dev<int>(o1, true);
    a.best_score = s.dev<T>(o2, true);
    rc = select_dev<T>(h, a, h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
int lattice_dev(VdynHandle *h, const vdyn::LatticeArgs<T> &a, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (a.E < 0 || a.P <= 0 || a.nwp < 2) return h->fail(VDYN_ERR_ARG, "plan_lattice: need E >= 0, P > 0, nwp >= 2");
    if (!std::isfinite(a.goal_v) || !std::isfinite(a.lookahead) || !std::isfinite(a.path_offset))
        return h->fail(VDYN_ERR_ARG, "plan_lattice: non-finite scalar");
    if (a.E == 0) return VDYN_OK;
    if (!a.px || !a.py || !a.ego || !a.closest_idx || !a.goal_idx || !a.goal_set || !a.params || !a.paths ||
        !a.validity || !a.cost)
        return h->fail(VDYN_ERR_ARG, "plan_lattice: null buffer");
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, vdyn::launch_plan_lattice<T>(a, (hipStream_t)stream));
    return VDYN_OK;
}

template <typename T>
int lattice_host(VdynHandle *h, vdyn::LatticeArgs<T> a)
{
    if (!h) return VDYN_ERR_ARG;
    if (a.E <= 0 || a.P <= 0 || a.nwp < 2 || !a.px || !a.py || !a.ego)
        return a.E == 0 ? VDYN_OK : lattice_dev<T>(h, a, h->stream);
    const size_t EP = (size_t)a.E * a.P;
    Stage s(h);
    const size_t i0 = s.in(a.px, sizeof(T) * (size_t)a.nwp), i1 = s.in(a.py, sizeof(T) * (size_t)a.nwp),
                 i2 = s.in(a.ego, sizeof(T) * 3 * (size_t)a.E), i3 = s.in(a.params_in, sizeof(T) * 3 * EP);
    // closest_len is needed on the device even when the caller does not want it back
    T dummy = 0;
    const bool want_len = a.closest_len != nullptr;
    const size_t o0 = s.out(a.closest_idx, sizeof(int32_t) * (size_t)a.E),
                 o1 = s.out(a.goal_idx, sizeof(int32_t) * (size_t)a.E),
                 o2 = s.out(want_len ? a.closest_len : &dummy, sizeof(T) * (want_len ? (size_t)a.E : 1)),
                 o3 = s.out(a.goal_set, sizeof(T) * 4 * EP), o4 = s.out(a.params, sizeof(T) * 3 * EP),
                 o5 = s.out(a.paths, sizeof(T) * 3 * 49 * EP), o6 = s.out(a.validity, sizeof(int32_t) * EP),
                 o7 = s.out(a.cost, sizeof(T) * EP);
    int rc = s.upload();
    if (rc) return rc;
    a.px = s.dev<T>(i0, false); a.py = s.dev<T>(i1, false); a.ego = s.dev<T>(i2, false);
    a.params_in = s.dev<T>(i3, false);
    a.closest_idx = s.dev<int>(o0, true); a.goal_idx = s.dev<int>(o1, true);
    a.closest_len = want_len ? s.dev<T>(o2, true) : nullptr;
    a.goal_set = s.dev<T>(o3, true); a.params = s.dev<T>(o4, true); a.paths = s.dev<T>(o5, true);
    a.validity = s.dev<int>(o6, true); a.cost = s.dev<T>(o7, true);
    rc = lattice_dev<T>(h, a, h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
vdyn::LatticeArgs<T> lattice_args(int32_t E, const T *px, const T *py, int32_t nwp, const T *ego, double goal_v,
                                  double lookahead, int32_t P, double path_offset, const T *params_in,
                                  int32_t *closest_idx, int32_t *goal_idx, T *closest_len, T *goal_set, T *params,
                                  T *paths, int32_t *validity, T *cost)
{
    vdyn::LatticeArgs<T> a;
    a.E = E; a.P = P; a.nwp = nwp; a.px = px; a.py = py; a.ego = ego; a.goal_v = goal_v; a.lookahead = lookahead;
    a.path_offset = path_offset; a.params_in = params_in; a.closest_idx = closest_idx; a.goal_idx = goal_idx;
    a.closest_len = closest_len; a.goal_set = goal_set; a.params = params; a.paths = paths; a.validity = validity;
    a.cost = cost;
    return a;
}

template <typename T>
int interp_dev(VdynHandle *h, int32_t E, int32_t P, int32_t L, const T *paths, const int32_t *best_idx, double res,
               int32_t Wmax, T *wp_out, int32_t *wcount, void *stream)
{
    if (!h) return VDYN_ERR_ARG;
    if (E < 0 || P <= 0 || L < 2 || Wmax < 2 || !(res > 0))
        return h->fail(VDYN_ERR_ARG, "interpolate_waypoints: need E >= 0, P > 0, L >= 2, Wmax >= 2, res > 0");
    if (E == 0) return VDYN_OK;
    if (!paths || !best_idx || !wp_out || !wcount) return h->fail(VDYN_ERR_ARG, "interpolate_waypoints: null buffer");
    VDYN_HIP(h, hipSetDevice(h->device));
    VDYN_HIP(h, vdyn::launch_interpolate_waypoints<T>(E, P, L, paths, best_idx, res, Wmax, wp_out, wcount,
                                                      (hipStream_t)stream));
    return VDYN_OK;
}

template <typename T>
int interp_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, const T *paths, const int32_t *best_idx, double res,
                int32_t Wmax, T *wp_out, int32_t *wcount)
{
    if (!h) return VDYN_ERR_ARG;
    if (E <= 0 || P <= 0 || L < 2 || Wmax < 2 || !paths || !best_idx || !wp_out || !wcount)
        return E == 0 ? VDYN_OK : interp_dev<T>(h, E, P, L, paths, best_idx, res, Wmax, wp_out, wcount, h->stream);
    Stage s(h);
    const size_t wpb = sizeof(T) * (size_t)E * Wmax * 2, wcb = sizeof(int32_t) * (size_t)E;
    const size_t i0 = s.in(paths, sizeof(T) * (size_t)E * P * 3 * L), i1 = s.in(best_idx, sizeof(int32_t) * (size_t)E),
                 i2 = s.in(wp_out, wpb), i3 = s.in(wcount, wcb);      // IN / OUT: egos without a path keep their table
    const size_t o0 = s.out(wp_out, wpb), o1 = s.out(wcount, wcb);
    int rc = s.upload();
    if (rc) return rc;
    VDYN_HIP(h, hipMemcpyAsync(s.dev<T>(o0, true), s.dev<T>(i2, false), wpb, hipMemcpyDeviceToDevice, h->stream));
    VDYN_HIP(h, hipMemcpyAsync(s.dev<int>(o1, true), s.dev<int>(i3, false), wcb, hipMemcpyDeviceToDevice, h->stream));
    rc = interp_dev<T>(h, E, P, L, s.dev<T>(i0, false), s.dev<int>(i1, false), res, Wmax, s.dev<T>(o0, true),
                       s.dev<int>(o1, true), h->stream);
    if (rc) return rc;
    return s.download();
}

template <typename T>
vdyn::RolloutArgs<T> step_args(int64_t n, const T *state_in, const T *ctrl, int k, double dt,
                               const double *mu4, T *state_out, T *state_dot, T *outputs)
{
    vdyn::RolloutArgs<T> a;
    a.n = n; a.H = 1; a.state0 = state_in; a.ctrl = ctrl; a.k = k; a.layout = VDYN_CTRL_PER_ROLLOUT;
    a.dt = dt; a.mu4 = mu4; a.terminal = state_out; a.state_dot = state_dot; a.outputs = outputs;
    return a;
}

template <typename T>
vdyn::RolloutArgs<T> rollout_args(int64_t n, int32_t H, const T *state0, const T *ctrl, int k, int layout,
                                  const int32_t *path_id, int32_t P, double dt, const double *mu4,
                                  T *terminal, T *traj, int32_t traj_stride)
{
    vdyn::RolloutArgs<T> a;
    a.n = n; a.H = H; a.state0 = state0; a.ctrl = ctrl; a.k = k; a.layout = layout; a.path_id = path_id;
    a.P = P; a.dt = dt; a.mu4 = mu4; a.terminal = terminal; a.traj = traj; a.traj_stride = traj_stride;
    return a;
}

}  // namespace

// ------------------------------------------------------------------ extern "C" --------

#define VDYN_DEFINE_ABI(S, T)                                                                            \
    extern "C" int vdyn_planar_model_##S##_dev(VdynHandle *h, int64_t n, const T *state, const T *ctrl12, \
                                               const T *acc_prev, T *state_dot, T *aux, T *outputs,     \
                                               T *acc, void *stream)                                     \
    {                                                                                                    \
        return planar_model_dev<T>(h, n, state, ctrl12, acc_prev, state_dot, aux, outputs, acc, stream); \
    }                                                                                                    \
    extern "C" int vdyn_planar_model_##S##_host(VdynHandle *h, int64_t n, const T *state,               \
                                                const T *ctrl12, const T *acc_prev, T *state_dot,       \
                                                T *aux, T *outputs, T *acc)                              \
    {                                                                                                    \
        return planar_model_host<T>(h, n, state, ctrl12, acc_prev, state_dot, aux, outputs, acc);        \
    }                                                                                                    \
    extern "C" int vdyn_step_##S##_dev(VdynHandle *h, int64_t n, const T *state_in, const T *ctrl, int k, \
                                       double dt, const double *mu4, T *state_out, T *state_dot,        \
                                       T *outputs, void *stream)                                         \
    {                                                                                                    \
        return rollout_dev<T>(h, step_args<T>(n, state_in, ctrl, k, dt, mu4, state_out, state_dot, outputs), \
                              stream, "step");                                                           \
    }                                                                                                    \
    extern "C" int vdyn_step_##S##_host(VdynHandle *h, int64_t n, const T *state_in, const T *ctrl, int k, \
                                        double dt, const double *mu4, T *state_out, T *state_dot,       \
                                        T *outputs)                                                      \
    {                                                                                                    \
        return rollout_host<T>(h, step_args<T>(n, state_in, ctrl, k, dt, mu4, state_out, state_dot, outputs), \
                               "step");                                                                  \
    }                                                                                                    \
    extern "C" int vdyn_rollout_##S##_dev(VdynHandle *h, int64_t n, int32_t H, const T *state0,         \
                                          const T *ctrl, int k, int layout, const int32_t *path_id,     \
                                          int32_t P, double dt, const double *mu4, T *terminal, T *traj, \
                                          int32_t traj_stride, void *stream)                             \
    {                                                                                                    \
        return rollout_dev<T>(h, rollout_args<T>(n, H, state0, ctrl, k, layout, path_id, P, dt, mu4,    \
                                                 terminal, traj, traj_stride), stream, "rollout");       \
    }                                                                                                    \
    extern "C" int vdyn_rollout_##S##_host(VdynHandle *h, int64_t n, int32_t H, const T *state0,        \
                                           const T *ctrl, int k, int layout, const int32_t *path_id,    \
                                           int32_t P, double dt, const double *mu4, T *terminal,        \
                                           T *traj, int32_t traj_stride)                                 \
    {                                                                                                    \
        return rollout_host<T>(h, rollout_args<T>(n, H, state0, ctrl, k, layout, path_id, P, dt, mu4,   \
                                                  terminal, traj, traj_stride), "rollout");              \
    }                                                                                                    \
    extern "C" int vdyn_mpc_argmin_##S##_dev(VdynHandle *h, int32_t E, int32_t C, int32_t H, const T *ego, \
                                             const T *cand, const T *goal, double dt, double w_delta,   \
                                             T *best_cost, int32_t *best_idx, T *cost_all, void *stream) \
    {                                                                                                    \
        return mpc_dev<T>(h, E, C, H, ego, cand, goal, dt, w_delta, best_cost, best_idx, cost_all, stream); \
    }                                                                                                    \
    extern "C" int vdyn_mpc_argmin_##S##_host(VdynHandle *h, int32_t E, int32_t C, int32_t H,           \
                                              const T *ego, const T *cand, const T *goal, double dt,    \
                                              double w_delta, T *best_cost, int32_t *best_idx,          \
                                              T *cost_all)                                               \
    {                                                                                                    \
        return mpc_host<T>(h, E, C, H, ego, cand, goal, dt, w_delta, best_cost, best_idx, cost_all);     \
    }

#define VDYN_DEFINE_CTRL_ABI(S, T)                                                                       \
    extern "C" int vdyn_controller_update_##S##_dev(                                                     \
        VdynHandle *h, const VdynCtrlGains *g, int64_t n, const T *state12, const T *cstate_in, const T *wp, \
        int32_t Wmax, const int32_t *wcount, const int32_t *path_id, int32_t P, double dt, T *cstate_out, \
        T *out, void *stream)                                                                            \
    {                                                                                                    \
        return closed_loop_dev<T>(h, g, cl_args<T>(n, 0, 1, 0, state12, cstate_in, wp, Wmax, wcount, path_id, \
                                                   P, dt, nullptr, cstate_out, nullptr, out), true, stream); \
    }                                                                                                    \
    extern "C" int vdyn_controller_update_##S##_host(                                                    \
        VdynHandle *h, const VdynCtrlGains *g, int64_t n, const T *state12, const T *cstate_in, const T *wp, \
        int32_t Wmax, const int32_t *wcount, const int32_t *path_id, int32_t P, double dt, T *cstate_out, \
        T *out)                                                                                          \
    {                                                                                                    \
        return closed_loop_host<T>(h, g, cl_args<T>(n, 0, 1, 0, state12, cstate_in, wp, Wmax, wcount, path_id, \
                                                    P, dt, nullptr, cstate_out, nullptr, out), true);    \
    }                                                                                                    \
    extern "C" int vdyn_closed_loop_##S##_dev(                                                           \
        VdynHandle *h, const VdynCtrlGains *g, int64_t n, int32_t H, int32_t ctrl_every, int32_t phase,  \
        const T *state0, const T *cstate_in, const T *wp, int32_t Wmax, const int32_t *wcount,           \
        const int32_t *path_id, int32_t P, double dt, T *terminal, T *cstate_out, T *log, T *datalog,    \
        void *stream)                                                                                    \
    {                                                                                                    \
        return closed_loop_dev<T>(h, g, cl_args<T>(n, H, ctrl_every, phase, state0, cstate_in, wp, Wmax, wcount, \
                                                   path_id, P, dt, terminal, cstate_out, log, nullptr,   \
                                                   datalog), false, stream);                             \
    }                                                                                                    \
    extern "C" int vdyn_closed_loop_##S##_host(                                                          \
        VdynHandle *h, const VdynCtrlGains *g, int64_t n, int32_t H, int32_t ctrl_every, int32_t phase,  \
        const T *state0, const T *cstate_in, const T *wp, int32_t Wmax, const int32_t *wcount,           \
        const int32_t *path_id, int32_t P, double dt, T *terminal, T *cstate_out, T *log, T *datalog)    \
    {                                                                                                    \
        return closed_loop_host<T>(h, g, cl_args<T>(n, H, ctrl_every, phase, state0, cstate_in, wp, Wmax, wcount, \
                                                    path_id, P, dt, terminal, cstate_out, log, nullptr,  \
                                                    datalog), false);                                    \
    }

#define VDYN_DEFINE_SELECT_ABI(S, T)                                                                     \
    extern "C" int vdyn_select_best_path_##S##_dev(                                                      \
        VdynHandle *h, int32_t E, int32_t P, int32_t L, const T *x, const T *y, const T *yaw,            \
        int64_t ego_stride, int64_t path_stride, int64_t point_stride, const T *obst, int32_t M,         \
        int32_t obst_per_ego, const double *circle_offsets, const double *circle_radii, int32_t nc,      \
        const T *goal, double weight, const int32_t *collision_in, const int32_t *validity,              \
        int32_t *collision_free, int32_t *best_idx, T *best_score, void *stream)                         \
    {                                                                                                    \
        vdyn::SelectArgs<T> a;                                                                           \
        a.E = E; a.P = P; a.L = L; a.x = x; a.y = y; a.yaw = yaw; a.ego_stride = ego_stride;             \
        a.path_stride = path_stride; a.point_stride = point_stride; a.obst = obst; a.M = M;              \
        a.obst_ego_stride = obst_per_ego ? 2 * (int64_t)M : 0; a.offsets = circle_offsets;               \
        a.radii = circle_radii; a.nc = nc; a.goal = goal; a.weight = weight;                             \
        a.collision_in = collision_in; a.validity = validity; a.collision_free = collision_free;         \
        a.best_idx = best_idx; a.best_score = best_score;                                                \
        return select_dev<T>(h, a, stream);                                                              \
    }                                                                                                    \
    extern "C" int vdyn_select_best_path_##S##_host(                                                     \
        VdynHandle *h, int32_t E, int32_t P, int32_t L, const T *paths, const T *obst, int32_t M,        \
        int32_t obst_per_ego, const double *circle_offsets, const double *circle_radii, int32_t nc,      \
        const T *goal, double weight, const int32_t *collision_in, const int32_t *validity,              \
        int32_t *collision_free, int32_t *best_idx, T *best_score)                                       \
    {                                                                                                    \
        return select_host<T>(h, E, P, L, paths, obst, M, obst_per_ego, circle_offsets, circle_radii, nc, \
                              goal, weight, collision_in, validity, collision_free, best_idx, best_score); \
    }

#define VDYN_DEFINE_FLEET_ABI(S, T)                                                                      \
    extern "C" int vdyn_rollout_fleet_##S##_dev(                                                         \
        VdynHandle *h, int64_t n, int32_t H, const T *state0, const T *ctrl, int k, int layout,          \
        const int32_t *path_id, int32_t P, const VdynParams *classes, int32_t V, const int32_t *vehicle_id, \
        double dt, const double *mu4, T *terminal, T *traj, int32_t traj_stride, void *stream)           \
    {                                                                                                    \
        vdyn::RolloutArgs<T> a = rollout_args<T>(n, H, state0, ctrl, k, layout, path_id, P, dt, mu4, terminal, \
                                                 traj, traj_stride);                                     \
        a.V = V; a.vehicle_id = vehicle_id;                                                              \
        return rollout_fleet_dev<T>(h, a, classes, stream);                                              \
    }                                                                                                    \
    extern "C" int vdyn_rollout_fleet_##S##_host(                                                        \
        VdynHandle *h, int64_t n, int32_t H, const T *state0, const T *ctrl, int k, int layout,          \
        const int32_t *path_id, int32_t P, const VdynParams *classes, int32_t V, const int32_t *vehicle_id, \
        double dt, const double *mu4, T *terminal, T *traj, int32_t traj_stride)                         \
    {                                                                                                    \
        vdyn::RolloutArgs<T> a = rollout_args<T>(n, H, state0, ctrl, k, layout, path_id, P, dt, mu4, terminal, \
                                                 traj, traj_stride);                                     \
        a.V = V; a.vehicle_id = vehicle_id;                                                              \
        return rollout_fleet_host<T>(h, a, classes);                                                     \
    }

#define VDYN_DEFINE_LATTICE_ABI(S, T)                                                                    \
    extern "C" int vdyn_plan_lattice_##S##_dev(                                                          \
        VdynHandle *h, int32_t E, const T *px, const T *py, int32_t nwp, const T *ego, double goal_v,    \
        double lookahead, int32_t P, double path_offset, const T *params_in, int32_t *closest_idx,       \
        int32_t *goal_idx, T *closest_len, T *goal_set, T *params, T *paths, int32_t *validity, T *cost, \
        void *stream)                                                                                    \
    {                                                                                                    \
        return lattice_dev<T>(h, lattice_args<T>(E, px, py, nwp, ego, goal_v, lookahead, P, path_offset, \
                                                 params_in, closest_idx, goal_idx, closest_len, goal_set, \
                                                 params, paths, validity, cost), stream);                \
    }                                                                                                    \
    extern "C" int vdyn_plan_lattice_##S##_host(                                                         \
        VdynHandle *h, int32_t E, const T *px, const T *py, int32_t nwp, const T *ego, double goal_v,    \
        double lookahead, int32_t P, double path_offset, const T *params_in, int32_t *closest_idx,       \
        int32_t *goal_idx, T *closest_len, T *goal_set, T *params, T *paths, int32_t *validity, T *cost) \
    {                                                                                                    \
        return lattice_host<T>(h, lattice_args<T>(E, px, py, nwp, ego, goal_v, lookahead, P, path_offset, \
                                                  params_in, closest_idx, goal_idx, closest_len, goal_set, \
                                                  params, paths, validity, cost));                       \
    }                                                                                                    \
    extern "C" int vdyn_interpolate_waypoints_##S##_dev(VdynHandle *h, int32_t E, int32_t P, int32_t L,  \
                                                        const T *paths, const int32_t *best_idx,         \
                                                        double res, int32_t Wmax, T *wp_out,             \
                                                        int32_t *wcount, void *stream)                   \
    {                                                                                                    \
        return interp_dev<T>(h, E, P, L, paths, best_idx, res, Wmax, wp_out, wcount, stream);            \
    }                                                                                                    \
    extern "C" int vdyn_interpolate_waypoints_##S##_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, \
                                                         const T *paths, const int32_t *best_idx,        \
                                                         double res, int32_t Wmax, T *wp_out,            \
                                                         int32_t *wcount)                                \
    {                                                                                                    \
        return interp_host<T>(h, E, P, L, paths, best_idx, res, Wmax, wp_out, wcount);                   \
    }

#define VDYN_DEFINE_SPIRAL_ABI(S, T)                                                                     \
    extern "C" int vdyn_rollout_spiral_##S##_dev(VdynHandle *h, int64_t n, int32_t H, const T *state0,   \
                                                 const T *spiral, double wheelbase, double max_steer,    \
                                                 double torque, double dt, const double *mu4,            \
                                                 T *terminal, T *traj, int32_t traj_stride, void *stream) \
    {                                                                                                    \
        return rollout_spiral_dev<T>(h, n, H, state0, spiral, wheelbase, max_steer, torque, dt, mu4, terminal, \
                                     traj, traj_stride, stream);                                         \
    }                                                                                                    \
    extern "C" int vdyn_rollout_spiral_##S##_host(VdynHandle *h, int64_t n, int32_t H, const T *state0,  \
                                                  const T *spiral, double wheelbase, double max_steer,   \
                                                  double torque, double dt, const double *mu4,           \
                                                  T *terminal, T *traj, int32_t traj_stride)             \
    {                                                                                                    \
        return rollout_spiral_host<T>(h, n, H, state0, spiral, wheelbase, max_steer, torque, dt, mu4, terminal, \
                                      traj, traj_stride);                                                \
    }

#define VDYN_DEFINE_DIAG_ABI(S, T)                                                                       \
    extern "C" int vdyn_nonfinite_lanes_##S##_dev(VdynHandle *h, int32_t rows, int64_t n, const T *x,    \
                                                  int32_t *status, int64_t *count, void *stream)         \
    {                                                                                                    \
        return nonfinite_dev<T>(h, rows, n, x, status, count, stream);                                   \
    }                                                                                                    \
    extern "C" int vdyn_nonfinite_lanes_##S##_host(VdynHandle *h, int32_t rows, int64_t n, const T *x,   \
                                                   int32_t *status, int64_t *count)                      \
    {                                                                                                    \
        return nonfinite_host<T>(h, rows, n, x, status, count);                                          \
    }                                                                                                    \
    extern "C" int vdyn_fastmath_eval_##S##_dev(VdynHandle *h, int32_t fn, int64_t n, const T *x, double c, \
                                                T *out0, T *out1, void *stream)                          \
    {                                                                                                    \
        return fastmath_dev<T>(h, fn, n, x, c, out0, out1, stream);                                      \
    }                                                                                                    \
    extern "C" int vdyn_fastmath_eval_##S##_host(VdynHandle *h, int32_t fn, int64_t n, const T *x, double c, \
                                                 T *out0, T *out1)                                       \
    {                                                                                                    \
        return fastmath_host<T>(h, fn, n, x, c, out0, out1);                                             \
    }

extern "C" int vdyn_tire_fit_f32(double C, float *coef)
{
    if (coef == nullptr) return VDYN_ERR_ARG;
    return vdyn::tire_fit_coefficients(C, coef) ? VDYN_OK : VDYN_ERR_ARG;   // no handle: nothing to hang a message on
}
extern "C" int vdyn_tire_fit_f64(double C, double *coef)
{
    if (coef == nullptr) return VDYN_ERR_ARG;
    return vdyn::tire_fit_coefficients64(C, coef) ? VDYN_OK : VDYN_ERR_ARG;
}

VDYN_DEFINE_ABI(f32, float)
VDYN_DEFINE_ABI(f64, double)
VDYN_DEFINE_DIAG_ABI(f32, float)
VDYN_DEFINE_DIAG_ABI(f64, double)
VDYN_DEFINE_SPIRAL_ABI(f32, float)
VDYN_DEFINE_SPIRAL_ABI(f64, double)
VDYN_DEFINE_LATTICE_ABI(f32, float)
VDYN_DEFINE_LATTICE_ABI(f64, double)
VDYN_DEFINE_FLEET_ABI(f32, float)
VDYN_DEFINE_FLEET_ABI(f64, double)
VDYN_DEFINE_SELECT_ABI(f32, float)
VDYN_DEFINE_SELECT_ABI(f64, double)
VDYN_DEFINE_CTRL_ABI(f32, float)
VDYN_DEFINE_CTRL_ABI(f64, double)
