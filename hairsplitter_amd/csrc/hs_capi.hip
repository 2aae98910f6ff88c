// hs_capi.hip -- the C ABI of include/hairsplitter_hip.h: device memory, kernel launches and the two
// stage drivers. There is no CPU fallback anywhere in this file: without a usable HIP device every entry
// point fails with HS_ENODEVICE.
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <sys/prctl.h>
#include <condition_variable>
#include <functional>
#include <thread>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hs_host.h"
#include "hs_host_sr.h"
#include "hs_driver.h"
#include "hs_kernels.hip"
#include "hs_kernels_graph.hip"
#include "hs_kernels_cw.hip"
#include "hs_kernels_myers.hip"
#include "hs_kernels_cols.hip"
#include "hs_kernels_loopa.hip"

namespace hs {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace hs

using hs::set_error;

#define HS_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            set_error(std::string(#call) + ": " + hipGetErrorString(e__));                               \
            return e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice ? HS_ENODEVICE : HS_EHIP;     \
        }                                                                                                \
    } while (0)

namespace {

// Size-class pool of device / pinned-host blocks: the stage drivers allocate dozens of temporaries per call and
// hipMalloc/hipFree (which synchronises) would dominate small batches. Blocks are kept for the life of the process.
// set when a host wait gave up on the device (stream_wait_impl): kernels may still be writing the blocks of the failed call, so
// from then on nothing is handed back to the pools (the blocks leak; the process is expected to report the error and end)
static std::atomic<bool> g_device_lost{false};
struct BlockPool {
    std::mutex mu;
    std::vector<std::pair<size_t, void*>> free_dev, free_host;
    static size_t round_up(size_t n) { size_t c = 4096; while (c < n) c <<= 1; return c; }
    int get(bool host, size_t n, void** out, size_t* cap) {
        const size_t c = round_up(n);
        {
            std::lock_guard<std::mutex> g(mu);
            auto& fl = host ? free_host : free_dev;
            for (size_t i = 0; i < fl.size(); ++i)
                if (fl[i].first == c) { *out = fl[i].second; *cap = c; fl[i] = fl.back(); fl.pop_back(); return HS_OK; }
        }
        if (host) HS_HIP(hipHostMalloc(out, c, hipHostMallocDefault)); else HS_HIP(hipMalloc(out, c));
        *cap = c;
        return HS_OK;
    }
    void put(bool host, void* p, size_t cap) {
        if (g_device_lost.load(std::memory_order_relaxed)) return;
        std::lock_guard<std::mutex> g(mu);
        (host ? free_host : free_dev).push_back(std::make_pair(cap, p));
    }
};
// one pool per device: a block belongs to the device (and, for pinned host blocks, is mapped for the device) that was current
// when it was allocated; the host threads of a shard stay on one device (hs_set_device / the sharded stage calls below)
BlockPool& pool() {
    static BlockPool* pools[64] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    dev &= 63;
    BlockPool* p = __atomic_load_n(&pools[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        std::lock_guard<std::mutex> g(mu);
        if (!pools[dev]) __atomic_store_n(&pools[dev], new BlockPool(), __ATOMIC_RELEASE);
        p = pools[dev];
    }
    return *p;
}

// HS_EXIT_PROBE=1 (diagnostic, called by the drop-in executables before they leave): gives the pooled blocks back and resets the
// device, timing every step -- what the process would otherwise leave to the kernel's teardown after _exit
extern "C" void hs_teardown_probe(void) {
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    BlockPool& p = pool();
    size_t db = 0, hb = 0, dn = 0, hn = 0;
    const double t0 = now();
    for (auto& b : p.free_dev) { db += b.first; dn++; (void)hipFree(b.second); }
    const double t1 = now();
    for (auto& b : p.free_host) { hb += b.first; hn++; (void)hipHostFree(b.second); }
    const double t2 = now();
    (void)hipDeviceReset();
    const double t3 = now();
    std::fprintf(stderr, "[hs exit probe] pooled device blocks: %zu, %.1f MB, hipFree %.1f ms; pooled pinned blocks: %zu, %.1f MB, hipHostFree %.1f ms; hipDeviceReset %.1f ms\n",
                 dn, db / 1e6, t1 - t0, hn, hb / 1e6, t2 - t1, t3 - t2);
}

// RAII device buffer (pooled). A block goes back to the pool it came from (`owner`), whatever device is current on the thread
// that releases it.
struct DBuf {
    void* p = nullptr;
    size_t bytes = 0, cap = 0;
    bool view = false;             // points into an UploadPack: not owned
    BlockPool* owner = nullptr;
    ~DBuf() { release(); }
    void release() { if (p && !view && owner) owner->put(false, p, cap); p = nullptr; view = false; }
    int alloc(size_t n) {
        release();
        bytes = n;
        owner = &pool();
        return owner->get(false, n ? n : 16, &p, &cap);
    }
    template <class T> int upload(const std::vector<T>& v);
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// pinned host buffer (pooled) for large downloads
struct HBuf {
    void* p = nullptr;
    size_t cap = 0;
    BlockPool* owner = nullptr;
    ~HBuf() { release(); }
    void release() { if (p && owner) owner->put(true, p, cap); p = nullptr; }
    int alloc(size_t n) {
        release();
        owner = &pool();
        return owner->get(true, n ? n : 16, &p, &cap);
    }
};

template <class T> int DBuf::upload(const std::vector<T>& v) {
    int rc = alloc(v.size() * sizeof(T));
    if (rc) return rc;
    const size_t n = v.size() * sizeof(T);
    if (n == 0) return HS_OK;
    if (n >= (128u << 10)) {     // large: stage through pinned memory (pageable sources get pinned page by page otherwise)
        HBuf h;
        if (int r2 = h.alloc(n)) return r2;
        std::memcpy(h.p, v.data(), n);
        HS_HIP(hipMemcpy(p, h.p, n, hipMemcpyHostToDevice));
    } else {
        HS_HIP(hipMemcpy(p, v.data(), n, hipMemcpyHostToDevice));
    }
    return HS_OK;
}

// Many small host arrays -> ONE pinned staging buffer -> ONE asynchronous host-to-device copy. Every hipMemcpy of a small
// pageable array costs 10-20 us of latency on the calling thread (a blit kernel each); a stage call used to issue a dozen.
// The DBufs handed to add() become views into the pack's device block; the pack (and its pinned buffer) must outlive the copy:
// keep it in the scope that ends with a synchronising call, or as a member.
// HS_COPY_STATS=1: how often every hipMemcpyAsync call site ran, printed when the library is unloaded (diagnostic for the copy count)
static std::atomic<long> g_copy_sites[4096];
static const bool g_copy_stats = [] {
    const bool on = std::getenv("HS_COPY_STATS") != nullptr;
    if (on) std::atexit([] { for (int i = 0; i < 4096; ++i) { const long n = g_copy_sites[i].load(); if (n) std::fprintf(stderr, "[hs copies] hs_capi.hip:%d %ld\n", i, n); } });
    return on;
}();
#define HS_COPY_ASYNC(...) ((g_copy_stats ? (void)g_copy_sites[__LINE__ & 4095].fetch_add(1, std::memory_order_relaxed) : (void)0), hipMemcpyAsync(__VA_ARGS__))

static int ship_kernel(void* dst, const void* src, size_t bytes, hipStream_t stream);      // (hsdev::k_ship, below the kernels)
struct UploadPack {
    struct Item { const void* src; size_t bytes, off; DBuf* dst; };
    std::vector<Item> items;
    size_t total = 0;
    DBuf dev;
    HBuf host;
    template <class T> void add(const std::vector<T>& v, DBuf& dst) {
        items.push_back(Item{v.data(), v.size() * sizeof(T), total, &dst});
        total = (total + v.size() * sizeof(T) + 255) & ~(size_t)255;
    }
    template <class T> void add(const hs::ArrayView<T>& v, DBuf& dst) {
        items.push_back(Item{v.data(), v.size() * sizeof(T), total, &dst});
        total = (total + v.size() * sizeof(T) + 255) & ~(size_t)255;
    }
    int commit(hipStream_t stream) {
        if (int rc = dev.alloc(total ? total : 256)) return rc;
        if (int rc = host.alloc(total ? total : 256)) return rc;
        for (const Item& it : items) {
            if (it.bytes) std::memcpy((char*)host.p + it.off, it.src, it.bytes);
            it.dst->release();
            it.dst->p = (char*)dev.p + it.off; it.dst->bytes = it.bytes; it.dst->cap = 0; it.dst->view = true;
        }
        if (total) { if (int rc = ship_kernel(dev.p, host.p, total, stream)) return rc; }
        items.clear(); total = 0;
        return HS_OK;
    }
};

// Host waits = polling hipStreamQuery with a 25-us sleep between two looks (HS_WAIT_SLEEP_US; the polling thread's timer slack set
// to 1 us, HS_TIMER_SLACK_NS, so that 25 us are 25 us and not 75): about 1 % of a core instead of 100 %, the host learns of the
// end of the work 30 us late -- which the other contig groups cover. No events, no interrupts, no device-wide
// scheduling flag. HS_SPIN_WAIT=1 polls back to back instead (measured on the 16-core box, 500-contig job: the same step time,
// 60-80 CPU-ms more per step; it only pays when a single chain owns the device). A wait that has lasted HS_WAIT_TIMEOUT_S
// seconds (default 1800, 0 = no limit) returns an error instead of hanging the caller for ever; the blocks of that call are
// then leaked, not recycled (g_device_lost), because the device may still be writing them.
static bool blocking_wait() { static const bool spin = std::getenv("HS_SPIN_WAIT") != nullptr && std::getenv("HS_BLOCKING_WAIT") == nullptr; return !spin; }
static bool spin_wait() { return !blocking_wait(); }
static std::atomic<long> g_waits{0}, g_wait_us{0};     // HS_TIMING: host waits and the wall time spent in them
static int stream_wait_impl(hipStream_t s);
static int stream_wait(hipStream_t s) {
    static const bool timed = std::getenv("HS_TIMING") != nullptr;
    if (!timed) { g_waits.fetch_add(1, std::memory_order_relaxed); return stream_wait_impl(s); }
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = stream_wait_impl(s);
    g_waits.fetch_add(1); g_wait_us.fetch_add((long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}
static long wait_timeout_s() {
    static const long v = []() { const char* e = std::getenv("HS_WAIT_TIMEOUT_S"); return e ? std::atol(e) : 1800l; }();
    return v;
}
static int stream_wait_impl(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    const long limit = wait_timeout_s();
    for (unsigned long looks = 1;; ++looks) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return HS_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); set_error(std::string("hipStreamQuery: ") + hipGetErrorString(e)); return HS_EHIP; }
        (void)hipGetLastError();      // (hipErrorNotReady is sticky otherwise)
        if (blocking_wait()) {
            static const long sleep_ns = []() { const char* e = std::getenv("HS_WAIT_SLEEP_US"); return e ? std::atol(e) * 1000l : 25000l; }();
            // the kernel's default timer slack adds up to 50 us to every sleep: a polling thread asks for 1 us (its own setting only)
            static const long slack_ns = []() { const char* e = std::getenv("HS_TIMER_SLACK_NS"); return e ? std::atol(e) : 1000l; }();
            static thread_local bool slack_set = false;
            if (slack_ns > 0 && !slack_set) { prctl(PR_SET_TIMERSLACK, (unsigned long)slack_ns); slack_set = true; }
            struct timespec ts = {0, sleep_ns}; nanosleep(&ts, nullptr);
        }
        else { for (int i = 0; i < 16; ++i) __builtin_ia32_pause(); }
        if (limit > 0 && (looks & 4095ul) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limit)) {
            g_device_lost.store(true);
            set_error("the device did not finish the queued work within " + std::to_string(limit) + " s (HS_WAIT_TIMEOUT_S)");
            return HS_EHIP;
        }
    }
}
static int copy_d2h(void* h, const void* d, size_t n, hipStream_t s) {
    if (n) HS_HIP(HS_COPY_ASYNC(h, d, n, hipMemcpyDeviceToHost, s));
    return stream_wait(s);
}

// Timing events are recycled per host thread: creating and destroying a few hundred (interrupt-backed) events per step makes
// the runtime stall for tens of milliseconds every few dozen steps.
// Device -> host into pageable memory through a pooled pinned buffer. A direct copy makes the runtime pin the destination on
// the fly (a "userptr" mapping); when that memory is later freed the kernel driver has to evict and restore the process's
// GPU queues to drop the mapping -- tens of milliseconds during which nothing runs.
static int d2h_pinned(void* dst, const void* d, size_t n, hipStream_t s) {
    if (!n) return stream_wait(s);
    HBuf h;
    if (int rc = h.alloc(n)) return rc;
    if (int rc = copy_d2h(h.p, d, n, s)) return rc;
    std::memcpy(dst, h.p, n);
    return HS_OK;
}

static int ship_kernel(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    hsdev::ShipList L; L.n = 1;
    L.seg[0].src = src; L.seg[0].dst = dst; L.seg[0].bytes = (long long)bytes; L.seg[0].count = nullptr; L.seg[0].stride = 0; L.seg[0].cap = 0; L.seg[0].extra = 0;
    hipLaunchKernelGGL(hsdev::k_ship, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(256, bytes / 4096 + 1))), dim3(256), 0, stream, L);
    HS_HIP(hipGetLastError());
    return HS_OK;
}
// a transfer queued as a kernel (hsdev::k_ship: pinned host memory is mapped into the device's address space); segments whose length
// is a count on the device take it from there when the kernel runs
struct Shipment {
    hsdev::ShipList L;
    Shipment() { L.n = 0; }
    void add(void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        hsdev::ShipSeg& g = L.seg[L.n++];
        g.src = src; g.dst = dst; g.bytes = (long long)bytes; g.count = nullptr; g.stride = 0; g.cap = 0; g.extra = 0;
    }
    void add_counted(void* dst, const void* src, const long long* d_count, size_t stride, long long cap, size_t extra = 0) {
        hsdev::ShipSeg& g = L.seg[L.n++];
        g.src = src; g.dst = dst; g.bytes = 0; g.count = d_count; g.stride = (long long)stride; g.cap = cap; g.extra = (long long)extra;
    }
    int launch(hipStream_t s, int blocks = 256) {
        if (L.n == 0) return HS_OK;
        hipLaunchKernelGGL(hsdev::k_ship, dim3((unsigned)blocks), dim3(256), 0, s, L);
        HS_HIP(hipGetLastError());
        L.n = 0;
        return HS_OK;
    }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    static std::vector<hipEvent_t>& cache() { static thread_local std::vector<hipEvent_t> c; return c; }
    ~EventPair() { if (a) cache().push_back(a); if (b) cache().push_back(b); }
    static int get(hipEvent_t* e) {
        std::vector<hipEvent_t>& c = cache();
        if (!c.empty()) { *e = c.back(); c.pop_back(); return HS_OK; }
        HS_HIP(hipEventCreateWithFlags(e, hipEventDefault));
        return HS_OK;
    }
    int init() { if (int rc = get(&a)) return rc; return get(&b); }
    // (called after the stream has been waited for: the events are complete and no synchronising call -- an ioctl in the
    // runtime -- is needed; one that is not yet complete is waited for the slow way)
    int ms(float* out) {
        hipError_t e = hipEventElapsedTime(out, a, b);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); HS_HIP(hipEventSynchronize(b)); e = hipEventElapsedTime(out, a, b); }
        HS_HIP(e);
        return HS_OK;
    }
};

// Per-kernel accounting (include/hairsplitter_hip.h: hs_kernel_stats). A KernelClock belongs to one host thread / stream;
// begin() / end() bracket a launch (or a short run of launches of one family) with events, flush() -- called once the stream
// has been waited for -- turns them into durations and adds them to the process-wide table.
struct KernelTable {
    std::mutex mu;
    hs_kernel_stats st;
    KernelTable() { std::memset(&st, 0, sizeof st); }
};
KernelTable& kernel_table() { static KernelTable* t = new KernelTable(); return *t; }
struct KernelClock {
    struct Item { int k; hipEvent_t a, b; int64_t bytes; };
    std::vector<Item> items;
    hipEvent_t open_a = nullptr;
    int open_k = -1;
    static bool enabled() { static const bool on = std::getenv("HS_NO_KERNEL_STATS") == nullptr; return on; }
    int begin(int k, hipStream_t s) {
        if (!enabled()) return HS_OK;
        if (int rc = EventPair::get(&open_a)) return rc;
        open_k = k;
        HS_HIP(hipEventRecord(open_a, s));
        return HS_OK;
    }
    int end(int64_t bytes, hipStream_t s) {
        if (!enabled()) return HS_OK;
        hipEvent_t b = nullptr;
        if (int rc = EventPair::get(&b)) return rc;
        HS_HIP(hipEventRecord(b, s));
        items.push_back(Item{open_k, open_a, b, bytes});
        open_a = nullptr; open_k = -1;
        return HS_OK;
    }
    // for launches whose closing event is recorded by the callee: the item is queued, *b is recorded by the caller's callee
    int end_prepare(int64_t bytes, hipEvent_t* b) {
        if (!enabled()) { *b = nullptr; return HS_OK; }
        if (int rc = EventPair::get(b)) return rc;
        items.push_back(Item{open_k, open_a, *b, bytes});
        open_a = nullptr; open_k = -1;
        return HS_OK;
    }
    static void add_bytes(int k, int64_t bytes) { KernelTable& t = kernel_table(); std::lock_guard<std::mutex> g(t.mu); t.st.bytes[k] += bytes; }
    void flush() {
        if (items.empty()) return;
        KernelTable& t = kernel_table();
        std::lock_guard<std::mutex> g(t.mu);
        for (Item& it : items) {
            float ms = 0;
            hipError_t e = hipEventElapsedTime(&ms, it.a, it.b);
            if (e == hipErrorNotReady) { (void)hipGetLastError(); e = hipEventSynchronize(it.b) == hipSuccess ? hipEventElapsedTime(&ms, it.a, it.b) : hipErrorUnknown; }
            if (e == hipSuccess) {
                t.st.ms[it.k] += ms; t.st.launches[it.k] += 1; t.st.bytes[it.k] += it.bytes;
            } else (void)hipGetLastError();
            EventPair::cache().push_back(it.a); EventPair::cache().push_back(it.b);
        }
        items.clear();
    }
    ~KernelClock() { flush(); if (open_a) EventPair::cache().push_back(open_a); }
};

static int host_threads() { return hs::host_threads(); }      // usable cores (hs_driver.cpp)

// K2 of one contig group at a time: the kernel fills the device on its own; eight of them side by side only slow each other
// down (0.29 ms alone, 1.0 ms each among eight) and every group would get its selection at the same late moment, whereas one
// after the other the groups' chains start 0.3 ms apart and their host work spreads out (44.5 -> 39.7 ms per C4 step). Taking
// turns for EVERY device phase of the groups was tried too: 52 ms per step -- the later phases are short, the lock then mostly
// adds its own queueing. HS_DEVICE_TURNS=0: no turns.
struct DeviceTurn {
    static std::mutex& mu() { static std::mutex m; return m; }
    static bool on() { static const bool v = []() { const char* e = std::getenv("HS_DEVICE_TURNS"); return !(e && e[0] == '0'); }(); return v; }
    std::unique_lock<std::mutex> lk;
    DeviceTurn() { if (on()) lk = std::unique_lock<std::mutex>(mu()); }
};

static void set_wait_policy() {}   // (the wait mode is no device state: see stream_wait)

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available: the HairSplitter MI355X path has no CPU fallback");
        return HS_ENODEVICE;
    }
    static std::once_flag once;
    std::call_once(once, set_wait_policy);
    return HS_OK;
}

// The HIP current device is a per-thread setting (0 on a new thread): every thread that works on a batch -- the caller's, the
// contig-group threads of a pipeline -- binds itself to the batch's device first
static int bind_device(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return HS_OK;
    HS_HIP(hipSetDevice(device));
    return HS_OK;
}

}  // namespace

extern "C" {

const char* hs_version(void) { return "hairsplitter_amd 0.2 (gfx950)"; }
const char* hs_kernel_name(int k) {
    static const char* names[HS_NKERNELS] = {"k_cigar_scan", "k_pileup_packed", "k_column_stats_tiled", "k_columns_compact", "k_gather_tiles", "k_column_top3_exact",
                                             "k_candidates_scan", "k_pack_flagged", "k_partition_transpose", "k_column_partition_lanes", "k_column_partition_test",
                                             "k_snp_flags", "k_window_masks", "k_snp_planes", "k_simdiff", "k_read_graph_rows", "k_read_graph_fill", "k_cw_visit_lists",
                                             "k_cw_seed_sets", "k_cw_seeded_lanes", "k_cw_seeded_rows", "k_window_tail", "k_cw_local", "k_loop_a", "other", "k_cand_bits", "k_ship"};
    return k >= 0 && k < HS_NKERNELS ? names[k] : "?";
}
void hs_kernel_stats_reset(void) { KernelTable& t = kernel_table(); std::lock_guard<std::mutex> g(t.mu); std::memset(&t.st, 0, sizeof t.st); }
void hs_kernel_stats_get(hs_kernel_stats* out) { if (!out) return; KernelTable& t = kernel_table(); std::lock_guard<std::mutex> g(t.mu); *out = t.st; }
const char* hs_last_error(void) { return hs::g_err.c_str(); }
// host waits for the device since the library was loaded (every one is a round trip of a contig group's chain); the wall time
// spent in them is only kept under HS_TIMING
void hs_host_wait_stats(int64_t* n_waits, double* ms_in_waits) { if (n_waits) *n_waits = g_waits.load(); if (ms_in_waits) *ms_in_waits = g_wait_us.load() / 1e3; }
int hs_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
static std::vector<int> device_list();
int hs_warmup(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    // every device the stage calls will use (HS_DEVICES / all visible), side by side: context + code object of each
    std::vector<int> devs = device_list();
    std::sort(devs.begin(), devs.end());
    devs.erase(std::unique(devs.begin(), devs.end()), devs.end());
    if (devs.empty()) devs.push_back(0);
    auto warm_one = [](int dev) {
        if (hipSetDevice(dev) != hipSuccess) return;
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return;
        hipLaunchKernelGGL(hsdev::k_swap_top2, dim3(1), dim3(64), 0, 0, (hsdev::hs_colstat_dev*)p, (const int64_t*)p, 0);   // loads the code object
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < devs.size(); ++i) th.emplace_back(warm_one, devs[i]);
    warm_one(devs[0]);
    for (auto& t : th) t.join();
    (void)hipSetDevice(devs[0]);
    return n;
}
int hs_set_device(int device) { HS_HIP(hipSetDevice(device)); set_wait_policy(); return HS_OK; }
int hs_device_synchronize(void) { HS_HIP(hipDeviceSynchronize()); return HS_OK; }
int hs_malloc(void** d_ptr, size_t bytes) { HS_HIP(hipMalloc(d_ptr, bytes ? bytes : 16)); return HS_OK; }
int hs_free(void* d_ptr) { HS_HIP(hipFree(d_ptr)); return HS_OK; }
int hs_memcpy_h2d(void* d, const void* h, size_t n) { if (n) HS_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return HS_OK; }
int hs_memcpy_d2h(void* h, const void* d, size_t n) { if (n) HS_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); return HS_OK; }
int hs_memset(void* d, int v, size_t n) { if (n) HS_HIP(hipMemset(d, v, n)); return HS_OK; }
int hs_event_create(void** ev) { hipEvent_t e; HS_HIP(hipEventCreate(&e)); *ev = (void*)e; return HS_OK; }
int hs_event_destroy(void* ev) { HS_HIP(hipEventDestroy((hipEvent_t)ev)); return HS_OK; }
int hs_event_record(void* ev, void* stream) { HS_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return HS_OK; }
int hs_event_elapsed_ms(void* a, void* b, float* ms) {
    HS_HIP(hipEventSynchronize((hipEvent_t)b));
    HS_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return HS_OK;
}

// ---------------------------------------------------------------------------------------------------
// kernel-level entry points
// ---------------------------------------------------------------------------------------------------
static int cigar_scan_launch(const int64_t* d_contig_off, const int32_t* d_rec_contig, const int32_t* d_rec_pos, const int64_t* d_rec_cig_off,
                             const uint32_t* d_cigar, const int64_t* d_rec_chunk_off, int32_t n_rec, int32_t* d_chunk_scratch, int32_t* d_rec_stats,
                             void* stream) {
    if (n_rec <= 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_cigar_scan, dim3((n_rec + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_contig_off, d_rec_contig,
                       d_rec_pos, d_rec_cig_off, d_cigar, d_rec_chunk_off, n_rec, d_chunk_scratch, d_rec_stats);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

// K1: the packed form over the task list, then the few records it leaves out (K0 flagged them) in the per-event form.
// HS_K1_PER_EVENT=1: the per-event form for everything (the kernel of rounds r01_a .. r01_j).
static int pileup_launch(const uint8_t* d_contig_seq, const int64_t* d_contig_off, const uint8_t* d_read_seq,
                         const int64_t* d_read_off, const int32_t* d_rec_read, const int32_t* d_rec_contig,
                         const int32_t* d_rec_pos, const uint8_t* d_rec_strand, const int64_t* d_rec_cig_off,
                         const uint32_t* d_cigar, const int64_t* d_pile_off, const int64_t* d_rec_chunk_off,
                         int32_t* d_chunk_scratch, const int32_t* d_task_rec, const int32_t* d_task_ev0, int32_t n_tasks,
                         int32_t ev_per_task, uint8_t* d_pile, int32_t* d_rec_stats, int32_t n_rec, void* stream) {
    if (n_tasks <= 0) return HS_OK;
    static const bool per_event = std::getenv("HS_K1_PER_EVENT") != nullptr;
    if (per_event) {
        hipLaunchKernelGGL(hsdev::k_pileup, dim3((n_tasks + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_contig_seq, d_contig_off,
                           d_read_seq, d_read_off, d_rec_read, d_rec_contig, d_rec_pos, d_rec_strand, d_rec_cig_off, d_cigar,
                           d_pile_off, d_rec_chunk_off, d_chunk_scratch, d_task_rec, d_task_ev0, n_tasks, ev_per_task, d_pile,
                           d_rec_stats);
    } else {
        hipLaunchKernelGGL(hsdev::k_pileup_packed, dim3((n_tasks + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_contig_seq, d_contig_off,
                           d_read_seq, d_read_off, d_rec_read, d_rec_contig, d_rec_pos, d_rec_strand, d_rec_cig_off, d_cigar,
                           d_pile_off, d_rec_chunk_off, d_chunk_scratch, d_task_rec, d_task_ev0, n_tasks, ev_per_task, d_pile,
                           d_rec_stats);
        const int blocks = std::max(1, std::min(256, (n_rec + 255) / 256));
        hipLaunchKernelGGL(hsdev::k_pileup_flagged_records, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_contig_seq, d_contig_off,
                           d_read_seq, d_read_off, d_rec_read, d_rec_contig, d_rec_pos, d_rec_strand, d_rec_cig_off, d_cigar,
                           d_pile_off, d_rec_chunk_off, d_chunk_scratch, n_rec, ev_per_task, d_pile, d_rec_stats);
    }
    HS_HIP(hipGetLastError());
    return HS_OK;
}

int hs_pileup(const uint8_t* d_contig_seq, const int64_t* d_contig_off, const uint8_t* d_read_seq,
              const int64_t* d_read_off, const int32_t* d_rec_read, const int32_t* d_rec_contig,
              const int32_t* d_rec_pos, const uint8_t* d_rec_strand, const int64_t* d_rec_cig_off,
              const uint32_t* d_cigar, const int64_t* d_pile_off, int32_t n_rec, const int64_t* d_rec_chunk_off,
              int32_t* d_chunk_scratch, const int32_t* d_task_rec, const int32_t* d_task_ev0, int32_t n_tasks,
              int32_t ev_per_task, uint8_t* d_pile, int32_t* d_rec_stats, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_rec <= 0) return HS_OK;
    if (int rc = cigar_scan_launch(d_contig_off, d_rec_contig, d_rec_pos, d_rec_cig_off, d_cigar, d_rec_chunk_off, n_rec, d_chunk_scratch,
                                   d_rec_stats, stream)) return rc;
    return pileup_launch(d_contig_seq, d_contig_off, d_read_seq, d_read_off, d_rec_read, d_rec_contig, d_rec_pos, d_rec_strand,
                         d_rec_cig_off, d_cigar, d_pile_off, d_rec_chunk_off, d_chunk_scratch, d_task_rec, d_task_ev0, n_tasks,
                         ev_per_task, d_pile, d_rec_stats, n_rec, stream);
}

int hs_pileup_plan(const int64_t* h_rec_cig_off, const uint32_t* h_cigar, int32_t n_rec, int32_t ev_per_task,
                   int64_t* h_rec_chunk_off, int32_t* n_tasks, int32_t** h_task_rec, int32_t** h_task_ev0) {
    if (ev_per_task <= 0 || !h_rec_chunk_off || !n_tasks || !h_task_rec || !h_task_ev0) { set_error("hs_pileup_plan: bad arguments"); return HS_EINVAL; }
    std::vector<int32_t> tr, te;
    h_rec_chunk_off[0] = 0;
    std::vector<int64_t> ev_of((size_t)std::max(n_rec, 0));
    {   // events per record: one pass over all CIGAR ops, blocks of records on the host threads
        const int nb = std::max(1, std::min(n_rec / 256 + 1, 256));
        hs::hs_parallel_for(nb, host_threads(), [&](int b) {
            const int r0 = (int)((int64_t)n_rec * b / nb), r1 = (int)((int64_t)n_rec * (b + 1) / nb);
            for (int r = r0; r < r1; ++r) {
                int64_t ev = 0;
                for (int64_t o = h_rec_cig_off[r]; o < h_rec_cig_off[r + 1]; ++o) {
                    const uint32_t op = h_cigar[o] & 15u;
                    if (op == 0 || op == 1 || op == 2 || op == 7 || op == 8) ev += h_cigar[o] >> 4;
                }
                ev_of[(size_t)r] = ev;
            }
        });
    }
    for (int r = 0; r < n_rec; ++r) {
        const int64_t ev = ev_of[(size_t)r];
        if (ev > 0x7fffffff) { set_error("alignment with more than 2^31 events"); return HS_EINVAL; }
        h_rec_chunk_off[r + 1] = h_rec_chunk_off[r] + (h_rec_cig_off[r + 1] - h_rec_cig_off[r] + 63) / 64;
        for (int64_t e = 0; e < ev; e += ev_per_task) { tr.push_back(r); te.push_back((int32_t)e); }
    }
    *n_tasks = (int32_t)tr.size();
    *h_task_rec = (int32_t*)std::malloc(std::max<size_t>(1, tr.size()) * sizeof(int32_t));
    *h_task_ev0 = (int32_t*)std::malloc(std::max<size_t>(1, te.size()) * sizeof(int32_t));
    if (!tr.empty()) { std::memcpy(*h_task_rec, tr.data(), tr.size() * sizeof(int32_t)); std::memcpy(*h_task_ev0, te.data(), te.size() * sizeof(int32_t)); }
    return HS_OK;
}
void hs_free_host(void* p) { std::free(p); }

// exclusive scan of n ints into n + 1 offsets (see hs_kernels_graph.hip); `scratch` must outlive the launches
static int exclusive_scan_launch(const int32_t* d_in, int n, int64_t* d_out, DBuf& scratch, hipStream_t stream) {
    const int n_tiles = (n + HS_SCAN_TILE - 1) / HS_SCAN_TILE;
    if (n_tiles == 0) { HS_HIP(hipMemsetAsync(d_out, 0, sizeof(int64_t), stream)); return HS_OK; }
    if (int rc = scratch.alloc((size_t)n_tiles * 16)) return rc;
    long long* tile_sum = scratch.as<long long>();
    long long* tile_off = tile_sum + n_tiles;
    hipLaunchKernelGGL(hsdev::k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(256), 0, stream, d_in, n, tile_sum);
    hipLaunchKernelGGL(hsdev::k_scan_tile_offsets, dim3(1), dim3(1024), 0, stream, tile_sum, n_tiles, tile_off, d_out + n);
    hipLaunchKernelGGL(hsdev::k_scan_apply, dim3((unsigned)n_tiles), dim3(256), 0, stream, d_in, n, tile_off, d_out);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

// K2's selection list: the kernels fill 256 scratch slots + one count per tile; scan + compact give the sorted list and its
// length (see column_stats_tail in hs_kernels.hip)
struct SelectionScratch {
    DBuf tile_cnt, tile_base, gpos, depth, scan_scratch;
    int64_t n_tiles = 0;
    int prepare(int64_t total_len) {
        n_tiles = (total_len + 255) / 256;
        if (int rc = tile_cnt.alloc((size_t)n_tiles * 4)) return rc;
        if (int rc = tile_base.alloc(((size_t)n_tiles + 1) * 8)) return rc;
        if (int rc = gpos.alloc((size_t)n_tiles * 256 * 8)) return rc;
        return depth.alloc((size_t)n_tiles * 256 * 4);
    }
    int finish(int32_t* d_sel_count, int64_t* d_sel_gpos, int32_t* d_sel_depth, int32_t sel_cap, hipStream_t stream) {
        if (n_tiles > 0x7fffffff) { set_error("too many tiles"); return HS_EINVAL; }
        if (int rc = exclusive_scan_launch(tile_cnt.as<int32_t>(), (int)n_tiles, tile_base.as<int64_t>(), scan_scratch, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_selection_compact, dim3((unsigned)n_tiles), dim3(256), 0, stream, tile_cnt.as<int32_t>(), tile_base.as<int64_t>(),
                           gpos.as<int64_t>(), depth.as<int32_t>(), n_tiles, d_sel_count, d_sel_gpos, d_sel_depth, sel_cap);
        HS_HIP(hipGetLastError());
        return HS_OK;
    }
};

static int column_stats_launch(const uint8_t* d_pile, const int64_t* d_pile_off, const int32_t* d_rec_pos,
                               const int32_t* d_rec_qend, const int32_t* d_contig_rec_off, const int64_t* d_contig_off,
                               int32_t n_contigs, int64_t total_len, hs_colstat* d_stats, int32_t min_second, int32_t* d_sel_count,
                               int64_t* d_sel_gpos, int32_t* d_sel_depth, int32_t sel_cap, int32_t max_depth, void* stream) {
    if (total_len <= 0) return HS_OK;
    const int64_t grid = (total_len + 255) / 256;
    const bool full = d_stats != nullptr;     // the stage driver passes no statistics buffer: selection only
    const bool narrow = max_depth > 0 && max_depth <= 255;
    using KernelT = void (*)(const uint8_t*, const int64_t*, const int32_t*, const int32_t*, const int32_t*, const int64_t*, int,
                             hsdev::hs_colstat_dev*, int, int32_t*, int64_t*, int32_t*, int);
    KernelT kernel = narrow ? (full ? (KernelT)hsdev::k_column_stats<1, true> : (KernelT)hsdev::k_column_stats<1, false>)
                            : (full ? (KernelT)hsdev::k_column_stats<2, true> : (KernelT)hsdev::k_column_stats<2, false>);
    SelectionScratch sc;
    if (d_sel_count) { if (int rc = sc.prepare(total_len)) return rc; }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, d_pile, d_pile_off, d_rec_pos, d_rec_qend,
                       d_contig_rec_off, d_contig_off, n_contigs, reinterpret_cast<hsdev::hs_colstat_dev*>(d_stats), min_second,
                       d_sel_count ? sc.tile_cnt.as<int32_t>() : nullptr, sc.gpos.as<int64_t>(), sc.depth.as<int32_t>(), sel_cap);
    HS_HIP(hipGetLastError());
    if (d_sel_count) {
        if (int rc = sc.finish(d_sel_count, d_sel_gpos, d_sel_depth, sel_cap, (hipStream_t)stream)) return rc;
        if (int rc_w = stream_wait((hipStream_t)stream)) return rc_w;   // the scratch goes back to the pool with this scope
    }
    return HS_OK;
}

int hs_column_stats(const uint8_t* d_pile, const int64_t* d_pile_off, const int32_t* d_rec_pos,
                    const int32_t* d_rec_qend, const int32_t* d_contig_rec_off, const int64_t* d_contig_off,
                    int32_t n_contigs, hs_colstat* d_stats, int32_t min_second, int32_t* d_sel_count, int64_t* d_sel_gpos,
                    int32_t* d_sel_depth, int32_t sel_cap, int32_t max_depth, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_contigs <= 0) return HS_OK;
    int64_t total = 0;
    HS_HIP(hipMemcpy(&total, d_contig_off + n_contigs, sizeof(int64_t), hipMemcpyDeviceToHost));
    return column_stats_launch(d_pile, d_pile_off, d_rec_pos, d_rec_qend, d_contig_rec_off, d_contig_off, n_contigs, total,
                               d_stats, min_second, d_sel_count, d_sel_gpos, d_sel_depth, sel_cap, max_depth, stream);
}

namespace {
// the tile plan of include/hairsplitter_hip.h from host vectors
void build_tile_plan(const int64_t* contig_off, int n_contigs, const int32_t* contig_rec_off, const int32_t* rec_pos, const int32_t* rec_qend,
                     const int64_t* pile_off, std::vector<int64_t>& tile_off, std::vector<hs_tile_entry>& ent, std::vector<int32_t>& rec) {
    const int64_t total = n_contigs > 0 ? contig_off[n_contigs] : 0;
    const int64_t n_tiles = (total + 255) / 256;
    tile_off.assign((size_t)n_tiles + 1, 0);
    for (int c = 0; c < n_contigs; ++c)
        for (int r = contig_rec_off[c]; r < contig_rec_off[c + 1]; ++r) {
            if (rec_qend[r] <= rec_pos[r]) continue;
            const int64_t gs = contig_off[c] + rec_pos[r], ge = contig_off[c] + rec_qend[r];
            for (int64_t t = gs >> 8; t <= (ge - 1) >> 8; ++t) tile_off[(size_t)t + 1]++;
        }
    for (int64_t t = 0; t < n_tiles; ++t) tile_off[(size_t)t + 1] += tile_off[(size_t)t];
    ent.resize((size_t)tile_off[(size_t)n_tiles]); rec.resize(ent.size());
    std::vector<int64_t> fill(tile_off.begin(), tile_off.end() - 1);
    for (int c = 0; c < n_contigs; ++c)
        for (int r = contig_rec_off[c]; r < contig_rec_off[c + 1]; ++r) {   // ascending record id inside every tile
            if (rec_qend[r] <= rec_pos[r]) continue;
            const int64_t gs = contig_off[c] + rec_pos[r], ge = contig_off[c] + rec_qend[r];
            for (int64_t t = gs >> 8; t <= (ge - 1) >> 8; ++t) {
                const int64_t k = fill[(size_t)t]++;
                const int32_t first = (int32_t)(gs - t * 256);
                ent[(size_t)k] = hs_tile_entry{first, (int32_t)(ge - gs), pile_off[r] - first};
                rec[(size_t)k] = r;
            }
        }
}
}  // namespace

int hs_tile_plan(const int64_t* h_contig_off, int32_t n_contigs, const int32_t* h_contig_rec_off, const int32_t* h_rec_pos,
                 const int32_t* h_rec_qend, const int64_t* h_pile_off, int64_t** tile_off, hs_tile_entry** tile_ent, int32_t** tile_rec,
                 int64_t* n_tiles) {
    if (!tile_off || !tile_ent || !tile_rec || !n_tiles || n_contigs < 0) { set_error("hs_tile_plan: bad arguments"); return HS_EINVAL; }
    std::vector<int64_t> to; std::vector<hs_tile_entry> en; std::vector<int32_t> rc;
    build_tile_plan(h_contig_off, n_contigs, h_contig_rec_off, h_rec_pos, h_rec_qend, h_pile_off, to, en, rc);
    *n_tiles = (int64_t)to.size() - 1;
    *tile_off = (int64_t*)std::malloc(to.size() * sizeof(int64_t));
    *tile_ent = (hs_tile_entry*)std::malloc((en.size() + 1) * sizeof(hs_tile_entry));
    *tile_rec = (int32_t*)std::malloc((rc.size() + 1) * sizeof(int32_t));
    if (!*tile_off || !*tile_ent || !*tile_rec) { set_error("hs_tile_plan: out of memory"); return HS_EINVAL; }
    std::memcpy(*tile_off, to.data(), to.size() * sizeof(int64_t));
    if (!en.empty()) { std::memcpy(*tile_ent, en.data(), en.size() * sizeof(hs_tile_entry)); std::memcpy(*tile_rec, rc.data(), rc.size() * sizeof(int32_t)); }
    return HS_OK;
}

// K2 on a tile plan; `sc` is caller-owned scratch (prepared); `after_main` (optional) is recorded right after the histogram
// kernel so that its duration can be told apart from the two small selection kernels
static int column_stats_tiled_launch(const uint8_t* d_pile, const int64_t* d_tile_off, const hs_tile_entry* d_tile_ent, int64_t total_len,
                                     hs_colstat* d_stats, int32_t min_second, int32_t* d_sel_count, int64_t* d_sel_gpos, int32_t* d_sel_depth,
                                     int32_t sel_cap, int32_t max_depth, SelectionScratch* sc, hipEvent_t after_main, hipStream_t stream,
                                     hipEvent_t after_main2 = nullptr, int64_t tile0 = 0, int64_t tile1 = -1 /* tiles [tile0, tile1) only */,
                                     int64_t g_lo = 0, int64_t g_hi = 0x7fffffffffffffffll, int32_t* d_tile_ent_sum = nullptr, bool compact = true,
                                     bool padded = false /* 256 readable bytes on both sides of the pileup */) {
    static_assert(sizeof(hs_tile_entry) == sizeof(int4), "hs_tile_entry is read as one 16-byte load");
    if (total_len <= 0) {   // nothing to count: an empty selection
        if (d_sel_count) HS_HIP(hipMemsetAsync(d_sel_count, 0, sizeof(int32_t), stream));
        if (after_main) HS_HIP(hipEventRecord(after_main, stream));
        if (after_main2) HS_HIP(hipEventRecord(after_main2, stream));
        return HS_OK;
    }
    const int64_t grid = tile1 >= 0 ? tile1 - tile0 : (total_len + 255) / 256;
    const bool full = d_stats != nullptr;
    const bool narrow = max_depth > 0 && max_depth <= 255;
    using KernelT = void (*)(const uint8_t*, const int64_t*, const int4*, int64_t, hsdev::hs_colstat_dev*, int, int32_t*, int64_t*, int32_t*, int, int64_t,
                             int64_t, int64_t, int32_t*);
    KernelT kernel = padded ? (narrow ? (full ? (KernelT)hsdev::k_column_stats_tiled<1, true, true> : (KernelT)hsdev::k_column_stats_tiled<1, false, true>)
                                      : (full ? (KernelT)hsdev::k_column_stats_tiled<2, true, true> : (KernelT)hsdev::k_column_stats_tiled<2, false, true>))
                            : (narrow ? (full ? (KernelT)hsdev::k_column_stats_tiled<1, true, false> : (KernelT)hsdev::k_column_stats_tiled<1, false, false>)
                                      : (full ? (KernelT)hsdev::k_column_stats_tiled<2, true, false> : (KernelT)hsdev::k_column_stats_tiled<2, false, false>));
    static const bool no_dw = std::getenv("HS_K2_BYTE_LOADS") != nullptr;      // (diagnostic: the one-byte-per-lane form for every launch)
    if (padded && narrow && !full && d_sel_count && !no_dw)      // the stage driver's launch: four positions per lane
        hipLaunchKernelGGL(hsdev::k_column_stats_tiled_dw, dim3((unsigned)grid), dim3(256), 0, stream, d_pile, d_tile_off, reinterpret_cast<const int4*>(d_tile_ent),
                           total_len, min_second, sc->tile_cnt.as<int32_t>(), sc->gpos.as<int64_t>(), sc->depth.as<int32_t>(), tile0, g_lo, g_hi, d_tile_ent_sum);
    else
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), 0, stream, d_pile, d_tile_off, reinterpret_cast<const int4*>(d_tile_ent),
                       total_len, reinterpret_cast<hsdev::hs_colstat_dev*>(d_stats), min_second, d_sel_count ? sc->tile_cnt.as<int32_t>() : nullptr,
                       d_sel_count ? sc->gpos.as<int64_t>() : nullptr, d_sel_count ? sc->depth.as<int32_t>() : nullptr, sel_cap, tile0, g_lo, g_hi,
                       d_tile_ent_sum);
    HS_HIP(hipGetLastError());
    if (after_main) HS_HIP(hipEventRecord(after_main, stream));
    if (after_main2) HS_HIP(hipEventRecord(after_main2, stream));
    if (d_sel_count && compact) return sc->finish(d_sel_count, d_sel_gpos, d_sel_depth, sel_cap, stream);
    return HS_OK;
}

int hs_column_stats_tiled(const uint8_t* d_pile, const int64_t* d_tile_off, const hs_tile_entry* d_tile_ent, int64_t total_len,
                          hs_colstat* d_stats, int32_t min_second, int32_t* d_sel_count, int64_t* d_sel_gpos, int32_t* d_sel_depth,
                          int32_t sel_cap, int32_t max_depth, void* stream) {
    if (int rc = require_device()) return rc;
    if (total_len <= 0) return HS_OK;
    SelectionScratch sc;
    if (d_sel_count) { if (int rc = sc.prepare(total_len)) return rc; }
    if (int rc = column_stats_tiled_launch(d_pile, d_tile_off, d_tile_ent, total_len, d_stats, min_second, d_sel_count, d_sel_gpos, d_sel_depth, sel_cap,
                                           max_depth, &sc, nullptr, (hipStream_t)stream)) return rc;
    if (d_sel_count) if (int rc_w = stream_wait((hipStream_t)stream)) return rc_w;   // the scratch goes back to the pool with this scope
    return HS_OK;
}

int hs_gather_columns_tiled(const uint8_t* d_pile, const int64_t* d_tile_off, const hs_tile_entry* d_tile_ent, const int32_t* d_tile_rec,
                            const int64_t* d_contig_off, const int32_t* d_contig_rec_off, const int32_t* d_sel_contig,
                            const int32_t* d_sel_pos, const int64_t* d_col_off, int32_t n_sel, int32_t* d_col_idx, uint8_t* d_col_code,
                            void* stream) {
    if (int rc = require_device()) return rc;
    if (n_sel <= 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_gather_columns_tiled, dim3((n_sel + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_pile, d_tile_off,
                       reinterpret_cast<const int4*>(d_tile_ent), d_tile_rec, d_contig_off, d_contig_rec_off, d_sel_contig, d_sel_pos, d_col_off,
                       n_sel, d_col_idx, d_col_code);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

int hs_gather_columns(const uint8_t* d_pile, const int64_t* d_pile_off, const int32_t* d_rec_pos,
                      const int32_t* d_rec_qend, const int32_t* d_contig_rec_off, const int32_t* d_sel_contig,
                      const int32_t* d_sel_pos, const int64_t* d_col_off, int32_t n_sel, int32_t* d_col_idx,
                      uint8_t* d_col_code, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_sel <= 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_gather_columns, dim3((n_sel + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_pile, d_pile_off,
                       d_rec_pos, d_rec_qend, d_contig_rec_off, d_sel_contig, d_sel_pos, d_col_off, n_sel, d_col_idx, d_col_code);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

int hs_column_top3(const int64_t* d_col_off, const uint8_t* d_col_code, int32_t n_cols, hs_coltop* d_out, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_cols <= 0) return HS_OK;
    static_assert(sizeof(hs_coltop) == sizeof(hsdev::hs_coltop_dev), "hs_coltop layout");
    hipLaunchKernelGGL(hsdev::k_column_top3, dim3((n_cols + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_col_off, d_col_code, n_cols,
                       reinterpret_cast<hsdev::hs_coltop_dev*>(d_out));
    HS_HIP(hipGetLastError());
    return HS_OK;
}

int hs_exclusive_scan_i32(const int32_t* d_in, int32_t n, int64_t* d_out, void* stream) {
    if (int rc = require_device()) return rc;
    if (n < 0) { set_error("hs_exclusive_scan_i32: negative length"); return HS_EINVAL; }
    DBuf scratch;
    if (int rc = exclusive_scan_launch(d_in, n, d_out, scratch, (hipStream_t)stream)) return rc;
    return stream_wait((hipStream_t)stream);   // the scratch goes back to the pool with this scope
}

int hs_pack_columns(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code, const int32_t* d_ids,
                    const int64_t* d_packed_off, int32_t n_ids, int32_t* d_out_idx, uint8_t* d_out_code, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_ids <= 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_pack_columns, dim3((n_ids + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_col_off, d_col_idx, d_col_code, d_ids,
                       d_packed_off, n_ids, d_out_idx, d_out_code);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

// K4 in two steps: the lanes-as-partitions kernel on a per-contig [read][partition] table (built on the device from the
// dense state arrays), then the exact one-partition-at-a-time kernel on the few columns the first one leaves undecided
static int partition_test_launch(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code, const int32_t* d_col_contig,
                                 const uint8_t* d_col_k0, const uint8_t* d_col_k1, const int32_t* d_col_c1, const uint8_t* d_col_is_cand, int32_t n_cols,
                                 DBuf& b_part_off, DBuf& b_part_state_off, DBuf& b_part_state /* (filled by pk.commit below) */,
                                 const int32_t* h_part_off, const int32_t* h_contig_n_reads, int32_t n_contigs, uint8_t* d_keep, hipStream_t stream,
                                 DBuf& d_tab, DBuf& d_tab_off, DBuf& d_ctg_n, DBuf& d_list, UploadPack& pk, KernelClock* kc = nullptr,
                                 int64_t col_entries = 0, int64_t state_bytes = 0) {
    std::vector<int64_t> tab_off((size_t)n_contigs + 1, 0);
    std::vector<int32_t> ctg_n(h_contig_n_reads, h_contig_n_reads + n_contigs);
    int64_t max_cells = 1;
    for (int c = 0; c < n_contigs; ++c) {
        const int P = h_part_off[c + 1] - h_part_off[c];
        const int64_t cells = (int64_t)ctg_n[(size_t)c] * ((P + 15) & ~15);
        tab_off[(size_t)c + 1] = tab_off[(size_t)c] + cells;
        max_cells = std::max(max_cells, cells);
    }
    pk.add(tab_off, d_tab_off); pk.add(ctg_n, d_ctg_n);
    if (int rc = pk.commit(stream)) return rc;
    const int32_t* d_part_off = b_part_off.as<int32_t>(); const int64_t* d_part_state_off = b_part_state_off.as<int64_t>(); const int8_t* d_part_state = b_part_state.as<int8_t>();
    if (int rc = d_tab.alloc(std::max<size_t>((size_t)tab_off.back(), 64))) return rc;
    if (tab_off.back() > 0) {
        const unsigned gx = (unsigned)std::min<int64_t>((max_cells + 255) / 256, 64);
        if (kc) { if (int rc = kc->begin(HS_K_PARTITION_TRANSPOSE, stream)) return rc; }
        hipLaunchKernelGGL(hsdev::k_partition_transpose, dim3(gx, (unsigned)n_contigs), dim3(256), 0, stream, d_part_off, d_part_state_off, d_part_state,
                           d_ctg_n.as<int32_t>(), d_tab_off.as<int64_t>(), n_contigs, d_tab.as<uint8_t>());
        if (kc) { if (int rc = kc->end(state_bytes + (int64_t)tab_off.back(), stream)) return rc; }      // the dense states in, the [read][partition] table out
    }
    if (int rc = d_list.alloc(((size_t)n_cols + 1) * 4)) return rc;     // [0] = number of undecided columns, then their indices
    hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, d_list.as<uint4>(), 1ll, 0u);
    if (kc) { if (int rc = kc->begin(HS_K_PARTITION_LANES, stream)) return rc; }
    hipLaunchKernelGGL(hsdev::k_column_partition_lanes, dim3((n_cols + 63) / 64), dim3(256), 0, stream, d_col_off, d_col_idx, d_col_code, d_col_contig,
                       d_col_k0, d_col_k1, d_col_c1, d_col_is_cand, n_cols, d_part_off, d_tab_off.as<int64_t>(), d_tab.as<uint8_t>(), d_keep,
                       d_list.as<int32_t>() + 1, d_list.as<int32_t>());
    if (kc) { if (int rc = kc->end(5 * col_entries + (int64_t)tab_off.back(), stream)) return rc; }      // the columns (idx + code) and the table, once each
    if (std::getenv("HS_K4_DEBUG")) {   // diagnostic: how many columns the first kernel leaves to the exact one
        int32_t nu = 0;
        if (int rc = d2h_pinned(&nu, d_list.p, 4, stream)) return rc;
        std::fprintf(stderr, "[hs k4] %d columns, %d undecided after the lanes kernel\n", n_cols, nu);
    }
    if (kc) { if (int rc = kc->begin(HS_K_PARTITION_TEST, stream)) return rc; }
    hipLaunchKernelGGL(hsdev::k_column_partition_test, dim3((unsigned)std::min(n_cols, 2048)), dim3(1024), 0, stream, d_col_off, d_col_idx,
                       d_col_code, d_col_contig, d_col_k0, d_col_k1, d_col_c1, d_col_is_cand, n_cols, d_part_off, d_part_state_off,
                       d_part_state, d_keep, d_list.as<int32_t>() + 1, d_list.as<int32_t>());
    HS_HIP(hipGetLastError());
    if (kc) { if (int rc = kc->end(state_bytes, stream)) return rc; }      // (the undecided columns against the dense states: a small share of them)
    return HS_OK;
}

int hs_partition_pair_distance(const int8_t* d_state, const int32_t* d_more, const int32_t* d_less, const int64_t* d_part_off, const int32_t* d_part_n,
                               const int32_t* d_pair_a, const int32_t* d_pair_b, int32_t n_pairs, int32_t threshold_p, const float* d_sigma3, int32_t* d_out,
                               void* stream) {
    if (int rc = require_device()) return rc;
    if (n_pairs <= 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_partition_pair_distance, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_state, d_more, d_less, d_part_off,
                       d_part_n, d_pair_a, d_pair_b, n_pairs, threshold_p, d_sigma3, d_out);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

int hs_column_partition_test(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code,
                             const int32_t* d_col_contig, const uint8_t* d_col_k0, const uint8_t* d_col_k1,
                             const int32_t* d_col_c1, const uint8_t* d_col_is_cand, int32_t n_cols,
                             const int32_t* d_part_off, const int64_t* d_part_state_off, const int8_t* d_part_state,
                             const int32_t* h_contig_n_reads, int32_t n_contigs, uint8_t* d_keep, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_cols <= 0) return HS_OK;
    if (!h_contig_n_reads || n_contigs <= 0) { set_error("hs_column_partition_test: the number of reads of every contig is needed"); return HS_EINVAL; }
    std::vector<int32_t> h_po((size_t)n_contigs + 1);
    if (int rc = d2h_pinned(h_po.data(), d_part_off, h_po.size() * 4, (hipStream_t)stream)) return rc;
    DBuf tab, tab_off, ctg_n, list, v_po, v_pso, v_ps;      // (the caller's device arrays as views)
    v_po.p = const_cast<int32_t*>(d_part_off); v_po.view = true; v_pso.p = const_cast<int64_t*>(d_part_state_off); v_pso.view = true; v_ps.p = const_cast<int8_t*>(d_part_state); v_ps.view = true;
    UploadPack pk;
    if (int rc = partition_test_launch(d_col_off, d_col_idx, d_col_code, d_col_contig, d_col_k0, d_col_k1, d_col_c1, d_col_is_cand, n_cols, v_po,
                                       v_pso, v_ps, h_po.data(), h_contig_n_reads, n_contigs, d_keep, (hipStream_t)stream, tab, tab_off, ctg_n, list, pk)) return rc;
    return stream_wait((hipStream_t)stream);   // the table goes back to the pool with this scope
}

// K5a: one workgroup per (contig, 4 words of its bit rows); the list of workgroups is made from the host copy of words[]
static void snp_planes_blocks(const int32_t* h_words, int n_contigs, std::vector<int32_t>& blk_c, std::vector<int32_t>& blk_w) {
    blk_c.clear(); blk_w.clear();
    for (int c = 0; c < n_contigs; ++c)
        for (int w0 = 0; w0 < h_words[c]; w0 += 4) { blk_c.push_back(c); blk_w.push_back(w0); }
}
static int snp_planes_launch(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code, const uint8_t* d_snp_ref,
                             const uint8_t* d_snp_alt, const int32_t* d_snp_contig, const int64_t* d_contig_snp_base, const int64_t* d_plane_off,
                             const int32_t* d_words, const int32_t* d_n_reads, const int32_t* d_blk_c, const int32_t* d_blk_w, size_t n_blocks,
                             int32_t n_snps, uint64_t* d_alt, uint64_t* d_ref, hipStream_t stream) {
    if (n_blocks == 0 || n_snps <= 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_snp_planes, dim3((unsigned)n_blocks), dim3(256), 0, stream, d_col_off, d_col_idx, d_col_code, d_snp_ref, d_snp_alt,
                       d_snp_contig, d_contig_snp_base, d_plane_off, d_words, d_n_reads, d_blk_c, d_blk_w, n_snps, (unsigned long long*)d_alt,
                       (unsigned long long*)d_ref);
    HS_HIP(hipGetLastError());
    return HS_OK;
}
int hs_snp_planes(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code, const uint8_t* d_snp_ref,
                  const uint8_t* d_snp_alt, const int32_t* d_snp_contig, const int64_t* d_contig_snp_base,
                  const int64_t* d_plane_off, const int32_t* d_words, const int32_t* d_n_reads, const int32_t* h_words, int32_t n_contigs,
                  int32_t n_snps, uint64_t* d_alt, uint64_t* d_ref, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_snps <= 0 || n_contigs <= 0) return HS_OK;
    if (!h_words) { set_error("hs_snp_planes: the host copy of words[] is needed"); return HS_EINVAL; }
    std::vector<int32_t> blk_c, blk_w;
    snp_planes_blocks(h_words, n_contigs, blk_c, blk_w);
    DBuf d_bc, d_bw;
    UploadPack pk;
    pk.add(blk_c, d_bc); pk.add(blk_w, d_bw);
    if (int rc = pk.commit((hipStream_t)stream)) return rc;
    if (int rc = snp_planes_launch(d_col_off, d_col_idx, d_col_code, d_snp_ref, d_snp_alt, d_snp_contig, d_contig_snp_base, d_plane_off, d_words, d_n_reads,
                                   d_bc.as<int32_t>(), d_bw.as<int32_t>(), blk_c.size(), n_snps, d_alt, d_ref, (hipStream_t)stream)) return rc;
    return stream_wait((hipStream_t)stream);   // the list goes back to the pool with this scope
}

// the upper-triangle 64 x 64 tiles of every contig's sim / diff matrices (the kernel mirrors)
static void simdiff_tiles(const std::vector<int32_t>& h_n_reads, std::vector<int32_t>& tc, std::vector<int32_t>& ti, std::vector<int32_t>& tj) {
    tc.clear(); ti.clear(); tj.clear();
    for (size_t c = 0; c < h_n_reads.size(); ++c) {
        const int nt = (h_n_reads[c] + 63) / 64;
        for (int i = 0; i < nt; ++i) for (int j = i; j < nt; ++j) { tc.push_back((int32_t)c); ti.push_back(i); tj.push_back(j); }
    }
}
static int simdiff_launch(const uint64_t* d_alt, const uint64_t* d_ref, const int64_t* d_plane_off, const int32_t* d_n_reads,
                          const int32_t* d_words, const int64_t* d_out_off, int32_t* d_sim, int32_t* d_diff, void* stream,
                          const int32_t* d_tc, const int32_t* d_ti, const int32_t* d_tj, size_t n_tiles, int es = 1 /* 2: (sim, diff) pairs, d_diff = d_sim + 1 */) {
    if (n_tiles == 0) return HS_OK;
    hipLaunchKernelGGL(hsdev::k_simdiff, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, d_alt, d_ref, d_plane_off,
                       d_n_reads, d_words, d_out_off, d_tc, d_ti, d_tj, d_sim, d_diff, es);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

int hs_simdiff(const uint64_t* d_alt, const uint64_t* d_ref, const int64_t* d_plane_off, const int32_t* d_n_reads,
               const int32_t* d_words, const int64_t* d_out_off, int32_t n_contigs, int32_t* d_sim, int32_t* d_diff,
               void* stream) {
    if (int rc = require_device()) return rc;
    if (n_contigs <= 0) return HS_OK;
    std::vector<int32_t> h_n((size_t)n_contigs);
    HS_HIP(hipMemcpy(h_n.data(), d_n_reads, sizeof(int32_t) * (size_t)n_contigs, hipMemcpyDeviceToHost));
    std::vector<int32_t> tc, ti, tj;
    simdiff_tiles(h_n, tc, ti, tj);
    DBuf a, b, c;
    UploadPack tiles;
    tiles.add(tc, a); tiles.add(ti, b); tiles.add(tj, c);
    if (int rc = tiles.commit((hipStream_t)stream)) return rc;
    if (int rc = simdiff_launch(d_alt, d_ref, d_plane_off, d_n_reads, d_words, d_out_off, d_sim, d_diff, stream, a.as<int32_t>(), b.as<int32_t>(), c.as<int32_t>(), tc.size())) return rc;
    if (int rc_w = stream_wait((hipStream_t)stream)) return rc_w;   // the tile lists die with this frame
    return HS_OK;
}

static int cw_launch(const int32_t* d_adj_off, const int32_t* d_adj, const int64_t* d_graph_off_base,
                     const int64_t* d_graph_adj_base, const int32_t* d_graph_n, const int32_t* d_perm, const int64_t* d_perm_base,
                     const uint8_t* d_mask, const int32_t* d_inst_graph, const int64_t* d_inst_label_base, int32_t n_inst,
                     int32_t max_n, int32_t* d_labels, int32_t* d_sweeps, void* stream) {
    if (n_inst <= 0) return HS_OK;
    size_t lds = (size_t)max_n * 8 + 1024;
    DBuf scratch;     // graphs whose labels + counters do not fit LDS keep them in global memory (2 * max_n ints per instance)
    const bool in_global = lds > 96 * 1024;
    if (in_global) { if (int rc = scratch.alloc((size_t)n_inst * 2 * (size_t)max_n * 4)) return rc; lds = 1024; }
    if (lds > 48 * 1024)
        HS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hsdev::k_chinese_whispers), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(hsdev::k_chinese_whispers, dim3((unsigned)n_inst), dim3(64), lds, (hipStream_t)stream, d_adj_off, d_adj,
                       d_graph_off_base, d_graph_adj_base, d_graph_n, d_perm, d_perm_base, d_mask, d_inst_graph,
                       d_inst_label_base, n_inst, d_labels, d_sweeps, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int32_t*)nullptr,
                       (const uint8_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, in_global ? scratch.as<int32_t>() : (int32_t*)nullptr, max_n);
    HS_HIP(hipGetLastError());
    if (in_global) { if (int rc = stream_wait((hipStream_t)stream)) return rc; }   // the scratch goes back to the pool with this scope
    return HS_OK;
}

int hs_chinese_whispers(const int32_t* d_adj_off, const int32_t* d_adj, const int64_t* d_graph_off_base,
                        const int64_t* d_graph_adj_base, const int32_t* d_graph_n, const int32_t* d_perm,
                        const int64_t* d_perm_base, const uint8_t* d_mask, const int32_t* d_inst_graph,
                        const int64_t* d_inst_label_base, int32_t n_inst, int32_t* d_labels, int32_t* d_sweeps,
                        void* stream) {
    if (int rc = require_device()) return rc;
    if (n_inst <= 0) return HS_OK;
    // the largest graph decides the LDS footprint: fetch the graph sizes of the instances
    std::vector<int32_t> ig((size_t)n_inst);
    HS_HIP(hipMemcpy(ig.data(), d_inst_graph, sizeof(int32_t) * (size_t)n_inst, hipMemcpyDeviceToHost));
    int gmax = 0;
    for (int g : ig) gmax = std::max(gmax, g);
    std::vector<int32_t> gn((size_t)gmax + 1);
    HS_HIP(hipMemcpy(gn.data(), d_graph_n, sizeof(int32_t) * gn.size(), hipMemcpyDeviceToHost));
    int max_n = 1;
    for (int g : ig) max_n = std::max(max_n, gn[(size_t)g]);
    return cw_launch(d_adj_off, d_adj, d_graph_off_base, d_graph_adj_base, d_graph_n, d_perm, d_perm_base, d_mask, d_inst_graph,
                     d_inst_label_base, n_inst, max_n, d_labels, d_sweeps, stream);
}

int hs_edit_distance(const uint8_t* d_query, const int64_t* d_query_off, const uint8_t* d_target,
                     const int64_t* d_target_off, int32_t n_pairs, int32_t mode, int32_t* d_dist, int32_t* d_end,
                     void* stream) {
    if (int rc = require_device()) return rc;
    if (n_pairs <= 0) return HS_OK;
    if (mode < 0 || mode > 2) { set_error("hs_edit_distance: mode must be 0 (NW), 1 (SHW) or 2 (HW)"); return HS_EINVAL; }
    // the offsets come back once: short queries share a wavefront (8 / 16 / 32 lanes per pair), the others take one each with the
    // hand-over row between two passes of 64 blocks as scratch (see hs_kernels_myers.hip)
    std::vector<int64_t> toff((size_t)n_pairs + 1), qoff((size_t)n_pairs + 1);
    HS_HIP(hipMemcpy(toff.data(), d_target_off, sizeof(int64_t) * toff.size(), hipMemcpyDeviceToHost));
    HS_HIP(hipMemcpy(qoff.data(), d_query_off, sizeof(int64_t) * qoff.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> cls[4];
    std::vector<int64_t> hs_off((size_t)n_pairs + 1, 0);
    const bool no_groups = std::getenv("HS_MYERS_NO_GROUPS") != nullptr;
    for (int i = 0; i < n_pairs; ++i) {
        const int64_t qn = qoff[(size_t)i + 1] - qoff[(size_t)i], tn = toff[(size_t)i + 1] - toff[(size_t)i];
        const int64_t nb = (qn + 63) / 64;
        const int c = (no_groups || nb > 32) ? 3 : nb <= 8 ? 0 : nb <= 16 ? 1 : 2;
        cls[c].push_back(i);
        hs_off[(size_t)i + 1] = hs_off[(size_t)i] + (c == 3 ? ((tn + 64 + 3) & ~(int64_t)3) + 4 * (tn + 64) : 0);
    }
    std::vector<int32_t> ids;
    size_t cls_off[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) { ids.insert(ids.end(), cls[c].begin(), cls[c].end()); cls_off[c + 1] = ids.size(); }
    DBuf scratch, d_ho, d_ids;
    UploadPack pk;
    pk.add(hs_off, d_ho); pk.add(ids, d_ids);
    if (int rc = pk.commit((hipStream_t)stream)) return rc;
    if (int rc = scratch.alloc(std::max<size_t>((size_t)hs_off.back(), 1))) return rc;
    const int32_t* idp = d_ids.as<int32_t>();
#define HS_MYERS_DIST_GROUPED(G, c)                                                                                                                       \
    if (!cls[c].empty())                                                                                                                                  \
        hipLaunchKernelGGL(hsdev::k_myers_distance_grouped<G>, dim3((unsigned)((cls[c].size() + 64 / G - 1) / (64 / G))), dim3(64), 0, (hipStream_t)stream, d_query, \
                           d_query_off, d_target, d_target_off, idp + cls_off[c], (int)cls[c].size(), mode, d_dist, d_end);
    HS_MYERS_DIST_GROUPED(8, 0)
    HS_MYERS_DIST_GROUPED(16, 1)
    HS_MYERS_DIST_GROUPED(32, 2)
#undef HS_MYERS_DIST_GROUPED
    if (!cls[3].empty())
        hipLaunchKernelGGL(hsdev::k_myers_distance, dim3((unsigned)cls[3].size()), dim3(64), 0, (hipStream_t)stream, d_query, d_query_off, d_target, d_target_off,
                           idp + cls_off[3], (int)cls[3].size(), mode, scratch.as<int8_t>(), d_ho.as<int64_t>(), d_dist, d_end);
    HS_HIP(hipGetLastError());
    if (int rc_w = stream_wait((hipStream_t)stream)) return rc_w;
    return HS_OK;
}

// A1 as the stage-5 call sites use edlib: HW mode, k = -1, TASK_PATH (see hs_kernels_myers.hip). Host offsets; the device
// buffers of the sequences and results are the caller's. d_ops may be NULL (locations only: edlib's TASK_LOC).
int hs_edlib_hw_align(const uint8_t* d_query, const int64_t* h_query_off, const uint8_t* d_target, const int64_t* h_target_off, int32_t n_pairs,
                      int32_t* d_dist, int32_t* d_start, int32_t* d_end, uint8_t* d_ops, const int64_t* h_ops_off, int32_t* d_ops_len, void* stream) {
    if (int rc = require_device()) return rc;
    if (n_pairs <= 0) return HS_OK;
    if (!h_query_off || !h_target_off || !d_dist || !d_start || !d_end || (d_ops && (!h_ops_off || !d_ops_len))) { set_error("hs_edlib_hw_align: bad arguments"); return HS_EINVAL; }
    const bool path = d_ops != nullptr;
    std::vector<int64_t> hs_off((size_t)n_pairs + 1, 0), st_off((size_t)n_pairs + 1, 0), qo(h_query_off, h_query_off + n_pairs + 1), to(h_target_off, h_target_off + n_pairs + 1),
        oo;
    // short queries whose matrix edlib keeps whole share a wavefront: 8 / 16 / 32 lanes per pair (k_myers_hw_path_grouped); the
    // others take a wavefront each
    std::vector<int32_t> cls[4];      // 0: 8 lanes, 1: 16, 2: 32, 3: a wavefront
    const bool no_groups = std::getenv("HS_MYERS_NO_GROUPS") != nullptr;      // (diagnostic, read at every call: every pair on a wavefront of its own)
    std::vector<int64_t> need_st((size_t)n_pairs, 0), need_hs((size_t)n_pairs, 0);
    for (int i = 0; i < n_pairs; ++i) {
        const int64_t qn = qo[(size_t)i + 1] - qo[(size_t)i], tn = to[(size_t)i + 1] - to[(size_t)i];
        const int64_t nb = (qn + 63) / 64;
        const bool one_leaf = 20 * nb * tn + 8 * tn < 1024 * 1024;      // edlib.cpp:1192-1196 on the whole target: on any part of it as well
        const int c = (no_groups || nb > 32 || (path && !one_leaf)) ? 3 : nb <= 8 ? 0 : nb <= 16 ? 1 : 2;
        cls[c].push_back(i);
        need_hs[(size_t)i] = c == 3 ? ((tn + 64 + 3) & ~(int64_t)3) + 4 * (tn + 64) : 0;      // deltas (bytes) and bottoms (ints) between two passes
        need_st[(size_t)i] = path ? std::min<int64_t>(tn * nb, MY_LEAF_CELLS) * 3 : 0;      // one leaf matrix (edlib's 1-MB rule), in 8-byte words
        if (path && h_ops_off[i + 1] - h_ops_off[i] < qn + tn) { set_error("hs_edlib_hw_align: an alignment needs room for query + target operations"); return HS_EINVAL; }
    }
    if (path) oo.assign(h_ops_off, h_ops_off + n_pairs + 1);
    // The pairs go out class by class, in chunks whose scratch (a leaf matrix per pair: up to 1.26 MB) stays within a budget: the
    // launches of a stream run one after the other, so the next chunk takes the same scratch again (HS_MYERS_SCRATCH_MB, default 16384:
    // 68 000 stage-5-sized pairs or 13 000 long ones at a time).
    const char* bud = std::getenv("HS_MYERS_SCRATCH_MB");
    const int64_t budget = std::max<int64_t>(2, bud ? std::atoll(bud) : 16384) * (1 << 20);
    struct Slice { int cls; size_t begin, end; };
    std::vector<Slice> slices;
    std::vector<int32_t> ids;
    int64_t cur_st = 0, cur_hs = 0, max_st = 0, max_hs = 0;
    for (int c = 0; c < 4; ++c) {
        size_t begin = ids.size();
        for (int32_t i : cls[c]) {
            if ((cur_st > 0 || cur_hs > 0) && (cur_st + need_st[(size_t)i]) * 8 + cur_hs + need_hs[(size_t)i] > budget) {      // the chunk is full: what came before goes out, the scratch starts over
                if (ids.size() > begin) slices.push_back(Slice{c, begin, ids.size()});
                begin = ids.size(); cur_st = 0; cur_hs = 0;
            }
            st_off[(size_t)i] = cur_st; hs_off[(size_t)i] = cur_hs;
            cur_st += need_st[(size_t)i]; cur_hs += need_hs[(size_t)i];
            max_st = std::max(max_st, cur_st); max_hs = std::max(max_hs, cur_hs);
            ids.push_back(i);
        }
        if (ids.size() > begin) slices.push_back(Slice{c, begin, ids.size()});
    }
    DBuf d_qo, d_to, d_ho, d_so, d_oo, d_hs, d_st, d_cols, d_ids;
    UploadPack pk;
    pk.add(qo, d_qo); pk.add(to, d_to); pk.add(hs_off, d_ho); pk.add(st_off, d_so); pk.add(ids, d_ids);
    if (path) pk.add(oo, d_oo);
    if (int rc = pk.commit((hipStream_t)stream)) return rc;
    if (int rc = d_hs.alloc(std::max<size_t>((size_t)max_hs, 1))) return rc;
    if (int rc = d_st.alloc(std::max<size_t>((size_t)max_st, 1) * 8)) return rc;
    if (int rc = d_cols.alloc(std::max<size_t>(path && !cls[3].empty() ? (size_t)(qo.back() - qo.front()) * 2 : 0, 1) * sizeof(int32_t))) return rc;      // Hirschberg's two columns
    const int32_t* idp = d_ids.as<int32_t>();
    const int64_t* oop = path ? d_oo.as<int64_t>() : nullptr;
    for (const Slice& sl : slices) {
        const int n = (int)(sl.end - sl.begin);
#define HS_MYERS_GROUPED(G)                                                                                                                               \
        hipLaunchKernelGGL(hsdev::k_myers_hw_path_grouped<G>, dim3((unsigned)((n + 64 / G - 1) / (64 / G))), dim3(64), 0, (hipStream_t)stream, d_query,        \
                           d_qo.as<int64_t>(), d_target, d_to.as<int64_t>(), idp + sl.begin, n, d_st.as<unsigned long long>(), d_so.as<int64_t>(),        \
                           path ? 1 : 0, d_dist, d_start, d_end, d_ops, oop, d_ops_len)
        if (sl.cls == 0) HS_MYERS_GROUPED(8);
        else if (sl.cls == 1) HS_MYERS_GROUPED(16);
        else if (sl.cls == 2) HS_MYERS_GROUPED(32);
        else
            hipLaunchKernelGGL(hsdev::k_myers_hw_path, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_query, d_qo.as<int64_t>(), d_target, d_to.as<int64_t>(),
                               idp + sl.begin, n, d_hs.as<int8_t>(), d_ho.as<int64_t>(), d_st.as<unsigned long long>(), d_so.as<int64_t>(), d_cols.as<int32_t>(),
                               path ? 1 : 0, d_dist, d_start, d_end, d_ops, oop, d_ops_len);
#undef HS_MYERS_GROUPED
    }
    HS_HIP(hipGetLastError());
    return stream_wait((hipStream_t)stream);   // the scratch goes back to the pool with this scope
}

// ---------------------------------------------------------------------------------------------------
// stage 3 driver
// ---------------------------------------------------------------------------------------------------
struct hs_cv_batch {
    int32_t n_contigs = 0, n_reads = 0, n_rec = 0;
    std::vector<int64_t> contig_off, pile_off;
    std::vector<int32_t> contig_rec_off, rec_pos, rec_qend, rec_contig;
    std::vector<int64_t> rec_refspan;
    std::vector<int32_t> ploidy;      // hs_cv_batch_set_ploidy
    int64_t total_len = 0, total_pile = 0;
    int32_t n_tasks = 0, ev_per_task = 4096, max_depth = 0;
    std::vector<int32_t> rec_task_off;      // [n_rec + 1] first pileup task of every record (tasks are listed record by record)
    int device = 0;                   // the device that was current when the batch was created: every buffer below lives there
    DBuf contig_seq, d_contig_off, read_seq, read_off, rec_read, d_rec_contig, d_rec_pos, rec_strand, rec_cig_off, cigar,
        d_pile_off, d_contig_rec_off, d_rec_qend, pile, rec_stats, rec_chunk_off, chunk_scratch, task_rec, task_ev0,
        tile_off, tile_ent, tile_lrec, d_rank_of, d_orig_of, d_read_end, d_rank_end;
    HBuf h_stage_a;   // pinned staging of the per-record counters
    // the pileup is padded by 256 bytes on both sides: k_gather_tiles loads the 256 bytes a record lays over a tile whole, also
    // where the record covers part of the tile only
    uint8_t* pile_ptr() const { return pile.as<uint8_t>() + 256; }
};

// The stage drivers allocate and free multi-megabyte arrays on many threads every call; with glibc's defaults those go
// through mmap/munmap and heap trimming, i.e. page faults and TLB shootdowns across all threads (20-ms stalls every few
// steps). Keep the memory in the process instead.
static void tune_allocator() {
    static std::once_flag once;
    std::call_once(once, [] {
        const int a = mallopt(M_MMAP_THRESHOLD, 32 << 20);      // the largest value glibc accepts
        const int b = mallopt(M_TRIM_THRESHOLD, 0x7fffffff);      // (never give freed memory back: the next call takes it again)
        const int c = mallopt(M_TOP_PAD, 64 << 20);
        if (std::getenv("HS_TIMING")) std::fprintf(stderr, "[hs timing] mallopt: mmap_threshold %d trim_threshold %d top_pad %d\n", a, b, c);
    });
}

int hs_cv_batch_create(const uint8_t* h_contig_seq, const int64_t* h_contig_off, int32_t n_contigs,
                       const uint8_t* h_read_seq, const int64_t* h_read_off, int32_t n_reads,
                       const int32_t* h_rec_read, const int32_t* h_rec_pos, const uint8_t* h_rec_strand,
                       const int64_t* h_rec_cig_off, const uint32_t* h_cigar, const int32_t* h_contig_rec_off,
                       hs_cv_batch** out) {
    if (int rc = require_device()) return rc;
    if (!out || n_contigs < 0) { set_error("hs_cv_batch_create: bad arguments"); return HS_EINVAL; }
    tune_allocator();
    hs_cv_batch* b = new hs_cv_batch();
    HS_HIP(hipGetDevice(&b->device));
    b->n_contigs = n_contigs; b->n_reads = n_reads;
    b->contig_off.assign(h_contig_off, h_contig_off + n_contigs + 1);
    b->contig_rec_off.assign(h_contig_rec_off, h_contig_rec_off + n_contigs + 1);
    b->n_rec = b->contig_rec_off[(size_t)n_contigs];
    b->total_len = b->contig_off[(size_t)n_contigs];
    const int n_rec = b->n_rec;
    b->rec_pos.assign(h_rec_pos, h_rec_pos + n_rec);
    b->rec_contig.resize((size_t)n_rec);
    b->rec_qend.resize((size_t)n_rec);
    b->rec_refspan.resize((size_t)n_rec);
    b->pile_off.assign((size_t)n_rec + 1, 0);
    std::vector<int64_t> readspan_of((size_t)n_rec);
    {   // reference / read span of every record: one pass over all CIGAR ops, blocks of records on the host threads
        const int nb = std::max(1, std::min(n_rec / 256 + 1, 256));
        hs::hs_parallel_for(nb, host_threads(), [&](int blk) {
            const int r0 = (int)((int64_t)n_rec * blk / nb), r1 = (int)((int64_t)n_rec * (blk + 1) / nb);
            for (int r = r0; r < r1; ++r) {
                int64_t refspan = 0, readspan = 0;
                for (int64_t o = h_rec_cig_off[r]; o < h_rec_cig_off[r + 1]; ++o) {
                    const uint32_t op = h_cigar[o] & 15u; const int64_t len = h_cigar[o] >> 4;
                    if (op == 0 || op == 2 || op == 7 || op == 8) refspan += len;
                    if (op == 0 || op == 1 || op == 4 || op == 5 || op == 7 || op == 8) readspan += len;
                }
                b->rec_refspan[(size_t)r] = refspan; readspan_of[(size_t)r] = readspan;
            }
        });
    }
    for (int c = 0; c < n_contigs; ++c) {
        const int64_t L = b->contig_off[(size_t)c + 1] - b->contig_off[(size_t)c];
        for (int r = b->contig_rec_off[(size_t)c]; r < b->contig_rec_off[(size_t)c + 1]; ++r) {
            b->rec_contig[(size_t)r] = c;
            const int64_t refspan = b->rec_refspan[(size_t)r], readspan = readspan_of[(size_t)r];
            const int64_t rl = h_read_off[h_rec_read[r] + 1] - h_read_off[h_rec_read[r]];
            const int64_t pos = h_rec_pos[r];
            if (pos < 0) { set_error("negative alignment start"); delete b; return HS_EINVAL; }
            int64_t qend = pos >= L ? pos : std::min(pos + refspan, L);
            // a CIGAR that runs past the read is only tolerated for the part that lies beyond the contig end
            if (readspan > rl && pos + refspan <= L) { set_error("CIGAR consumes more bases than the read has"); delete b; return HS_EINVAL; }
            b->rec_qend[(size_t)r] = (int32_t)qend;
            b->pile_off[(size_t)r + 1] = b->pile_off[(size_t)r] + (qend - pos);
        }
    }
    b->total_pile = b->pile_off[(size_t)n_rec];
    {   // deepest position of the batch (sweep over record starts / ends): picks the histogram counter width of K2
        std::vector<std::pair<int32_t, int32_t>> ev;
        for (int c = 0; c < n_contigs; ++c) {
            ev.clear();
            for (int r = b->contig_rec_off[(size_t)c]; r < b->contig_rec_off[(size_t)c + 1]; ++r)
                if (b->rec_qend[(size_t)r] > b->rec_pos[(size_t)r]) { ev.push_back(std::make_pair(b->rec_pos[(size_t)r], 1)); ev.push_back(std::make_pair(b->rec_qend[(size_t)r], -1)); }
            std::sort(ev.begin(), ev.end());
            int d = 0;
            for (auto& e : ev) { d += e.second; b->max_depth = std::max(b->max_depth, d); }
        }
        // what the kernels need is a bound on the depth of ONE position (K2 counts in 16-bit lanes; the reference's own loop
        // over a column's reads is a `short`, call_variants.cpp:479); the number of records on a contig is unbounded
        if (b->max_depth > 65535) { set_error("a position is covered by more than 65535 alignment records"); delete b; return HS_EINVAL; }
    }
    int rc = 0;
    // Large pageable arrays (read bases, CIGAR ops: gigabytes) go up through two pinned 32-MB buffers: a chunk is copied into one on
    // several host threads while the other is on its way (a plain hipMemcpy from pageable memory stages through the runtime's own
    // buffers with one copying thread: 12-15 GB/s)
    HBuf stage[2];
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    auto up_big = [&](void* dst, const char* src, size_t bytes) -> int {
        const size_t CH = (size_t)32 << 20;
        for (int k = 0; k < 2; ++k) {
            if (!stage[k].p) { if (int r = stage[k].alloc(CH)) return r; }
            if (!stage_done[k]) HS_HIP(hipEventCreateWithFlags(&stage_done[k], hipEventDisableTiming));
        }
        int k = 0;
        for (size_t off = 0; off < bytes; off += CH, k ^= 1) {
            const size_t n = std::min(CH, bytes - off);
            HS_HIP(hipEventSynchronize(stage_done[k]));      // (the copy that last used this buffer; returns at once the first time)
            const int parts = 8;
            hs::hs_parallel_for(parts, std::min(parts, host_threads()), [&](int q) {
                const size_t a = n * (size_t)q / parts, e = n * ((size_t)q + 1) / parts;
                std::memcpy((char*)stage[k].p + a, src + off + a, e - a);
            });
            HS_HIP(hipMemcpyAsync((char*)dst + off, stage[k].p, n, hipMemcpyHostToDevice, nullptr));
            HS_HIP(hipEventRecord(stage_done[k], nullptr));
        }
        HS_HIP(hipStreamSynchronize(nullptr));
        return HS_OK;
    };
    auto up = [&](DBuf& d, const void* src, size_t bytes) {
        if (rc) return;
        rc = d.alloc(bytes);
        if (rc || !bytes) return;
        if (bytes >= ((size_t)64 << 20)) { rc = up_big(d.p, (const char*)src, bytes); return; }
        hipError_t e = hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = HS_EHIP; }
    };
    up(b->contig_seq, h_contig_seq, (size_t)b->total_len);
    up(b->d_contig_off, b->contig_off.data(), sizeof(int64_t) * b->contig_off.size());
    up(b->read_seq, h_read_seq, (size_t)h_read_off[n_reads]);
    up(b->read_off, h_read_off, sizeof(int64_t) * ((size_t)n_reads + 1));
    up(b->rec_read, h_rec_read, sizeof(int32_t) * (size_t)n_rec);
    up(b->d_rec_contig, b->rec_contig.data(), sizeof(int32_t) * (size_t)n_rec);
    up(b->d_rec_pos, h_rec_pos, sizeof(int32_t) * (size_t)n_rec);
    up(b->rec_strand, h_rec_strand, (size_t)n_rec);
    up(b->rec_cig_off, h_rec_cig_off, sizeof(int64_t) * ((size_t)n_rec + 1));
    up(b->cigar, h_cigar, sizeof(uint32_t) * (size_t)h_rec_cig_off[n_rec]);
    up(b->d_pile_off, b->pile_off.data(), sizeof(int64_t) * b->pile_off.size());
    {   // tile plan of K2 / K3
        std::vector<int64_t> to; std::vector<hs_tile_entry> en; std::vector<int32_t> rc_;
        build_tile_plan(b->contig_off.data(), n_contigs, b->contig_rec_off.data(), b->rec_pos.data(), b->rec_qend.data(), b->pile_off.data(), to, en, rc_);
        up(b->tile_off, to.data(), sizeof(int64_t) * to.size());
        up(b->tile_ent, en.data(), sizeof(hs_tile_entry) * en.size());
        for (size_t k = 0; k < rc_.size(); ++k) rc_[k] -= b->contig_rec_off[(size_t)b->rec_contig[(size_t)rc_[k]]];      // the read's index on its contig: what a column lists
        up(b->tile_lrec, rc_.data(), sizeof(int32_t) * rc_.size());
    }
    up(b->d_contig_rec_off, b->contig_rec_off.data(), sizeof(int32_t) * b->contig_rec_off.size());
    {   // what loop A on the device (k_loop_a) needs of the reads: their rank by start position on the contig (ties by index: the
        // bit order of the partition bit sets, as rank_reads() of the host glue), the inverse, the end of their alignment
        std::vector<int32_t> rank_of((size_t)n_rec), orig_of((size_t)n_rec), read_end((size_t)n_rec);
        hs::hs_parallel_for(n_contigs, host_threads(), [&](int c) {
            const int r0 = b->contig_rec_off[(size_t)c], n = b->contig_rec_off[(size_t)c + 1] - r0;
            std::vector<int32_t> order((size_t)n);
            for (int r = 0; r < n; ++r) order[(size_t)r] = r;
            std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return h_rec_pos[r0 + x] != h_rec_pos[r0 + y] ? h_rec_pos[r0 + x] < h_rec_pos[r0 + y] : x < y; });
            for (int k = 0; k < n; ++k) { rank_of[(size_t)(r0 + order[(size_t)k])] = k; orig_of[(size_t)(r0 + k)] = order[(size_t)k]; }
            for (int k = 0; k < n; ++k) {      // (in rank order: the kernel indexes everything per read by rank)
                const int r = order[(size_t)k];
                read_end[(size_t)(r0 + k)] = (int32_t)std::min<int64_t>((int64_t)h_rec_pos[r0 + r] + b->rec_refspan[(size_t)(r0 + r)], 0x7fffffff);
            }
        });
        up(b->d_rank_of, rank_of.data(), sizeof(int32_t) * (size_t)n_rec);
        up(b->d_orig_of, orig_of.data(), sizeof(int32_t) * (size_t)n_rec);
        up(b->d_read_end, read_end.data(), sizeof(int32_t) * (size_t)n_rec);
        {   // per READ of a contig: {its rank, the end of its alignment} (k_cand_bits: one gather per column entry)
            std::vector<int32_t> rank_end((size_t)n_rec * 2);
            for (int r = 0; r < n_rec; ++r) { rank_end[(size_t)r * 2] = rank_of[(size_t)r]; rank_end[(size_t)r * 2 + 1] = (int32_t)std::min<int64_t>((int64_t)h_rec_pos[r] + b->rec_refspan[(size_t)r], 0x7fffffff); }
            up(b->d_rank_end, rank_end.data(), sizeof(int32_t) * 2 * (size_t)n_rec);
        }
    }
    up(b->d_rec_qend, b->rec_qend.data(), sizeof(int32_t) * (size_t)n_rec);
    if (!rc) {   // launch plan of the pileup kernel
        std::vector<int64_t> chunk_off((size_t)n_rec + 1);
        int32_t* tr = nullptr; int32_t* te = nullptr;
        rc = hs_pileup_plan(h_rec_cig_off, h_cigar, n_rec, b->ev_per_task, chunk_off.data(), &b->n_tasks, &tr, &te);
        if (!rc) {
            up(b->rec_chunk_off, chunk_off.data(), sizeof(int64_t) * chunk_off.size());
            up(b->task_rec, tr, sizeof(int32_t) * (size_t)b->n_tasks);
            up(b->task_ev0, te, sizeof(int32_t) * (size_t)b->n_tasks);
            b->rec_task_off.assign((size_t)n_rec + 1, 0);
            for (int32_t t = 0; t < b->n_tasks; ++t) b->rec_task_off[(size_t)tr[t] + 1]++;
            for (int r = 0; r < n_rec; ++r) b->rec_task_off[(size_t)r + 1] += b->rec_task_off[(size_t)r];
            if (!rc) rc = b->chunk_scratch.alloc(sizeof(int32_t) * 4 * (size_t)chunk_off[(size_t)n_rec]);
        }
        std::free(tr); std::free(te);
    }
    if (!rc) rc = b->pile.alloc((size_t)b->total_pile + 512);
    if (!rc) rc = b->rec_stats.alloc(sizeof(int32_t) * 4 * (size_t)n_rec);
    for (int k = 0; k < 2; ++k) if (stage_done[k]) (void)hipEventDestroy(stage_done[k]);
    if (rc) { delete b; return rc; }
    *out = b;
    return HS_OK;
}

void hs_cv_batch_destroy(hs_cv_batch* b) { delete b; }
int64_t hs_cv_batch_aligned_bp(const hs_cv_batch* b) { return b ? b->total_pile : 0; }

void hs_cv_result_destroy(hs_cv_result* r) { hs::free_cv_result(r); }
void hs_sr_result_destroy(hs_sr_result* r) { hs::free_sr_result(r); }

}  // extern "C"

namespace {

// HIP implementation of the stage-3 device interface (the only one the product has). The columns of a contig range live in
// this object from extract_candidates() to finish_columns(); the SNP columns it leaves (snp_*) are what stage 4 reads.
struct HipCvOps : hs::CvDeviceOps {
    hs_cv_batch* b;
    hipStream_t stream = nullptr;
    explicit HipCvOps(hs_cv_batch* batch) : b(batch) {}
    KernelClock kc;

    // ---- K0 + K1 over the whole batch ----
    int pileup(std::vector<int32_t>& rec_stats, float k_ms[4]) override {
        EventPair e0, e1;
        if (int rc = e0.init()) return rc;
        if (int rc = e1.init()) return rc;
        HS_HIP(hipEventRecord(e0.a, stream));
        if (int rc = kc.begin(HS_K_CIGAR_SCAN, stream)) return rc;
        if (int rc = cigar_scan_launch(b->d_contig_off.as<int64_t>(), b->d_rec_contig.as<int32_t>(), b->d_rec_pos.as<int32_t>(),
                                       b->rec_cig_off.as<int64_t>(), b->cigar.as<uint32_t>(), b->rec_chunk_off.as<int64_t>(), b->n_rec,
                                       b->chunk_scratch.as<int32_t>(), b->rec_stats.as<int32_t>(), stream)) return rc;
        if (int rc = kc.end((int64_t)b->cigar.bytes + (int64_t)b->chunk_scratch.bytes + 16 * (int64_t)b->n_rec, stream)) return rc;   // ops in, chunk table + counters out
        HS_HIP(hipEventRecord(e0.b, stream));
        HS_HIP(hipEventRecord(e1.a, stream));
        if (int rc = kc.begin(HS_K_PILEUP, stream)) return rc;
        if (int rc = pileup_launch(b->contig_seq.as<uint8_t>(), b->d_contig_off.as<int64_t>(), b->read_seq.as<uint8_t>(), b->read_off.as<int64_t>(),
                                   b->rec_read.as<int32_t>(), b->d_rec_contig.as<int32_t>(), b->d_rec_pos.as<int32_t>(), b->rec_strand.as<uint8_t>(),
                                   b->rec_cig_off.as<int64_t>(), b->cigar.as<uint32_t>(), b->d_pile_off.as<int64_t>(),
                                   b->rec_chunk_off.as<int64_t>(), b->chunk_scratch.as<int32_t>(), b->task_rec.as<int32_t>(), b->task_ev0.as<int32_t>(),
                                   b->n_tasks, b->ev_per_task, b->pile_ptr(), b->rec_stats.as<int32_t>(), b->n_rec, stream)) return rc;
        if (int rc = kc.end(2 * b->total_pile, stream)) return rc;      // one read base in + one code out per aligned bp
        HS_HIP(hipEventRecord(e1.b, stream));
        auto grow = [](HBuf& h, size_t need) -> int { if (h.cap >= need && h.p) return HS_OK; return h.alloc(need + need / 4); };
        if (!rec_stats.empty()) {
            if (int rc = grow(b->h_stage_a, rec_stats.size() * sizeof(int32_t))) return rc;
            HS_HIP(HS_COPY_ASYNC(b->h_stage_a.p, b->rec_stats.p, rec_stats.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        }
        if (int rc = stream_wait(stream)) return rc;
        if (!rec_stats.empty()) std::memcpy(rec_stats.data(), b->h_stage_a.p, rec_stats.size() * sizeof(int32_t));
        if (int rc = e1.ms(&k_ms[0])) return rc;
        if (int rc = e0.ms(&k_ms[3])) return rc;
        kc.flush();
        return HS_OK;
    }

    // K0 + K1 for the records of the contigs [c0, c1) only (the fused pipeline: every contig group brings up its own share of the
    // pileup, one group at a time, so that the first group's host work starts a quarter of a millisecond into the step instead
    // of after the whole batch); rec_stats: the batch-wide array, this range's slice is filled in
    int pileup_range(int c0, int c1, std::vector<int32_t>& rec_stats, float k_ms[4]) {
        const int r0 = b->contig_rec_off[(size_t)c0], r1 = b->contig_rec_off[(size_t)c1];
        const int nr = r1 - r0;
        if (nr <= 0) return HS_OK;
        if ((int)b->rec_task_off.size() != b->n_rec + 1) { set_error("pileup_range: the batch has no task index"); return HS_EINVAL; }
        const int t0 = b->rec_task_off[(size_t)r0], t1 = b->rec_task_off[(size_t)r1];
        EventPair e0, e1;
        if (int rc = e0.init()) return rc;
        if (int rc = e1.init()) return rc;
        {
            DeviceTurn turn;
            HS_HIP(hipEventRecord(e0.a, stream));
            if (int rc = kc.begin(HS_K_CIGAR_SCAN, stream)) return rc;
            if (int rc = cigar_scan_launch(b->d_contig_off.as<int64_t>(), b->d_rec_contig.as<int32_t>() + r0, b->d_rec_pos.as<int32_t>() + r0,
                                           b->rec_cig_off.as<int64_t>() + r0, b->cigar.as<uint32_t>(), b->rec_chunk_off.as<int64_t>() + r0, nr,
                                           b->chunk_scratch.as<int32_t>(), b->rec_stats.as<int32_t>() + 4 * (size_t)r0, stream)) return rc;
            const double share = b->n_rec > 0 ? (double)nr / (double)b->n_rec : 0.0;
            if (int rc = kc.end((int64_t)(share * ((double)b->cigar.bytes + (double)b->chunk_scratch.bytes)) + 16 * (int64_t)nr, stream)) return rc;
            HS_HIP(hipEventRecord(e0.b, stream));
            HS_HIP(hipEventRecord(e1.a, stream));
            if (int rc = kc.begin(HS_K_PILEUP, stream)) return rc;
            static const bool per_event = std::getenv("HS_K1_PER_EVENT") != nullptr;
            int64_t range_pile = 0;
            if (t1 > t0) {
                if (per_event)
                    hipLaunchKernelGGL(hsdev::k_pileup, dim3((t1 - t0 + 3) / 4), dim3(256), 0, stream, b->contig_seq.as<uint8_t>(), b->d_contig_off.as<int64_t>(),
                                       b->read_seq.as<uint8_t>(), b->read_off.as<int64_t>(), b->rec_read.as<int32_t>(), b->d_rec_contig.as<int32_t>(), b->d_rec_pos.as<int32_t>(),
                                       b->rec_strand.as<uint8_t>(), b->rec_cig_off.as<int64_t>(), b->cigar.as<uint32_t>(), b->d_pile_off.as<int64_t>(), b->rec_chunk_off.as<int64_t>(),
                                       b->chunk_scratch.as<int32_t>(), b->task_rec.as<int32_t>() + t0, b->task_ev0.as<int32_t>() + t0, t1 - t0, b->ev_per_task, b->pile_ptr(),
                                       b->rec_stats.as<int32_t>());
                else {
                    hipLaunchKernelGGL(hsdev::k_pileup_packed, dim3((t1 - t0 + 3) / 4), dim3(256), 0, stream, b->contig_seq.as<uint8_t>(), b->d_contig_off.as<int64_t>(),
                                       b->read_seq.as<uint8_t>(), b->read_off.as<int64_t>(), b->rec_read.as<int32_t>(), b->d_rec_contig.as<int32_t>(), b->d_rec_pos.as<int32_t>(),
                                       b->rec_strand.as<uint8_t>(), b->rec_cig_off.as<int64_t>(), b->cigar.as<uint32_t>(), b->d_pile_off.as<int64_t>(), b->rec_chunk_off.as<int64_t>(),
                                       b->chunk_scratch.as<int32_t>(), b->task_rec.as<int32_t>() + t0, b->task_ev0.as<int32_t>() + t0, t1 - t0, b->ev_per_task, b->pile_ptr(),
                                       b->rec_stats.as<int32_t>());
                    const int blocks = std::max(1, std::min(256, (nr + 255) / 256));
                    hipLaunchKernelGGL(hsdev::k_pileup_flagged_records, dim3(blocks), dim3(256), 0, stream, b->contig_seq.as<uint8_t>(), b->d_contig_off.as<int64_t>(),
                                       b->read_seq.as<uint8_t>(), b->read_off.as<int64_t>(), b->rec_read.as<int32_t>() + r0, b->d_rec_contig.as<int32_t>() + r0,
                                       b->d_rec_pos.as<int32_t>() + r0, b->rec_strand.as<uint8_t>() + r0, b->rec_cig_off.as<int64_t>() + r0, b->cigar.as<uint32_t>(),
                                       b->d_pile_off.as<int64_t>() + r0, b->rec_chunk_off.as<int64_t>() + r0, b->chunk_scratch.as<int32_t>(), nr, b->ev_per_task, b->pile_ptr(),
                                       b->rec_stats.as<int32_t>() + 4 * (size_t)r0);
                }
                HS_HIP(hipGetLastError());
                range_pile = b->pile_off[(size_t)r1] - b->pile_off[(size_t)r0];
            }
            if (int rc = kc.end(2 * range_pile, stream)) return rc;      // one read base in + one code out per aligned bp
            HS_HIP(hipEventRecord(e1.b, stream));
            const size_t bytes = (size_t)nr * 4 * sizeof(int32_t);
            if (int rc = h_info_grow(h_rec_stats, bytes)) return rc;
            HS_HIP(HS_COPY_ASYNC(h_rec_stats.p, b->rec_stats.as<int32_t>() + 4 * (size_t)r0, bytes, hipMemcpyDeviceToHost, stream));
            if (int rc = stream_wait(stream)) return rc;      // (inside the turn: the next group's share starts when this one is through)
            std::memcpy(rec_stats.data() + 4 * (size_t)r0, h_rec_stats.p, bytes);
        }
        float m = 0;
        if (int rc = e1.ms(&m)) return rc; k_ms[0] += m;
        if (int rc = e0.ms(&m)) return rc; k_ms[3] += m;
        kc.flush();
        return HS_OK;
    }
    HBuf h_rec_stats;
    // K0 + K1 of the contigs [c0, c1) queued on the stream, nothing waited for, and the contigs' mean distances formed on the device
    // (k_contig_error: call_variants.cpp:434 from K1's integer counters, and the read minimum of :463-466 that follows from it) into the
    // info block / d_min_reads: the fused pipeline's column pass starts from its own share of the pileup without the host in between
    bool own_pileup = false;
    int pileup_range_launch(int c0, int c1, float* d_mean_distance, int32_t* d_min_reads_out) {
        const int r0 = b->contig_rec_off[(size_t)c0], r1 = b->contig_rec_off[(size_t)c1];
        const int nr = r1 - r0;
        if ((int)b->rec_task_off.size() != b->n_rec + 1) { set_error("pileup_range: the batch has no task index"); return HS_EINVAL; }
        if (nr > 0) {
            const int t0 = b->rec_task_off[(size_t)r0], t1 = b->rec_task_off[(size_t)r1];
            if (int rc = kc.begin(HS_K_CIGAR_SCAN, stream)) return rc;
            if (int rc = cigar_scan_launch(b->d_contig_off.as<int64_t>(), b->d_rec_contig.as<int32_t>() + r0, b->d_rec_pos.as<int32_t>() + r0,
                                           b->rec_cig_off.as<int64_t>() + r0, b->cigar.as<uint32_t>(), b->rec_chunk_off.as<int64_t>() + r0, nr,
                                           b->chunk_scratch.as<int32_t>(), b->rec_stats.as<int32_t>() + 4 * (size_t)r0, stream)) return rc;
            const double share = b->n_rec > 0 ? (double)nr / (double)b->n_rec : 0.0;
            if (int rc = kc.end((int64_t)(share * ((double)b->cigar.bytes + (double)b->chunk_scratch.bytes)) + 16 * (int64_t)nr, stream)) return rc;
            if (int rc = kc.begin(HS_K_PILEUP, stream)) return rc;
            int64_t range_pile = 0;
            if (t1 > t0) {
                hipLaunchKernelGGL(hsdev::k_pileup_packed, dim3((t1 - t0 + 3) / 4), dim3(256), 0, stream, b->contig_seq.as<uint8_t>(), b->d_contig_off.as<int64_t>(),
                                   b->read_seq.as<uint8_t>(), b->read_off.as<int64_t>(), b->rec_read.as<int32_t>(), b->d_rec_contig.as<int32_t>(), b->d_rec_pos.as<int32_t>(),
                                   b->rec_strand.as<uint8_t>(), b->rec_cig_off.as<int64_t>(), b->cigar.as<uint32_t>(), b->d_pile_off.as<int64_t>(), b->rec_chunk_off.as<int64_t>(),
                                   b->chunk_scratch.as<int32_t>(), b->task_rec.as<int32_t>() + t0, b->task_ev0.as<int32_t>() + t0, t1 - t0, b->ev_per_task, b->pile_ptr(),
                                   b->rec_stats.as<int32_t>());
                const int blocks = std::max(1, std::min(256, (nr + 255) / 256));
                hipLaunchKernelGGL(hsdev::k_pileup_flagged_records, dim3(blocks), dim3(256), 0, stream, b->contig_seq.as<uint8_t>(), b->d_contig_off.as<int64_t>(),
                                   b->read_seq.as<uint8_t>(), b->read_off.as<int64_t>(), b->rec_read.as<int32_t>() + r0, b->d_rec_contig.as<int32_t>() + r0,
                                   b->d_rec_pos.as<int32_t>() + r0, b->rec_strand.as<uint8_t>() + r0, b->rec_cig_off.as<int64_t>() + r0, b->cigar.as<uint32_t>(),
                                   b->d_pile_off.as<int64_t>() + r0, b->rec_chunk_off.as<int64_t>() + r0, b->chunk_scratch.as<int32_t>(), nr, b->ev_per_task, b->pile_ptr(),
                                   b->rec_stats.as<int32_t>() + 4 * (size_t)r0);
                HS_HIP(hipGetLastError());
                range_pile = b->pile_off[(size_t)r1] - b->pile_off[(size_t)r0];
            }
            if (int rc = kc.end(2 * range_pile, stream)) return rc;      // one read base in + one code out per aligned bp
        }
        hipLaunchKernelGGL(hsdev::k_contig_error, dim3((unsigned)((c1 - c0 + 3) / 4)), dim3(256), 0, stream, b->rec_stats.as<int32_t>(), b->d_contig_rec_off.as<int32_t>(), c0, c1 - c0,
                           d_mean_distance, d_min_reads_out);
        HS_HIP(hipGetLastError());
        return HS_OK;
    }
    static int h_info_grow(HBuf& h, size_t need) { if (h.cap >= need && h.p) return HS_OK; return h.alloc(need + need / 4); }

    // ---- the columns of the current contig range ----
    int range_c0 = 0, range_c1 = 0;
    int64_t n_cols = 0, n_entries = 0;            // extracted columns / their entries
    SelectionScratch range_scratch;               // K2's per-tile slots
    DBuf d_tile_ent_sum, d_tile_ebase, d_scan2;
    DBuf d_col_gpos, d_col_rec, d_co, d_col_len, d_ci, d_cc;      // d_co / d_ci / d_cc: the CSR of the columns (also read by k_loop_a_prepare)
    DBuf d_col_ctg, d_k0, d_k1, d_c1, d_cand, d_ctg_col_off, d_min_reads, d_blk_cnt, d_blk_ent;
    // what the host reads between the phases, one small block = one download: [ColumnsHeader 64 B][tie counters 16 B][pad][per-contig counts 4 C]
    DBuf d_info; HBuf h_info;
    hsdev::ColumnsHeader* dev_header() const { return d_info.as<hsdev::ColumnsHeader>(); }
    unsigned long long* dev_tie() const { return (unsigned long long*)((char*)d_info.p + 64); }
    // behind the two counters: per contig of the range the candidates [C], the SNPs [C] and the SNP bounds [2 C], cleared with the header
    int32_t* dev_ctg_n() const { return (int32_t*)((char*)d_info.p + 128); }
    int32_t* dev_ctg_snp() const { return dev_ctg_n() + (range_c1 - range_c0); }
    int32_t* dev_snp_bounds() const { return dev_ctg_n() + 2 * (range_c1 - range_c0); }
    const hsdev::ColumnsHeader& host_header() const { return *(const hsdev::ColumnsHeader*)h_info.p; }
    const int32_t* host_ctg_n() const { return (const int32_t*)((const char*)h_info.p + 128); }
    const int32_t* host_ctg_snp() const { return host_ctg_n() + (range_c1 - range_c0); }
    // ... and, for a range that brought up its own pileup, the contigs' mean distances [C floats]
    static size_t info_bytes(int C) { return (128 + (size_t)C * 20 + 15) & ~(size_t)15; }
    float* dev_ctg_md() const { return (float*)(dev_ctg_n() + 4 * (range_c1 - range_c0)); }
    const float* host_ctg_md() const { return (const float*)(host_ctg_n() + 4 * (range_c1 - range_c0)); }
    // the flagged columns (candidates, later the SNPs) packed into ONE block = one download: [records 16 nf][column index 4 nf][offsets 8 (nf + 1)]
    // [read indices 4 ne][codes ne], every part 256-byte aligned; the SNP block is what stage 4 takes over (HipSrOps::adopt_columns)
    DBuf d_pk; HBuf h_pk;
    struct PackLayout { size_t rec, col, off, idx, code, total, head; };
    static PackLayout pack_layout(int64_t nf, int64_t ne) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        PackLayout L;
        L.rec = 0; L.col = up((size_t)nf * 16); L.off = L.col + up((size_t)nf * 4); L.idx = L.off + up(((size_t)nf + 1) * 8);
        L.head = L.idx; L.code = L.idx + up((size_t)ne * 4); L.total = L.code + up((size_t)ne + 1);
        return L;
    }
    PackLayout pk_layout{};
    UploadPack range_pack;
    int n_gathered = 0;                           // (k_loop_a_prepare checks its column indices against it)
    int64_t gathered_entries = 0;
    static int grow(HBuf& h, size_t need) { if (h.cap >= need && h.p) return HS_OK; return h.alloc(need + need / 4); }
    static int grow(DBuf& d, size_t need) { if (d.cap >= need && d.p && !d.view) { d.bytes = need; return HS_OK; } return d.alloc(need + need / 4); }

    // What a call learnt about the sizes of this contig range, kept by the caller from step to step (a pipeline group runs the same
    // contigs again and again; a service runs batches of the same shape): with it the arrays are sized ahead and the whole column
    // pass, K2 to the candidates' bit sets, is queued without a host round trip; every kernel checks the capacities it was given and
    // a pass whose numbers did not fit (the header says so) is run again the careful way, sizes first.
    struct Keep {
        bool valid = false;
        int64_t cols = 0, entries = 0, cand = 0, cand_entries = 0, cb_words = 0, snp = 0, snp_entries = 0;
        hipEvent_t k2_done = nullptr;      // K2 of the groups one after the other on the device (DeviceTurn without the host in it)
        ~Keep() { if (k2_done) (void)hipEventDestroy(k2_done); }
    };
    // per pipeline: the groups queue their K2 in group order (group 0 first: the groups are cut so that the early ones are the big ones),
    // each behind the event of the one before
    struct K2Order {
        std::mutex mu; std::condition_variable cv; hipEvent_t last = nullptr; int next = 0;
        void reset() { std::lock_guard<std::mutex> lk(mu); next = 0; }
        void abort() { { std::lock_guard<std::mutex> lk(mu); next = 1 << 30; } cv.notify_all(); }      // (a group failed: nobody waits for its turn)
    };
    struct K2Turn {      // holds the order's lock from this group's turn on; passes the turn on when it goes out of scope at the latest
        K2Order* o = nullptr; std::unique_lock<std::mutex> lk; bool passed = true;
        void take(K2Order* order, int ticket) {
            o = order; lk = std::unique_lock<std::mutex>(o->mu);
            if (ticket >= 0) o->cv.wait(lk, [&] { return o->next >= ticket; });
            passed = false;
        }
        void pass() { if (!passed) { passed = true; o->next++; lk.unlock(); o->cv.notify_all(); } }
        bool held() const { return !passed; }
        ~K2Turn() { pass(); }
    };
    int order_ticket = -1;             // this group's place in the order (-1: whoever comes first)
    Keep* keep = nullptr;
    K2Order* k2_order = nullptr;
    static int64_t with_margin(int64_t v) { return v + v / 8 + 1024; }
    static bool order_phase1() { static const bool k2_only = []() { const char* e = std::getenv("HS_ORDER_SCOPE"); return e && std::string(e) == "k2"; }(); return !k2_only; }
    static bool hints_on() { static const bool off = std::getenv("HS_NO_SIZE_HINTS") != nullptr; return !off; }

    int fetch_info() {      // the info block from the device (one transfer + wait)
        const size_t bytes = info_bytes(range_c1 - range_c0);
        if (int rc = grow(h_info, bytes)) return rc;
        Shipment sh; sh.add(h_info.p, d_info.p, bytes);
        if (int rc = sh.launch(stream, 1)) return rc;
        return stream_wait(stream);
    }
    // the columns carrying `flag`: block sums + offsets (the header then holds their number and their entries) ...
    int flag_sums_launch(int flag, int64_t cols_cap) {
        const int n_blocks = (int)((cols_cap + HS_FP_BLOCK - 1) / HS_FP_BLOCK);
        if (int rc = grow(d_blk_cnt, std::max<size_t>(1, (size_t)n_blocks) * 8)) return rc;
        if (int rc = grow(d_blk_ent, std::max<size_t>(1, (size_t)n_blocks) * 8)) return rc;
        if (int rc = kc.begin(HS_K_PACK_COLUMNS, stream)) return rc;
        if (n_blocks > 0)
            hipLaunchKernelGGL(hsdev::k_flag_block_sums, dim3((unsigned)n_blocks), dim3(256), 0, stream, d_col_rec.as<hsdev::hs_colrec_dev>(), d_col_len.as<int32_t>(),
                               dev_header(), flag, d_blk_cnt.as<long long>(), d_blk_ent.as<long long>());
        hipLaunchKernelGGL(hsdev::k_flag_block_offsets, dim3(1), dim3(1024), 0, stream, d_blk_cnt.as<long long>(), d_blk_ent.as<long long>(), n_blocks, dev_header());
        HS_HIP(hipGetLastError());
        return kc.end(20 * cols_cap, stream);      // record + length of every column in
    }
    // ... and packed on the device (d_pk, laid out for the capacities cap_f columns / cap_e entries)
    int pack_launch(int flag, int64_t cols_cap, int64_t cap_f, int64_t cap_e) {
        const int n_blocks = (int)((cols_cap + HS_FP_BLOCK - 1) / HS_FP_BLOCK);
        pk_layout = pack_layout(cap_f, cap_e);
        const PackLayout& L = pk_layout;
        if (int rc = grow(d_pk, L.total)) return rc;
        char* base = (char*)d_pk.p;
        hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, reinterpret_cast<uint4*>(base + L.off), 1ll, 0u);      // (no flagged column: offsets[0] = 0)
        if (n_blocks == 0) { HS_HIP(hipGetLastError()); return HS_OK; }
        if (int rc = kc.begin(HS_K_PACK_COLUMNS, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_pack_flagged, dim3((unsigned)n_blocks), dim3(256), 0, stream, d_col_rec.as<hsdev::hs_colrec_dev>(), d_co.as<int64_t>(),
                           d_col_len.as<int32_t>(), d_ci.as<int32_t>(), d_cc.as<uint8_t>(), dev_header(), flag, d_blk_cnt.as<long long>(),
                           d_blk_ent.as<long long>(), (hsdev::hs_colrec_dev*)(base + L.rec), (int32_t*)(base + L.col), (int64_t*)(base + L.off), (int32_t*)(base + L.idx),
                           (uint8_t*)(base + L.code), cap_f, cap_e);
        HS_HIP(hipGetLastError());
        return kc.end(10 * cap_e + 60 * cap_f, stream);      // the flagged columns' entries in and out, their records
    }

    // The packed candidates as bit sets (k_cand_bits) straight into pinned host memory, with their records and the info block: what
    // loop A reads. The word blocks are bump-allocated on the device (cap_words).
    DBuf d_cb_bits, d_cb_words, d_cb_counter;
    HBuf h_cb;
    struct CbLayout { size_t rec, bits, words, total; int64_t cap_cand, cap_words; } cbl{};
    int cand_bits_launch(int64_t cap_cand, int64_t cap_cand_entries, int64_t cap_words) {
        static_assert(sizeof(hs::CandBits) == sizeof(hsdev::CandBitsDev), "CandBits layout");
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        cbl.cap_cand = cap_cand; cbl.cap_words = cap_words;
        cbl.rec = 256; cbl.bits = cbl.rec + up((size_t)cap_cand * 16); cbl.words = cbl.bits + up((size_t)cap_cand * 32); cbl.total = cbl.words + up((size_t)cap_words * 8);
        if (int rc = grow(d_cb_bits, std::max<size_t>(1, (size_t)cap_cand) * 32)) return rc;
        if (int rc = grow(d_cb_words, std::max<size_t>(1, (size_t)cap_words) * 8)) return rc;
        if (int rc = grow(d_cb_counter, 256)) return rc;
        if (int rc = grow(h_cb, cbl.total)) return rc;
        const char* cb = (const char*)d_cand_pk.p;
        hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, d_cb_counter.as<uint4>(), 16ll, 0u);
        if (cap_cand > 0) {
            if (int rc = kc.begin(HS_K_CAND_BITS, stream)) return rc;
            hipLaunchKernelGGL(hsdev::k_cand_bits, dim3((unsigned)((cap_cand + HS_CB_WAVES - 1) / HS_CB_WAVES)), dim3(64 * HS_CB_WAVES), 0, stream,
                               (const hsdev::hs_colrec_dev*)(cb + cand_layout.rec), (const int64_t*)(cb + cand_layout.off), (const int32_t*)(cb + cand_layout.idx),
                               (const uint8_t*)(cb + cand_layout.code), dev_header(), (long long)cap_cand, b->d_contig_rec_off.as<int32_t>(), b->d_rank_end.as<int2>(),
                               d_cb_bits.as<hsdev::CandBitsDev>(), d_cb_words.as<unsigned long long>(), (long long)cap_words,
                               d_cb_counter.as<unsigned long long>(), (long long)cap_cand_entries);
            HS_HIP(hipGetLastError());
            if (int rc = kc.end(5 * cap_cand_entries + 48 * cap_cand, stream)) return rc;      // the candidates' entries in; record, header and block out
        }
        Shipment sh;
        sh.add(h_cb.p, d_cb_counter.p, 16);
        const long long* n_flagged = reinterpret_cast<const long long*>((const char*)d_info.p + offsetof(hsdev::ColumnsHeader, n_flagged));
        sh.add_counted((char*)h_cb.p + cbl.rec, cb + cand_layout.rec, n_flagged, 16, cap_cand);
        sh.add_counted((char*)h_cb.p + cbl.bits, d_cb_bits.p, n_flagged, 32, cap_cand);
        sh.add_counted((char*)h_cb.p + cbl.words, d_cb_words.p, d_cb_counter.as<long long>(), 8, cap_words);
        if (int rc = grow(h_info, info_bytes(range_c1 - range_c0))) return rc;
        sh.add(h_info.p, d_info.p, info_bytes(range_c1 - range_c0));
        if (int rc = kc.begin(HS_K_SHIP, stream)) return rc;
        if (int rc = sh.launch(stream)) return rc;
        return kc.end(0, stream);
    }
    // after the wait: did the capacities hold?
    bool cand_bits_fit() const { const unsigned long long* cnt = (const unsigned long long*)h_cb.p; return cnt[1] == 0; }
    int64_t cand_bits_words() const { return (int64_t)((const unsigned long long*)h_cb.p)[0]; }
    void cand_bits_result(hs::CvCandidates& out) const {
        out.rec = (const hs_colrec*)((const char*)h_cb.p + cbl.rec);
        out.bits = (const hs::CandBits*)((const char*)h_cb.p + cbl.bits);
        out.words = (const uint64_t*)((const char*)h_cb.p + cbl.words);
    }
    int fetch_candidates(hs::CvCandidates& out) override {      // the candidates of the last extract_candidates() for the host's loop A, after all
        out.bits = nullptr; out.words = nullptr;
        if (cand_count == 0) return HS_OK;
        int64_t cap_words = std::max<int64_t>(with_margin(keep ? keep->cb_words : 0), 16 * cand_count + cand_entries / 4 + 64);
        for (int attempt = 0;; ++attempt) {
            if (int rc = cand_bits_launch(cand_count, cand_entries, cap_words)) return rc;
            if (int rc = stream_wait(stream)) return rc;
            if (cand_bits_fit()) { if (keep) keep->cb_words = cand_bits_words(); cand_bits_result(out); return HS_OK; }
            if (attempt >= 4 || cand_bits_words() <= cap_words) { set_error("candidate bit sets: the blocks do not fit (a column over more than 65535 words of reads?)"); return HS_EINVAL; }
            cap_words = cand_bits_words() + 64;      // (the counter ran on past the capacity: it is the exact need)
        }
    }
    DBuf d_cand_pk;                   // the packed candidates (kept beside d_pk, which the SNPs take later): k_cand_bits and k_loop_a read them
    PackLayout cand_layout{};
    int64_t cand_count = 0, cand_entries = 0;
    std::vector<int32_t> cand_per_contig;
    int extract_candidates(int c0, int c1, const std::vector<int32_t>& min_reads, float thr, hs::CvCandidates& out, float k_ms[3], bool want_entries) override {
        const bool hinted = keep && keep->valid && want_entries && hints_on();
        if (hinted) {
            const int rc = extract_candidates_impl(c0, c1, min_reads, thr, out, k_ms, want_entries, true);
            if (rc != HS_EAGAIN_SIZES) return rc;
            keep->valid = false;      // (the sizes of this range have changed: the careful way, which also takes the new ones)
        }
        return extract_candidates_impl(c0, c1, min_reads, thr, out, k_ms, want_entries, false);
    }
    static constexpr int HS_EAGAIN_SIZES = -1000;      // (internal: a capacity did not hold)
    int extract_candidates_impl(int c0, int c1, const std::vector<int32_t>& min_reads, float thr, hs::CvCandidates& out, float k_ms[3], bool want_entries, bool hinted) {
        static_assert(sizeof(hs_colrec) == sizeof(hsdev::hs_colrec_dev), "hs_colrec layout");
        static_assert(sizeof(hsdev::ColumnsHeader) == 64, "info block layout");
        const int C = c1 - c0;
        range_c0 = c0; range_c1 = c1; n_cols = 0; n_entries = 0; n_gathered = 0; gathered_entries = 0;
        cand_count = 0; cand_entries = 0;
        out = hs::CvCandidates();
        out.contig_n_cand.assign((size_t)C, 0);
        k_ms[0] = k_ms[1] = k_ms[2] = 0;
        const int64_t g0 = b->contig_off[(size_t)c0], g1 = b->contig_off[(size_t)c1];
        if (g1 <= g0 || C <= 0) return HS_OK;
        const int64_t t0 = g0 >> 8, t1 = (g1 + 255) >> 8, nt = t1 - t0;
        if (nt > 0x7fffffff) { set_error("too many tiles in one contig range"); return HS_EINVAL; }
        EventPair e_k2, e_k3, e_k3b;
        if (int rc = e_k2.init()) return rc;
        if (int rc = e_k3.init()) return rc;
        if (int rc = e_k3b.init()) return rc;
        const size_t info_b = info_bytes(C);
        if (int rc = grow(d_info, info_b)) return rc;
        if (own_pileup) { if (int rc = grow(d_min_reads, std::max<size_t>(1, (size_t)C) * 4)) return rc; }
        if (int rc = range_scratch.prepare(nt * 256)) return rc;
        if (int rc = grow(d_tile_ent_sum, (size_t)nt * 4)) return rc;
        if (int rc = grow(d_tile_ebase, ((size_t)nt + 1) * 8)) return rc;
        const int64_t range_pile = b->total_len > 0 ? (int64_t)((double)b->total_pile * (double)(g1 - g0) / (double)b->total_len) : 0;      // one code in per aligned bp of the range (its share of the batch)
        auto k2_launch = [&]() -> int {   // ---- K2 over the tiles of the range: per tile its selected positions (second count >= 4), their depths and the sum of those ----
            hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(256), 0, stream, d_info.as<uint4>(), (long long)(info_b / 16), 0u);      // (header, tie counters, candidates / SNPs / SNP bounds per contig)
            if (own_pileup) { if (int rc = pileup_range_launch(c0, c1, dev_ctg_md(), d_min_reads.as<int32_t>())) return rc; }
            HS_HIP(hipEventRecord(e_k2.a, stream));
            if (int rc = kc.begin(HS_K_COLUMN_STATS, stream)) return rc;
            hipEvent_t k2_done = nullptr;
            if (int rc = kc.end_prepare(range_pile, &k2_done)) return rc;
            if (int rc = column_stats_tiled_launch(b->pile_ptr(), b->tile_off.as<int64_t>(), b->tile_ent.as<hs_tile_entry>(), b->total_len, nullptr, 4,
                                                   range_scratch.tile_cnt.as<int32_t>() /* (only "selection wanted") */, nullptr, nullptr, 0, b->max_depth, &range_scratch,
                                                   e_k2.b, stream, k2_done, t0, t1, g0, g1, d_tile_ent_sum.as<int32_t>(), false, true)) return rc;
            if (int rc = exclusive_scan_launch(range_scratch.tile_cnt.as<int32_t>(), (int)nt, range_scratch.tile_base.as<int64_t>(), range_scratch.scan_scratch, stream)) return rc;
            return exclusive_scan_launch(d_tile_ent_sum.as<int32_t>(), (int)nt, d_tile_ebase.as<int64_t>(), d_scan2, stream);
        };
        int64_t cap_cols = 0, cap_entries = 0;
        K2Turn order_lock;
        if (!hinted) {
            DeviceTurn turn;
            if (int rc = k2_launch()) return rc;
            // the two totals (the last elements of the scans) -> sizes of the column arrays
            hipLaunchKernelGGL(hsdev::k_columns_totals, dim3(1), dim3(64), 0, stream, range_scratch.tile_base.as<int64_t>() + nt, d_tile_ebase.as<int64_t>() + nt, dev_header());
            HS_HIP(hipGetLastError());
            if (int rc = fetch_info()) return rc;
            cap_cols = host_header().n_cols; cap_entries = host_header().n_entries;
        } else {
            // the groups' K2 launches one after the other ON THE DEVICE (each fills it on its own, see DeviceTurn): this group's
            // launch queues behind the event of the group that came before it, no host thread waits for anything
            // HS_ORDER_SCOPE=phase1 (default): not only K2 but the group's whole chain up to the shipment of the candidates runs behind the
            // previous group's -- the groups then reach the host one after the other (candidates every ~0.8 ms) instead of all at the same
            // late moment, and loops A / B of one group run while the device works on the next one's columns. HS_ORDER_SCOPE=k2: K2 only.
            if (k2_order && DeviceTurn::on()) {
                order_lock.take(k2_order, order_ticket);
                if (!keep->k2_done) HS_HIP(hipEventCreateWithFlags(&keep->k2_done, hipEventDisableTiming));
                if (k2_order->last && k2_order->last != keep->k2_done) HS_HIP(hipStreamWaitEvent(stream, k2_order->last, 0));
                if (int rc = k2_launch()) return rc;
                if (!order_phase1()) { HS_HIP(hipEventRecord(keep->k2_done, stream)); k2_order->last = keep->k2_done; order_lock.pass(); }
            } else if (int rc = k2_launch()) return rc;
            cap_cols = with_margin(keep->cols); cap_entries = with_margin(keep->entries);
        }
        if (cap_cols > 0x7fffffff) { set_error("more than 2^31 columns in one contig range"); return HS_EINVAL; }
        if (int rc = grow(d_col_gpos, std::max<size_t>(1, (size_t)cap_cols) * 8)) return rc;
        if (int rc = grow(d_col_rec, std::max<size_t>(1, (size_t)cap_cols) * sizeof(hs_colrec))) return rc;
        if (int rc = grow(d_co, ((size_t)cap_cols + 1) * 8)) return rc;
        if (int rc = grow(d_col_len, std::max<size_t>(1, (size_t)cap_cols) * 4)) return rc;
        if (int rc = grow(d_ci, std::max<size_t>(1, (size_t)cap_entries) * 4)) return rc;
        if (int rc = grow(d_cc, std::max<size_t>(1, (size_t)cap_entries))) return rc;
        if (int rc = grow(d_col_ctg, std::max<size_t>(1, (size_t)cap_cols) * 4)) return rc;
        if (int rc = grow(d_k0, std::max<size_t>(1, (size_t)cap_cols))) return rc;
        if (int rc = grow(d_k1, std::max<size_t>(1, (size_t)cap_cols))) return rc;
        if (int rc = grow(d_c1, std::max<size_t>(1, (size_t)cap_cols) * 4)) return rc;
        if (int rc = grow(d_cand, std::max<size_t>(1, (size_t)cap_cols))) return rc;
        if (int rc = grow(d_ctg_col_off, ((size_t)C + 1) * 8)) return rc;
        if (!own_pileup) { range_pack.add(min_reads, d_min_reads); if (int rc = range_pack.commit(stream)) return rc; }
        // ---- the column list with its CSR offsets, K3 (tile-cooperative gather), K3b (leading codes, reference order), V1 ----
        if (int rc = kc.begin(HS_K_COLUMNS_COMPACT, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_columns_compact, dim3((unsigned)nt), dim3(256), 0, stream, range_scratch.tile_cnt.as<int32_t>(), range_scratch.tile_base.as<int64_t>(),
                           d_tile_ebase.as<int64_t>(), range_scratch.gpos.as<int64_t>(), range_scratch.depth.as<int32_t>(), nt, b->d_contig_off.as<int64_t>(), b->n_contigs,
                           d_col_gpos.as<int64_t>(), d_col_rec.as<hsdev::hs_colrec_dev>(), d_co.as<int64_t>(), d_col_len.as<int32_t>(), dev_header(), cap_cols, cap_entries);
        HS_HIP(hipGetLastError());
        if (int rc = kc.end(20 * nt + 48 * cap_cols, stream)) return rc;      // tile counts and bases in; slot in, position + record + offset + length out per column
        HS_HIP(hipEventRecord(e_k3.a, stream));
        if (cap_cols > 0) {
            if (int rc = kc.begin(HS_K_GATHER_COLUMNS, stream)) return rc;
            hipLaunchKernelGGL(hsdev::k_gather_tiles, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, stream, b->pile_ptr(), b->tile_off.as<int64_t>(),
                               reinterpret_cast<const int4*>(b->tile_ent.as<hs_tile_entry>()), b->tile_lrec.as<int32_t>(), t0, nt, range_scratch.tile_cnt.as<int32_t>(),
                               range_scratch.tile_base.as<int64_t>(), d_col_gpos.as<int64_t>(), d_co.as<int64_t>(), d_ci.as<int32_t>(), d_cc.as<uint8_t>(), dev_header());
            HS_HIP(hipGetLastError());
            // the pileup bytes of the range's tiles in (every one of them once), read index + code out per column entry
            if (int rc = kc.end(range_pile + 5 * (hinted ? keep->entries : cap_entries), stream)) return rc;
        }
        HS_HIP(hipEventRecord(e_k3.b, stream));
        HS_HIP(hipEventRecord(e_k3b.a, stream));
        if (cap_cols > 0) {
            if (int rc = kc.begin(HS_K_COLUMN_TOP3, stream)) return rc;
            const unsigned grid = (unsigned)std::min<int64_t>((cap_cols + 3) / 4, 16384);
            hipLaunchKernelGGL(hsdev::k_column_top3_exact, dim3(grid), dim3(256), 0, stream, d_co.as<int64_t>(), d_col_len.as<int32_t>(), d_cc.as<uint8_t>(),
                               dev_header(), d_col_rec.as<hsdev::hs_colrec_dev>(), dev_tie());
            HS_HIP(hipGetLastError());
            if (int rc = kc.end((hinted ? keep->entries : cap_entries) + 16 * (hinted ? keep->cols : cap_cols), stream)) return rc;
        }
        if (int rc = kc.begin(HS_K_CANDIDATES_SCAN, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_candidates_scan, dim3((unsigned)((std::max<int64_t>(cap_cols, C) + 255) / 256)), dim3(256), 0, stream, d_col_gpos.as<int64_t>(), dev_header(),
                           b->d_contig_off.as<int64_t>(), c0, C, d_min_reads.as<int32_t>(), thr, d_col_rec.as<hsdev::hs_colrec_dev>(), d_col_ctg.as<int32_t>(),
                           d_k0.as<uint8_t>(), d_k1.as<uint8_t>(), d_c1.as<int32_t>(), d_cand.as<uint8_t>(), d_ctg_col_off.as<int64_t>(), dev_ctg_n());
        HS_HIP(hipGetLastError());
        if (int rc = kc.end(43 * (hinted ? keep->cols : cap_cols), stream)) return rc;      // record in and out, the four arrays K4 reads out
        HS_HIP(hipEventRecord(e_k3b.b, stream));
        // ---- the candidates, packed (they stay on the device); the host's loop A gets them as bit sets ----
        if (int rc = flag_sums_launch(HS_COL_CAND, cap_cols)) return rc;
        int64_t cap_cand = 0, cap_cand_entries = 0, cap_words = 0;
        if (!hinted) {
            if (int rc = fetch_info()) return rc;      // (carries the per-contig counts and the tie counters too)
            cap_cand = host_header().n_flagged; cap_cand_entries = host_header().n_flagged_entries;
            cap_words = std::max<int64_t>(with_margin(keep ? keep->cb_words : 0), 16 * cap_cand + cap_cand_entries / 4 + 64);
        } else { cap_cand = with_margin(keep->cand); cap_cand_entries = with_margin(keep->cand_entries); cap_words = with_margin(keep->cb_words); }
        if (int rc = pack_launch(HS_COL_CAND, cap_cols, cap_cand, cap_cand_entries)) return rc;
        cand_layout = pk_layout;
        std::swap(d_cand_pk, d_pk);      // (d_pk is packed again for the SNPs)
        if (want_entries) {
            for (int attempt = 0;; ++attempt) {
                if (int rc = cand_bits_launch(cap_cand, cap_cand_entries, cap_words)) return rc;
                if (order_lock.held()) { HS_HIP(hipEventRecord(keep->k2_done, stream)); k2_order->last = keep->k2_done; order_lock.pass(); }
                if (int rc = stream_wait(stream)) return rc;
                if (cand_bits_fit()) break;
                if (hinted) return HS_EAGAIN_SIZES;
                if (attempt >= 4 || cand_bits_words() <= cap_words) { set_error("candidate bit sets: the blocks do not fit (a column over more than 65535 words of reads?)"); return HS_EINVAL; }
                cap_words = cand_bits_words() + 64;      // (the counter ran on past the capacity: it is the exact need)
            }
        } else if (int rc = fetch_info()) return rc;
        const hsdev::ColumnsHeader& H = host_header();
        if (hinted && (!H.ok || H.n_cols > cap_cols || H.n_entries > cap_entries || H.n_flagged > cap_cand || H.n_flagged_entries > cap_cand_entries)) return HS_EAGAIN_SIZES;
        n_cols = H.n_cols; n_entries = H.n_entries;
        out.n_columns = n_cols; out.n_entries = n_entries;
        n_gathered = (int)n_cols; gathered_entries = n_entries;
        std::memcpy(out.contig_n_cand.data(), host_ctg_n(), (size_t)C * 4);
        if (own_pileup) out.contig_mean_distance.assign(host_ctg_md(), host_ctg_md() + C);
        cand_per_contig = out.contig_n_cand; cand_count = H.n_flagged; cand_entries = H.n_flagged_entries;
        { unsigned long long t2[2]; std::memcpy(t2, (const char*)h_info.p + 64, 16); out.n_tie = (int64_t)t2[0]; out.n_tie_big = (int64_t)t2[1]; }
        out.n_cand = cand_count;
        { static const int64_t zero_off[1] = {0}; out.off = zero_off; }
        if (want_entries && cand_count > 0) cand_bits_result(out);
        if (keep) {
            keep->cols = n_cols; keep->entries = n_entries; keep->cand = cand_count; keep->cand_entries = cand_entries;
            if (want_entries) keep->cb_words = cand_bits_words();
            keep->valid = want_entries;
        }
        kc.flush();
        if (int rc = e_k2.ms(&k_ms[0])) return rc;
        if (int rc = e_k3.ms(&k_ms[1])) return rc;
        return e_k3b.ms(&k_ms[2]);
    }

    // ---- K4 + the merge of the SNP lists + the SNP columns packed (they stay in d_pk for stage 4) ----
    DBuf d_keep;
    int64_t snp_count = 0, snp_entries = 0;       // what d_pk holds after finish_columns (HipSrOps::adopt_columns)
    int finish_columns(const hs::CvPartitionTest& t, bool want_entries, hs::CvSnpSet& out, float* k_ms) override {
        const int C = range_c1 - range_c0;
        out = hs::CvSnpSet();
        out.contig_n_snp.assign((size_t)C, 0);
        if (k_ms) *k_ms = 0;
        if ((int)t.contig_n_reads.size() != C) { set_error("finish_columns: partitions of another contig range"); return HS_EINVAL; }
        snp_count = 0; snp_entries = 0;
        static const int64_t zero_off[1] = {0};
        if (n_cols == 0 || C == 0) {
            pk_layout = pack_layout(0, 0);
            if (int rc = grow(d_pk, pk_layout.total)) return rc;
            hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, reinterpret_cast<uint4*>((char*)d_pk.p + pk_layout.off), 1ll, 0u);
            out.off = zero_off;
            return stream_wait(stream);
        }
        const int n = (int)n_cols;
        DBuf d_po, d_pso, d_ps;
        DBuf d_tab, d_tab_off, d_ctg_nr, d_list;
        UploadPack pk_tab;      // one upload: the partitions and the table offsets (partition_test_launch adds its arrays and commits)
        pk_tab.add(t.part_off, d_po);
        pk_tab.add(t.part_state_off, d_pso);
        pk_tab.add(t.part_state, d_ps);
        if (int rc = grow(d_keep, (size_t)n)) return rc;
        EventPair e; if (int rc = e.init()) return rc;
        HS_HIP(hipEventRecord(e.a, stream));
        if (int rc = partition_test_launch(d_co.as<int64_t>(), d_ci.as<int32_t>(), d_cc.as<uint8_t>(), d_col_ctg.as<int32_t>(), d_k0.as<uint8_t>(),
                                           d_k1.as<uint8_t>(), d_c1.as<int32_t>(), d_cand.as<uint8_t>(), n, d_po, d_pso,
                                           d_ps, t.part_off.data(), t.contig_n_reads.data(), C, d_keep.as<uint8_t>(),
                                           stream, d_tab, d_tab_off, d_ctg_nr, d_list, pk_tab, &kc, gathered_entries, (int64_t)t.part_state.size())) return rc;
        HS_HIP(hipEventRecord(e.b, stream));
        if (int rc = kc.begin(HS_K_SNP_SELECT, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_snp_bounds, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, stream, d_col_rec.as<hsdev::hs_colrec_dev>(), d_col_ctg.as<int32_t>(),
                           d_keep.as<uint8_t>(), n_cols, C, dev_snp_bounds());
        hipLaunchKernelGGL(hsdev::k_snp_flags, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, stream, d_col_rec.as<hsdev::hs_colrec_dev>(), d_col_ctg.as<int32_t>(),
                           d_keep.as<uint8_t>(), n_cols, C, dev_snp_bounds(), dev_ctg_snp());
        HS_HIP(hipGetLastError());
        if (int rc = kc.end(33 * n_cols, stream)) return rc;      // records in (twice) and out, the verdicts in
        if (int rc = flag_sums_launch(HS_COL_SNP, n_cols)) return rc;
        bool hinted = keep && keep->valid && keep->snp > 0 && hints_on();
        for (;;) {
            int64_t cap_snp, cap_snp_entries;
            if (!hinted) {
                if (int rc = fetch_info()) return rc;      // (the partition tables, uploads and lists of this scope are done with: it waits)
                cap_snp = host_header().n_flagged; cap_snp_entries = host_header().n_flagged_entries;
            } else { cap_snp = with_margin(keep->snp); cap_snp_entries = with_margin(keep->snp_entries); }
            if (int rc = pack_launch(HS_COL_SNP, n_cols, cap_snp, cap_snp_entries)) return rc;
            // the SNPs' records and offsets (and, for a caller that writes the .col file, their entries) into pinned memory, with the info block
            const PackLayout& L = pk_layout;
            if (int rc = grow(h_pk, std::max<size_t>(want_entries ? L.total : L.head, 256))) return rc;
            if (int rc = grow(h_info, info_bytes(C))) return rc;
            const long long* n_flagged = reinterpret_cast<const long long*>((const char*)d_info.p + offsetof(hsdev::ColumnsHeader, n_flagged));
            const long long* n_flagged_e = reinterpret_cast<const long long*>((const char*)d_info.p + offsetof(hsdev::ColumnsHeader, n_flagged_entries));
            Shipment sh;
            const char* base = (const char*)d_pk.p;
            sh.add_counted((char*)h_pk.p + L.rec, base + L.rec, n_flagged, 16, cap_snp);
            sh.add_counted((char*)h_pk.p + L.off, base + L.off, n_flagged, 8, cap_snp, 8);
            if (want_entries) {
                sh.add_counted((char*)h_pk.p + L.idx, base + L.idx, n_flagged_e, 4, cap_snp_entries);
                sh.add_counted((char*)h_pk.p + L.code, base + L.code, n_flagged_e, 1, cap_snp_entries);
            }
            sh.add(h_info.p, d_info.p, info_bytes(C));
            if (int rc = kc.begin(HS_K_SHIP, stream)) return rc;
            if (int rc = sh.launch(stream)) return rc;
            if (int rc = kc.end(0, stream)) return rc;
            if (int rc = stream_wait(stream)) return rc;
            if (host_header().n_flagged <= cap_snp && host_header().n_flagged_entries <= cap_snp_entries) break;
            if (!hinted) { set_error("finish_columns: the packed SNP block does not hold the SNPs"); return HS_EINVAL; }
            hinted = false;      // (more SNPs than last time: once more with the counts known)
        }
        const int64_t n_snp = host_header().n_flagged, e_snp = host_header().n_flagged_entries;
        std::memcpy(out.contig_n_snp.data(), host_ctg_snp(), (size_t)C * 4);
        out.n_snp = n_snp; out.n_entries = e_snp;
        snp_count = n_snp; snp_entries = e_snp;
        if (keep) { keep->snp = n_snp; keep->snp_entries = e_snp; }
        if (n_snp > 0) {
            const char* hb = (const char*)h_pk.p;
            out.rec = (const hs_colrec*)(hb + pk_layout.rec); out.off = (const int64_t*)(hb + pk_layout.off);
            if (want_entries) { out.idx = (const int32_t*)(hb + pk_layout.idx); out.code = (const uint8_t*)(hb + pk_layout.code); }
        } else out.off = zero_off;
        kc.flush();
        return k_ms ? e.ms(k_ms) : HS_OK;
    }
    // ---- loop A on the device (hs_kernels_loopa.hip) on the candidates of the last extract_candidates() ----
    bool has_partition_pairs() const override { return true; }
    int partition_pairs(const std::vector<int8_t>& state, const std::vector<int32_t>& more, const std::vector<int32_t>& less, const std::vector<int64_t>& part_off,
                        const std::vector<int32_t>& part_n, const std::vector<int32_t>& pair_a, const std::vector<int32_t>& pair_b, const std::vector<float>& sigma3,
                        std::vector<int32_t>& out) override {
        const int n_pairs = (int)pair_a.size();
        out.assign((size_t)n_pairs * 8, 0);
        if (n_pairs == 0) return HS_OK;
        DBuf d_st, d_mo, d_le, d_po, d_pn, d_pa, d_pb, d_sg, d_out;
        UploadPack pk;
        pk.add(state, d_st); pk.add(more, d_mo); pk.add(less, d_le); pk.add(part_off, d_po); pk.add(part_n, d_pn); pk.add(pair_a, d_pa); pk.add(pair_b, d_pb); pk.add(sigma3, d_sg);
        if (int rc = pk.commit(stream)) return rc;
        if (int rc = d_out.alloc((size_t)n_pairs * 32)) return rc;
        if (int rc = hs_partition_pair_distance(d_st.as<int8_t>(), d_mo.as<int32_t>(), d_le.as<int32_t>(), d_po.as<int64_t>(), d_pn.as<int32_t>(), d_pa.as<int32_t>(),
                                                d_pb.as<int32_t>(), n_pairs, 2, d_sg.as<float>(), d_out.as<int32_t>(), stream)) return rc;
        return d2h_pinned(out.data(), d_out.p, (size_t)n_pairs * 32, stream);
    }
    bool has_robust_partitions() const override { return true; }
    DBuf d_la_parts, d_la_bits, d_la_cnt, d_la_np, d_la_pb, d_la_out_rec, d_la_out_bits, d_la_out_cnt, d_la_diag, d_la_rc, d_la_hdr, d_la_words, d_la_ends;
    HBuf h_la_np, h_la_rec, h_la_bits, h_la_cnt;
    int robust_partitions(const std::vector<int32_t>& contig_n_reads, hs::CvLoopAResult& out, float* k_ms) override {
        static_assert(sizeof(hs::CvPartRecord) == sizeof(hsdev::LoopAPartition), "partition record layouts differ");
        const int C = (int)contig_n_reads.size();
        out = hs::CvLoopAResult();
        out.part_base.assign((size_t)C + 1, 0); out.failed.assign((size_t)C, 0); out.bits_base.assign((size_t)C, 0); out.cnt_base.assign((size_t)C, 0);
        if (k_ms) *k_ms = 0;
        if (C == 0 || cand_count == 0) return HS_OK;
        if (C != range_c1 - range_c0 || (int)cand_per_contig.size() != C) { set_error("robust_partitions: contigs of another range"); return HS_EINVAL; }
        // pools: a contig gets room for a share of its candidates as partitions (most candidates join a partition; one that needs
        // more is done by the host). HS_LOOP_A_POOL_DIV sets the share (default: candidates / 6 + 48)
        static const int pool_div = []() { const char* e = std::getenv("HS_LOOP_A_POOL_DIV"); return e && std::atoi(e) > 0 ? std::atoi(e) : 6; }();
        std::vector<int64_t> cand_off((size_t)C + 1, 0), cap_off((size_t)C + 1, 0), bits_off((size_t)C + 1, 0), cnt_off((size_t)C + 1, 0);
        std::vector<std::pair<int64_t, int>> weight;
        int w_max = 1;
        for (int c = 0; c < C; ++c) {
            const int64_t k = cand_per_contig[(size_t)c];
            const int N = contig_n_reads[(size_t)c];
            const int W = (N + 63) >> 6;
            const bool fits = W <= HS_LA_MAXW;
            const int64_t cap = (k == 0 || !fits) ? 0 : std::min<int64_t>(k, k / pool_div + 48);
            cand_off[(size_t)c + 1] = cand_off[(size_t)c] + k;
            cap_off[(size_t)c + 1] = cap_off[(size_t)c] + cap;
            bits_off[(size_t)c + 1] = bits_off[(size_t)c] + cap * 3 * W;
            cnt_off[(size_t)c + 1] = cnt_off[(size_t)c] + cap * N;
            if (fits) w_max = std::max(w_max, W);
            weight.push_back(std::make_pair(-k, c));
        }
        std::sort(weight.begin(), weight.end());
        std::vector<int32_t> order((size_t)C);
        for (int i = 0; i < C; ++i) order[(size_t)i] = weight[(size_t)i].second;
        DBuf d_co_, d_cap, d_bo, d_cno, d_ord;
        UploadPack pk;
        pk.add(cand_off, d_co_); pk.add(cap_off, d_cap); pk.add(bits_off, d_bo); pk.add(cnt_off, d_cno); pk.add(order, d_ord);
        if (int rc = pk.commit(stream)) return rc;
        if (int rc = grow(d_la_parts, std::max<size_t>(1, (size_t)cap_off.back()) * sizeof(hsdev::LoopAPartition))) return rc;
        if (int rc = grow(d_la_bits, std::max<size_t>(1, (size_t)bits_off.back()) * 8)) return rc;
        if (int rc = grow(d_la_cnt, std::max<size_t>(1, (size_t)cnt_off.back()) * 4)) return rc;
        if (int rc = grow(d_la_np, (size_t)C * 8)) return rc;      // [C] partitions, [C] failed
        if (int rc = grow(d_la_pb, ((size_t)C + 1) * 8)) return rc;
        const char* cb = (const char*)d_cand_pk.p;
        if (int rc = grow(d_la_diag, 128)) return rc;
        HS_HIP(hipMemsetAsync(d_la_diag.p, 0, 128, stream));
        if (int rc = grow(d_la_rc, std::max<size_t>(1, (size_t)cand_count) * 512)) return rc;      // a row of 128 (rank << 8 | code) per candidate
        if (int rc = grow(d_la_ends, std::max<size_t>(1, (size_t)cand_count) * 8)) return rc;
        if (int rc = grow(d_la_hdr, std::max<size_t>(1, (size_t)cand_count) * sizeof(hsdev::LoopAColumn))) return rc;
        if (int rc = grow(d_la_words, std::max<size_t>(1, (size_t)cand_count) * 512)) return rc;
        const size_t lds = (size_t)3 * w_max * HS_LA_SLOTS * 8;
        if (lds > 32 * 1024)
            HS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hsdev::k_loop_a), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        EventPair e; if (int rc = e.init()) return rc;
        HS_HIP(hipEventRecord(e.a, stream));
        if (int rc = kc.begin(HS_K_ROBUST_PARTITIONS, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_loop_a_prepare, dim3((unsigned)((cand_count + 3) / 4)), dim3(256), 0, stream, (const hsdev::hs_colrec_dev*)(cb + cand_layout.rec),
                           (const int64_t*)(cb + cand_layout.off), (const int32_t*)(cb + cand_layout.idx), (const uint8_t*)(cb + cand_layout.code), cand_count,
                           b->d_contig_rec_off.as<int32_t>(), b->d_rank_of.as<int32_t>(), d_la_rc.as<int32_t>(), d_la_hdr.as<hsdev::LoopAColumn>(), d_la_words.as<unsigned long long>(),
                           d_la_ends.as<int32_t>());
        hipLaunchKernelGGL(hsdev::k_loop_a, dim3((unsigned)C), dim3(64), lds, stream, (const hsdev::hs_colrec_dev*)(cb + cand_layout.rec), d_la_rc.as<int32_t>(),
                           d_la_ends.as<int32_t>(), d_la_hdr.as<hsdev::LoopAColumn>(), d_la_words.as<unsigned long long>(),
                           d_co_.as<int64_t>(), range_c0, C, b->d_contig_rec_off.as<int32_t>(),
                           b->d_orig_of.as<int32_t>(), b->d_read_end.as<int32_t>(), d_ord.as<int32_t>(), d_cap.as<int64_t>(), d_bo.as<int64_t>(),
                           d_cno.as<int64_t>(), d_la_parts.as<hsdev::LoopAPartition>(), d_la_bits.as<unsigned long long>(), d_la_cnt.as<int32_t>(), d_la_np.as<int32_t>(),
                           d_la_np.as<int32_t>() + C, w_max, d_la_diag.as<unsigned long long>());
        HS_HIP(hipGetLastError());
        if (int rc = kc.end(5 * cand_entries + 16 * cand_count, stream)) return rc;      // the candidates' entries (idx + code) and records once
        hipLaunchKernelGGL(hsdev::k_loop_a_scan, dim3(1), dim3(64), 0, stream, d_la_np.as<int32_t>(), C, d_la_pb.as<int64_t>());
        HS_HIP(hipGetLastError());
        HS_HIP(hipEventRecord(e.b, stream));
        // the counts first, then the partitions themselves, packed
        if (int rc = grow(h_la_np, (size_t)C * 8 + ((size_t)C + 1) * 8)) return rc;
        HS_HIP(HS_COPY_ASYNC(h_la_np.p, d_la_np.p, (size_t)C * 8, hipMemcpyDeviceToHost, stream));
        HS_HIP(HS_COPY_ASYNC((char*)h_la_np.p + (size_t)C * 8, d_la_pb.p, ((size_t)C + 1) * 8, hipMemcpyDeviceToHost, stream));
        if (int rc = stream_wait(stream)) return rc;
        const int32_t* h_np = (const int32_t*)h_la_np.p;
        std::memcpy(out.failed.data(), h_np + C, (size_t)C * 4);
        std::memcpy(out.part_base.data(), (const char*)h_la_np.p + (size_t)C * 8, ((size_t)C + 1) * 8);
        const int64_t n_parts = out.part_base[(size_t)C];
        int64_t tb = 0, tc = 0;
        for (int c = 0; c < C; ++c) {
            const int64_t P = h_np[c];
            const int N = contig_n_reads[(size_t)c];
            out.bits_base[(size_t)c] = tb; out.cnt_base[(size_t)c] = tc;
            tb += P * 3 * ((N + 63) >> 6); tc += P * N;
        }
        if (int rc = grow(d_la_out_rec, std::max<size_t>(1, (size_t)n_parts) * sizeof(hsdev::LoopAPartition))) return rc;
        if (int rc = grow(d_la_out_bits, std::max<size_t>(1, (size_t)tb) * 8)) return rc;
        if (int rc = grow(d_la_out_cnt, std::max<size_t>(1, (size_t)tc) * 4)) return rc;
        DBuf d_bb, d_cb2;
        UploadPack pk2;
        pk2.add(out.bits_base, d_bb); pk2.add(out.cnt_base, d_cb2);
        if (int rc = pk2.commit(stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_loop_a_pack, dim3((unsigned)C), dim3(256), 0, stream, d_la_np.as<int32_t>(), range_c0, b->d_contig_rec_off.as<int32_t>(), d_cap.as<int64_t>(),
                           d_bo.as<int64_t>(), d_cno.as<int64_t>(), d_la_parts.as<hsdev::LoopAPartition>(), d_la_bits.as<unsigned long long>(), d_la_cnt.as<int32_t>(),
                           d_la_pb.as<int64_t>(), d_bb.as<int64_t>(), d_cb2.as<int64_t>(), d_la_out_rec.as<hsdev::LoopAPartition>(), d_la_out_bits.as<unsigned long long>(),
                           d_la_out_cnt.as<int32_t>());
        HS_HIP(hipGetLastError());
        if (int rc = grow(h_la_rec, std::max<size_t>(1, (size_t)n_parts) * sizeof(hs::CvPartRecord))) return rc;
        if (int rc = grow(h_la_bits, std::max<size_t>(1, (size_t)tb) * 8)) return rc;
        if (int rc = grow(h_la_cnt, std::max<size_t>(1, (size_t)tc) * 4)) return rc;
        if (n_parts) HS_HIP(HS_COPY_ASYNC(h_la_rec.p, d_la_out_rec.p, (size_t)n_parts * sizeof(hs::CvPartRecord), hipMemcpyDeviceToHost, stream));
        if (tb) HS_HIP(HS_COPY_ASYNC(h_la_bits.p, d_la_out_bits.p, (size_t)tb * 8, hipMemcpyDeviceToHost, stream));
        if (tc) HS_HIP(HS_COPY_ASYNC(h_la_cnt.p, d_la_out_cnt.p, (size_t)tc * 4, hipMemcpyDeviceToHost, stream));
        if (int rc = stream_wait(stream)) return rc;      // (the offset tables of this scope are done with)
        out.rec = (const hs::CvPartRecord*)h_la_rec.p; out.bits = (const uint64_t*)h_la_bits.p; out.cnt = (const int32_t*)h_la_cnt.p;
#ifdef HS_LA_DIAG
        {
            unsigned long long dg[10];
            if (int rc = d2h_pinned(dg, d_la_diag.p, sizeof dg, stream)) return rc;
            std::fprintf(stderr, "[hs la] cycles: loads+window %llu, build %llu, compare %llu, exact %llu, verdict+evict %llu, augment %llu, create %llu, end %llu; candidates %llu, exact lanes %llu\n",
                         dg[0], dg[1], dg[2], dg[3], dg[4], dg[5], dg[6], dg[7], dg[8], dg[9]);
        }
#endif
        kc.flush();
        return k_ms ? e.ms(k_ms) : HS_OK;
    }
};

// ---------------------------------------------------------------------------------------------------
// stage 4 on the device: every clustering window in its local index space (hs_driver.h: SrWindowSet)
// ---------------------------------------------------------------------------------------------------
// The read graphs of a call: ONE CSR over the rows (window, masked read) with local neighbour ids, the visiting order of
// every window, and the window tables the clustering kernels index. Built by K6 from the resident sim / diff matrices (the
// few rows whose result depends on std::sort's arrangement of equal distances are resolved on the host and patched in) plus
// the rows a caller brings for windows on the low-memory path.
struct GraphRows {
    DBuf d_oo, d_n, d_wc, d_row0, d_ids, d_rw, d_bo, d_fe, d_rank, d_rank_off;   // views into `pack`
    UploadPack pack;
    DBuf d_off, d_nbr, d_visit, d_visit_n, d_prog_info, d_prog_bytes, d_prog_steps, d_prog_adj;
    int64_t rows = 0, rows_dev = 0, total = 0;      // total: neighbour entries of the CSR -- an upper bound while !total_exact (the exact number is d_off[rows])
    bool total_exact = true;
    int W = 0, max_m = 1;
    std::vector<int32_t> win_m;    // [W]
    // what the build leaves behind on the stream: the object keeps it until it dies (or builds again), so that no host wait is needed
    // just to give temporaries back to the pool
    DBuf d_bits, d_amb, d_deg, d_scan, d_wmo, d_ltw, d_lti, d_ltj, d_wsim, d_wdiff, d_stage, d_src, d_len, d_dst, d_os, d_od, d_pb, d_pm, d_pi, d_pj;
    UploadPack pk_amb, pk_patch;
    HBuf h_amb, h_rows, h_deg, h_nbr;
    std::vector<int32_t> row_win;
    std::vector<int64_t> win_bits_off, win_mat_off;
    std::vector<int32_t> lt_w, lt_i, lt_j;
    int Wd = 0, Wmx = 0, rows_mx = 0;
    size_t amb_cap = 0;
    int64_t stage_cap = 0, stage_hint = 0;      // entries (sim / diff pairs) the undecided rows may stage / staged last time
    EventPair ev;
    bool timed = false, begun = false;
    const int32_t* d_sim = nullptr; const int32_t* d_diff = nullptr; int es = 1;
    std::vector<int64_t> ctg_out_off; std::vector<int32_t> ctg_n;
    float* k_ms = nullptr;
};

// the bit rows of the call (K5a), for the windows of the low-memory path
struct PlaneRows { const uint64_t* d_alt = nullptr; const uint64_t* d_ref = nullptr; const int64_t* d_plane_off = nullptr; const int32_t* d_words = nullptr; };

// K6 in two halves. begin: the window tables go up, the row kernels run and -- in the same pass -- stage the sim / diff entries of the
// rows they cannot decide; count, list and staged entries are shipped to pinned memory. Nothing is waited for: the caller has host
// work of its own to do meanwhile (the Chinese-Whispers chain of the call is planned from the window plans alone).
static int graph_rows_begin(const int32_t* d_sim, const int32_t* d_diff, const std::vector<int64_t>& ctg_out_off, const std::vector<int32_t>& ctg_n_matrix,
                            const hs::SrWindowSet& ws, GraphRows& G, hipStream_t stream, float* k_ms, KernelClock* kc = nullptr,
                            const PlaneRows* planes = nullptr, int es = 1 /* element stride of d_sim / d_diff (2: pairs in one array) */) {
    G.ctg_n = ws.ctg_reads.empty() ? ctg_n_matrix : ws.ctg_reads;      // reads of every contig (the matrix list has 0 for low-memory contigs)
    G.ctg_out_off = ctg_out_off;
    const std::vector<int32_t>& ctg_n = G.ctg_n;
    const int W = (int)ws.win_contig.size();
    G.W = W; G.rows = ws.rows(); G.total = 0; G.total_exact = true; G.max_m = 1; G.begun = true; G.timed = false; G.k_ms = k_ms;
    G.d_sim = d_sim; G.d_diff = d_diff; G.es = es;
    if (G.rows > 0x7fffffff) { set_error("read graphs: too many rows"); return HS_EINVAL; }
    const int rows = (int)G.rows;
    const int Wd = ws.n_dev_windows;
    const int Wmx = ws.ctg_reads.empty() ? Wd : ws.n_matrix_windows;      // [0, Wmx): sim / diff of the contig; [Wmx, Wd): window-local matrices (low-memory path)
    const int rows_dev = (int)ws.win_row0[(size_t)Wd];
    const int rows_mx = (int)ws.win_row0[(size_t)Wmx];
    G.rows_dev = rows_dev; G.Wd = Wd; G.Wmx = Wmx; G.rows_mx = rows_mx;
    if (Wmx < Wd && (!planes || !planes->d_alt)) { set_error("read graphs: low-memory windows without bit rows"); return HS_EINVAL; }
    if ((int64_t)ws.host_off.size() != (int64_t)(rows - rows_dev) + 1 && rows != rows_dev) { set_error("read graphs: host rows do not match the window set"); return HS_EINVAL; }
    std::vector<int32_t>& row_win = G.row_win; row_win.assign((size_t)rows_dev, 0);
    std::vector<int64_t>& win_bits_off = G.win_bits_off; win_bits_off.assign((size_t)Wd + 1, 0);
    G.win_m.resize((size_t)W);
    int max_m_dev = 1, max_len = 1;
    for (int w = 0; w < W; ++w) {
        const int64_t m0 = ws.win_row0[(size_t)w], m = ws.win_row0[(size_t)w + 1] - m0;
        G.win_m[(size_t)w] = (int32_t)m;
        G.max_m = std::max(G.max_m, (int)m);
        if (w < Wd) {
            for (int64_t r = 0; r < m; ++r) row_win[(size_t)(m0 + r)] = w;
            win_bits_off[(size_t)w + 1] = win_bits_off[(size_t)w] + m * ((m + 63) >> 6);
            max_m_dev = std::max(max_m_dev, (int)m);
            max_len = std::max(max_len, w < Wmx ? ctg_n[(size_t)ws.win_contig[(size_t)w]] : (int)m);
        }
    }
    // low-memory windows: an m x m sim / diff per window, every 64 x 64 tile of it one workgroup
    std::vector<int64_t>& win_mat_off = G.win_mat_off; win_mat_off.assign((size_t)Wd + 1, 0);
    std::vector<int32_t>& lt_w = G.lt_w; std::vector<int32_t>& lt_i = G.lt_i; std::vector<int32_t>& lt_j = G.lt_j;
    lt_w.clear(); lt_i.clear(); lt_j.clear();
    for (int w = 0; w < Wd; ++w) {
        const int64_t m = w >= Wmx ? G.win_m[(size_t)w] : 0;
        win_mat_off[(size_t)w + 1] = win_mat_off[(size_t)w] + m * m;
        const int nt = (int)((m + 63) / 64);
        for (int i = 0; i < nt; ++i) for (int j = 0; j < nt; ++j) { lt_w.push_back(w); lt_i.push_back(i); lt_j.push_back(j); }
    }
    if (Wmx < Wd) { G.pack.add(win_mat_off, G.d_wmo); G.pack.add(lt_w, G.d_ltw); G.pack.add(lt_i, G.d_lti); G.pack.add(lt_j, G.d_ltj); }
    G.pack.add(ctg_out_off, G.d_oo);
    G.pack.add(ctg_n, G.d_n);
    G.pack.add(ws.win_contig, G.d_wc);
    G.pack.add(ws.win_row0, G.d_row0);
    G.pack.add(ws.mask_ids, G.d_ids);
    G.pack.add(row_win, G.d_rw);
    G.pack.add(win_bits_off, G.d_bo);
    G.pack.add(ws.win_final_empty, G.d_fe);
    G.pack.add(ws.rank, G.d_rank);
    G.pack.add(ws.ctg_rank_off, G.d_rank_off);
    if (int rc = G.pack.commit(stream)) return rc;
    if (int rc = G.d_off.alloc(((size_t)rows + 1) * 8 + 32)) return rc;      // (+ room for the 16-byte granules of a shipment of its last element)
    if (int rc = G.d_visit.alloc(std::max<size_t>((size_t)rows, 1) * 4)) return rc;
    if (int rc = G.d_visit_n.alloc(std::max<size_t>((size_t)W, 1) * 4)) return rc;
    if (rows == 0) { hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, G.d_off.as<uint4>(), 1ll, 0u); HS_HIP(hipGetLastError()); return HS_OK; }
    if (int rc = G.d_deg.alloc((size_t)rows * 4)) return rc;
    if (int rc = G.ev.init()) return rc;
    if (rows_dev > 0) {
        if (rows_mx > 0 && !d_sim) { set_error("read graphs before simdiff"); return HS_EINVAL; }
        const size_t bits_bytes = ((size_t)win_bits_off.back() * 8 + 15) & ~(size_t)15;
        if (int rc = G.d_bits.alloc(std::max<size_t>(bits_bytes, 16))) return rc;
        // the block the host reads: [rows left to it][entries staged] | rows [amb_cap] | stage offsets [amb_cap] ; then the staged entries
        G.amb_cap = (size_t)rows_dev;
        G.stage_cap = std::max<int64_t>(G.stage_hint + G.stage_hint / 2, 64 * (int64_t)max_len);
        const size_t amb_bytes = 16 + (((size_t)G.amb_cap * 4 + 15) & ~(size_t)15) + (((size_t)G.amb_cap * 8 + 15) & ~(size_t)15);      // (every part a whole number of 16-byte granules: the parts of a shipment must not share one)
        if (int rc = G.d_amb.alloc(amb_bytes)) return rc;
        if (int rc = G.d_stage.alloc((size_t)G.stage_cap * 8 + 32)) return rc;
        hipLaunchKernelGGL(hsdev::k_fill16, dim3((unsigned)std::min<size_t>(256, bits_bytes / 4096 + 1)), dim3(256), 0, stream, G.d_bits.as<uint4>(), (long long)(std::max<size_t>(bits_bytes, 16) / 16), 0u);
        hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, G.d_amb.as<uint4>(), 1ll, 0u);
        unsigned long long* amb_head = G.d_amb.as<unsigned long long>();
        int32_t* amb_rows = reinterpret_cast<int32_t*>((char*)G.d_amb.p + 16);
        long long* amb_so = reinterpret_cast<long long*>((char*)G.d_amb.p + 16 + (((size_t)G.amb_cap * 4 + 15) & ~(size_t)15));
        int32_t* st_sim = G.d_stage.as<int32_t>(); int32_t* st_diff = st_sim + (((size_t)G.stage_cap + 3) & ~(size_t)3);
        // per-wave LDS: cap distances + cap totals. Four waves per workgroup while they fit, else one; windows wider than that
        // (m > 7168 masked reads) send their rows to the host
        int cap = ((max_m_dev + 63) / 64) * 64, waves = 4;
        if ((size_t)cap * 8 * 4 > 57344) waves = 1;
        if ((size_t)cap * 8 > 57344) cap = 7168;
        const float below = 1 - ws.error_rate * 2;   // :778
        HS_HIP(hipEventRecord(G.ev.a, stream));
        if (rows_mx > 0) {
            if (kc) { if (int rc = kc->begin(HS_K_GRAPH_ROWS, stream)) return rc; }
            hipLaunchKernelGGL(hsdev::k_read_graph_rows<false>, dim3((rows_mx + waves - 1) / waves), dim3(64 * waves), (size_t)cap * 8 * waves, stream, d_sim, d_diff,
                               G.d_oo.as<int64_t>(), G.d_n.as<int32_t>(), G.d_wc.as<int32_t>(), G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(), G.d_rw.as<int32_t>(),
                               G.d_bo.as<int64_t>(), 0, rows_mx, below, cap, G.d_bits.as<unsigned long long>(), amb_head, amb_rows, (int)G.amb_cap,
                               (const int64_t*)nullptr, es, amb_so, st_sim, st_diff, (long long)G.stage_cap);
            HS_HIP(hipGetLastError());
            if (kc) {   // per row: the sim and diff entries of the window's m reads in, m link bits out
                int64_t by = 0;
                for (int w = 0; w < Wmx; ++w) { const int64_t m = G.win_m[(size_t)w]; by += m * (8 * m + (m + 7) / 8); }
                if (int rc = kc->end(by, stream)) return rc;
            }
        }
        if (rows_dev > rows_mx) {
            // create_read_graph_low_memory: the window-local matrices from the bit rows, then the same row kernel with that path's distance
            const size_t mat = (size_t)win_mat_off.back();
            if (int rc = G.d_wsim.alloc(std::max<size_t>(mat, 1) * 8)) return rc;      // (sim, diff) pairs: a row kernel's two reads of a pair share a line
            G.d_wdiff.release(); G.d_wdiff.p = (char*)G.d_wsim.p + 4; G.d_wdiff.bytes = 0; G.d_wdiff.cap = 0; G.d_wdiff.view = true;
            if (kc) { if (int rc = kc->begin(HS_K_SIMDIFF, stream)) return rc; }
            hipLaunchKernelGGL(hsdev::k_simdiff_windows, dim3((unsigned)lt_w.size()), dim3(256), 0, stream, planes->d_alt, planes->d_ref, planes->d_plane_off, planes->d_words,
                               G.d_wc.as<int32_t>(), G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(), G.d_wmo.as<int64_t>(), G.d_ltw.as<int32_t>(), G.d_lti.as<int32_t>(),
                               G.d_ltj.as<int32_t>(), G.d_wsim.as<int32_t>(), G.d_wdiff.as<int32_t>(), 2);
            HS_HIP(hipGetLastError());
            if (kc) { if (int rc = kc->end(8 * (int64_t)mat, stream)) return rc; }
            if (kc) { if (int rc = kc->begin(HS_K_GRAPH_ROWS, stream)) return rc; }
            const int n_lm = rows_dev - rows_mx;
            hipLaunchKernelGGL(hsdev::k_read_graph_rows<true>, dim3((n_lm + waves - 1) / waves), dim3(64 * waves), (size_t)cap * 8 * waves, stream, G.d_wsim.as<int32_t>(),
                               G.d_wdiff.as<int32_t>(), G.d_oo.as<int64_t>(), G.d_n.as<int32_t>(), G.d_wc.as<int32_t>(), G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(),
                               G.d_rw.as<int32_t>(), G.d_bo.as<int64_t>(), rows_mx, n_lm, below, cap, G.d_bits.as<unsigned long long>(), amb_head, amb_rows, (int)G.amb_cap,
                               G.d_wmo.as<int64_t>(), 2, amb_so, st_sim, st_diff, (long long)G.stage_cap);
            HS_HIP(hipGetLastError());
            if (kc) { if (int rc = kc->end(8 * (int64_t)mat, stream)) return rc; }
        }
        HS_HIP(hipEventRecord(G.ev.b, stream));
        G.timed = true;
        // count, list, stage offsets and the staged entries in one shipment
        const size_t rows_part = ((size_t)G.amb_cap * 4 + 15) & ~(size_t)15, stage_half = (((size_t)G.stage_cap + 3) & ~(size_t)3) * 4;
        if (int rc = G.h_amb.alloc(amb_bytes + 2 * stage_half + 64)) return rc;
        Shipment sh;
        char* hb = (char*)G.h_amb.p;
        sh.add(hb, G.d_amb.p, 16);
        sh.add_counted(hb + 16, (char*)G.d_amb.p + 16, reinterpret_cast<const long long*>(amb_head), 4, (long long)G.amb_cap);
        sh.add_counted(hb + 16 + rows_part, (char*)G.d_amb.p + 16 + rows_part, reinterpret_cast<const long long*>(amb_head), 8, (long long)G.amb_cap);
        sh.add_counted(hb + amb_bytes, st_sim, reinterpret_cast<const long long*>(amb_head + 1), 4, G.stage_cap);
        sh.add_counted(hb + amb_bytes + stage_half, st_diff, reinterpret_cast<const long long*>(amb_head + 1), 4, G.stage_cap);
        if (kc) { if (int rc = kc->begin(HS_K_SHIP, stream)) return rc; }
        if (int rc = sh.launch(stream, 16)) return rc;
        if (kc) { if (int rc = kc->end(0, stream)) return rc; }
    }
    return HS_OK;
}

// end: the one wait of K6; the undecided rows are resolved with std::sort itself on the staged entries and patched in; degrees, scan,
// fill (the neighbour array sized by its upper bound sum m^2 where that is moderate: no round trip for the total), visiting orders.
// Nothing is waited for at the end: what the stream still reads lives in G.
static int graph_rows_end(const hs::SrWindowSet& ws, GraphRows& G, hipStream_t stream, int64_t* rows_on_host, KernelClock* kc = nullptr) {
    if (rows_on_host) *rows_on_host = 0;
    const int rows = (int)G.rows, rows_dev = (int)G.rows_dev, rows_mx = G.rows_mx, Wd = G.Wd, Wmx = G.Wmx, W = G.W;
    if (rows == 0) return HS_OK;
    const std::vector<int32_t>& ctg_n = G.ctg_n;
    const std::vector<int32_t>& row_win = G.row_win;
    const std::vector<int64_t>& win_bits_off = G.win_bits_off;
    const std::vector<int64_t>& win_mat_off = G.win_mat_off;
    if (rows_dev > 0) {
        if (int rc = stream_wait(stream)) return rc;
        const size_t rows_part = ((size_t)G.amb_cap * 4 + 15) & ~(size_t)15, amb_bytes = 16 + rows_part + (((size_t)G.amb_cap * 8 + 15) & ~(size_t)15), stage_half = (((size_t)G.stage_cap + 3) & ~(size_t)3) * 4;
        const char* hb = (const char*)G.h_amb.p;
        const int64_t n_amb = (int64_t)((const unsigned long long*)hb)[0];
        const int64_t staged = (int64_t)((const unsigned long long*)hb)[1];
        G.stage_hint = staged;
        if (n_amb > (int64_t)G.amb_cap) { set_error("read graphs: more undecided rows than rows"); return HS_EINVAL; }
        if (n_amb > 0) {
            // rows where std::sort's arrangement of equal distances decides: exactly what the reference does, on their sim / diff entries
            const int32_t* amb_rows = (const int32_t*)(hb + 16);
            const long long* amb_so = (const long long*)(hb + 16 + rows_part);
            const int32_t* st_sim = (const int32_t*)(hb + amb_bytes); const int32_t* st_diff = (const int32_t*)(hb + amb_bytes + stage_half);
            std::vector<int32_t> order((size_t)n_amb);
            for (int64_t k = 0; k < n_amb; ++k) order[(size_t)k] = (int32_t)k;
            std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return amb_rows[x] < amb_rows[y]; });      // (the links are ORed in: any order gives the same bits; sorted for the fetch below)
            // rows whose entries found no room in the staging area (a first call, or more of them than ever before): fetched the old way
            std::vector<int32_t> late;
            for (int32_t k : order) if (amb_so[k] < 0) late.push_back(k);
            std::vector<int64_t> late_dst((size_t)late.size() + 1, 0);
            const int32_t* late_sim = nullptr; const int32_t* late_diff = nullptr;
            if (!late.empty()) {
                std::vector<int64_t> src(late.size()); std::vector<int32_t> len(late.size());
                int n_mx = 0;
                for (size_t q = 0; q < late.size(); ++q) {
                    const int row = amb_rows[late[q]];
                    const int w = row_win[(size_t)row];
                    const int c = ws.win_contig[(size_t)w];
                    const int N = ctg_n[(size_t)c];
                    const int r1 = ws.mask_ids[(size_t)row];
                    if (row < rows_mx) { src[q] = G.ctg_out_off[(size_t)c] + (int64_t)r1 * N; len[q] = N; n_mx++; }
                    else { const int64_t m = G.win_m[(size_t)w]; src[q] = win_mat_off[(size_t)w] + (int64_t)(row - ws.win_row0[(size_t)w]) * m; len[q] = (int32_t)m; }
                    late_dst[q + 1] = late_dst[q] + len[q];
                }
                G.pk_amb.add(src, G.d_src); G.pk_amb.add(len, G.d_len); G.pk_amb.add(late_dst, G.d_dst);
                if (int rc = G.pk_amb.commit(stream)) return rc;
                if (int rc = G.d_os.alloc((size_t)late_dst.back() * 4 + 16)) return rc;
                if (int rc = G.d_od.alloc((size_t)late_dst.back() * 4 + 16)) return rc;
                if (n_mx > 0)
                    hipLaunchKernelGGL(hsdev::k_read_graph_fetch_rows, dim3(n_mx), dim3(256), 0, stream, G.d_sim, G.d_diff, G.d_src.as<int64_t>(), G.d_len.as<int32_t>(),
                                       G.d_dst.as<int64_t>(), G.d_os.as<int32_t>(), G.d_od.as<int32_t>(), G.es);
                if ((int)late.size() > n_mx)
                    hipLaunchKernelGGL(hsdev::k_read_graph_fetch_rows, dim3((unsigned)late.size() - n_mx), dim3(256), 0, stream, G.d_wsim.as<int32_t>(), G.d_wdiff.as<int32_t>(),
                                       G.d_src.as<int64_t>() + n_mx, G.d_len.as<int32_t>() + n_mx, G.d_dst.as<int64_t>() + n_mx, G.d_os.as<int32_t>(), G.d_od.as<int32_t>(), 2);
                HS_HIP(hipGetLastError());
                const size_t row_bytes = (((size_t)late_dst.back() * 4) + 15) & ~(size_t)15;
                if (int rc = G.h_rows.alloc(2 * row_bytes + 16)) return rc;
                Shipment sh; sh.add(G.h_rows.p, G.d_os.p, row_bytes); sh.add((char*)G.h_rows.p + row_bytes, G.d_od.p, row_bytes);
                if (int rc = sh.launch(stream, 16)) return rc;
                if (int rc_w = stream_wait(stream)) return rc_w;
                late_sim = (const int32_t*)G.h_rows.p; late_diff = (const int32_t*)((const char*)G.h_rows.p + row_bytes);
                G.stage_hint = std::max<int64_t>(G.stage_hint, staged + late_dst.back());
            }
            std::vector<int64_t> pbase; std::vector<int32_t> pmw, pi, pj;
            std::vector<uint8_t> mask;
            std::vector<int> picked;
            size_t late_q = 0;
            for (int32_t k : order) {
                const int row = amb_rows[k];
                const int w = row_win[(size_t)row];
                const int64_t m0 = ws.win_row0[(size_t)w];
                const int m = (int)(ws.win_row0[(size_t)w + 1] - m0);
                const int32_t* ids = ws.mask_ids.data() + m0;
                const int N = ctg_n[(size_t)ws.win_contig[(size_t)w]];
                const int32_t* rs; const int32_t* rd;
                if (late_q < late.size() && late[late_q] == k) { rs = late_sim + late_dst[late_q]; rd = late_diff + late_dst[late_q]; late_q++; }
                else { rs = st_sim + amb_so[k]; rd = st_diff + amb_so[k]; }
                mask.assign((size_t)N, 0);
                for (int j = 0; j < m; ++j) mask[(size_t)ids[j]] = 1;
                if (row < rows_mx) hs::sr_pick_row_sorted(rs, rd, N, ids[row - m0], mask.data(), ws.error_rate, picked);
                else hs::sr_pick_row_sorted_low_memory(rs, rd, ids, m, (int)(row - m0), N, mask.data(), ws.error_rate, picked);
                for (int nb : picked) {
                    const int j = (int)(std::lower_bound(ids, ids + m, nb) - ids);
                    pbase.push_back(win_bits_off[(size_t)w]); pmw.push_back((m + 63) >> 6); pi.push_back((int32_t)(row - m0)); pj.push_back(j);
                }
            }
            if (!pi.empty()) {
                G.pk_patch.add(pbase, G.d_pb); G.pk_patch.add(pmw, G.d_pm); G.pk_patch.add(pi, G.d_pi); G.pk_patch.add(pj, G.d_pj);
                if (int rc = G.pk_patch.commit(stream)) return rc;
                const int np = (int)pi.size();
                hipLaunchKernelGGL(hsdev::k_read_graph_patch, dim3((np + 255) / 256), dim3(256), 0, stream, G.d_pb.as<int64_t>(), G.d_pm.as<int32_t>(), G.d_pi.as<int32_t>(),
                                   G.d_pj.as<int32_t>(), np, G.d_bits.as<unsigned long long>());
                HS_HIP(hipGetLastError());
            }
            if (rows_on_host) *rows_on_host = n_amb;
            if (std::getenv("HS_TIMING") && Wmx < Wd) {
                int64_t n_lm = 0; for (int64_t k = 0; k < n_amb; ++k) n_lm += amb_rows[k] >= rows_mx;
                std::fprintf(stderr, "[hs timing] sr: low-memory path on the device: %d windows, %d rows, %ld of them resolved on the host (NaN distances / std::sort ties)\n",
                             Wd - Wmx, rows_dev - rows_mx, (long)n_lm);
            }
        } else if (std::getenv("HS_TIMING") && Wmx < Wd)
            std::fprintf(stderr, "[hs timing] sr: low-memory path on the device: %d windows, %d rows, none resolved on the host\n", Wd - Wmx, rows_dev - rows_mx);
        if (kc) { if (int rc = kc->begin(HS_K_GRAPH_CSR, stream)) return rc; }
        hipLaunchKernelGGL(hsdev::k_read_graph_degrees, dim3((rows_dev + 255) / 256), dim3(256), 0, stream, G.d_bits.as<unsigned long long>(), G.d_rw.as<int32_t>(),
                           G.d_row0.as<int64_t>(), G.d_bo.as<int64_t>(), rows_dev, G.d_deg.as<int32_t>());
        HS_HIP(hipGetLastError());
        if (kc) { if (int rc = kc->end((int64_t)win_bits_off.back() * 8 + 4 * (int64_t)rows_dev, stream)) return rc; }
    }
    if (rows > rows_dev) {   // degrees of the rows the host brings
        const size_t nh = (size_t)(rows - rows_dev);
        if (int rc = G.h_deg.alloc(nh * 4 + 16)) return rc;
        int32_t* hd = (int32_t*)G.h_deg.p;
        for (size_t r = 0; r < nh; ++r) hd[r] = (int32_t)(ws.host_off[r + 1] - ws.host_off[r]);
        HS_HIP(HS_COPY_ASYNC(G.d_deg.as<int32_t>() + rows_dev, hd, nh * 4, hipMemcpyHostToDevice, stream));
    }
    if (int rc = exclusive_scan_launch(G.d_deg.as<int32_t>(), rows, G.d_off.as<int64_t>(), G.d_scan, stream)) return rc;
    // the neighbour array: at most sum m^2 entries for the device rows; the exact number only has to be known here when host rows have to
    // be placed behind them (contigs whose low-memory graphs the host builds) or when the bound is unreasonable
    int64_t bound = (int64_t)ws.host_nbr.size();
    for (int w = 0; w < Wd; ++w) bound += (int64_t)G.win_m[(size_t)w] * G.win_m[(size_t)w];
    int64_t total = bound;
    G.total_exact = false;
    if (!ws.host_nbr.empty() || bound > ((int64_t)1 << 28)) {
        if (int rc = d2h_pinned(&total, G.d_off.as<int64_t>() + rows, 8, stream)) return rc;
        G.total_exact = true;
    }
    G.total = total;
    if (int rc = G.d_nbr.alloc(std::max<size_t>((size_t)total, 1) * 4)) return rc;
    if (total > 0) {
        if (rows_dev > 0) {
            if (kc) { if (int rc = kc->begin(HS_K_GRAPH_CSR, stream)) return rc; }
            hipLaunchKernelGGL(hsdev::k_read_graph_fill, dim3((rows_dev + 255) / 256), dim3(256), 0, stream, G.d_bits.as<unsigned long long>(), G.d_rw.as<int32_t>(),
                               G.d_row0.as<int64_t>(), G.d_bo.as<int64_t>(), G.d_off.as<int64_t>(), rows_dev, G.d_nbr.as<int32_t>());
            HS_HIP(hipGetLastError());
            if (kc) { if (int rc = kc->end((int64_t)win_bits_off.back() * 8 + 4 * (G.total_exact ? total : (int64_t)rows_dev * 20), stream)) return rc; }
        }
        if (!ws.host_nbr.empty()) {   // the host rows sit behind the device rows: their lists start at total - |host_nbr|
            const size_t nb = ws.host_nbr.size() * 4;
            if (int rc = G.h_nbr.alloc(nb)) return rc;
            std::memcpy(G.h_nbr.p, ws.host_nbr.data(), nb);
            HS_HIP(HS_COPY_ASYNC(G.d_nbr.as<int32_t>() + (total - (int64_t)ws.host_nbr.size()), G.h_nbr.p, nb, hipMemcpyHostToDevice, stream));
        }
    }
    {   // visiting order of every window (hs_kernels_cw.hip)
        const int cap = std::min(((G.max_m + 63) / 64) * 64, 8192);
        // the visit programs of the row-packed Chinese-Whispers kernel (windows with m <= 255): nnz + 15 m bytes per window
        if (int rc = G.d_prog_info.alloc(std::max<size_t>((size_t)rows, 1) * 4)) return rc;
        if (int rc = G.d_prog_bytes.alloc((size_t)total + 15 * (size_t)rows + 64)) return rc;
        const size_t steps_bytes = (std::max<size_t>((size_t)W, 1) * 4 + 15) & ~(size_t)15;
        if (int rc = G.d_prog_steps.alloc(steps_bytes)) return rc;
        if (int rc = G.d_prog_adj.alloc(std::max<size_t>((size_t)rows, 1) * 8)) return rc;
        hipLaunchKernelGGL(hsdev::k_fill16, dim3((unsigned)std::min<size_t>(64, steps_bytes / 4096 + 1)), dim3(256), 0, stream, G.d_prog_steps.as<uint4>(), (long long)(steps_bytes / 16), 0u);
        if (kc) { if (int rc = kc->begin(HS_K_VISIT_LISTS, stream)) return rc; }
        hipLaunchKernelGGL(hsdev::k_cw_visit_lists, dim3((unsigned)W), dim3(256), (size_t)cap * 4, stream, G.d_off.as<int64_t>(), G.d_nbr.as<int32_t>(), G.d_row0.as<int64_t>(),
                           G.d_ids.as<int32_t>(), G.d_wc.as<int32_t>(), G.d_rank_off.as<int64_t>(), G.d_rank.as<int32_t>(), W, cap, G.d_visit.as<int32_t>(),
                           G.d_visit_n.as<int32_t>(), G.d_prog_info.as<uint32_t>(), G.d_prog_bytes.as<uint8_t>(), G.d_prog_steps.as<int32_t>(),
                           G.d_prog_adj.as<unsigned long long>());
        HS_HIP(hipGetLastError());
        if (kc) { if (int rc = kc->end(20 * (int64_t)rows, stream)) return rc; }   // row offsets + read id + rank in, visiting slot out
    }
    return HS_OK;
}
// after a wait of the stream: the kernel clocks and the K6 time of the call
static int graph_rows_settle(GraphRows& G, KernelClock* kc) {
    if (kc) kc->flush();
    if (G.timed) { G.timed = false; float m = 0; if (int rc = G.ev.ms(&m)) return rc; if (G.k_ms) *G.k_ms += m; }
    return HS_OK;
}
// both halves and a wait: the graphs are complete and G.total is their exact size (the kernel-level entry point)
static int graph_rows_build(const int32_t* d_sim, const int32_t* d_diff, const std::vector<int64_t>& ctg_out_off, const std::vector<int32_t>& ctg_n_matrix,
                            const hs::SrWindowSet& ws, GraphRows& G, hipStream_t stream, int64_t* rows_on_host, float* k_ms, KernelClock* kc = nullptr,
                            const PlaneRows* planes = nullptr, int es = 1) {
    if (int rc = graph_rows_begin(d_sim, d_diff, ctg_out_off, ctg_n_matrix, ws, G, stream, k_ms, kc, planes, es)) return rc;
    if (int rc = graph_rows_end(ws, G, stream, rows_on_host, kc)) return rc;
    if (G.rows > 0 && !G.total_exact) { int64_t t = 0; if (int rc = d2h_pinned(&t, G.d_off.as<int64_t>() + G.rows, 8, stream)) return rc; G.total = t; G.total_exact = true; }
    else if (int rc_w = stream_wait(stream)) return rc_w;
    return graph_rows_settle(G, kc);
}

struct HipSrOps : hs::SrDeviceOps {
    hipStream_t stream = nullptr;
    DBuf d_sim, d_diff;                       // K5 results stay in HBM for K6
    std::vector<int64_t> sd_out_off;
    std::vector<int32_t> sd_n;
    GraphRows G;
    KernelClock kc;

    // K5 runs on while the host plans the windows: its temporaries and its timing events are parked here until the next
    // call that waits for the stream anyway
    struct SimdiffInFlight {
        DBuf d_alt, d_ref, d_sr, d_sa, d_sc, d_cb, d_po, d_n, d_pn, d_w, d_oo, t_c, t_i, t_j, d_bc, d_bw;
        std::vector<int32_t> tc, ti, tj, blk_c, blk_w;      // (the lists ride in the one upload of the call)
        UploadPack pk;
        EventPair ev;
        float* k_ms = nullptr;
    };
    std::unique_ptr<SimdiffInFlight> sd_flight;
    int settle_simdiff(bool account = true) {   // waits for K5 if it is still running, accounts its time, releases its temporaries
        if (!sd_flight) return HS_OK;
        float m = 0;
        const int rc = sd_flight->ev.ms(&m);
        if (!rc && account && sd_flight->k_ms) *sd_flight->k_ms += m;
        sd_flight.reset();
        return rc;
    }
    ~HipSrOps() override { (void)settle_simdiff(false); }   // the caller's counter may be gone by now

    DBuf d_col_off, d_col_idx, d_col_code;    // SNP columns of the batch: uploaded once (or taken over from stage 3), read by K5a and the seeded CW runs
    UploadPack col_pack;
    const hs::CwChain* resident_cols = nullptr;
    // stage 3 -> 4 on the device: the packed SNP columns HipCvOps::finish_columns left (offsets, read indices, codes) become this
    // object's columns; nothing is uploaded
    bool adopted = false;
    int64_t adopted_cols = 0, adopted_entries = 0;
    DBuf cols_block;      // (adopted: the packed SNP block of stage 3; d_col_off / d_col_idx / d_col_code are views into it)
    void adopt_columns(HipCvOps& cv) {
        std::swap(cols_block, cv.d_pk);
        char* base = (char*)cols_block.p;
        auto view = [&](DBuf& d, size_t off, size_t bytes) { d.release(); d.p = base + off; d.bytes = bytes; d.cap = 0; d.view = true; };
        view(d_col_off, cv.pk_layout.off, ((size_t)cv.snp_count + 1) * 8);
        view(d_col_idx, cv.pk_layout.idx, (size_t)cv.snp_entries * 4);
        view(d_col_code, cv.pk_layout.code, (size_t)cv.snp_entries);
        adopted = true; adopted_cols = cv.snp_count; adopted_entries = cv.snp_entries;
    }
    bool columns_resident() const override { return adopted; }
    // (HS_LOW_MEMORY_GRAPHS_ON_HOST=1: the windows of the low-memory path keep the host builder)
    bool low_memory_graphs() const override { static const bool host = std::getenv("HS_LOW_MEMORY_GRAPHS_ON_HOST") != nullptr; return !host; }
    void drop_resident_columns() override { adopted = false; resident_cols = nullptr; }
    int fetch_columns(std::vector<int32_t>& idx, std::vector<uint8_t>& code) override {
        if (!adopted) { set_error("fetch_columns: no resident columns"); return HS_EINVAL; }
        idx.resize((size_t)adopted_entries); code.resize((size_t)adopted_entries);
        if (adopted_entries == 0) return HS_OK;
        if (int rc = d2h_pinned(idx.data(), d_col_idx.p, (size_t)adopted_entries * 4, stream)) return rc;
        return d2h_pinned(code.data(), d_col_code.p, (size_t)adopted_entries, stream);
    }
    int window_masks(const std::vector<int64_t>& col_a, const std::vector<int64_t>& col_b, const std::vector<int64_t>& slot_off,
                     std::vector<int32_t>& ids, std::vector<int32_t>& win_m) override {
        if (!adopted) { set_error("window_masks: no resident columns"); return HS_EINVAL; }
        const int W = (int)col_a.size();
        const int64_t total = slot_off.back();
        ids.resize((size_t)total); win_m.assign((size_t)W, 0);
        if (W == 0) return HS_OK;
        DBuf d_a, d_b, d_so, d_ids;
        UploadPack pk;
        pk.add(col_a, d_a); pk.add(col_b, d_b); pk.add(slot_off, d_so);
        if (int rc = pk.commit(stream)) return rc;
        // the ids and the per-window counts in one block: one copy back
        const size_t ids_bytes = ((size_t)total * 4 + 255) & ~(size_t)255;
        if (int rc = d_ids.alloc(ids_bytes + (size_t)W * 4 + 16)) return rc;
        int32_t* const dm = reinterpret_cast<int32_t*>((char*)d_ids.p + ids_bytes);
        if (int rc = kc.begin(HS_K_WINDOW_MASKS, stream)) return rc;
        hipLaunchKernelGGL(hsdev::k_window_masks, dim3((unsigned)((W + 3) / 4)), dim3(256), 0, stream, d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(), d_a.as<int64_t>(),
                           d_b.as<int64_t>(), d_so.as<int64_t>(), W, d_ids.as<int32_t>(), dm);
        HS_HIP(hipGetLastError());
        if (int rc = kc.end(8 * total + 4 * (int64_t)W, stream)) return rc;
        HBuf h_ids;
        if (int rc = h_ids.alloc(ids_bytes + (size_t)W * 4 + 16)) return rc;
        { Shipment sh; sh.add(h_ids.p, d_ids.p, ids_bytes + (size_t)W * 4); if (int rc = sh.launch(stream, 32)) return rc; }
        if (int rc = stream_wait(stream)) return rc;
        if (total) std::memcpy(ids.data(), h_ids.p, (size_t)total * 4);
        std::memcpy(win_m.data(), (const char*)h_ids.p + ids_bytes, (size_t)W * 4);
        kc.flush();
        return HS_OK;
    }
    int simdiff_columns(const hs::SimdiffJob& job, float* k_ms) override {
        const hs::CwChain& ch = *job.cols;
        if (!adopted) {
            col_pack.add(ch.col_off, d_col_off); col_pack.add(ch.col_idx, d_col_idx); col_pack.add(ch.col_code, d_col_code);
            if (int rc = col_pack.commit(stream)) return rc;
        } else if ((int64_t)ch.col_off.size() != adopted_cols + 1) { set_error("simdiff_columns: the resident columns are not those of this call"); return HS_EINVAL; }
        resident_cols = job.cols;
        sd_out_off = job.out_off; sd_n = job.n_reads;
        if (job.out_total <= 0 && job.plane_total <= 0) return HS_OK;
        if (int rc = settle_simdiff()) return rc;
        sd_flight.reset(new SimdiffInFlight());
        SimdiffInFlight& f = *sd_flight;
        f.k_ms = k_ms;
        f.pk.add(job.snp_ref, f.d_sr);
        f.pk.add(job.snp_alt, f.d_sa);
        f.pk.add(job.snp_contig, f.d_sc);
        f.pk.add(job.contig_snp_base, f.d_cb);
        f.pk.add(job.plane_off, f.d_po);
        f.pk.add(job.n_reads, f.d_n);
        f.pk.add(job.plane_n, f.d_pn);
        f.pk.add(job.words, f.d_w);
        f.pk.add(job.out_off, f.d_oo);
        snp_planes_blocks(job.words.data(), (int)job.words.size(), f.blk_c, f.blk_w);
        simdiff_tiles(job.n_reads, f.tc, f.ti, f.tj);
        f.pk.add(f.blk_c, f.d_bc); f.pk.add(f.blk_w, f.d_bw); f.pk.add(f.tc, f.t_c); f.pk.add(f.ti, f.t_i); f.pk.add(f.tj, f.t_j);
        if (int rc = f.pk.commit(stream)) return rc;
        const size_t pbytes = (size_t)job.plane_total * sizeof(uint64_t);
        if (int rc = f.d_alt.alloc(pbytes)) return rc;
        if (int rc = f.d_ref.alloc(pbytes)) return rc;
        // (sim, diff) of a read pair side by side: K6 reads both for every masked read of a row, one 64-byte line instead of two
        if (int rc = d_sim.alloc(std::max<size_t>((size_t)job.out_total, 1) * 2 * sizeof(int32_t))) return rc;
        d_diff.release(); d_diff.p = (char*)d_sim.p + sizeof(int32_t); d_diff.bytes = 0; d_diff.cap = 0; d_diff.view = true;
        if (int rc = kc.begin(HS_K_SNP_PLANES, stream)) return rc;
        if (int rc = snp_planes_launch(d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(), d_col_code.as<uint8_t>(), f.d_sr.as<uint8_t>(), f.d_sa.as<uint8_t>(),
                                       f.d_sc.as<int32_t>(), f.d_cb.as<int64_t>(), f.d_po.as<int64_t>(), f.d_w.as<int32_t>(), f.d_pn.as<int32_t>(), f.d_bc.as<int32_t>(),
                                       f.d_bw.as<int32_t>(), f.blk_c.size(), (int32_t)job.snp_ref.size(), f.d_alt.as<uint64_t>(), f.d_ref.as<uint64_t>(), stream)) return rc;
        if (int rc = kc.end(5 * (adopted ? adopted_entries : (int64_t)ch.col_idx.size()) + 2 * (int64_t)pbytes, stream)) return rc;
        if (int rc = f.ev.init()) return rc;
        HS_HIP(hipEventRecord(f.ev.a, stream));
        if (int rc = kc.begin(HS_K_SIMDIFF, stream)) return rc;
        if (int rc = simdiff_launch(f.d_alt.as<uint64_t>(), f.d_ref.as<uint64_t>(), f.d_po.as<int64_t>(), f.d_n.as<int32_t>(), f.d_w.as<int32_t>(),
                                    f.d_oo.as<int64_t>(), d_sim.as<int32_t>(), d_diff.as<int32_t>(), stream, f.t_c.as<int32_t>(), f.t_i.as<int32_t>(), f.t_j.as<int32_t>(), f.tc.size(), 2)) return rc;
        if (int rc = kc.end(2 * (int64_t)pbytes + 8 * job.out_total, stream)) return rc;   // the two bit-planes in, sim + diff out
        HS_HIP(hipEventRecord(f.ev.b, stream));
        return HS_OK;   // not waited for: the stream orders K6 behind it, the host goes on planning the windows
    }

    struct Keep { int64_t stage_hint = 0; };      // kept by the caller from step to step: how much the undecided graph rows staged
    Keep* keep = nullptr;
    PlaneRows plane_rows() const {
        PlaneRows pr;
        if (sd_flight) { pr.d_alt = sd_flight->d_alt.as<uint64_t>(); pr.d_ref = sd_flight->d_ref.as<uint64_t>(); pr.d_plane_off = sd_flight->d_po.as<int64_t>(); pr.d_words = sd_flight->d_w.as<int32_t>(); }
        return pr;
    }
    int build_graphs(const hs::SrWindowSet& ws, int64_t* rows_on_host, float* k_ms) override {
        if (int rc = build_graphs_begin(ws, k_ms)) return rc;
        return build_graphs_end(ws, rows_on_host);
    }
    bool two_phase_graphs() const override { return true; }
    int build_graphs_begin(const hs::SrWindowSet& ws, float* k_ms) override {
        const PlaneRows pr = plane_rows();
        if (keep) G.stage_hint = keep->stage_hint;
        return graph_rows_begin(d_sim.as<int32_t>(), d_diff.as<int32_t>(), sd_out_off, sd_n, ws, G, stream, k_ms, &kc, &pr, 2);
    }
    int build_graphs_end(const hs::SrWindowSet& ws, int64_t* rows_on_host) override {
        const int rc = graph_rows_end(ws, G, stream, rows_on_host, &kc);
        if (keep) keep->stage_hint = G.stage_hint;
        return rc;      // (not waited for: K5's temporaries and the clocks are settled behind the wait of cw_chain / the destructor)
    }
    int fetch_graphs(std::vector<int64_t>& off, std::vector<int32_t>& nbr) override {
        if (G.rows > 0 && !G.total_exact) { int64_t t = 0; if (int rc = d2h_pinned(&t, G.d_off.as<int64_t>() + G.rows, 8, stream)) return rc; G.total = t; G.total_exact = true; }
        off.assign((size_t)G.rows + 1, 0); nbr.assign((size_t)G.total, 0);
        if (int rc = d2h_pinned(off.data(), G.d_off.p, off.size() * 8, stream)) return rc;
        if (G.total > 0) { if (int rc = d2h_pinned(nbr.data(), G.d_nbr.p, nbr.size() * 4, stream)) return rc; }
        return HS_OK;
    }

    // dynamic LDS of the one-wavefront-per-instance kernels: nodes kept in LDS (wider windows use global scratch)
    static int lds_nodes(int max_m, int ints_per_node, int limit_bytes) {
        const int cap = ((std::max(max_m, 1) + 63) / 64) * 64;
        return std::min(cap, (limit_bytes / (ints_per_node * 4)) / 64 * 64);
    }

    int cw_chain(const hs::CwChain& ch, std::vector<int32_t>& labels, std::vector<int32_t>& final_labels, std::vector<uint8_t>& final_ok,
                 float k_ms[3], hs::SrChainStats* stats) override {
        if (int rc = settle_simdiff()) return rc;   // (a batch without graph windows never built graphs)
        const int Wc = (int)ch.win.size();
        const int64_t n_inst = (int64_t)ch.seed_col.size();
        const int64_t total_m = ch.chain_row0.back();
        if (n_inst > 0x7fffffff) { set_error("Chinese Whispers: too many runs in one call"); return HS_EINVAL; }
        // per-SNP runs: window, slab offset (the runs of a window are contiguous: K * m labels); small windows go to the
        // row-packed kernel, the others to the one-wavefront-per-run kernel
        std::vector<int32_t> inst_win((size_t)n_inst), list_big, list_lanes, unit_win, unit_inst0, unit_n;
        const int lanes_cap = std::getenv("HS_CW_NO_LANES") ? 0 : 64;     // windows up to 64 reads: one run per lane (k_cw_seeded_lanes)
        int max_m_small = 1;
        std::vector<int64_t> inst_slab((size_t)n_inst), chain_slab0((size_t)Wc), big_scr, tail_scr((size_t)Wc, 0);
        int64_t slab = 0, big_scr_total = 0, tail_scr_total = 0;
        int max_m_big = 1, max_m_chain = 1;
        for (int k = 0; k < Wc; ++k) {
            const int w = ch.win[(size_t)k];
            const int m = G.win_m[(size_t)w];
            chain_slab0[(size_t)k] = slab;
            max_m_chain = std::max(max_m_chain, m);
            for (int64_t i = ch.win_seed_begin[(size_t)k]; i < ch.win_seed_begin[(size_t)k + 1]; ++i) {
                inst_win[(size_t)i] = w; inst_slab[(size_t)i] = slab; slab += m;
                if (m > HS_CWR_CAP) { list_big.push_back((int32_t)i); max_m_big = std::max(max_m_big, m); }
                else if (m <= lanes_cap) list_lanes.push_back((int32_t)i);
            }
            if (m <= HS_CWR_CAP && m > lanes_cap) {      // units of up to eight runs of this window for the row-packed kernel
                max_m_small = std::max(max_m_small, m);
                for (int64_t i = ch.win_seed_begin[(size_t)k]; i < ch.win_seed_begin[(size_t)k + 1]; i += 8) {
                    unit_win.push_back(w); unit_inst0.push_back((int32_t)i);
                    unit_n.push_back((int32_t)std::min<int64_t>(8, ch.win_seed_begin[(size_t)k + 1] - i));
                }
            }
        }
        const int cap_big = lds_nodes(max_m_big, 2, 96 * 1024);
        for (int32_t i : list_big) {
            const int m = G.win_m[(size_t)inst_win[(size_t)i]];
            big_scr.push_back(big_scr_total);
            if (m > cap_big) big_scr_total += 2 * (int64_t)m;
        }
        const int cap_tail = lds_nodes(max_m_chain, 7, 56 * 1024);
        for (int k = 0; k < Wc; ++k) {
            const int m = G.win_m[(size_t)ch.win[(size_t)k]];
            tail_scr[(size_t)k] = tail_scr_total;
            if (m > cap_tail) tail_scr_total += 7 * (int64_t)m + (m & 1);      // keeps the next window's doubles 8-byte aligned
        }
        DBuf d_ll, d_sets, d_names, d_slots, d_alive, d_ovf, d_ovf_n, d_iw, d_is, d_seed, d_uw, d_ui, d_un, d_lb, d_bs, d_cw, d_cr0, d_csb, d_cs0, d_ts, d_slab, d_gs, d_l3, d_final, d_ok, d_stat,
            d_cpos, d_sf, d_sl, d_plo, d_phi;
        if (resident_cols != &ch && !adopted) {   // normally uploaded by simdiff_columns already
            col_pack.add(ch.col_off, d_col_off); col_pack.add(ch.col_idx, d_col_idx); col_pack.add(ch.col_code, d_col_code);
            if (int rc = col_pack.commit(stream)) return rc;
            resident_cols = &ch;
        }
        const bool finish = ch.finish_on_device && !std::getenv("HS_FINISH_ON_HOST");
        UploadPack pk;
        pk.add(inst_win, d_iw); pk.add(inst_slab, d_is); pk.add(ch.seed_col, d_seed); pk.add(unit_win, d_uw); pk.add(unit_inst0, d_ui); pk.add(unit_n, d_un); pk.add(list_big, d_lb); pk.add(list_lanes, d_ll);
        pk.add(big_scr, d_bs); pk.add(ch.win, d_cw); pk.add(ch.chain_row0, d_cr0); pk.add(ch.win_seed_begin, d_csb); pk.add(chain_slab0, d_cs0);
        pk.add(tail_scr, d_ts);
        if (finish) { pk.add(ch.col_pos, d_cpos); pk.add(ch.win_snp_first, d_sf); pk.add(ch.win_snp_last, d_sl); pk.add(ch.win_pos_lo, d_plo); pk.add(ch.win_pos_hi, d_phi); }
        if (int rc = pk.commit(stream)) return rc;
        if (int rc = d_slab.alloc(std::max<size_t>((size_t)slab, 1) * 4)) return rc;
        if (int rc = d_gs.alloc(std::max<size_t>((size_t)std::max(big_scr_total, tail_scr_total), 2) * 4)) return rc;
        if (int rc = d_l3.alloc(std::max<size_t>((size_t)total_m, 1) * 4)) return rc;
        if (int rc = d_final.alloc(std::max<size_t>((size_t)total_m, 1) * 4)) return rc;
        if (int rc = d_ok.alloc(std::max<size_t>((size_t)Wc, 1))) return rc;
        if (int rc = d_stat.alloc(416)) return rc;    // {sweeps, bytes} of the per-SNP runs, {sweeps, bytes} of the window tails, histogram of sweeps per run [16]
        hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, d_stat.as<uint4>(), 26ll, 0u);
        EventPair e1, e2;
        if (int rc = e1.init()) return rc;
        if (int rc = e2.init()) return rc;
        // ---- per-SNP runs, seeded on the device from the SNP columns ----
        HS_HIP(hipEventRecord(e1.a, stream));
        if (!list_lanes.empty()) {
            const int n = (int)list_lanes.size();
            const size_t npad = ((size_t)n + 63) & ~(size_t)63;
            if (int rc = d_sets.alloc(npad * 128)) return rc;
            if (int rc = d_names.alloc(npad * 16)) return rc;
            if (int rc = d_slots.alloc(npad * 64)) return rc;
            if (int rc = d_alive.alloc(npad)) return rc;
            if (int rc = d_ovf.alloc((size_t)n * 4)) return rc;
            if (int rc = d_ovf_n.alloc(16)) return rc;
            hipLaunchKernelGGL(hsdev::k_fill16, dim3(1), dim3(64), 0, stream, d_ovf_n.as<uint4>(), 1ll, 0u);
            if (int rc = kc.begin(HS_K_CW_SEED_SETS, stream)) return rc;
            hipLaunchKernelGGL(hsdev::k_cw_seed_sets, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(), d_ll.as<int32_t>(), n,
                               d_iw.as<int32_t>(), d_seed.as<int64_t>(), d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(), d_col_code.as<uint8_t>(),
                               d_sets.as<unsigned long long>(), d_names.as<uint8_t>(), d_slots.as<uint8_t>(), d_alive.as<uint8_t>(), d_ovf.as<int32_t>(), d_ovf_n.as<int32_t>());
            HS_HIP(hipGetLastError());
            if (int rc = kc.end((int64_t)npad * 209 + 5 * (int64_t)n * 40, stream)) return rc;      // sets, names, slots, alive out per run; the seeding column (about 40 entries) in
            if (int rc = kc.begin(HS_K_CW_SEEDED, stream)) return rc;
            hipLaunchKernelGGL(hsdev::k_cw_seeded_lanes, dim3((unsigned)(npad / 64)), dim3(64), 0, stream, G.d_off.as<int64_t>(), G.d_row0.as<int64_t>(), G.d_visit_n.as<int32_t>(),
                               G.d_prog_info.as<uint32_t>(), G.d_prog_adj.as<unsigned long long>(), d_ll.as<int32_t>(), n, d_iw.as<int32_t>(), d_is.as<int64_t>(),
                               d_sets.as<unsigned long long>(), d_names.as<uint8_t>(), d_slots.as<uint8_t>(), d_alive.as<uint8_t>(), d_slab.as<int32_t>(), d_stat.as<unsigned long long>());
            HS_HIP(hipGetLastError());
            if (int rc = kc.end((int64_t)npad * 209, stream)) return rc;      // + sweeps * (4 nnz + 8 m) per run, counted by the kernel itself (added below)
            if (std::getenv("HS_TIMING")) {
                int32_t n_ovf = 0;
                if (int rc = d2h_pinned(&n_ovf, d_ovf_n.p, 4, stream)) return rc;
                std::fprintf(stderr, "[hs timing] sr: %d per-SNP runs one per lane, %d of them with more than 16 labels alive (-> one wavefront each); %zu runs in units of 8, %zu one wavefront each\n",
                             n, n_ovf, unit_win.size() * 8, list_big.size());
            }
            // the few runs with more labels alive than slots: one wavefront each, the list and its length are on the device
            if (int rc = kc.begin(HS_K_CW_SEEDED_WIDE, stream)) return rc;
            hipLaunchKernelGGL(hsdev::k_cw_seeded_wave, dim3((unsigned)std::min(n, 1024)), dim3(64), (size_t)64 * 8, stream, G.d_off.as<int64_t>(), G.d_nbr.as<int32_t>(),
                               G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(), G.d_visit.as<int32_t>(), G.d_visit_n.as<int32_t>(), d_ovf.as<int32_t>(), 0, d_ovf_n.as<int32_t>(),
                               d_iw.as<int32_t>(), d_seed.as<int64_t>(), d_is.as<int64_t>(), d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(),
                               d_col_code.as<uint8_t>(), 64, (int32_t*)nullptr, (const int64_t*)nullptr, d_slab.as<int32_t>(), d_stat.as<unsigned long long>());
            HS_HIP(hipGetLastError());
            if (int rc = kc.end(0, stream)) return rc;
        }
        if (!unit_win.empty() || !list_big.empty()) { if (int rc = kc.begin(HS_K_CW_SEEDED_WIDE, stream)) return rc; }
        if (!unit_win.empty()) {
            const int n = (int)unit_win.size();
            const int m_cap = std::max(16, (max_m_small + 15) & ~15);
            const int cnt_cap = std::max(m_cap, 256);
            const int prog_cap = std::max(2048, m_cap * 32);
            const size_t lds = (size_t)2 * m_cap * 4 + (size_t)8 * cnt_cap * 4 + (size_t)prog_cap + (size_t)8 * m_cap;
            if (lds > 48 * 1024)
                HS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hsdev::k_cw_seeded_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(hsdev::k_cw_seeded_rows, dim3((unsigned)n), dim3(128), lds, stream, G.d_off.as<int64_t>(), G.d_nbr.as<int32_t>(),
                               G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(), G.d_visit.as<int32_t>(), G.d_visit_n.as<int32_t>(), G.d_prog_info.as<uint32_t>(),
                               G.d_prog_bytes.as<uint8_t>(), G.d_prog_steps.as<int32_t>(), d_uw.as<int32_t>(), d_ui.as<int32_t>(), d_un.as<int32_t>(), n, d_seed.as<int64_t>(), d_is.as<int64_t>(), d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(),
                               d_col_code.as<uint8_t>(), m_cap, prog_cap, d_slab.as<int32_t>(), d_stat.as<unsigned long long>());
            HS_HIP(hipGetLastError());
        }
        if (!list_big.empty()) {
            const int n = (int)list_big.size();
            const size_t lds = (size_t)cap_big * 8;
            if (lds > 48 * 1024)
                HS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hsdev::k_cw_seeded_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(hsdev::k_cw_seeded_wave, dim3((unsigned)n), dim3(64), lds, stream, G.d_off.as<int64_t>(), G.d_nbr.as<int32_t>(),
                               G.d_row0.as<int64_t>(), G.d_ids.as<int32_t>(), G.d_visit.as<int32_t>(), G.d_visit_n.as<int32_t>(), d_lb.as<int32_t>(), n, (const int32_t*)nullptr,
                               d_iw.as<int32_t>(), d_seed.as<int64_t>(), d_is.as<int64_t>(), d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(),
                               d_col_code.as<uint8_t>(), cap_big, d_gs.as<int32_t>(), d_bs.as<int64_t>(), d_slab.as<int32_t>(), d_stat.as<unsigned long long>());
            HS_HIP(hipGetLastError());
        }
        if (!unit_win.empty() || !list_big.empty()) { if (int rc = kc.end(0, stream)) return rc; }
        HS_HIP(hipEventRecord(e1.b, stream));
        // ---- the rest of the window's chain, labels in LDS from here to the finished clusters ----
        HS_HIP(hipEventRecord(e2.a, stream));
        if (int rc = kc.begin(HS_K_WINDOW_TAIL, stream)) return rc;
        if (Wc > 0) {
            const size_t lds = (size_t)cap_tail * 7 * 4;
            if (lds > 32 * 1024)
                HS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hsdev::k_window_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(hsdev::k_window_tail, dim3((unsigned)Wc), dim3(64), lds, stream, G.d_off.as<int64_t>(), G.d_nbr.as<int32_t>(), G.d_row0.as<int64_t>(),
                               G.d_ids.as<int32_t>(), G.d_visit.as<int32_t>(), G.d_visit_n.as<int32_t>(), G.d_fe.as<uint8_t>(), G.d_prog_info.as<uint32_t>(),
                               G.d_prog_adj.as<unsigned long long>(), d_cw.as<int32_t>(), d_cr0.as<int64_t>(),
                               d_csb.as<int64_t>(), d_cs0.as<int64_t>(), d_slab.as<int32_t>(), Wc, d_col_off.as<int64_t>(), d_col_idx.as<int32_t>(),
                               d_col_code.as<uint8_t>(), d_cpos.as<int32_t>(), d_sf.as<int64_t>(), d_sl.as<int64_t>(), d_plo.as<int32_t>(), d_phi.as<int32_t>(),
                               finish ? 1 : 0, cap_tail, d_gs.as<int32_t>(), d_ts.as<int64_t>(), d_l3.as<int32_t>(), d_final.as<int32_t>(), d_ok.as<uint8_t>(),
                               d_stat.as<unsigned long long>() + 2);
            HS_HIP(hipGetLastError());
        }
        if (int rc = kc.end(4 * slab + 8 * total_m, stream)) return rc;   // the runs' labels in, the two label arrays out (+ its two runs, below)
        HS_HIP(hipEventRecord(e2.b, stream));
        {
            // the finished labels and the per-window verdict come back in one shipment with the counters (and the size of the graphs, which
            // the host has not asked for until now); the labels of the third run are only fetched when some window has to be finished by the
            // host code (few or none)
            HBuf h, h4;
            const size_t o_nnz = 416, o_final = 512, o_ok = o_final + (((size_t)total_m * 4 + 255) & ~(size_t)255);
            if (int rc = h4.alloc(o_ok + (size_t)Wc + 256)) return rc;
            Shipment sh;
            sh.add(h4.p, d_stat.p, 416);
            sh.add((char*)h4.p + o_nnz, G.d_off.as<int64_t>() + (G.rows & ~(int64_t)1), 16);      // (the granule that holds d_off[rows])
            bool need_chain_labels = !finish;
            if (finish) {
                sh.add((char*)h4.p + o_final, d_final.p, (size_t)total_m * sizeof(int32_t));
                sh.add((char*)h4.p + o_ok, d_ok.p, (size_t)Wc);
            }
            if (int rc = kc.begin(HS_K_SHIP, stream)) return rc;
            if (int rc = sh.launch(stream, 64)) return rc;
            if (int rc = kc.end(0, stream)) return rc;
            if (int rc = stream_wait(stream)) return rc;
            if (finish) {
                final_labels.resize((size_t)total_m); final_ok.resize((size_t)Wc);
                std::memcpy(final_labels.data(), (const char*)h4.p + o_final, (size_t)total_m * sizeof(int32_t));
                std::memcpy(final_ok.data(), (const char*)h4.p + o_ok, (size_t)Wc);
                for (uint8_t ok : final_ok) if (!ok) { need_chain_labels = true; break; }
            }
            if (need_chain_labels) {
                labels.resize((size_t)total_m);
                if (int rc = h.alloc(std::max<size_t>((size_t)total_m, 1) * sizeof(int32_t))) return rc;
                if (int rc = copy_d2h(h.p, d_l3.p, (size_t)total_m * sizeof(int32_t), stream)) return rc;
                std::memcpy(labels.data(), h.p, (size_t)total_m * sizeof(int32_t));
            } else labels.clear();
            if (!G.total_exact) { G.total = ((const int64_t*)((const char*)h4.p + o_nnz))[G.rows & 1]; G.total_exact = true; }
            if (int rc = graph_rows_settle(G, &kc)) return rc;
            const unsigned long long* st = (const unsigned long long*)h4.p;
            if (stats) { stats->n_instances = n_inst + 2 * (int64_t)Wc; stats->sweeps = (int64_t)(st[0] + st[2]); stats->bytes = (int64_t)(st[1] + st[3]); stats->graph_nnz = G.total; }
            if (std::getenv("HS_TIMING")) {
                std::string h;
                for (int k = 0; k < 16; ++k) h += " " + std::to_string((unsigned long long)st[4 + k]);
                std::fprintf(stderr, "[hs timing] sr: per-SNP Chinese-Whispers runs by number of sweeps (0..15+):%s\n", h.c_str());
#ifdef HS_TAIL_DIAG      // (build with -DHS_TAIL_DIAG: shader cycles of the sections of k_window_tail, summed over the windows)
                std::fprintf(stderr, "[hs timing] sr: k_window_tail cycles: merged ids %llu, two runs + small clusters %llu, renumbering %llu, merge_close_clusters %llu, merge_wrongly_split_haplotypes: slots + map %llu, SNPs %llu, link counts %llu, links + output %llu; SNPs fast %llu slow %llu, keys %llu, windows with map %llu, clusters %llu\n",
                             (unsigned long long)st[20], (unsigned long long)st[21], (unsigned long long)st[22], (unsigned long long)st[23], (unsigned long long)st[25], (unsigned long long)st[26],
                             (unsigned long long)st[27], (unsigned long long)st[24], (unsigned long long)st[28], (unsigned long long)st[29], (unsigned long long)st[30], (unsigned long long)st[31], (unsigned long long)st[32]);
#endif
#ifdef HS_CW_DIAG
                h.clear(); for (int k = 0; k < 16; ++k) h += " " + std::to_string((unsigned long long)st[20 + k]);
                std::fprintf(stderr, "[hs timing] sr: runs by labels alive after seeding (0..15+):%s\n", h.c_str());
                h.clear(); for (int k = 0; k < 16; ++k) h += " " + std::to_string((unsigned long long)st[36 + k]);
                std::fprintf(stderr, "[hs timing] sr: runs by m/16:%s\n", h.c_str());
#endif
            }
            KernelClock::add_bytes(HS_K_CW_SEEDED, (int64_t)st[1]);
            KernelClock::add_bytes(HS_K_WINDOW_TAIL, (int64_t)st[3]);
            kc.flush();
        }
        float m = 0;
        if (int rc = e1.ms(&m)) return rc; k_ms[0] += m;
        if (int rc = e2.ms(&m)) return rc; k_ms[1] += m;
        return HS_OK;
    }
    int cw(hs::CwWave& wv, float* k_ms) override {
        const int n_inst = (int)wv.inst_win.size();
        if (n_inst == 0) return HS_OK;
        int max_m = 1;
        for (int w : wv.inst_win) max_m = std::max(max_m, G.win_m[(size_t)w]);
        const int cap = lds_nodes(max_m, 2, 96 * 1024);
        std::vector<int64_t> scr((size_t)n_inst, 0);
        int64_t scr_total = 0;
        for (int k = 0; k < n_inst; ++k) { const int m = G.win_m[(size_t)wv.inst_win[(size_t)k]]; scr[(size_t)k] = scr_total; if (m > cap) scr_total += 2 * (int64_t)m; }
        DBuf d_iw, d_lo, d_lab, d_scr, d_gs;
        UploadPack pk;
        pk.add(wv.inst_win, d_iw); pk.add(wv.inst_label_off, d_lo); pk.add(wv.labels, d_lab); pk.add(scr, d_scr);
        if (int rc = pk.commit(stream)) return rc;
        if (int rc = d_gs.alloc(std::max<size_t>((size_t)scr_total, 2) * 4)) return rc;
        EventPair ev; if (int rc = ev.init()) return rc;
        HS_HIP(hipEventRecord(ev.a, stream));
        if (int rc = kc.begin(HS_K_CW_LOCAL, stream)) return rc;
        const size_t lds = (size_t)cap * 8;
        if (lds > 48 * 1024)
            HS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hsdev::k_cw_local), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(hsdev::k_cw_local, dim3((unsigned)n_inst), dim3(64), lds, stream, G.d_off.as<int64_t>(), G.d_nbr.as<int32_t>(), G.d_row0.as<int64_t>(),
                           G.d_visit.as<int32_t>(), G.d_visit_n.as<int32_t>(), G.d_fe.as<uint8_t>(), d_iw.as<int32_t>(), d_lo.as<int64_t>(), n_inst, cap,
                           d_gs.as<int32_t>(), d_scr.as<int64_t>(), d_lab.as<int32_t>());
        HS_HIP(hipGetLastError());
        if (int rc = kc.end(8 * (int64_t)wv.labels.size(), stream)) return rc;
        HS_HIP(hipEventRecord(ev.b, stream));
        if (int rc = d2h_pinned(wv.labels.data(), d_lab.p, wv.labels.size() * sizeof(int32_t), stream)) return rc;
        float m = 0; if (int rc = ev.ms(&m)) return rc;
        if (k_ms) *k_ms += m;
        return HS_OK;
    }
};

}  // namespace

extern "C" {

static void fill_meta(const hs_cv_batch* b, hs::CvMeta& meta) {
    meta.n_contigs = b->n_contigs; meta.n_rec = b->n_rec; meta.contig_off = b->contig_off; meta.contig_rec_off = b->contig_rec_off;
    meta.pile_off = b->pile_off; meta.total_len = b->total_len; meta.rec_pos = b->rec_pos; meta.rec_refspan = b->rec_refspan; meta.ploidy = b->ploidy;
}

int hs_cv_batch_set_ploidy(hs_cv_batch* b, const int32_t* ploidy) {
    if (!b) { set_error("hs_cv_batch_set_ploidy: null batch"); return HS_EINVAL; }
    if (ploidy) b->ploidy.assign(ploidy, ploidy + b->n_contigs); else b->ploidy.clear();
    return HS_OK;
}

int hs_cv_run(hs_cv_batch* b, float automatic_snp_threshold, int32_t n_threads, hs_cv_result** out) {
    if (int rc = require_device()) return rc;
    if (!b || !out) { set_error("hs_cv_run: null argument"); return HS_EINVAL; }
    if (int rc = bind_device(b->device)) return rc;
    hs::CvMeta meta; fill_meta(b, meta);
    HipCvOps ops(b);
    return hs::cv_run(ops, meta, automatic_snp_threshold, n_threads, out);
}

int hs_read_graphs(const int32_t* d_sim, const int32_t* d_diff, const int64_t* ctg_out_off, const int32_t* ctg_n_reads,
                   int32_t n_contigs, const int32_t* win_contig, const int64_t* win_mask_off, const int32_t* mask_ids,
                   int32_t n_windows, float error_rate, int64_t** nbr_off, int32_t** nbr, int64_t* n_rows_host, void* stream) {
    if (int rc = require_device()) return rc;
    if (!nbr_off || !nbr || n_windows < 0 || n_contigs < 0) { set_error("hs_read_graphs: bad arguments"); return HS_EINVAL; }
    hs::SrWindowSet ws;
    ws.error_rate = error_rate;
    ws.win_contig.assign(win_contig, win_contig + n_windows);
    ws.win_row0.assign(win_mask_off, win_mask_off + n_windows + 1);
    ws.mask_ids.assign(mask_ids, mask_ids + win_mask_off[n_windows]);
    ws.n_dev_windows = n_windows;
    ws.win_final_empty.assign((size_t)n_windows, 0);
    ws.ctg_rank_off.assign((size_t)n_contigs, 0);
    int64_t ro = 0;
    for (int c = 0; c < n_contigs; ++c) { ws.ctg_rank_off[(size_t)c] = ro; ro += ctg_n_reads[c]; }
    ws.rank.resize((size_t)ro);         // the visiting order is of no interest here: the identity permutation per contig
    for (int c = 0; c < n_contigs; ++c) for (int32_t r = 0; r < ctg_n_reads[c]; ++r) ws.rank[(size_t)(ws.ctg_rank_off[(size_t)c] + r)] = r;
    for (int w = 0; w < n_windows; ++w) if (win_contig[w] < 0 || win_contig[w] >= n_contigs) { set_error("hs_read_graphs: window contig out of range"); return HS_EINVAL; }
    GraphRows G;
    float ms = 0;
    int64_t on_host = 0;
    if (int rc = graph_rows_build(d_sim, d_diff, std::vector<int64_t>(ctg_out_off, ctg_out_off + n_contigs), std::vector<int32_t>(ctg_n_reads, ctg_n_reads + n_contigs),
                                  ws, G, (hipStream_t)stream, &on_host, &ms)) return rc;
    *nbr_off = (int64_t*)std::malloc(((size_t)G.rows + 1) * sizeof(int64_t));
    *nbr = (int32_t*)std::malloc(((size_t)G.total + 1) * sizeof(int32_t));
    if (!*nbr_off || !*nbr) { set_error("hs_read_graphs: out of memory"); return HS_EINVAL; }
    if (int rc = d2h_pinned(*nbr_off, G.d_off.p, ((size_t)G.rows + 1) * 8, (hipStream_t)stream)) return rc;
    if (G.total > 0) { if (int rc = d2h_pinned(*nbr, G.d_nbr.p, (size_t)G.total * 4, (hipStream_t)stream)) return rc; }
    // the device keeps neighbours as window-local indices; this entry point reports read ids
    for (int w = 0; w < n_windows; ++w)
        for (int64_t e = (*nbr_off)[win_mask_off[w]]; e < (*nbr_off)[win_mask_off[w + 1]]; ++e) (*nbr)[e] = mask_ids[win_mask_off[w] + (*nbr)[e]];
    if (n_rows_host) *n_rows_host = on_host;
    return HS_OK;
}

int hs_cv_select(hs_cv_batch* b, hs_cv_selection** out) {
    if (int rc = require_device()) return rc;
    if (!b || !out) { set_error("hs_cv_select: null argument"); return HS_EINVAL; }
    if (int rc = bind_device(b->device)) return rc;
    hs::CvMeta meta; fill_meta(b, meta);
    HipCvOps ops(b);
    hs::CvSelection* sel = new hs::CvSelection();
    if (int rc = hs::cv_pileup(ops, meta, *sel)) { delete sel; return rc; }
    hs_cv_selection* o = (hs_cv_selection*)std::calloc(1, sizeof(hs_cv_selection));
    o->n_selected = 0;
    for (int k = 0; k < 4; ++k) o->t_kernel_ms[k] = sel->k_ms[k];
    o->t_device_ms = sel->t_device_ms; o->t_host_ms = sel->t_host_ms;
    o->impl = sel;
    *out = o;
    return HS_OK;
}
void hs_cv_selection_destroy(hs_cv_selection* s) {
    if (!s) return;
    delete (hs::CvSelection*)s->impl;
    std::free(s);
}
int hs_cv_run_range(hs_cv_batch* b, const hs_cv_selection* sel, int32_t c0, int32_t c1, float automatic_snp_threshold, int32_t n_threads,
                    hs_cv_result** out) {
    if (int rc = require_device()) return rc;
    if (!b || !sel || !sel->impl || !out) { set_error("hs_cv_run_range: null argument"); return HS_EINVAL; }
    if (int rc = bind_device(b->device)) return rc;
    hs::CvMeta meta; fill_meta(b, meta);
    HipCvOps ops(b);
    return hs::cv_run_range(ops, meta, ((const hs::CvSelection*)sel->impl)->rec_stats, c0, c1, automatic_snp_threshold, n_threads, out);
}

int hs_sr_run_cv_range(const hs_cv_batch* b, int32_t c0, int32_t c1, const hs_cv_result* cv, float error_rate, float rarest_strain_abundance,
                       int32_t low_memory, int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size, hs_sr_result** out) {
    if (int rc = require_device()) return rc;
    if (!b || !cv || !out) { set_error("hs_sr_run_cv_range: null argument"); return HS_EINVAL; }
    if (int rc = bind_device(b->device)) return rc;
    hs::CvMeta meta; fill_meta(b, meta);
    HipSrOps ops;
    return hs::sr_run_from_cv(ops, meta, c0, c1, cv, error_rate, rarest_strain_abundance, low_memory, amplicon, seed, n_threads, window_size, out);
}
int hs_sr_run_cv(const hs_cv_batch* b, const hs_cv_result* cv, float error_rate, float rarest_strain_abundance, int32_t low_memory,
                 int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size, hs_sr_result** out) {
    if (!b) { set_error("hs_sr_run_cv: null argument"); return HS_EINVAL; }
    return hs_sr_run_cv_range(b, 0, b->n_contigs, cv, error_rate, rarest_strain_abundance, low_memory, amplicon, seed, n_threads, window_size, out);
}

// ---- contig groups on persistent threads (see the header) ----
struct hs_pipeline {
    hs_cv_batch* batch = nullptr;
    std::vector<std::pair<int, int>> ranges;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<int(int)> job;
    std::function<void()> on_fail;      // called by a group thread whose job failed (or that never bound to the device): releases whoever waits for that group
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    std::vector<int> rcs;
    std::vector<std::string> errs;
    hs_cv_selection* sel = nullptr;
    std::vector<hs_cv_result*> cv;
    std::vector<hs::SrWorkspace> sr_keep;      // per group: the stage-4 plans and visiting orders live from step to step
    std::vector<std::unique_ptr<HipCvOps::Keep>> cv_keep;      // per group: the sizes of its column pass (the next step queues it without asking)
    std::vector<HipSrOps::Keep> sr_dev_keep;
    HipCvOps::K2Order k2_order;
    bool keep_columns = false;      // HS_PIPELINE_KEEP_COLUMNS

    int device = 0;        // = batch->device: the group threads bind themselves to it (a new thread starts on device 0)
    std::vector<int> thread_device;   // what every group thread found current after binding (hs_pipeline_thread_devices)
    void worker(int g) {
        uint64_t seen = 0;
        const bool bound = hipSetDevice(device) == hipSuccess;
        { int cur = -1; if (!bound || hipGetDevice(&cur) != hipSuccess) cur = -1; std::lock_guard<std::mutex> lk(mu); thread_device[(size_t)g] = cur; }
        for (;;) {
            std::function<int(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen; f = job;
            }
            int rc;
            if (!bound) { set_error("a contig-group thread could not bind to the device of its batch"); rc = HS_EHIP; }
            else rc = f(g);
            if (rc) { k2_order.abort(); std::function<void()> h; { std::lock_guard<std::mutex> lk(mu); h = on_fail; } if (h) h(); }      // (nobody waits for a group that is gone: its turn, its share of the error rate)
            {
                std::lock_guard<std::mutex> lk(mu);
                rcs[(size_t)g] = rc;
                if (rc) errs[(size_t)g] = hs_last_error();
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    int run(const std::function<int(int)>& f) {   // f(group) on every group thread; first failure wins
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f; pending = (int)threads.size(); gen++;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        for (size_t g = 0; g < rcs.size(); ++g) if (rcs[g]) { set_error(errs[g]); return rcs[g]; }
        return HS_OK;
    }
    void drop_cv() {
        for (hs_cv_result*& r : cv) { if (r) hs::free_cv_result(r); r = nullptr; }
        if (sel) { hs_cv_selection_destroy(sel); sel = nullptr; }
    }
};

int hs_pipeline_create(hs_cv_batch* b, int32_t n_groups, hs_pipeline** out) {
    if (int rc = require_device()) return rc;
    if (!b || !out) { set_error("hs_pipeline_create: null argument"); return HS_EINVAL; }
    tune_allocator();
    hs_pipeline* p = new hs_pipeline();
    p->batch = b;
    const int C = b->n_contigs;
    const int G = std::max(1, std::min<int>(n_groups, std::max(C, 1)));
    {
        // consecutive contigs per group, cut where the aligned bases (= pileup bytes) reach g / G of the batch: the groups' chains are
        // as long as their data, and the longest one ends the step
        std::vector<int> cut((size_t)G + 1, 0);
        cut[(size_t)G] = C;
        const int64_t total = b->pile_off.empty() ? 0 : b->pile_off.back();
        if (total > 0 && (int)b->contig_rec_off.size() == C + 1) {
            // The groups reach the host one after the other (the device hands out their candidate columns about 0.8 ms apart, K2 by K2),
            // and the step ends when the LAST group's chain -- loops A / B, K4, all of stage 4 -- is through: the later a group starts,
            // the smaller it is made, so that the chains end together. HS_GROUP_TAPER = share of the last group relative to the first
            // (1 = equal groups, the default: on the 16-core box 0.2 - 0.5 measured within the noise of equal groups, 18.9 - 21.5 ms), linear in between.
            static const double taper = []() { const char* e = std::getenv("HS_GROUP_TAPER"); const double v = e ? std::atof(e) : 1.0; return v > 0 && v <= 1 ? v : 1.0; }();
            std::vector<double> upto((size_t)G + 1, 0.0);
            for (int g = 0; g < G; ++g) upto[(size_t)g + 1] = upto[(size_t)g] + (G > 1 ? 1.0 + (taper - 1.0) * g / (G - 1) : 1.0);
            int c = 0;
            for (int g = 1; g < G; ++g) {
                const int64_t want = (int64_t)((double)total * upto[(size_t)g] / upto[(size_t)G]);
                while (c < C && b->pile_off[(size_t)b->contig_rec_off[(size_t)c + 1]] <= want) ++c;
                // the contig that crosses the mark goes to the side it lies more on
                if (c < C) {
                    const int64_t lo = b->pile_off[(size_t)b->contig_rec_off[(size_t)c]], hi = b->pile_off[(size_t)b->contig_rec_off[(size_t)c + 1]];
                    if (hi - want < want - lo) ++c;
                }
                cut[(size_t)g] = std::max(cut[(size_t)g - 1], std::min(c, C));
            }
        } else for (int g = 1; g < G; ++g) cut[(size_t)g] = (int)((int64_t)C * g / G);
        // every group keeps at least one contig while there are enough of them
        for (int g = 1; g < G; ++g) cut[(size_t)g] = std::max(cut[(size_t)g], std::min(g, C));
        for (int g = G - 1; g >= 1; --g) cut[(size_t)g] = std::min(cut[(size_t)g], C - (G - g));
        for (int g = 1; g < G; ++g) cut[(size_t)g] = std::max(cut[(size_t)g], cut[(size_t)g - 1]);
        for (int g = 0; g < G; ++g) p->ranges.push_back(std::make_pair(cut[(size_t)g], cut[(size_t)g + 1]));
    }
    p->rcs.assign((size_t)G, 0); p->errs.assign((size_t)G, std::string()); p->cv.assign((size_t)G, nullptr);
    p->device = b->device; p->thread_device.assign((size_t)G, -1);
    p->sr_keep.resize((size_t)G);
    for (int g = 0; g < G; ++g) p->cv_keep.emplace_back(new HipCvOps::Keep());
    p->sr_dev_keep.resize((size_t)G);
    for (int g = 0; g < G; ++g) p->threads.emplace_back([p, g] { p->worker(g); });
    *out = p;
    return HS_OK;
}

int hs_cv_batch_device(const hs_cv_batch* b) { return b ? b->device : -1; }
int hs_pipeline_set_option(hs_pipeline* p, int32_t option, int64_t value) {
    if (!p) { set_error("hs_pipeline_set_option: null pipeline"); return HS_EINVAL; }
    if (option == HS_PIPELINE_KEEP_COLUMNS) { p->keep_columns = value != 0; return HS_OK; }
    set_error("hs_pipeline_set_option: unknown option"); return HS_EINVAL;
}
int hs_pipeline_groups(const hs_pipeline* p) { return p ? (int)p->ranges.size() : 0; }
int hs_pipeline_group_range(const hs_pipeline* p, int32_t g, int32_t* c0, int32_t* c1) {
    if (!p || g < 0 || g >= (int32_t)p->ranges.size()) { set_error("hs_pipeline_group_range: no such group"); return HS_EINVAL; }
    if (c0) *c0 = p->ranges[(size_t)g].first; if (c1) *c1 = p->ranges[(size_t)g].second;
    return HS_OK;
}
const hs_cv_result* hs_pipeline_group_cv(const hs_pipeline* p, int32_t g) { return (p && g >= 0 && g < (int32_t)p->cv.size()) ? p->cv[(size_t)g] : nullptr; }
// the device every contig-group thread is bound to (-1: not bound yet / failed); returns the number of groups
int hs_pipeline_thread_devices(hs_pipeline* p, int32_t* out, int32_t cap) {
    if (!p) return 0;
    // the threads bind themselves when they start: an empty job makes sure every one of them has got that far
    (void)p->run([](int) { return HS_OK; });
    std::lock_guard<std::mutex> lk(p->mu);
    for (size_t g = 0; g < p->thread_device.size() && (int32_t)g < cap; ++g) out[g] = p->thread_device[g];
    return (int)p->thread_device.size();
}

void hs_pipeline_destroy(hs_pipeline* p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->quit = true; }
    p->cv_go.notify_all();
    for (std::thread& t : p->threads) t.join();
    p->drop_cv();
    delete p;
}

static hs_sr_result* concat_sr_parts(hs_pipeline* p, std::vector<hs_sr_result*>& parts, const std::vector<hs::SrSparseLabels>& sparse, hs_pipeline_stats* st) {
    // concatenate in contig order
    hs_sr_result* R = (hs_sr_result*)std::calloc(1, sizeof(hs_sr_result));
    int64_t W = 0, NL = 0;
    for (hs_sr_result* r : parts) { W += r->win_off[r->n_contigs]; NL += r->label_off[r->win_off[r->n_contigs]]; }
    R->n_contigs = p->batch->n_contigs;
    R->win_off = (int64_t*)std::malloc(((size_t)R->n_contigs + 1) * sizeof(int64_t));
    R->win_start = (int32_t*)std::malloc((size_t)std::max<int64_t>(1, W) * sizeof(int32_t));
    R->win_end = (int32_t*)std::malloc((size_t)std::max<int64_t>(1, W) * sizeof(int32_t));
    R->label_off = (int64_t*)std::malloc(((size_t)W + 1) * sizeof(int64_t));
    R->labels = hs::sr_labels_alloc((size_t)NL);
    int64_t w0 = 0, l0 = 0; int c0 = 0;
    R->win_off[0] = 0; R->label_off[0] = 0;
    std::vector<int64_t> part_l0(parts.size(), 0);
    for (size_t pi = 0; pi < parts.size(); ++pi) {
        hs_sr_result* r = parts[pi];
        const int64_t w = r->win_off[r->n_contigs], nl = r->label_off[w];
        for (int c = 0; c < r->n_contigs; ++c) R->win_off[c0 + c + 1] = w0 + r->win_off[c + 1];
        if (w) { std::memcpy(R->win_start + w0, r->win_start, (size_t)w * sizeof(int32_t)); std::memcpy(R->win_end + w0, r->win_end, (size_t)w * sizeof(int32_t)); }
        for (int64_t k = 0; k < w; ++k) R->label_off[w0 + k + 1] = l0 + r->label_off[k + 1];
        part_l0[pi] = l0;
        R->t_device_ms += r->t_device_ms; R->t_host_ms += r->t_host_ms; R->n_cw_instances += r->n_cw_instances;
        for (int k = 0; k < 4; ++k) R->t_kernel_ms[k] += r->t_kernel_ms[k];
        R->t_kernel_graph_ms += r->t_kernel_graph_ms; R->n_graph_rows_host += r->n_graph_rows_host; R->n_windows_finished_on_host += r->n_windows_finished_on_host;
        R->n_cw_sweeps += r->n_cw_sweeps; R->cw_bytes += r->cw_bytes; R->graph_nnz += r->graph_nnz; R->n_graph_rows += r->n_graph_rows; R->simdiff_bytes += r->simdiff_bytes;
        w0 += w; l0 += nl; c0 += r->n_contigs;
    }
    // the labels (tens of MB per batch) are spread over the reads of every window here, once, in their final place: blocks of
    // windows on the caller's worker threads
    {
        struct Blk { int part; int64_t w0, w1; };
        std::vector<Blk> blks;
        for (size_t pi = 0; pi < parts.size(); ++pi) {
            const int64_t w = parts[pi]->win_off[parts[pi]->n_contigs];
            for (int64_t a = 0; a < w; a += 64) blks.push_back(Blk{(int)pi, a, std::min<int64_t>(w, a + 64)});
        }
        hs::hs_parallel_for((int)blks.size(), host_threads(), [&](int i) {
            const Blk& b = blks[(size_t)i];
            const hs_sr_result* r = parts[(size_t)b.part];
            hs::sr_expand_labels(sparse[(size_t)b.part], r->label_off, b.w0, b.w1, R->labels + part_l0[(size_t)b.part]);
        });
    }
    for (hs_sr_result* r : parts) hs::free_sr_result(r);
    if (st) {
        st->n_cw_instances = R->n_cw_instances; st->n_graph_rows_host = R->n_graph_rows_host;
        st->t_device_ms += R->t_device_ms; st->t_host_ms += R->t_host_ms;
        for (int k = 0; k < 4; ++k) st->t_kernel_sr_ms[k] = R->t_kernel_ms[k];
        st->t_kernel_graph_ms = R->t_kernel_graph_ms;
        st->n_cw_sweeps = R->n_cw_sweeps; st->cw_bytes = R->cw_bytes; st->graph_nnz = R->graph_nnz; st->n_graph_rows = R->n_graph_rows; st->simdiff_bytes = R->simdiff_bytes;
    }
    return R;
}

int hs_pipeline_select(hs_pipeline* p, float* mean_distance, hs_pipeline_stats* st) {
    if (!p || !mean_distance) { set_error("hs_pipeline_select: null argument"); return HS_EINVAL; }
    if (int rc = bind_device(p->batch->device)) return rc;
    p->drop_cv();
    if (int rc = hs_cv_select(p->batch, &p->sel)) return rc;
    const hs::CvSelection& sel = *(const hs::CvSelection*)p->sel->impl;
    const hs_cv_batch* b = p->batch;
    for (int c = 0; c < b->n_contigs; ++c) {   // call_variants.cpp:434 per contig, from the integer counters of K1
        int64_t nerr = 0, nlen = 0;
        for (int r = b->contig_rec_off[(size_t)c]; r < b->contig_rec_off[(size_t)c + 1]; ++r) { nerr += sel.rec_stats[(size_t)r * 4 + 1]; nlen += sel.rec_stats[(size_t)r * 4 + 2]; }
        mean_distance[c] = hs::mean_distance_from_counts(nerr, nlen);
    }
    if (st) {
        std::memset(st, 0, sizeof *st);
        st->t_device_ms = p->sel->t_device_ms; st->t_host_ms = p->sel->t_host_ms;
        st->t_kernel_cv_ms[0] = p->sel->t_kernel_ms[0]; st->t_kernel_cv_ms[3] = p->sel->t_kernel_ms[3];
    }
    return HS_OK;
}

int hs_pipeline_run(hs_pipeline* p, float automatic_snp_threshold, float error_rate, float rarest_strain_abundance, int32_t low_memory,
                    int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size, hs_sr_result** out, hs_pipeline_stats* st) {
    if (!p || !out || !p->sel) { set_error("hs_pipeline_run: run hs_pipeline_select first"); return HS_EINVAL; }
    if (int rc = bind_device(p->batch->device)) return rc;
    const int G = (int)p->ranges.size();
    // n_threads <= 0: three workers per usable core over all groups -- a group is on the host for part of its chain only (the
    // rest it waits for the device), so that many threads keep the cores busy without queueing behind each other (500-contig
    // bench, 16 usable cores: 16 threads 74 ms per step, 32: 46, 48: 44, 64: 47, 128: 51)
    if (n_threads <= 0) n_threads = 3 * host_threads();
    const int per = std::max(1, n_threads / G);
    if (window_size <= 0) {
        // choose the window size over the WHOLE batch (separate_reads.cpp:1466-1498 looks at every read of every contig): READ
        // limits are (POS-1, POS + reference span) (input_output.cpp:503-511), `int sumLength` wraps
        const hs_cv_batch* b = p->batch;
        if (amplicon) { window_size = 0; for (int c = 0; c < b->n_contigs; ++c) window_size = std::max<int32_t>(window_size, (int32_t)(b->contig_off[(size_t)c + 1] - b->contig_off[(size_t)c])); }
        else {
            uint32_t sum = 0; int above = 0;
            for (int64_t v : b->rec_refspan) { const int len = (int)v + 2; sum += (uint32_t)len; above += len > 4000; }
            const double mean = b->rec_refspan.empty() ? 4000.0 : (int32_t)sum / double(b->rec_refspan.size());
            window_size = 2000;
            if (above < 20 && mean < 4000 && mean > 2000) window_size = 1000;
            else if (above < 20 && mean < 2000) window_size = 500;
        }
    }
    std::vector<hs_sr_result*> parts((size_t)G, nullptr);
    std::vector<hs::SrSparseLabels> sparse((size_t)G);      // the groups leave their labels per window; concat_sr_parts spreads them
    hs::set_trace_origin();
    p->k2_order.reset();
    const int rc = p->run([&](int g) {
        const int c0 = p->ranges[(size_t)g].first, c1 = p->ranges[(size_t)g].second;
        hs::CvMeta meta; fill_meta(p->batch, meta);
        HipCvOps cv_ops(p->batch);
        cv_ops.keep = p->cv_keep[(size_t)g].get(); cv_ops.k2_order = &p->k2_order; cv_ops.order_ticket = g;
        // the SNP columns stay on the device: stage 4 takes them over where stage 3 packed them (HS_COLUMNS_VIA_HOST=1: down and up again)
        static const bool via_host = std::getenv("HS_COLUMNS_VIA_HOST") != nullptr;
        if (p->cv[(size_t)g]) { hs::free_cv_result(p->cv[(size_t)g]); p->cv[(size_t)g] = nullptr; }
        if (int r = hs::cv_run_range(cv_ops, meta, ((const hs::CvSelection*)p->sel->impl)->rec_stats, c0, c1, automatic_snp_threshold, per, &p->cv[(size_t)g], !via_host && !p->keep_columns)) return r;
        HipSrOps ops;
        ops.keep = &p->sr_dev_keep[(size_t)g];
        if (!via_host) ops.adopt_columns(cv_ops);
        return hs::sr_run_from_cv(ops, meta, c0, c1, p->cv[(size_t)g], error_rate, rarest_strain_abundance, low_memory, amplicon, seed, per, window_size,
                                  &parts[(size_t)g], &sparse[(size_t)g], &p->sr_keep[(size_t)g]);
    });
    if (rc) { for (hs_sr_result* r : parts) if (r) hs::free_sr_result(r); return rc; }
    if (st) {
        for (int g = 0; g < G; ++g) {
            const hs_cv_result* r = p->cv[(size_t)g];
            st->n_snps += r->snp_off[r->n_contigs];
            st->t_device_ms += r->t_device_ms; st->t_host_ms += r->t_host_ms;
            st->t_kernel_cv_ms[2] += r->t_kernel_ms[2]; st->t_kernel_k4_ms += r->t_kernel_k4_ms;
            st->n_columns_extracted += r->n_columns_extracted; st->n_columns_downloaded += r->n_columns_downloaded;
            st->n_columns_downloaded_late += r->n_columns_downloaded_late;
            st->t_kernel_cv_ms[1] += r->t_kernel_ms[1];
        }
    }
    if (std::getenv("HS_TIMING")) std::fprintf(stderr, "[hs timing] host waits so far: %ld, %.1f ms in them\n", g_waits.load(), g_wait_us.load() / 1e3);
    hs_sr_result* R = concat_sr_parts(p, parts, sparse, st);
    *out = R;      // (the groups' stage-3 results stay until the next call: hs_pipeline_group_cv)
    return HS_OK;
}

// hs_pipeline_select + the error rate + hs_pipeline_run in ONE call for a job that lives in one process: every contig group brings
// up its own share of the pileup (one group at a time on the device), so its host work starts a quarter of a millisecond into the
// step instead of after the pileup of the whole batch; the job-wide error rate (call_variants.cpp:1312-1315: float sum of the
// positive per-contig mean distances in contig order, then what hairsplitter.py makes of the printed value: %g, capped at 0.15,
// hairsplitter.py:686-692,725) is formed when the last group has its counters, which is long before the first one needs it for
// stage 4. mean_distance [n_contigs] and error_rate_out (optional) receive what hs_pipeline_select would have returned / what
// the caller would have computed.
int hs_pipeline_run_fused(hs_pipeline* p, float automatic_snp_threshold, float rarest_strain_abundance, int32_t low_memory, int32_t amplicon, uint32_t seed,
                          int32_t n_threads, int32_t window_size, float* mean_distance, float* error_rate_out, hs_sr_result** out, hs_pipeline_stats* st) {
    if (!p || !out) { set_error("hs_pipeline_run_fused: null argument"); return HS_EINVAL; }
    if (int rc = bind_device(p->batch->device)) return rc;
    hs_cv_batch* b = p->batch;
    const int G = (int)p->ranges.size();
    const int C = b->n_contigs;
    if (n_threads <= 0) n_threads = 3 * host_threads();
    const int per = std::max(1, n_threads / G);
    if (window_size <= 0) {
        if (amplicon) { window_size = 0; for (int c = 0; c < C; ++c) window_size = std::max<int32_t>(window_size, (int32_t)(b->contig_off[(size_t)c + 1] - b->contig_off[(size_t)c])); }
        else {
            uint32_t sum = 0; int above = 0;
            for (int64_t v : b->rec_refspan) { const int len = (int)v + 2; sum += (uint32_t)len; above += len > 4000; }
            const double mean = b->rec_refspan.empty() ? 4000.0 : (int32_t)sum / double(b->rec_refspan.size());
            window_size = 2000;
            if (above < 20 && mean < 4000 && mean > 2000) window_size = 1000;
            else if (above < 20 && mean < 2000) window_size = 500;
        }
    }
    p->drop_cv();
    hs::CvSelection* sel = new hs::CvSelection();
    sel->rec_stats.assign((size_t)b->n_rec * 4, 0);
    for (int k = 0; k < 4; ++k) sel->k_ms[k] = 0;
    p->sel = (hs_cv_selection*)std::calloc(1, sizeof(hs_cv_selection));
    p->sel->impl = sel;
    std::vector<float> md((size_t)std::max(C, 1), 0.f);
    std::vector<std::array<float, 4>> k_ms_g((size_t)G, std::array<float, 4>{0, 0, 0, 0});
    // the meeting point of the groups before stage 4
    std::mutex bm; std::condition_variable bcv;
    int arrived = 0; bool aborted = false;
    float error_rate = 0;
    std::vector<hs_sr_result*> parts((size_t)G, nullptr);
    std::vector<hs::SrSparseLabels> sparse((size_t)G);
    hs::set_trace_origin();
    p->k2_order.reset();
    // HS_FUSED_HOST_PILEUP=1: the round-3 form (every group waits for its share of the pileup and forms the distances on the host)
    static const bool host_pileup = std::getenv("HS_FUSED_HOST_PILEUP") != nullptr;
    const std::vector<int32_t> no_stats;
    { std::lock_guard<std::mutex> lk(p->mu); p->on_fail = [&] { { std::lock_guard<std::mutex> lk2(bm); aborted = true; } bcv.notify_all(); }; }
    const int rc = p->run([&](int g) {
        const int c0 = p->ranges[(size_t)g].first, c1 = p->ranges[(size_t)g].second;
        hs::CvMeta meta; fill_meta(b, meta);
        HipCvOps cv_ops(b);
        cv_ops.keep = p->cv_keep[(size_t)g].get(); cv_ops.k2_order = &p->k2_order; cv_ops.order_ticket = g;
        auto fail = [&](int r) { { std::lock_guard<std::mutex> lk(bm); aborted = true; } bcv.notify_all(); return r; };
        // the group's mean distances are known (from the host's sums, or with its candidate columns from the device): the last group
        // to get here forms the job's error rate
        auto arrive = [&](const float* md_g) {
            std::memcpy(md.data() + c0, md_g, (size_t)(c1 - c0) * sizeof(float));
            std::lock_guard<std::mutex> lk(bm);
            if (++arrived == G) {
                float total = 0; int n = 0;
                for (int c = 0; c < C; ++c) if (md[(size_t)c] > 0) { total += md[(size_t)c]; n++; }      // :1312-1315, contig order
                const float er32 = total / n;
                char buf[64];
                std::snprintf(buf, sizeof buf, "%g", (double)er32);      // what stage 3 prints and hairsplitter.py reads back
                double e = std::strtod(buf, nullptr);
                if (e > 0.15) e = 0.15;
                error_rate = (float)e;
                bcv.notify_all();
            }
        };
        static const bool via_host = std::getenv("HS_COLUMNS_VIA_HOST") != nullptr;
        if (host_pileup) {
            if (int r = cv_ops.pileup_range(c0, c1, sel->rec_stats, k_ms_g[(size_t)g].data())) return fail(r);
            std::vector<float> md_g((size_t)(c1 - c0));
            for (int c = c0; c < c1; ++c) {   // call_variants.cpp:434 per contig, from the integer counters of K1
                int64_t nerr = 0, nlen = 0;
                for (int r = b->contig_rec_off[(size_t)c]; r < b->contig_rec_off[(size_t)c + 1]; ++r) { nerr += sel->rec_stats[(size_t)r * 4 + 1]; nlen += sel->rec_stats[(size_t)r * 4 + 2]; }
                md_g[(size_t)(c - c0)] = hs::mean_distance_from_counts(nerr, nlen);
            }
            arrive(md_g.data());
            if (int r = hs::cv_run_range(cv_ops, meta, sel->rec_stats, c0, c1, automatic_snp_threshold, per, &p->cv[(size_t)g], !via_host && !p->keep_columns)) return fail(r);
        } else {
            // the group's share of the pileup is the head of its column pass on the device (K0, K1, the contigs' distances, K2 ...): one chain,
            // queued behind the previous group's, no host wait before the candidates
            cv_ops.own_pileup = true;
            const std::function<void(const float*)> on_md = arrive;
            if (int r = hs::cv_run_range(cv_ops, meta, no_stats, c0, c1, automatic_snp_threshold, per, &p->cv[(size_t)g], !via_host && !p->keep_columns, &on_md)) return fail(r);
        }
        {
            std::unique_lock<std::mutex> lk(bm);
            bcv.wait(lk, [&] { return arrived == G || aborted; });
            if (aborted) { set_error("hs_pipeline_run_fused: another contig group failed"); return HS_EINVAL; }
        }
        HipSrOps ops;
        ops.keep = &p->sr_dev_keep[(size_t)g];
        if (!via_host) ops.adopt_columns(cv_ops);
        const int r = hs::sr_run_from_cv(ops, meta, c0, c1, p->cv[(size_t)g], error_rate, rarest_strain_abundance, low_memory, amplicon, seed, per, window_size,
                                         &parts[(size_t)g], &sparse[(size_t)g], &p->sr_keep[(size_t)g]);
        return r ? fail(r) : r;
    });
    { std::lock_guard<std::mutex> lk(p->mu); p->on_fail = nullptr; }
    if (rc) { for (hs_sr_result* r : parts) if (r) hs::free_sr_result(r); return rc; }
    if (mean_distance) std::memcpy(mean_distance, md.data(), (size_t)C * sizeof(float));
    if (error_rate_out) *error_rate_out = error_rate;
    if (st) {
        std::memset(st, 0, sizeof *st);
        for (int g = 0; g < G; ++g) {
            st->t_kernel_cv_ms[0] += k_ms_g[(size_t)g][0]; st->t_kernel_cv_ms[3] += k_ms_g[(size_t)g][3];
            const hs_cv_result* r = p->cv[(size_t)g];
            st->n_snps += r->snp_off[r->n_contigs];
            st->t_device_ms += r->t_device_ms; st->t_host_ms += r->t_host_ms;
            st->t_kernel_cv_ms[2] += r->t_kernel_ms[2]; st->t_kernel_k4_ms += r->t_kernel_k4_ms;
            st->n_columns_extracted += r->n_columns_extracted; st->n_columns_downloaded += r->n_columns_downloaded;
            st->n_columns_downloaded_late += r->n_columns_downloaded_late;
            st->t_kernel_cv_ms[1] += r->t_kernel_ms[1];
        }
    }
    hs_sr_result* R = concat_sr_parts(p, parts, sparse, st);
    *out = R;
    return HS_OK;
}

int32_t hs_sr_window_size(const hs_sr_contig* contigs, int32_t n_contigs, int32_t amplicon) {
    return hs::sr_window_size(contigs, n_contigs, amplicon != 0);
}

// ---------------------------------------------------------------------------------------------------
// Several GPUs in one process: contigs are independent in both stages (call_variants.cpp:1276-1280, separate_reads.cpp:
// 1506-1508), so the stage-level calls shard them over the devices of hs_devices() -- longest-processing-time on a weight
// per contig -- one host thread (with its worker pool and HIP stream) per device; the per-shard results come back over each
// device's own PCIe link and are merged in contig order. HS_DEVICES="0,1,.." picks the devices (default: every visible
// one); a device may be listed more than once (two shards on one GPU: how the path is exercised on a single-GPU box).
// ---------------------------------------------------------------------------------------------------
static std::vector<int> device_list() {   // (extern "C" linkage block: plain static function)
    std::vector<int> d;
    if (const char* e = std::getenv("HS_DEVICES")) {
        for (const char* p = e; *p;) { char* end = nullptr; const long v = std::strtol(p, &end, 10); if (end == p) break; d.push_back((int)v); p = *end == ',' ? end + 1 : end; }
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (d.empty()) for (int i = 0; i < n; ++i) d.push_back(i);
    for (int& v : d) if (v < 0 || v >= n) v = 0;
    return d;
}
int hs_devices(int32_t* out, int32_t cap) {
    const std::vector<int> d = device_list();
    for (size_t i = 0; i < d.size() && (int32_t)i < cap; ++i) out[i] = d[i];
    return (int)d.size();
}
// longest-processing-time assignment, deterministic (ties by index); shards hold ascending contig ids
static std::vector<std::vector<int>> lpt_shards(const std::vector<double>& w, int parts) {
    std::vector<int> order(w.size());
    for (size_t i = 0; i < w.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return w[(size_t)a] != w[(size_t)b] ? w[(size_t)a] > w[(size_t)b] : a < b; });
    std::vector<double> load((size_t)parts, 0.0);
    std::vector<std::vector<int>> sh((size_t)parts);
    for (int i : order) {
        int best = 0;
        for (int k = 1; k < parts; ++k) if (load[(size_t)k] < load[(size_t)best]) best = k;
        sh[(size_t)best].push_back(i); load[(size_t)best] += w[(size_t)i];
    }
    for (auto& v : sh) std::sort(v.begin(), v.end());
    return sh;
}
// f(shard) on one host thread per shard, each bound to its device; first failure wins
static int run_on_devices(const std::vector<int>& devs, const std::function<int(int)>& f) {
    const int D = (int)devs.size();
    std::vector<int> rcs((size_t)D, 0);
    std::vector<std::string> errs((size_t)D);
    std::vector<std::thread> th;
    for (int k = 0; k < D; ++k)
        th.emplace_back([&, k] {
            if (hipSetDevice(devs[(size_t)k]) != hipSuccess) { rcs[(size_t)k] = HS_EHIP; errs[(size_t)k] = "hipSetDevice failed"; return; }
            set_wait_policy();
            rcs[(size_t)k] = f(k);
            if (rcs[(size_t)k]) errs[(size_t)k] = hs_last_error();
        });
    for (auto& t : th) t.join();
    for (int k = 0; k < D; ++k) if (rcs[(size_t)k]) { set_error(errs[(size_t)k]); return rcs[(size_t)k]; }
    return HS_OK;
}

int hs_sr_run(const hs_sr_contig* contigs, int32_t n_contigs, int32_t window_size, float error_rate, int32_t low_memory,
              uint32_t seed, int32_t n_threads, hs_sr_result** out) {
    if (int rc = require_device()) return rc;
    if (!out || n_contigs < 0) { set_error("hs_sr_run: bad arguments"); return HS_EINVAL; }
    const std::vector<int> devs = device_list();
    if (devs.size() <= 1 || n_contigs < 2) {
        HipSrOps ops;
        return hs::sr_run(ops, contigs, n_contigs, window_size, error_rate, low_memory, seed, n_threads, out);
    }
    // weight of a contig in stage 4: its sim / diff matrices (N^2) and its windows (~ length x depth)
    std::vector<double> w((size_t)n_contigs);
    for (int c = 0; c < n_contigs; ++c) w[(size_t)c] = 16.0 * contigs[c].n_reads * (double)contigs[c].n_reads + (double)contigs[c].n_snps * contigs[c].n_reads + 1.0;
    const int D = (int)std::min<size_t>(devs.size(), (size_t)n_contigs);
    const std::vector<std::vector<int>> shards = lpt_shards(w, D);
    std::vector<hs_sr_result*> parts((size_t)D, nullptr);
    const int per = n_threads > 0 ? std::max(1, n_threads / D) : 0;
    const int rc = run_on_devices(std::vector<int>(devs.begin(), devs.begin() + D), [&](int k) {
        std::vector<hs_sr_contig> sub;
        for (int c : shards[(size_t)k]) sub.push_back(contigs[c]);
        HipSrOps ops;
        return hs::sr_run(ops, sub.data(), (int32_t)sub.size(), window_size, error_rate, low_memory, seed, per, &parts[(size_t)k]);
    });
    if (rc) { for (hs_sr_result* r : parts) if (r) hs::free_sr_result(r); return rc; }
    // merge in contig order
    std::vector<std::pair<int, int>> where((size_t)n_contigs);   // contig -> (shard, index in shard)
    for (int k = 0; k < D; ++k) for (size_t i = 0; i < shards[(size_t)k].size(); ++i) where[(size_t)shards[(size_t)k][i]] = std::make_pair(k, (int)i);
    hs_sr_result* R = (hs_sr_result*)std::calloc(1, sizeof(hs_sr_result));
    int64_t W = 0, NL = 0;
    for (hs_sr_result* r : parts) { W += r->win_off[r->n_contigs]; NL += r->label_off[r->win_off[r->n_contigs]]; }
    R->n_contigs = n_contigs;
    R->win_off = (int64_t*)std::malloc(((size_t)n_contigs + 1) * sizeof(int64_t));
    R->win_start = (int32_t*)std::malloc((size_t)std::max<int64_t>(1, W) * sizeof(int32_t));
    R->win_end = (int32_t*)std::malloc((size_t)std::max<int64_t>(1, W) * sizeof(int32_t));
    R->label_off = (int64_t*)std::malloc(((size_t)W + 1) * sizeof(int64_t));
    R->labels = hs::sr_labels_alloc((size_t)NL);
    // contig order: where every contig's windows and labels go (prefix sums over the contigs), then the copies on all host threads
    R->win_off[0] = 0; R->label_off[0] = 0;
    std::vector<int64_t> lab0((size_t)n_contigs + 1, 0);
    for (int c = 0; c < n_contigs; ++c) {
        const hs_sr_result* r = parts[(size_t)where[(size_t)c].first];
        const int i = where[(size_t)c].second;
        R->win_off[c + 1] = R->win_off[c] + (r->win_off[i + 1] - r->win_off[i]);
        lab0[(size_t)c + 1] = lab0[(size_t)c] + (r->label_off[r->win_off[i + 1]] - r->label_off[r->win_off[i]]);
    }
    R->label_off[W] = NL;
    hs::hs_parallel_for(n_contigs, host_threads(), [&](int c) {
        const hs_sr_result* r = parts[(size_t)where[(size_t)c].first];
        const int i = where[(size_t)c].second;
        const int64_t q0 = r->win_off[i], nq = r->win_off[i + 1] - q0, w0 = R->win_off[c];
        if (nq == 0) return;
        std::memcpy(R->win_start + w0, r->win_start + q0, (size_t)nq * sizeof(int32_t));
        std::memcpy(R->win_end + w0, r->win_end + q0, (size_t)nq * sizeof(int32_t));
        const int64_t shift = lab0[(size_t)c] - r->label_off[q0];
        for (int64_t q = 0; q < nq; ++q) R->label_off[w0 + q] = r->label_off[q0 + q] + shift;
        std::memcpy(R->labels + lab0[(size_t)c], r->labels + r->label_off[q0], (size_t)(r->label_off[q0 + nq] - r->label_off[q0]) * sizeof(int32_t));
    });
    for (hs_sr_result* r : parts) {
        R->t_device_ms += r->t_device_ms; R->t_host_ms += r->t_host_ms; R->n_cw_instances += r->n_cw_instances;
        for (int k = 0; k < 4; ++k) R->t_kernel_ms[k] += r->t_kernel_ms[k];
        R->t_kernel_graph_ms += r->t_kernel_graph_ms; R->n_graph_rows_host += r->n_graph_rows_host; R->n_windows_finished_on_host += r->n_windows_finished_on_host;
        R->n_cw_sweeps += r->n_cw_sweeps; R->cw_bytes += r->cw_bytes; R->graph_nnz += r->graph_nnz; R->n_graph_rows += r->n_graph_rows; R->simdiff_bytes += r->simdiff_bytes;
        hs::free_sr_result(r);
    }
    *out = R;
    return HS_OK;
}

// Stage 3 from host buffers (what parse_reads / parse_assembly / parse_SAM produce, flattened as for hs_cv_batch_create),
// sharded over hs_devices(): every shard uploads its contigs, the reads its records refer to and their CIGARs to its device,
// runs hs_cv_run there and the results are merged in contig order (error rate over the whole job, call_variants.cpp:1312-1315).
int hs_cv_run_host(const uint8_t* h_contig_seq, const int64_t* h_contig_off, int32_t n_contigs, const uint8_t* h_read_seq, const int64_t* h_read_off,
                   int32_t n_reads, const int32_t* h_rec_read, const int32_t* h_rec_pos, const uint8_t* h_rec_strand, const int64_t* h_rec_cig_off,
                   const uint32_t* h_cigar, const int32_t* h_contig_rec_off, float automatic_snp_threshold, int32_t n_threads, hs_cv_result** out) {
    if (int rc = require_device()) return rc;
    if (!out || n_contigs < 0) { set_error("hs_cv_run_host: bad arguments"); return HS_EINVAL; }
    const std::vector<int> devs = device_list();
    if (devs.size() <= 1 || n_contigs < 2) {
        hs_cv_batch* b = nullptr;
        if (int rc = hs_cv_batch_create(h_contig_seq, h_contig_off, n_contigs, h_read_seq, h_read_off, n_reads, h_rec_read, h_rec_pos, h_rec_strand,
                                        h_rec_cig_off, h_cigar, h_contig_rec_off, &b)) return rc;
        const int rc = hs_cv_run(b, automatic_snp_threshold, n_threads, out);
        hs_cv_batch_destroy(b);
        return rc;
    }
    // weight of a contig in stage 3 ~ its aligned bp ~ the bases of the reads aligned to it
    std::vector<double> w((size_t)n_contigs, 1.0);
    for (int c = 0; c < n_contigs; ++c)
        for (int r = h_contig_rec_off[c]; r < h_contig_rec_off[c + 1]; ++r) w[(size_t)c] += (double)(h_read_off[h_rec_read[r] + 1] - h_read_off[h_rec_read[r]]);
    const int D = (int)std::min<size_t>(devs.size(), (size_t)n_contigs);
    const std::vector<std::vector<int>> shards = lpt_shards(w, D);
    std::vector<hs_cv_result*> parts((size_t)D, nullptr);
    const int per = n_threads > 0 ? std::max(1, n_threads / D) : 0;
    const int rc = run_on_devices(std::vector<int>(devs.begin(), devs.begin() + D), [&](int k) {
        // the shard's own flat arrays: contigs, the reads its records use (first-use order), records, CIGARs
        const std::vector<int>& ids = shards[(size_t)k];
        std::vector<int64_t> c_off(1, 0), r_off(1, 0), cig_off(1, 0);
        std::vector<int32_t> rec_off(1, 0), rec_read, rec_pos;
        std::vector<uint8_t> rec_strand;
        std::vector<uint8_t, hs::NoInitAlloc<uint8_t>> c_seq, r_seq;
        std::vector<uint32_t, hs::NoInitAlloc<uint32_t>> cig;
        std::vector<int32_t> new_id((size_t)n_reads, -1);
        int n_sub_reads = 0;
        int64_t c_total = 0, r_total = 0, cig_total = 0, rec_total = 0;
        for (int c : ids) {
            c_total += h_contig_off[c + 1] - h_contig_off[c];
            for (int r = h_contig_rec_off[c]; r < h_contig_rec_off[c + 1]; ++r) {
                rec_total++; cig_total += h_rec_cig_off[r + 1] - h_rec_cig_off[r];
                if (new_id[(size_t)h_rec_read[r]] < 0) { new_id[(size_t)h_rec_read[r]] = n_sub_reads++; r_total += h_read_off[h_rec_read[r] + 1] - h_read_off[h_rec_read[r]]; }
            }
        }
        c_seq.resize((size_t)c_total); r_seq.resize((size_t)r_total); cig.resize((size_t)cig_total);
        rec_read.reserve((size_t)rec_total); rec_pos.reserve((size_t)rec_total); rec_strand.reserve((size_t)rec_total);
        r_off.assign((size_t)n_sub_reads + 1, 0);
        std::fill(new_id.begin(), new_id.end(), -1);
        n_sub_reads = 0;
        int64_t cw = 0, rw = 0, gw = 0;
        for (int c : ids) {
            const int64_t L = h_contig_off[c + 1] - h_contig_off[c];
            std::memcpy(c_seq.data() + cw, h_contig_seq + h_contig_off[c], (size_t)L); cw += L; c_off.push_back(cw);
            for (int r = h_contig_rec_off[c]; r < h_contig_rec_off[c + 1]; ++r) {
                const int g = h_rec_read[r];
                if (new_id[(size_t)g] < 0) {
                    new_id[(size_t)g] = n_sub_reads;
                    const int64_t n = h_read_off[g + 1] - h_read_off[g];
                    std::memcpy(r_seq.data() + rw, h_read_seq + h_read_off[g], (size_t)n); rw += n;
                    r_off[(size_t)++n_sub_reads] = rw;
                }
                rec_read.push_back(new_id[(size_t)g]); rec_pos.push_back(h_rec_pos[r]); rec_strand.push_back(h_rec_strand[r]);
                const int64_t n = h_rec_cig_off[r + 1] - h_rec_cig_off[r];
                std::memcpy(cig.data() + gw, h_cigar + h_rec_cig_off[r], (size_t)n * sizeof(uint32_t)); gw += n; cig_off.push_back(gw);
            }
            rec_off.push_back((int32_t)rec_read.size());
        }
        hs_cv_batch* b = nullptr;
        if (int rc2 = hs_cv_batch_create(c_seq.data(), c_off.data(), (int32_t)ids.size(), r_seq.data(), r_off.data(), n_sub_reads, rec_read.data(), rec_pos.data(),
                                         rec_strand.data(), cig_off.data(), cig.data(), rec_off.data(), &b)) return rc2;
        const int rc2 = hs_cv_run(b, automatic_snp_threshold, per, &parts[(size_t)k]);
        hs_cv_batch_destroy(b);
        return rc2;
    });
    if (rc) { for (hs_cv_result* r : parts) if (r) hs::free_cv_result(r); return rc; }
    std::vector<std::pair<int, int>> where((size_t)n_contigs);
    for (int k = 0; k < D; ++k) for (size_t i = 0; i < shards[(size_t)k].size(); ++i) where[(size_t)shards[(size_t)k][i]] = std::make_pair(k, (int)i);
    hs_cv_result* R = (hs_cv_result*)std::calloc(1, sizeof(hs_cv_result));
    int64_t S = 0, E = 0;
    for (hs_cv_result* r : parts) { S += r->snp_off[r->n_contigs]; E += r->col_off[r->snp_off[r->n_contigs]]; }
    R->n_contigs = n_contigs;
    R->mean_distance = (float*)std::malloc(std::max<size_t>(1, (size_t)n_contigs) * sizeof(float));
    R->depth = (float*)std::malloc(std::max<size_t>(1, (size_t)n_contigs) * sizeof(float));
    R->snp_off = (int64_t*)std::malloc(((size_t)n_contigs + 1) * sizeof(int64_t));
    R->snp_pos = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->snp_ref = (uint8_t*)std::malloc(std::max<int64_t>(1, S));
    R->snp_alt = (uint8_t*)std::malloc(std::max<int64_t>(1, S));
    R->snp_n_ref = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->snp_n_alt = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->col_off = (int64_t*)std::malloc(((size_t)S + 1) * sizeof(int64_t));
    R->col_idx = (int32_t*)std::malloc(std::max<int64_t>(1, E) * sizeof(int32_t));
    R->col_code = (uint8_t*)std::malloc(std::max<int64_t>(1, E));
    // contig order: where every contig's SNPs and column entries go (prefix sums over the contigs), then the copies on all host threads
    R->snp_off[0] = 0; R->col_off[0] = 0;
    std::vector<int64_t> ent0((size_t)n_contigs + 1, 0);
    float total_error = 0; int n_err = 0;
    for (int c = 0; c < n_contigs; ++c) {
        const hs_cv_result* r = parts[(size_t)where[(size_t)c].first];
        const int i = where[(size_t)c].second;
        R->mean_distance[c] = r->mean_distance[i]; R->depth[c] = r->depth[i];
        if (r->mean_distance[i] > 0) { total_error += r->mean_distance[i]; n_err++; }      // call_variants.cpp:1312-1315, contig order
        R->snp_off[c + 1] = R->snp_off[c] + (r->snp_off[i + 1] - r->snp_off[i]);
        ent0[(size_t)c + 1] = ent0[(size_t)c] + (r->col_off[r->snp_off[i + 1]] - r->col_off[r->snp_off[i]]);
    }
    R->col_off[S] = E;
    hs::hs_parallel_for(n_contigs, host_threads(), [&](int c) {
        const hs_cv_result* r = parts[(size_t)where[(size_t)c].first];
        const int i = where[(size_t)c].second;
        const int64_t q0 = r->snp_off[i], nq = r->snp_off[i + 1] - q0, s0 = R->snp_off[c];
        if (nq == 0) return;
        std::memcpy(R->snp_pos + s0, r->snp_pos + q0, (size_t)nq * sizeof(int32_t));
        std::memcpy(R->snp_ref + s0, r->snp_ref + q0, (size_t)nq);
        std::memcpy(R->snp_alt + s0, r->snp_alt + q0, (size_t)nq);
        std::memcpy(R->snp_n_ref + s0, r->snp_n_ref + q0, (size_t)nq * sizeof(int32_t));
        std::memcpy(R->snp_n_alt + s0, r->snp_n_alt + q0, (size_t)nq * sizeof(int32_t));
        const int64_t shift = ent0[(size_t)c] - r->col_off[q0];
        for (int64_t q = 0; q < nq; ++q) R->col_off[s0 + q] = r->col_off[q0 + q] + shift;
        const int64_t ne = r->col_off[q0 + nq] - r->col_off[q0];
        std::memcpy(R->col_idx + ent0[(size_t)c], r->col_idx + r->col_off[q0], (size_t)ne * sizeof(int32_t));
        std::memcpy(R->col_code + ent0[(size_t)c], r->col_code + r->col_off[q0], (size_t)ne);
    });
    R->error_rate = total_error / n_err; R->n_contigs_with_error_rate = n_err;
    for (hs_cv_result* r : parts) {
        R->t_device_ms += r->t_device_ms; R->t_host_ms += r->t_host_ms;
        for (int k = 0; k < 4; ++k) R->t_kernel_ms[k] += r->t_kernel_ms[k];
        R->t_kernel_k4_ms += r->t_kernel_k4_ms;
        R->n_columns_extracted += r->n_columns_extracted; R->n_columns_downloaded += r->n_columns_downloaded; R->n_columns_downloaded_late += r->n_columns_downloaded_late;
        hs::free_cv_result(r);
    }
    *out = R;
    return HS_OK;
}

}  // extern "C"
