// hs_driver.cpp -- stage drivers: order of device launches and host glue for HS_call_variants and
// HS_separate_reads, independent of how the device interface is implemented (see hs_driver.h).
#include "hs_driver.h"

#include <sys/prctl.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace hs {

int host_threads();

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Persistent worker pool: the drivers issue a dozen short parallel sections per call and spawning 200+ std::threads
// for each of them costs more than the sections themselves. One pool per calling thread: independent batches driven from
// different host threads (each on its own HIP stream) overlap their serial sections and device waits.
class WorkerPool {
public:
    static WorkerPool& get() { static thread_local WorkerPool* p = new WorkerPool(); return *p; }
    void run(int n, int n_threads, const std::function<void(int)>& f) {
        if (n <= 0) return;
        if (n_threads <= 1 || n == 1) { for (int i = 0; i < n; ++i) f(i); return; }
        std::lock_guard<std::mutex> serial(run_mu_);          // one parallel section at a time
        ensure(std::min(n_threads, n) - 1);
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = &f; n_ = n; next_.store(0); active_ = std::min((int)workers_.size(), std::min(n_threads, n) - 1); pending_ = active_; gen_++;
            // indices are handed out in runs: with thousands of tiny items (one per clustering window) a shared counter bumped
            // once per item costs more than the items
            grain_ = std::max(1, n / (8 * (active_ + 1)));
        }
        cv_.notify_all();
        const int grain = grain_;
        for (;;) { const int i = next_.fetch_add(grain); if (i >= n) break; for (int k = i; k < std::min(n, i + grain); ++k) f(k); }
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    void ensure(int want) {
        want = std::min(want, 255);
        while ((int)workers_.size() < want) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] { ::prctl(PR_SET_NAME, "hs-pool", 0, 0, 0); loop(id); });      // (named: per-thread CPU accounting, bench.py HS_BENCH_THREAD_CPU=1)
            workers_.back().detach();
        }
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job; int n, grain;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen && id < active_; });
                seen = gen_; job = job_; n = n_; grain = grain_;
            }
            for (;;) { const int i = next_.fetch_add(grain); if (i >= n) break; for (int k = i; k < std::min(n, i + grain); ++k) (*job)(k); }
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* job_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0, pending_ = 0, grain_ = 1;
    uint64_t gen_ = 0;
};

// One pool for the whole process (HS_SHARED_POOL=1; default: a private pool per calling thread -- measured on the 16-core box: the same
// step time, 10 % more CPU time in wake-ups). The contig groups of a
// pipeline are on the host at different moments -- the device hands them their candidate columns one after the other -- and the
// step ends with the LAST group's chain: with a private pool of n_threads / groups workers each, that group walks its contigs on six
// threads while the cores the other groups have left stand idle. Here every parallel section is a job in one list; the
// workers (as many as the process has usable cores) take chunks of the oldest job that has any left, the caller works on its own
// job too. No oversubscription when all groups are on the host at once, all cores for a group that is there alone.
class SharedPool {
public:
    static SharedPool& get() { static SharedPool* p = new SharedPool(); return *p; }
    static bool on() { static const bool v = []() { const char* e = std::getenv("HS_SHARED_POOL"); return e && e[0] == '1'; }(); return v; }
    void run(int n, int n_threads, const std::function<void(int)>& f) {
        if (n <= 0) return;
        if (n_threads <= 1 || n == 1) { for (int i = 0; i < n; ++i) f(i); return; }
        Job job;
        job.f = &f; job.n = n;
        {
            std::lock_guard<std::mutex> g(mu_);
            ensure();
            job.grain = std::max(1, n / (8 * (std::min(n_threads, (int)workers_.size() + 1))));      // (workers_ is read under the lock: ensure() may grow it)
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        for (;;) { const int i = job.next.fetch_add(job.grain); if (i >= n) break; const int e = std::min(n, i + job.grain); for (int k = i; k < e; ++k) f(k); job.done.fetch_add(e - i); }
        std::unique_lock<std::mutex> g(mu_);
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job));      // (nothing left to hand out; chunks in flight finish below)
        done_cv_.wait(g, [&] { return job.done.load() >= n; });
    }
private:
    struct Job { const std::function<void(int)>* f = nullptr; int n = 0, grain = 1; std::atomic<int> next{0}, done{0}; };
    void ensure() {
        static const int want = []() { const char* e = std::getenv("HS_POOL_THREADS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : host_threads(); }();
        while ((int)workers_.size() < want) { workers_.emplace_back([this] { ::prctl(PR_SET_NAME, "hs-spool", 0, 0, 0); loop(); }); workers_.back().detach(); }
    }
    void loop() {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            Job* job = nullptr;
            for (Job* j : jobs_) if (j->next.load(std::memory_order_relaxed) < j->n) { job = j; break; }
            if (!job) { cv_.wait(g); continue; }
            const int i = job->next.fetch_add(job->grain);
            if (i >= job->n) continue;
            const int n = job->n, e = std::min(n, i + job->grain);
            const std::function<void(int)>* f = job->f;
            g.unlock();
            for (int k = i; k < e; ++k) (*f)(k);
            const bool last = job->done.fetch_add(e - i) + (e - i) >= n;      // (the job may be gone after this line)
            g.lock();
            if (last) done_cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    std::vector<Job*> jobs_;
};

template <class F>
void parallel_for(int n, int n_threads, F f) {
    std::function<void(int)> fn = f;
    if (SharedPool::on()) SharedPool::get().run(n, n_threads, fn); else WorkerPool::get().run(n, n_threads, fn);
}

template <class T> T* dup_vec(const std::vector<T>& v) {
    T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

void hs_parallel_for(int n, int n_threads, const std::function<void(int)>& f) { if (SharedPool::on()) SharedPool::get().run(n, n_threads, f); else WorkerPool::get().run(n, n_threads, f); }

// Worker threads for host-side passes when the caller names no count: the cores this process may actually use -- the cgroup
// CPU quota when there is one (a container limited to 16 of 256 CPUs must not start 255 workers) -- and never more than 32
int host_threads() {
    static const int n = [] {
        long q = -1, per = 100000;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { char a[32]; if (std::fscanf(f, "%31s %ld", a, &per) == 2 && std::strcmp(a, "max") != 0) q = std::atol(a); std::fclose(f); }
        int t = (int)std::thread::hardware_concurrency();
        if (q > 0 && per > 0) t = std::min<int>(t, (int)((q + per - 1) / per));
        return std::max(1, std::min(t, 32));
    }();
    return n;
}

void free_cv_result(hs_cv_result* r) {
    if (!r) return;
    std::free(r->mean_distance); std::free(r->depth); std::free(r->snp_off); std::free(r->snp_pos); std::free(r->snp_ref);
    std::free(r->snp_alt); std::free(r->snp_n_ref); std::free(r->snp_n_alt); std::free(r->col_off);
    if (!r->entries_borrowed) { std::free(r->col_idx); std::free(r->col_code); }
    std::free(r);
}
// The dense label array of a result is tens of megabytes, above the largest mmap threshold glibc accepts: malloc'ed, every
// call would map it, fault it in page by page on all threads and unmap it again (15 % of the host's CPU time on the
// 500-contig bench). A few big blocks are therefore kept and handed out again; a 64-byte header carries the capacity.
namespace {
struct LabelBlockHeader { uint64_t magic; size_t cap; char pad[48]; };
constexpr uint64_t kLabelMagic = 0x48534c4142454c53ull;
constexpr size_t kLabelKeepMin = 8u << 20, kLabelKeepBlocks = 4;
std::mutex g_label_mu;
std::vector<LabelBlockHeader*> g_label_free;
}  // namespace

int32_t* sr_labels_alloc(size_t n_labels) {
    const size_t need = std::max<size_t>(1, n_labels) * sizeof(int32_t);
    {
        std::lock_guard<std::mutex> g(g_label_mu);
        for (size_t i = 0; i < g_label_free.size(); ++i) {
            LabelBlockHeader* h = g_label_free[i];
            if (h->cap >= need && h->cap / 2 <= need) { g_label_free.erase(g_label_free.begin() + (long)i); return (int32_t*)(h + 1); }
        }
    }
    const size_t cap = need >= kLabelKeepMin ? need + need / 8 : need;      // room for the next batch to be a little bigger
    LabelBlockHeader* h = (LabelBlockHeader*)std::malloc(sizeof(LabelBlockHeader) + cap);
    if (!h) return nullptr;
    h->magic = kLabelMagic; h->cap = cap;
    return (int32_t*)(h + 1);
}

void sr_labels_free(int32_t* labels) {
    if (!labels) return;
    LabelBlockHeader* h = (LabelBlockHeader*)labels - 1;
    if (h->magic != kLabelMagic) { std::free(labels); return; }      // (not ours: a plain malloc)
    if (h->cap >= kLabelKeepMin) {
        std::lock_guard<std::mutex> g(g_label_mu);
        if (g_label_free.size() < kLabelKeepBlocks) { g_label_free.push_back(h); return; }
    }
    std::free(h);
}

void free_sr_result(hs_sr_result* r) {
    if (!r) return;
    std::free(r->win_off); std::free(r->win_start); std::free(r->win_end); std::free(r->label_off); sr_labels_free(r->labels);
    std::free(r);
}

static std::atomic<double> g_trace_origin{0.0};
void set_trace_origin() { g_trace_origin.store(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()); }
double trace_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - g_trace_origin.load(); }
namespace {
struct Laps {   // HS_TIMING: fine-grained wall clock of a stage driver (HS_TIMING=cpu: wall/CPU time of the calling thread; HS_TIMING=abs: the
                // end of every lap in ms since the start of the pipeline call, so that the chains of the contig groups can be laid side by side)
    bool on = std::getenv("HS_TIMING") != nullptr;
    bool cpu = on && std::string(std::getenv("HS_TIMING")) == "cpu";
    bool abs_ = on && std::string(std::getenv("HS_TIMING")) == "abs";
    double t = now_ms(), c = cpu_ms();
    std::string line;
    const char* tag;
    explicit Laps(const char* t_) : tag(t_) {}
    static double cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    void lap(const char* what) {
        if (!on) return;
        const double n = now_ms();
        char buf[96];
        if (abs_) std::snprintf(buf, sizeof buf, " %s @%.2f", what, n - g_trace_origin.load());
        else if (cpu) { const double cn = cpu_ms(); std::snprintf(buf, sizeof buf, " %s %.2f/%.2f", what, n - t, cn - c); c = cn; }
        else std::snprintf(buf, sizeof buf, " %s %.2f", what, n - t);
        line += buf; t = n;
    }
    ~Laps() { if (on) std::fprintf(stderr, "[hs timing] %s laps (ms) thread %zu:%s\n", tag, std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000, line.c_str()); }
};
}  // namespace

// ---------------------------------------------------------------------------------------------------
// stage 3
// ---------------------------------------------------------------------------------------------------
// The whole-batch streaming pass (K0 CIGAR scan, K1 pileup): one launch each over every contig of the batch; the per-record
// counters come back (error rate of every contig, call_variants.cpp:434).
int cv_pileup(CvDeviceOps& dev, const CvMeta& b, CvSelection& sel) {
    const double t_start = now_ms();
    for (int k = 0; k < 4; ++k) sel.k_ms[k] = 0;
    sel.rec_stats.assign((size_t)b.n_rec * 4, 0);
    if (int rc = dev.pileup(sel.rec_stats, sel.k_ms)) return rc;
    sel.t_device_ms = now_ms() - t_start; sel.t_host_ms = 0;
    return HS_OK;
}

// Stage 3 for the contigs [c0, c1) of the batch on top of the pileup. The columns never leave the device: it extracts them,
// names their leading codes, picks the candidates (K2, K3, K3b, V1) and hands the candidates over; the partition logic
// (loops A and B: sequential per contig, libm inside) runs here; the final partitions go back for loops C / D and the merge
// of the SNP lists, and the SNPs come out. Ranges are independent: several may run concurrently from different host threads,
// each with its own device interface (stream).
int cv_run_range(CvDeviceOps& dev, const CvMeta& b, const std::vector<int32_t>& rec_stats, int c0, int c1, float automatic_snp_threshold, int n_threads,
                 hs_cv_result** out, bool resident, const std::function<void(const float*)>* on_mean_distance) {
    if (n_threads <= 0) n_threads = host_threads();
    if (c0 < 0 || c1 > b.n_contigs || c0 > c1) { set_error("cv_run_range: bad contig range"); return HS_EINVAL; }
    const int C = c1 - c0;
    const double t_start = now_ms();
    float k_ms[4] = {0, 0, 0, 0};
    float k_ms_k4 = 0;              // column x partition test
    std::vector<ContigCvResult> res((size_t)C);
    const bool derive = rec_stats.empty() && b.n_rec > 0;      // the device forms the mean distances (and the read minima) from its own counters
    std::vector<int32_t> min_reads((size_t)(derive ? 0 : C), 5);
    for (int c = 0; c < C; ++c) {   // call_variants.cpp:434 and :463-466 from the integer counters of K1; :565
        const int gc = c0 + c;
        ContigCvResult& o = res[(size_t)c];
        if (!derive) {
            int64_t nerr = 0, nlen = 0;
            for (int r = b.contig_rec_off[(size_t)gc]; r < b.contig_rec_off[(size_t)gc + 1]; ++r) { nerr += rec_stats[(size_t)r * 4 + 1]; nlen += rec_stats[(size_t)r * 4 + 2]; }
            o.mean_distance = mean_distance_from_counts(nerr, nlen);
            min_reads[(size_t)c] = o.mean_distance < 0.015 ? 3 : 5;
        }
        const int64_t L = b.contig_off[(size_t)gc + 1] - b.contig_off[(size_t)gc];
        const int64_t entries = b.pile_off[(size_t)b.contig_rec_off[(size_t)gc + 1]] - b.pile_off[(size_t)b.contig_rec_off[(size_t)gc]];
        o.depth = (float)((double)entries / (double)L);
    }
    // loop A of keep_only_robust_variants: the device interface may keep the contigs its kernel (k_loop_a) holds for itself -- it queues
    // their walk inside extract_candidates() and marks them in contig_on_device; the others' candidates arrive as bit sets and are walked
    // here meanwhile, then the device's partitions are collected and every contig goes through loop B (libm) on the host threads.
    CvCandidates cand;
    float k_ms_x[3] = {0, 0, 0};
    if (int rc = dev.extract_candidates(c0, c1, min_reads, automatic_snp_threshold, cand, k_ms_x, true)) return rc;
    k_ms[1] = k_ms_x[0]; k_ms[2] = k_ms_x[1] + k_ms_x[2];
    if (derive) {
        if ((int)cand.contig_mean_distance.size() != C) { set_error("cv_run_range: the device interface did not report the contigs' mean distances"); return HS_EINVAL; }
        for (int c = 0; c < C; ++c) res[(size_t)c].mean_distance = cand.contig_mean_distance[(size_t)c];
    }
    if (on_mean_distance) { std::vector<float> md((size_t)C); for (int c = 0; c < C; ++c) md[(size_t)c] = res[(size_t)c].mean_distance; (*on_mean_distance)(md.data()); }
    const double t_dev_done = now_ms();
    Laps laps("cv glue");
    std::vector<int64_t> cand_base((size_t)C + 1, 0);
    for (int c = 0; c < C; ++c) cand_base[(size_t)c + 1] = cand_base[(size_t)c] + cand.contig_n_cand[(size_t)c];
    if (cand_base[(size_t)C] != cand.n_cand) { set_error("cv_run_range: candidate counts do not add up"); return HS_EINVAL; }

    // ---- the partition logic: loop A (device, or host), loop B on the host ----
    std::vector<CvContigState*> cst((size_t)C, nullptr);
    for (int c = 0; c < C; ++c) cst[(size_t)c] = cv_state_new();
    std::vector<int32_t> n_reads_of((size_t)C);
    parallel_for(C, n_threads, [&](int c) {
        const int gc = c0 + c;   // index in the batch
        const int r0 = b.contig_rec_off[(size_t)gc];
        const int n_reads_c = b.contig_rec_off[(size_t)gc + 1] - r0;
        n_reads_of[(size_t)c] = n_reads_c;
        cv_phase_begin(*cst[(size_t)c], n_reads_c, cand.contig_n_cand[(size_t)c], res[(size_t)c].mean_distance, res[(size_t)c]);
    });
    laps.lap("begin");
    const bool some_on_device = (int)cand.contig_on_device.size() == C;
    auto on_device = [&](int c) { return some_on_device && cand.contig_on_device[(size_t)c] != 0; };
    bool host_has_bits = cand.bits != nullptr || cand.n_cand == 0;
    auto candidates_of = [&](int c) {
        CandidateSet cs;
        cs.n = cand.contig_n_cand[(size_t)c];
        cs.rec = cand.rec + cand_base[(size_t)c]; cs.bits = cand.bits + cand_base[(size_t)c]; cs.words = cand.words;
        if (cand.idx) { cs.off = cand.off + cand_base[(size_t)c]; cs.idx = cand.idx; cs.code = cand.code; }      // (raw entries: the harness's cross-check only)
        return cs;
    };
    static const bool loop_b_pairs = []() { const char* e = std::getenv("HS_LOOP_B_PAIRS_ON_DEVICE"); return e && e[0] != '0'; }();
    const bool pairs_on_device = loop_b_pairs && dev.has_partition_pairs();
    const bool ranked = (int)b.rank_of.size() == b.n_rec && (int)b.orig_of.size() == b.n_rec;
    auto walk_on_host = [&](int c) {
        const int r0 = b.contig_rec_off[(size_t)(c0 + c)];
        cv_phase_a_host(*cst[(size_t)c], candidates_of(c), b.rec_pos.data() + r0, ranked ? b.rank_of.data() + r0 : nullptr, ranked ? b.orig_of.data() + r0 : nullptr);
        if (!pairs_on_device) cv_phase_b(*cst[(size_t)c], res[(size_t)c]);
    };
    // the host's contigs, those with the most candidates first: the walk is sequential per contig, and a large contig that starts last is
    // what the other threads of the group then wait for
    std::vector<int> host_list, dev_list;
    for (int c = 0; c < C; ++c) (on_device(c) ? dev_list : host_list).push_back(c);
    std::stable_sort(host_list.begin(), host_list.end(), [&](int x, int y) { return cand.contig_n_cand[(size_t)x] > cand.contig_n_cand[(size_t)y]; });
    if (!host_list.empty() && !host_has_bits) { set_error("cv_run_range: the device interface handed no bit sets of the candidate columns"); return HS_EINVAL; }
    parallel_for((int)host_list.size(), n_threads, [&](int k) { walk_on_host(host_list[(size_t)k]); });
    laps.lap("loop_a_host");
    int n_dev_failed = 0;
    if (!dev_list.empty()) {
        CvLoopAResult la;
        if (int rc = dev.collect_partitions(la)) return rc;
        if ((int)la.failed.size() != C || (int)la.part_base.size() != C + 1) { set_error("cv_run_range: the device interface did not hand over the partitions of its contigs"); return HS_EINVAL; }
        laps.lap("loop_a_device");
        std::vector<int> redo;
        for (int c : dev_list) if (la.failed[(size_t)c]) redo.push_back(c);
        n_dev_failed = (int)redo.size();
        if (!redo.empty()) {      // (rare: a contig the kernel's tables did not hold after all -- its candidates as bit sets, walked here)
            if (int rc = dev.fetch_candidates(cand)) return rc;
            if (cand.n_cand > 0 && !cand.bits) { set_error("cv_run_range: the device interface handed no bit sets of the candidate columns"); return HS_EINVAL; }
        }
        std::stable_sort(dev_list.begin(), dev_list.end(), [&](int x, int y) { return cand.contig_n_cand[(size_t)x] > cand.contig_n_cand[(size_t)y]; });
        parallel_for((int)dev_list.size(), n_threads, [&](int k) {
            const int c = dev_list[(size_t)k];
            if (la.failed[(size_t)c]) { walk_on_host(c); return; }
            const int r0 = b.contig_rec_off[(size_t)(c0 + c)];
            cv_phase_a_import(*cst[(size_t)c], b.rec_pos.data() + r0, (int)(la.part_base[(size_t)c + 1] - la.part_base[(size_t)c]), la.rec + la.part_base[(size_t)c], la.bits, la.cnt,
                              ranked ? b.rank_of.data() + r0 : nullptr, ranked ? b.orig_of.data() + r0 : nullptr);
            if (!pairs_on_device) cv_phase_b(*cst[(size_t)c], res[(size_t)c]);
        });
    }
    if (pairs_on_device) {
        // loop B with distance(Partition, Partition) from the device (HS_LOOP_B_PAIRS_ON_DEVICE=1): every pair of the partitions that
        // pass loop B's gate, in one launch for the range; the host walks loop B with the table and only recomputes a pair whose
        // final partition has been merged into meanwhile
        std::vector<int32_t> n_surv((size_t)C, 0);
        parallel_for(C, n_threads, [&](int c) { n_surv[(size_t)c] = cv_loop_b_survivors(*cst[(size_t)c]); });
        std::vector<int64_t> part_first((size_t)C + 1, 0), elem_first((size_t)C + 1, 0), pair_first((size_t)C + 1, 0);
        for (int c = 0; c < C; ++c) {
            part_first[(size_t)c + 1] = part_first[(size_t)c] + n_surv[(size_t)c];
            elem_first[(size_t)c + 1] = elem_first[(size_t)c] + (int64_t)n_surv[(size_t)c] * n_reads_of[(size_t)c];
            pair_first[(size_t)c + 1] = pair_first[(size_t)c] + (int64_t)n_surv[(size_t)c] * (n_surv[(size_t)c] - 1) / 2;
        }
        std::vector<int8_t> pstate((size_t)elem_first[(size_t)C]);
        std::vector<int32_t> pmore((size_t)elem_first[(size_t)C]), pless((size_t)elem_first[(size_t)C]), part_n((size_t)part_first[(size_t)C]);
        std::vector<int64_t> part_off((size_t)part_first[(size_t)C]);
        std::vector<int32_t> pair_a((size_t)pair_first[(size_t)C]), pair_b((size_t)pair_first[(size_t)C]);
        parallel_for(C, n_threads, [&](int c) {
            const int S = n_surv[(size_t)c], N = n_reads_of[(size_t)c];
            cv_export_survivors(*cst[(size_t)c], pstate.data() + elem_first[(size_t)c], pmore.data() + elem_first[(size_t)c], pless.data() + elem_first[(size_t)c]);
            for (int k = 0; k < S; ++k) { part_off[(size_t)part_first[(size_t)c] + k] = elem_first[(size_t)c] + (int64_t)k * N; part_n[(size_t)part_first[(size_t)c] + k] = N; }
            int64_t q = pair_first[(size_t)c];
            for (int j = 1; j < S; ++j) for (int i = 0; i < j; ++i, ++q) { pair_a[(size_t)q] = (int32_t)(part_first[(size_t)c] + i); pair_b[(size_t)q] = (int32_t)(part_first[(size_t)c] + j); }
        });
        std::vector<int32_t> table;
        if (int rc = dev.partition_pairs(pstate, pmore, pless, part_off, part_n, pair_a, pair_b, cv_three_sigma_table(), table)) return rc;
        parallel_for(C, n_threads, [&](int c) { cv_phase_b(*cst[(size_t)c], res[(size_t)c], table.data() + 8 * pair_first[(size_t)c]); });
        if (std::getenv("HS_TIMING")) std::fprintf(stderr, "[hs timing] cv loop B: %ld partition pairs of %ld gated partitions from the device\n", (long)pair_first[(size_t)C], (long)part_first[(size_t)C]);
    }
    if (std::getenv("HS_TIMING") && !dev_list.empty()) std::fprintf(stderr, "[hs timing] cv loop A: %zu of %d contigs on the device (%d handed back), %zu on the host\n", dev_list.size(), C, n_dev_failed, host_list.size());
    laps.lap("phase_ab");
    // ---- loops C and D and the merge of the SNP lists on the device, against the final partitions ----
    CvSnpSet snps;
    {
        CvPartitionTest t;
        t.part_off.assign((size_t)C + 1, 0);
        t.contig_n_reads.resize((size_t)C);
        std::vector<int64_t> st_base((size_t)C + 1, 0);
        for (int c = 0; c < C; ++c) {
            t.contig_n_reads[(size_t)c] = b.contig_rec_off[(size_t)(c0 + c) + 1] - b.contig_rec_off[(size_t)(c0 + c)];
            const int nf = cv_final_partitions(*cst[(size_t)c]);
            t.part_off[(size_t)c + 1] = t.part_off[(size_t)c] + nf;
            st_base[(size_t)c + 1] = st_base[(size_t)c] + (int64_t)nf * t.contig_n_reads[(size_t)c];
        }
        t.part_state.resize((size_t)st_base[(size_t)C]); t.part_state_off.resize((size_t)t.part_off[(size_t)C]);
        parallel_for(C, n_threads, [&](int c) {
            cv_export_partitions(*cst[(size_t)c], t.part_state.data() + st_base[(size_t)c], st_base[(size_t)c], t.part_state_off.data() + t.part_off[(size_t)c]);
            cv_state_free(cst[(size_t)c]);
        });
        laps.lap("k4_prep");
        if (int rc = dev.finish_columns(t, !resident, snps, &k_ms_k4)) return rc;
        laps.lap("k4+snps");
    }
    const double t_glue_done = now_ms();

    hs_cv_result* R = (hs_cv_result*)std::calloc(1, sizeof(hs_cv_result));
    R->n_contigs = C;
    std::vector<float> md((size_t)C), dp((size_t)C);
    std::vector<int64_t> snp_off((size_t)C + 1, 0);
    float total_error = 0; int n_err_contigs = 0;
    for (int c = 0; c < C; ++c) {
        md[(size_t)c] = res[(size_t)c].mean_distance; dp[(size_t)c] = res[(size_t)c].depth;
        if (res[(size_t)c].mean_distance > 0) { total_error += res[(size_t)c].mean_distance; n_err_contigs++; }   // call_variants.cpp:1312-1315
        snp_off[(size_t)c + 1] = snp_off[(size_t)c] + snps.contig_n_snp[(size_t)c];
    }
    const int64_t S = snps.n_snp, E = snps.n_entries;
    if (snp_off[(size_t)C] != S) { set_error("cv_run_range: SNP counts do not add up"); return HS_EINVAL; }
    R->snp_pos = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->snp_ref = (uint8_t*)std::malloc(std::max<int64_t>(1, S));
    R->snp_alt = (uint8_t*)std::malloc(std::max<int64_t>(1, S));
    R->snp_n_ref = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->snp_n_alt = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->col_off = (int64_t*)std::malloc((S + 1) * sizeof(int64_t));
    const bool with_entries = snps.idx != nullptr || E == 0;
    R->col_idx = with_entries ? (int32_t*)std::malloc(std::max<int64_t>(1, E) * sizeof(int32_t)) : nullptr;
    R->col_code = with_entries ? (uint8_t*)std::malloc(std::max<int64_t>(1, E)) : nullptr;
    R->n_columns_extracted = cand.n_columns;
    R->n_columns_downloaded = cand.n_cand;
    R->n_columns_downloaded_late = cand.n_tie;
    {
        const int nb = (int)std::min<int64_t>(std::max<int64_t>(1, S / 4096), 4 * (int64_t)n_threads);
        parallel_for(nb, n_threads, [&](int blk) {
            const int64_t a = S * blk / nb, e = S * (blk + 1) / nb;
            for (int64_t s = a; s < e; ++s) {
                const hs_colrec& r = snps.rec[s];
                R->snp_pos[s] = r.pos; R->snp_ref[s] = r.k0; R->snp_alt[s] = r.k1; R->snp_n_ref[s] = r.c0; R->snp_n_alt[s] = r.c1;
            }
            if (S) std::memcpy(R->col_off + a, snps.off + a, (size_t)(e - a) * sizeof(int64_t));
            if (with_entries && e > a) {
                const int64_t ea = snps.off[a], ee = snps.off[e];
                std::memcpy(R->col_idx + ea, snps.idx + ea, (size_t)(ee - ea) * sizeof(int32_t));
                std::memcpy(R->col_code + ea, snps.code + ea, (size_t)(ee - ea));
            }
        });
        R->col_off[S] = E;
        if (S == 0) R->col_off[0] = 0;
    }
    laps.lap("result");
    R->mean_distance = dup_vec(md); R->depth = dup_vec(dp); R->snp_off = dup_vec(snp_off);
    R->error_rate = total_error / n_err_contigs;      // call_variants.cpp:1377 (float / int)
    R->n_contigs_with_error_rate = n_err_contigs;
    R->t_kernel_ms[0] = k_ms[0]; R->t_kernel_ms[1] = k_ms[1]; R->t_kernel_ms[2] = k_ms[2]; R->t_kernel_ms[3] = k_ms[3]; R->t_kernel_k4_ms = k_ms_k4;
    R->t_device_ms = t_dev_done - t_start;
    R->t_host_ms = now_ms() - t_dev_done;
    if (std::getenv("HS_TIMING"))
        std::fprintf(stderr, "[hs timing] cv range [%d,%d): columns + candidates %.2f ms, host glue %.2f ms (up to the SNPs %.2f); columns: %lld extracted, %lld candidates, %lld with tied counts (%lld beyond 16 keys), %lld SNPs\n",
                     c0, c1, t_dev_done - t_start, R->t_host_ms, t_glue_done - t_dev_done, (long long)cand.n_columns,
                     (long long)cand.n_cand, (long long)cand.n_tie, (long long)cand.n_tie_big, (long long)S);
    *out = R;
    return HS_OK;
}


int cv_attach_entries(CvDeviceOps& dev, hs_cv_result* R, int n_threads, bool borrow) {
    if (!R || R->col_idx) return HS_OK;      // (nothing deferred: the entries came with the result, or there are none)
    const int64_t S = R->snp_off[R->n_contigs], E = R->col_off[S];
    CvSnpSet snps;
    if (int rc = dev.late_entries(snps)) return rc;
    if (E > 0 && (!snps.idx || !snps.code)) return HS_OK;      // (the implementation kept them on the device only)
    if (borrow && E > 0) {      // the caller keeps the implementation's block alive as long as the result: no copy, no pages to fault in
        R->col_idx = const_cast<int32_t*>(snps.idx); R->col_code = const_cast<uint8_t*>(snps.code); R->entries_borrowed = 1;
        return HS_OK;
    }
    R->col_idx = (int32_t*)std::malloc(std::max<int64_t>(1, E) * sizeof(int32_t));
    R->col_code = (uint8_t*)std::malloc(std::max<int64_t>(1, E));
    if (n_threads <= 0) n_threads = host_threads();
    const int nb = (int)std::min<int64_t>(std::max<int64_t>(1, E >> 20), 4 * (int64_t)n_threads);
    parallel_for(nb, n_threads, [&](int blk) {
        const int64_t a = E * blk / nb, e = E * (blk + 1) / nb;
        if (e > a) { std::memcpy(R->col_idx + a, snps.idx + a, (size_t)(e - a) * sizeof(int32_t)); std::memcpy(R->col_code + a, snps.code + a, (size_t)(e - a)); }
    });
    return HS_OK;
}

// results of two consecutive contig ranges put together (both are consumed); the error rate over all contigs, in contig order
// (call_variants.cpp:1312-1315,1377)
hs_cv_result* cv_concat_results(hs_cv_result* a, hs_cv_result* b) {
    hs_cv_result* R = (hs_cv_result*)std::calloc(1, sizeof(hs_cv_result));
    const int Ca = a->n_contigs, Cb = b->n_contigs, C = Ca + Cb;
    const int64_t Sa = a->snp_off[Ca], Sb = b->snp_off[Cb], S = Sa + Sb;
    const int64_t Ea = a->col_off[Sa], Eb = b->col_off[Sb], E = Ea + Eb;
    R->n_contigs = C;
    R->mean_distance = (float*)std::malloc(std::max(1, C) * sizeof(float)); R->depth = (float*)std::malloc(std::max(1, C) * sizeof(float));
    R->snp_off = (int64_t*)std::malloc(((size_t)C + 1) * sizeof(int64_t));
    R->snp_pos = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->snp_ref = (uint8_t*)std::malloc(std::max<int64_t>(1, S)); R->snp_alt = (uint8_t*)std::malloc(std::max<int64_t>(1, S));
    R->snp_n_ref = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t)); R->snp_n_alt = (int32_t*)std::malloc(std::max<int64_t>(1, S) * sizeof(int32_t));
    R->col_off = (int64_t*)std::malloc(((size_t)S + 1) * sizeof(int64_t));
    const bool with_entries = (a->col_idx || Ea == 0) && (b->col_idx || Eb == 0);
    R->col_idx = with_entries ? (int32_t*)std::malloc(std::max<int64_t>(1, E) * sizeof(int32_t)) : nullptr;
    R->col_code = with_entries ? (uint8_t*)std::malloc(std::max<int64_t>(1, E)) : nullptr;
    float total_error = 0; int n_err = 0;
    auto put = [&](const hs_cv_result* r, int cbase, int64_t sbase, int64_t ebase) {
        const int Cr = r->n_contigs; const int64_t Sr = r->snp_off[Cr], Er = r->col_off[Sr];
        for (int c = 0; c < Cr; ++c) {
            R->mean_distance[cbase + c] = r->mean_distance[c]; R->depth[cbase + c] = r->depth[c];
            if (r->mean_distance[c] > 0) { total_error += r->mean_distance[c]; n_err++; }
            R->snp_off[cbase + c] = sbase + r->snp_off[c];
        }
        if (Sr) {
            std::memcpy(R->snp_pos + sbase, r->snp_pos, (size_t)Sr * sizeof(int32_t)); std::memcpy(R->snp_ref + sbase, r->snp_ref, (size_t)Sr); std::memcpy(R->snp_alt + sbase, r->snp_alt, (size_t)Sr);
            std::memcpy(R->snp_n_ref + sbase, r->snp_n_ref, (size_t)Sr * sizeof(int32_t)); std::memcpy(R->snp_n_alt + sbase, r->snp_n_alt, (size_t)Sr * sizeof(int32_t));
        }
        for (int64_t q = 0; q < Sr; ++q) R->col_off[sbase + q] = ebase + r->col_off[q];
        if (with_entries && Er) { std::memcpy(R->col_idx + ebase, r->col_idx, (size_t)Er * sizeof(int32_t)); std::memcpy(R->col_code + ebase, r->col_code, (size_t)Er); }
        R->t_device_ms += r->t_device_ms; R->t_host_ms += r->t_host_ms;
        for (int k = 0; k < 4; ++k) R->t_kernel_ms[k] += r->t_kernel_ms[k];
        R->t_kernel_k4_ms += r->t_kernel_k4_ms;
        R->n_columns_extracted += r->n_columns_extracted; R->n_columns_downloaded += r->n_columns_downloaded; R->n_columns_downloaded_late += r->n_columns_downloaded_late;
    };
    put(a, 0, 0, 0); put(b, Ca, Sa, Ea);
    R->snp_off[C] = S; R->col_off[S] = E;
    R->error_rate = total_error / n_err; R->n_contigs_with_error_rate = n_err;
    free_cv_result(a); free_cv_result(b);
    return R;
}

int cv_run(CvDeviceOps& dev, const CvMeta& b, float automatic_snp_threshold, int n_threads, hs_cv_result** out) {
    CvSelection sel;
    if (int rc = cv_pileup(dev, b, sel)) return rc;
    if (int rc = cv_run_range(dev, b, sel.rec_stats, 0, b.n_contigs, automatic_snp_threshold, n_threads, out)) return rc;
    hs_cv_result* R = *out;
    R->t_kernel_ms[0] = sel.k_ms[0]; R->t_kernel_ms[3] = sel.k_ms[3];
    R->t_device_ms += sel.t_device_ms; R->t_host_ms += sel.t_host_ms;
    return HS_OK;
}

// ---------------------------------------------------------------------------------------------------
// stage 4
// ---------------------------------------------------------------------------------------------------
int sr_run(SrDeviceOps& dev, const hs_sr_contig* contigs, int32_t n_contigs, int32_t window_size, float error_rate,
           int32_t low_memory, uint32_t seed, int32_t n_threads, hs_sr_result** out, SrSparseLabels* sparse, SrWorkspace* keep, SrTaps* taps) {
    Laps laps("sr");
    if (n_threads <= 0) n_threads = host_threads();
    const int C = n_contigs;
    const bool lowmem = low_memory != 0;
    const double t_start = now_ms();
    double dev_ms = 0;
    float k_ms[4] = {0, 0, 0, 0};

    // the SNP columns may be with the device already (stage 3 left them there): the host then only has their offsets, until
    // something needs to walk them here (need_columns)
    const bool resident = dev.columns_resident();
    std::vector<hs_sr_contig> cs_local(contigs, contigs + C);      // (the column pointers are filled in if the columns are fetched)
    std::vector<SrContigState> st_local;
    if (keep) keep->st.resize((size_t)C); else st_local.resize((size_t)C);
    std::vector<SrContigState>& st = keep ? keep->st : st_local;
    parallel_for(C, n_threads, [&](int c) {
        SrContigState& s = st[(size_t)c];
        s.c = &cs_local[(size_t)c];
        s.N = contigs[c].n_reads;
        s.low_memory_now = lowmem || sr_coverage_above_1000(contigs[c]);   // separate_reads.cpp:1515-1518
        if (contigs[c].n_snps == 0) { for (SrWindowPlan& old : s.windows) s.spare.push_back(std::move(old)); s.windows.clear(); return; }   // :1522-1524
        s.words = (contigs[c].n_snps + 63) / 64;
        if (!s.low_memory_now && ((int)s.pos_key.size() != s.N || std::memcmp(s.pos_key.data(), contigs[c].read_start, (size_t)s.N * 4) != 0)) {
            s.pos_key.assign(contigs[c].read_start, contigs[c].read_start + s.N);
            s.pos_orig.resize((size_t)s.N); s.pos_rank.resize((size_t)s.N);
            for (int k = 0; k < s.N; ++k) s.pos_orig[(size_t)k] = k;
            const int32_t* st0 = contigs[c].read_start;
            std::sort(s.pos_orig.begin(), s.pos_orig.end(), [st0](int32_t a, int32_t b2) { return st0[a] != st0[b2] ? st0[a] < st0[b2] : a < b2; });
            for (int k = 0; k < s.N; ++k) s.pos_rank[(size_t)s.pos_orig[(size_t)k]] = k;
        }
        if (s.perm_n != s.N || s.perm_seed != seed) {      // (a kept state has the order of these N reads already)
            s.perm = shuffled_order(s.N, seed);
            s.rank.resize((size_t)s.N);
            for (int k = 0; k < s.N; ++k) s.rank[(size_t)s.perm[(size_t)k]] = k;
            s.perm_n = s.N; s.perm_seed = seed;
        }
    });

    laps.lap("perm");
    const bool lm_on_device = dev.low_memory_graphs();
    bool matrices_by_position = false;
    // ---- the SNP columns of the batch, concatenated once: K5a/K5 (sim / diff) now, Chinese-Whispers seeding later ----
    CwChain ch;
    std::vector<int64_t> col_base_of_contig((size_t)C, 0);
    {
        SimdiffJob job;
        job.cols = &ch;
        job.contig_snp_base.assign((size_t)C, 0); job.plane_off.assign((size_t)C, 0); job.out_off.assign((size_t)C, 0);
        job.n_reads.assign((size_t)C, 0); job.words.assign((size_t)C, 0); job.plane_n.assign((size_t)C, 0); job.read_base.assign((size_t)C, 0);
        int64_t n_ent = 0, n_col = 0;
        for (int c = 0; c < C; ++c) { n_col += contigs[c].n_snps; if (contigs[c].n_snps) n_ent += contigs[c].col_off[contigs[c].n_snps] - contigs[c].col_off[0]; }
        // The columns of the call as ONE CSR in contig order. When the caller's arrays already are that (stage 3 hands its result
        // over in memory: one offset array starting at 0, entries back to back) they are used where they lie.
        bool contiguous = n_col > 0;
        {
            const int64_t* off_next = nullptr; const int32_t* idx0 = nullptr; const uint8_t* code0 = nullptr;
            bool first = true;
            for (int c = 0; c < C && contiguous; ++c) {
                const hs_sr_contig& hc = contigs[c];
                if (hc.n_snps == 0) continue;
                if (first) { first = false; idx0 = hc.col_idx; code0 = hc.col_code; if (hc.col_off[0] != 0) contiguous = false; }
                else if (hc.col_idx != idx0 || hc.col_code != code0 || hc.col_off != off_next) contiguous = false;
                off_next = hc.col_off + hc.n_snps;
            }
        }
        if (resident && n_col > 0 && !contiguous) { set_error("sr_run: resident columns need one offset array in contig order"); return HS_EINVAL; }
        int64_t* own_off = nullptr; int32_t* own_idx = nullptr; uint8_t* own_code = nullptr;
        if (contiguous) {
            const hs_sr_contig* first = nullptr;
            for (int c = 0; c < C && !first; ++c) if (contigs[c].n_snps) first = &contigs[c];
            ch.col_off.view(first->col_off, (size_t)n_col + 1);
            if (!resident) { ch.col_idx.view(first->col_idx, (size_t)n_ent); ch.col_code.view(first->col_code, (size_t)n_ent); }
        } else {
            own_off = ch.col_off.alloc((size_t)n_col + 1); own_idx = ch.col_idx.alloc((size_t)n_ent); own_code = ch.col_code.alloc((size_t)n_ent);
            own_off[0] = 0;
        }
        job.snp_ref.resize((size_t)n_col); job.snp_alt.resize((size_t)n_col); job.snp_contig.resize((size_t)n_col);
        std::vector<int64_t> ent_base_of_contig((size_t)C, 0);
        {
            int64_t cb = 0, eb = 0, rows_total = 0;
            for (int c = 0; c < C; ++c) {
                col_base_of_contig[(size_t)c] = cb; ent_base_of_contig[(size_t)c] = eb;
                job.contig_snp_base[(size_t)c] = cb;
                const hs_sr_contig& hc = contigs[c];
                if (hc.n_snps == 0) continue;
                cb += hc.n_snps; eb += hc.col_off[hc.n_snps] - hc.col_off[0];
                if (st[(size_t)c].low_memory_now && !lm_on_device) continue;
                // bit rows for the matrix path and for the low-memory path (whose windows compare their own reads only: no N x N)
                job.plane_off[(size_t)c] = job.plane_total; job.words[(size_t)c] = st[(size_t)c].words; job.plane_n[(size_t)c] = st[(size_t)c].N;
                job.plane_total += (int64_t)st[(size_t)c].N * st[(size_t)c].words;
                job.read_base[(size_t)c] = rows_total; rows_total += st[(size_t)c].N;
                if (st[(size_t)c].low_memory_now) continue;
                job.out_off[(size_t)c] = job.out_total;
                job.n_reads[(size_t)c] = st[(size_t)c].N;
                job.out_total += (int64_t)st[(size_t)c].N * st[(size_t)c].N;
            }
        }
        {   // the matrices' rows in the order of the reads' start positions
            const bool read_order = false;
            int64_t rows_all = 0;
            for (int c = 0; c < C; ++c) rows_all = std::max<int64_t>(rows_all, job.read_base[(size_t)c] + job.plane_n[(size_t)c]);
            if (!read_order) job.pos_orig.assign((size_t)rows_all, 0);
            if (!read_order) for (int c = 0; c < C; ++c) if (job.plane_n[(size_t)c] > 0 && job.n_reads[(size_t)c] == 0)      // (bit rows only: the low-memory path keeps read order)
                for (int k = 0; k < job.plane_n[(size_t)c]; ++k) job.pos_orig[(size_t)(job.read_base[(size_t)c] + k)] = k;
        }
        parallel_for(C, n_threads, [&](int c) {
            const hs_sr_contig& hc = contigs[c];
            if (hc.n_snps == 0) return;
            const int64_t cb = col_base_of_contig[(size_t)c], e_base = ent_base_of_contig[(size_t)c], o0 = hc.col_off[0];   // col_off need not start at 0
            if (!contiguous) {
                for (int s = 0; s < hc.n_snps; ++s) own_off[(size_t)(cb + s) + 1] = e_base + hc.col_off[s + 1] - o0;
                const int64_t n = hc.col_off[hc.n_snps] - o0;
                std::memcpy(own_idx + e_base, hc.col_idx + o0, (size_t)n * sizeof(int32_t));
                std::memcpy(own_code + e_base, hc.col_code + o0, (size_t)n);
            }
            if (!job.pos_orig.empty() && job.n_reads[(size_t)c] > 0) {
                if ((int)st[(size_t)c].pos_orig.size() == st[(size_t)c].N) std::memcpy(job.pos_orig.data() + job.read_base[(size_t)c], st[(size_t)c].pos_orig.data(), (size_t)st[(size_t)c].N * 4);
                else for (int k = 0; k < st[(size_t)c].N; ++k) job.pos_orig[(size_t)(job.read_base[(size_t)c] + k)] = k;
            }
            std::memcpy(job.snp_ref.data() + cb, hc.snp_ref, (size_t)hc.n_snps);
            std::memcpy(job.snp_alt.data() + cb, hc.snp_alt, (size_t)hc.n_snps);
            std::fill(job.snp_contig.begin() + cb, job.snp_contig.begin() + cb + hc.n_snps, c);
        });
        laps.lap("columns");
        matrices_by_position = !job.pos_orig.empty();
        const double t0 = now_ms();
        if (int rc = dev.simdiff_columns(job, &k_ms[0])) return rc;
        dev_ms += now_ms() - t0;
    }

    const double t_simdiff_done = now_ms();
    laps.lap("simdiff");
    // the columns on the host, for the few things that walk them here (resident case: fetched once, on demand)
    std::vector<int32_t> fetched_idx; std::vector<uint8_t> fetched_code;
    bool columns_here = !resident;
    auto need_columns = [&]() -> int {
        if (columns_here) return HS_OK;
        const double t0 = now_ms();
        if (int rc = dev.fetch_columns(fetched_idx, fetched_code)) return rc;
        dev_ms += now_ms() - t0;
        for (int c = 0; c < C; ++c) { cs_local[(size_t)c].col_idx = fetched_idx.data(); cs_local[(size_t)c].col_code = fetched_code.data(); }
        ch.col_idx.view(fetched_idx.data(), fetched_idx.size()); ch.col_code.view(fetched_code.data(), fetched_code.size());
        columns_here = true;
        return HS_OK;
    };
    // ---- window plans (host) ----
    parallel_for(C, n_threads, [&](int c) {
        if (contigs[c].n_snps == 0) return;
        sr_plan_windows(st[(size_t)c], window_size, error_rate, lowmem, !resident);
    });
    laps.lap("plan_windows");
    if (resident) {   // the reads of every window from the device: those present at its first and its last SNP column
        struct WR { int c, w; };
        std::vector<WR> wr;
        for (int c = 0; c < C; ++c)
            for (size_t w = 0; w < st[(size_t)c].windows.size(); ++w) if (st[(size_t)c].windows[w].has_snps) wr.push_back(WR{c, (int)w});
        const size_t W = wr.size();
        std::vector<int64_t> col_a(W), col_b(W), slot_off(W + 1, 0);
        for (size_t i = 0; i < W; ++i) {
            const SrWindowPlan& wp = st[(size_t)wr[i].c].windows[(size_t)wr[i].w];
            col_a[i] = col_base_of_contig[(size_t)wr[i].c] + wp.col_a; col_b[i] = col_base_of_contig[(size_t)wr[i].c] + wp.col_b;
            slot_off[i + 1] = slot_off[i] + (ch.col_off[(size_t)col_a[i] + 1] - ch.col_off[(size_t)col_a[i]]);
        }
        std::vector<int32_t> ids, win_m;
        if (W) {
            const double t0 = now_ms();
            if (int rc = dev.window_masks(col_a, col_b, slot_off, ids, win_m)) return rc;
            dev_ms += now_ms() - t0;
        }
        parallel_for((int)W, n_threads, [&](int i) {
            SrWindowPlan& wp = st[(size_t)wr[(size_t)i].c].windows[(size_t)wr[(size_t)i].w];
            wp.ids.assign(ids.begin() + slot_off[(size_t)i], ids.begin() + slot_off[(size_t)i] + win_m[(size_t)i]);
        });
        laps.lap("window_masks");
    }
    // ---- the window set of the call: matrix-path windows first (their graphs come from K6), then the low-memory ones ----
    struct WRef { int c, w; };
    std::vector<WRef> wrefs;            // index = window index in the set
    SrWindowSet ws;
    ws.error_rate = error_rate;
    {
        // Low-memory contigs on the device: the reference indexes a read's 0/1/2 vector by the ORDER of its appearances (:545-575),
        // the bit rows by SNP; the two agree when every read is present at every SNP between its first and its last one (always so
        // for columns that stage 3 wrote: a column lists every record that covers the position). Checked here; a contig where it
        // does not hold keeps the host builder.
        std::vector<uint8_t> lm_dev((size_t)C, 0);
        bool any_lm = false;
        for (int c = 0; c < C; ++c) if (contigs[c].n_snps > 0 && st[(size_t)c].low_memory_now) any_lm = true;
        if (any_lm && lm_on_device) {
            if (int rc = need_columns()) return rc;
            parallel_for(C, n_threads, [&](int c) {
                const hs_sr_contig& hc = cs_local[(size_t)c];
                if (hc.n_snps == 0 || !st[(size_t)c].low_memory_now) return;
                std::vector<int32_t> first((size_t)st[(size_t)c].N, -1), last((size_t)st[(size_t)c].N, -1), cnt((size_t)st[(size_t)c].N, 0);
                for (int s = 0; s < hc.n_snps; ++s)
                    for (int64_t e = hc.col_off[s]; e < hc.col_off[s + 1]; ++e) {
                        const int r = hc.col_idx[e];
                        if (first[(size_t)r] < 0) first[(size_t)r] = s;
                        last[(size_t)r] = s; cnt[(size_t)r]++;
                    }
                bool ok = true;
                for (int r = 0; r < st[(size_t)c].N && ok; ++r) if (cnt[(size_t)r] > 0 && cnt[(size_t)r] != last[(size_t)r] - first[(size_t)r] + 1) ok = false;
                lm_dev[(size_t)c] = ok ? 1 : 0;
            });
        }
        if (any_lm && std::getenv("HS_TIMING")) {
            int n_lm = 0, n_ok = 0;
            for (int c = 0; c < C; ++c) if (contigs[c].n_snps > 0 && st[(size_t)c].low_memory_now) { n_lm++; n_ok += lm_dev[(size_t)c]; }
            std::fprintf(stderr, "[hs timing] sr: %d contigs on the low-memory path, %d of them with graphs from the device\n", n_lm, n_ok);
        }
        std::vector<WRef> lm_w, host_w;
        for (int c = 0; c < C; ++c)
            for (size_t w = 0; w < st[(size_t)c].windows.size(); ++w) {
                if (!st[(size_t)c].windows[w].has_snps) continue;
                (!st[(size_t)c].low_memory_now ? wrefs : (lm_dev[(size_t)c] ? lm_w : host_w)).push_back(WRef{c, (int)w});
            }
        ws.n_matrix_windows = (int32_t)wrefs.size();
        wrefs.insert(wrefs.end(), lm_w.begin(), lm_w.end());
        ws.n_dev_windows = (int32_t)wrefs.size();
        wrefs.insert(wrefs.end(), host_w.begin(), host_w.end());
        ws.ctg_reads.resize((size_t)C);
        for (int c = 0; c < C; ++c) ws.ctg_reads[(size_t)c] = st[(size_t)c].N;
        const size_t W = wrefs.size();
        ws.win_contig.resize(W); ws.win_row0.assign(W + 1, 0); ws.win_final_empty.resize(W);
        for (size_t i = 0; i < W; ++i) {
            SrWindowPlan& wp = st[(size_t)wrefs[i].c].windows[(size_t)wrefs[i].w];
            wp.row0 = ws.win_row0[i];
            ws.win_contig[i] = wrefs[i].c;
            ws.win_final_empty[i] = wp.final_graph_empty ? 1 : 0;
            ws.win_row0[i + 1] = ws.win_row0[i] + (int64_t)wp.ids.size();
        }
        ws.mask_ids.resize((size_t)ws.win_row0[W]);
        parallel_for((int)W, n_threads, [&](int i) {
            const SrWindowPlan& wp = st[(size_t)wrefs[(size_t)i].c].windows[(size_t)wrefs[(size_t)i].w];
            std::copy(wp.ids.begin(), wp.ids.end(), ws.mask_ids.begin() + ws.win_row0[(size_t)i]);
        });
        ws.ctg_rank_off.assign((size_t)C, 0);
        int64_t ro = 0;
        for (int c = 0; c < C; ++c) { ws.ctg_rank_off[(size_t)c] = ro; ro += (int64_t)st[(size_t)c].rank.size(); }
        ws.rank.resize((size_t)ro);
        parallel_for(C, n_threads, [&](int c) { std::copy(st[(size_t)c].rank.begin(), st[(size_t)c].rank.end(), ws.rank.begin() + ws.ctg_rank_off[(size_t)c]); });
        if (matrices_by_position) {      // (the same layout: rank.size() == N for every contig with SNPs)
            ws.pos_rank.assign((size_t)ro, 0);
            parallel_for(C, n_threads, [&](int c) {
                const SrContigState& s = st[(size_t)c];
                if (s.rank.empty()) return;
                int32_t* o = ws.pos_rank.data() + ws.ctg_rank_off[(size_t)c];
                if (!s.low_memory_now && (int)s.pos_rank.size() == s.N) std::copy(s.pos_rank.begin(), s.pos_rank.end(), o);
                else for (int k = 0; k < (int)s.rank.size(); ++k) o[k] = k;
            });
        }
        // create_read_graph_low_memory (-l, or coverage > 1000): O(m^2 S) per window on the host, one task per window
        const int n_host = (int)host_w.size();
        if (n_host > 0) {
            if (int rc = need_columns()) return rc;
            std::vector<std::vector<std::vector<int32_t>>> lists((size_t)n_host);
            parallel_for(n_host, n_threads, [&](int i) {
                const WRef& r = wrefs[(size_t)ws.n_dev_windows + (size_t)i];
                sr_build_window_graph_low_memory(st[(size_t)r.c], st[(size_t)r.c].windows[(size_t)r.w], error_rate, lists[(size_t)i]);
            });
            ws.host_off.assign(1, 0);
            for (auto& wl : lists)
                for (auto& row : wl) { ws.host_nbr.insert(ws.host_nbr.end(), row.begin(), row.end()); ws.host_off.push_back((int64_t)ws.host_nbr.size()); }
        }
    }
    laps.lap("window_set");
    int64_t rows_on_host = 0, n_finish_host = 0;
    float k6_ms = 0;
    const bool two_phase = !wrefs.empty() && dev.two_phase_graphs();
    if (!wrefs.empty()) {
        const double t0 = now_ms();
        if (two_phase) { if (int rc = dev.build_graphs_begin(ws, &k6_ms)) return rc; }      // (the device works on the rows while the chain is planned below)
        else if (int rc = dev.build_graphs(ws, &rows_on_host, &k6_ms)) return rc;
        dev_ms += now_ms() - t0;
    }
    const double t_plan_done = now_ms();
    laps.lap("build_graphs");

    // ---- the dependent Chinese-Whispers runs, device resident (see CwChain) ----
    int64_t n_cw = 0;
    std::vector<int64_t> chain_index(wrefs.size(), -1);   // position of the window in the chain
    {
        ch.win_seed_begin.assign(1, 0);
        ch.chain_row0.assign(1, 0);
        ch.finish_on_device = !lowmem;
        for (int c = 0; c < C && ch.finish_on_device; ++c)
            if (contigs[c].n_snps > 0 && !st[(size_t)c].snp_pos_sorted) ch.finish_on_device = false;
        if (ch.finish_on_device) {
            ch.col_pos.reserve(ch.col_off.size());
            for (int c = 0; c < C; ++c) ch.col_pos.insert(ch.col_pos.end(), contigs[c].snp_pos, contigs[c].snp_pos + contigs[c].n_snps);
        }
        // the windows of the chain and where their seeds / rows begin (serial, a few thousand additions), then every window's seeds and SNP
        // range filled in by the group's threads (this sits between K6's launch and the wait for its rows: on the group's chain)
        for (size_t i = 0; i < wrefs.size(); ++i) {
            const SrWindowPlan& w = st[(size_t)wrefs[i].c].windows[(size_t)wrefs[i].w];
            if (w.local_snps.empty()) continue;          // finalize_clustering :909-919
            chain_index[i] = (int64_t)ch.win.size();
            ch.win.push_back((int32_t)i);
            ch.win_seed_begin.push_back(ch.win_seed_begin.back() + (int64_t)w.local_snps.size());
            ch.chain_row0.push_back(ch.chain_row0.back() + (int64_t)w.ids.size());
            n_cw += (int64_t)w.local_snps.size() + 2;
        }
        const size_t Wc = ch.win.size();
        ch.seed_col.resize((size_t)ch.win_seed_begin.back());
        if (ch.finish_on_device) { ch.win_snp_first.resize(Wc); ch.win_snp_last.resize(Wc); ch.win_pos_lo.resize(Wc); ch.win_pos_hi.resize(Wc); }
        parallel_for((int)Wc, n_threads, [&](int k) {
            const size_t i = (size_t)ch.win[(size_t)k];
            const SrWindowPlan& w = st[(size_t)wrefs[i].c].windows[(size_t)wrefs[i].w];
            const int64_t base = col_base_of_contig[(size_t)wrefs[i].c];
            int64_t at = ch.win_seed_begin[(size_t)k];
            for (int snp : w.local_snps) ch.seed_col[(size_t)at++] = base + snp;
            if (ch.finish_on_device) {
                const hs_sr_contig& hc = contigs[wrefs[i].c];
                ch.win_snp_first[(size_t)k] = base + (std::lower_bound(hc.snp_pos, hc.snp_pos + hc.n_snps, w.final_lo) - hc.snp_pos);
                ch.win_snp_last[(size_t)k] = base + (std::lower_bound(hc.snp_pos, hc.snp_pos + hc.n_snps, w.final_hi) - hc.snp_pos);
                ch.win_pos_lo[(size_t)k] = w.final_lo; ch.win_pos_hi[(size_t)k] = w.final_hi;
            }
        });
    }
    laps.lap("chain_build");
    if (two_phase) {
        const double t0 = now_ms();
        if (int rc = dev.build_graphs_end(ws, &rows_on_host)) return rc;
        dev_ms += now_ms() - t0;
        laps.lap("graphs_end");
    }
    std::vector<int32_t> chain_labels, final_labels;
    std::vector<uint8_t> final_ok;
    SrChainStats cst;
    {
        const double t0 = now_ms();
        if (taps) dev.tap_chain(&taps->run_off, &taps->run_labels);
        if (!ch.win.empty()) { if (int rc = dev.cw_chain(ch, chain_labels, final_labels, final_ok, &k_ms[1], &cst)) return rc; }
        dev_ms += now_ms() - t0;
    }
    if (taps) {
        dev.tap_chain(nullptr, nullptr);
        taps->win_row0.assign(1, 0); taps->run_begin.assign(1, 0);
        for (size_t k = 0; k < ch.win.size(); ++k) {
            const WRef& wr = wrefs[(size_t)ch.win[k]];
            const SrWindowPlan& w = st[(size_t)wr.c].windows[(size_t)wr.w];
            taps->win_contig.push_back(wr.c); taps->win_start.push_back(w.start);
            taps->mask_ids.insert(taps->mask_ids.end(), w.ids.begin(), w.ids.end());
            taps->win_row0.push_back((int64_t)taps->mask_ids.size());
            for (int snp : w.local_snps) taps->run_snp.push_back(snp);
            taps->run_begin.push_back((int64_t)taps->run_snp.size());
        }
        taps->third = chain_labels;
    }

    const double t_waves_done = now_ms();
    laps.lap("cw_chain");
    // ---- tail of finalize_clustering on the host for the windows the device left: needs the graphs here ----
    std::vector<int64_t> g_off; std::vector<int32_t> g_nbr;
    bool graphs_here = false;
    auto need_graphs = [&]() -> int {
        if (graphs_here) return HS_OK;
        const double t0 = now_ms();
        if (int rc = dev.fetch_graphs(g_off, g_nbr)) return rc;
        dev_ms += now_ms() - t0;
        graphs_here = true;
        return HS_OK;
    };
    {
        bool any_host = false;
        for (size_t i = 0; i < wrefs.size() && !any_host; ++i)
            if (chain_index[i] >= 0 && (final_ok.empty() || !final_ok[(size_t)chain_index[i]])) any_host = true;
        if (any_host) { if (int rc = need_graphs()) return rc; if (int rc = need_columns()) return rc; }
    }
    parallel_for((int)wrefs.size(), n_threads, [&](int i) {
        SrContigState& s = st[(size_t)wrefs[(size_t)i].c];
        SrWindowPlan& w = s.windows[(size_t)wrefs[(size_t)i].w];
        const int64_t k = chain_index[(size_t)i];
        if (k < 0) w.labels.assign(w.ids.size(), -1);                    // no seeding SNP: every read of the window unclustered
        else if (!final_ok.empty() && final_ok[(size_t)k]) {
            const int32_t* f = final_labels.data() + ch.chain_row0[(size_t)k];      // finished on the device (K8)
            w.labels.assign(f, f + w.ids.size());
        } else {
            SrLocalGraph g; g.off = g_off.data() + w.row0; g.nbr = g_nbr.data();
            sr_finish_window(s, w, chain_labels.data() + ch.chain_row0[(size_t)k], g, lowmem);
        }
    });
    for (size_t i = 0; i < wrefs.size(); ++i)
        if (chain_index[i] >= 0 && (final_ok.empty() || !final_ok[(size_t)chain_index[i]])) n_finish_host++;

    const double t_finish_done = now_ms();
    laps.lap("finish");
    // ---- optional ploidy cap (separate_reads.cpp:1711-1715, :1341-1396) ----
    {
        CwWave w4;
        std::vector<size_t> who;
        for (size_t i = 0; i < wrefs.size(); ++i) {
            SrContigState& s = st[(size_t)wrefs[i].c];
            SrWindowPlan& w = s.windows[(size_t)wrefs[i].w];
            if (contigs[wrefs[i].c].ploidy <= 0) continue;
            std::vector<int32_t> init(w.ids.size());
            if (!sr_ploidy_init_labels(w, contigs[wrefs[i].c].ploidy, init.data())) continue;
            who.push_back(i);
            w4.inst_win.push_back((int32_t)i);
            w4.inst_label_off.push_back((int64_t)w4.labels.size());
            w4.labels.insert(w4.labels.end(), init.begin(), init.end());
        }
        if (!w4.inst_win.empty()) {
            const double t0 = now_ms();
            if (int rc = dev.cw(w4, &k_ms[3])) return rc;
            n_cw += (int64_t)w4.inst_win.size();
            dev_ms += now_ms() - t0;
        }
        for (size_t k = 0; k < who.size(); ++k) {
            SrWindowPlan& w = st[(size_t)wrefs[who[k]].c].windows[(size_t)wrefs[who[k]].w];
            w.labels.assign(w4.labels.begin() + w4.inst_label_off[k], w4.labels.begin() + w4.inst_label_off[k] + (int64_t)w.ids.size());
        }
    }

    // ---- result: N labels per window (-2 for the reads the window does not hold) ----
    hs_sr_result* R = (hs_sr_result*)std::calloc(1, sizeof(hs_sr_result));
    R->n_contigs = C;
    std::vector<int64_t> win_off((size_t)C + 1, 0), label_off(1, 0);
    std::vector<int32_t> ws_, we_;
    std::vector<const SrWindowPlan*> wl;
    for (int c = 0; c < C; ++c) {
        for (auto& w : st[(size_t)c].windows) {
            ws_.push_back(w.start); we_.push_back(w.end);
            wl.push_back(&w);
            label_off.push_back(label_off.back() + (int64_t)st[(size_t)c].N);
        }
        win_off[(size_t)c + 1] = (int64_t)ws_.size();
    }
    R->win_off = dup_vec(win_off); R->win_start = dup_vec(ws_); R->win_end = dup_vec(we_); R->label_off = dup_vec(label_off);
    if (sparse) {      // the caller spreads the labels itself (sr_expand_labels)
        sparse->off.assign(wl.size() + 1, 0);
        for (size_t i = 0; i < wl.size(); ++i) sparse->off[i + 1] = sparse->off[i] + (int64_t)wl[i]->ids.size();
        sparse->ids.resize((size_t)sparse->off.back()); sparse->labels.resize((size_t)sparse->off.back());
        parallel_for((int)wl.size(), n_threads, [&](int i) {
            const SrWindowPlan& w = *wl[(size_t)i];
            std::copy(w.ids.begin(), w.ids.end(), sparse->ids.begin() + sparse->off[(size_t)i]);
            std::copy(w.labels.begin(), w.labels.end(), sparse->labels.begin() + sparse->off[(size_t)i]);
        });
        R->labels = nullptr;
    } else {
        R->labels = sr_labels_alloc((size_t)label_off.back());
        parallel_for((int)wl.size(), n_threads, [&](int i) {
            int32_t* o = R->labels + label_off[(size_t)i];
            const int64_t n = label_off[(size_t)i + 1] - label_off[(size_t)i];
            std::fill(o, o + n, -2);
            const SrWindowPlan& w = *wl[(size_t)i];
            for (size_t j = 0; j < w.ids.size(); ++j) o[w.ids[j]] = w.labels[j];
        });
    }
    laps.lap("result");
    R->n_cw_instances = n_cw;
    R->t_kernel_graph_ms = k6_ms; R->n_graph_rows_host = rows_on_host; R->n_windows_finished_on_host = n_finish_host;
    R->n_cw_sweeps = cst.sweeps; R->cw_bytes = cst.bytes; R->graph_nnz = cst.graph_nnz; R->n_graph_rows = ws.rows();
    for (int c = 0; c < C; ++c)
        if (contigs[c].n_snps > 0 && !st[(size_t)c].low_memory_now)
            R->simdiff_bytes += (int64_t)st[(size_t)c].N * contigs[c].n_snps / 4 + 16 * (int64_t)st[(size_t)c].N * st[(size_t)c].N;
    for (int k = 0; k < 4; ++k) R->t_kernel_ms[k] = k_ms[k];
    R->t_device_ms = dev_ms;
    R->t_host_ms = (now_ms() - t_start) - dev_ms;
    if (std::getenv("HS_TIMING"))
        std::fprintf(stderr, "[hs timing] sr: graph rows resolved by std::sort on the host: %ld, k_read_graph_rows %.3f ms; windows finished on the host: %ld of %zu\n",
                     (long)rows_on_host, k6_ms, (long)n_finish_host, ch.win.size());
    if (std::getenv("HS_TIMING"))
        std::fprintf(stderr, "[hs timing] sr: planes+simdiff %.2f ms, plan windows+graphs %.2f ms, clustering chain %.2f ms, finish %.2f ms, total %.2f ms (device %.2f)\n",
                     t_simdiff_done - t_start, t_plan_done - t_simdiff_done, t_waves_done - t_plan_done, t_finish_done - t_waves_done, now_ms() - t_start, dev_ms);
    *out = R;
    return HS_OK;
}

void sr_expand_labels(const SrSparseLabels& sp, const int64_t* label_off, int64_t w0, int64_t w1, int32_t* dense) {
    for (int64_t w = w0; w < w1; ++w) {
        int32_t* o = dense + (label_off[w] - label_off[0]);
        std::fill(o, o + (label_off[w + 1] - label_off[w]), -2);
        for (int64_t e = sp.off[(size_t)w]; e < sp.off[(size_t)w + 1]; ++e) o[sp.ids[(size_t)e]] = sp.labels[(size_t)e];
    }
}

// ---------------------------------------------------------------------------------------------------
// stage 3 -> stage 4 hand-over without the .col text round trip
// ---------------------------------------------------------------------------------------------------
int sr_run_from_cv(SrDeviceOps& dev, const CvMeta& b, int c0, int c1, const hs_cv_result* cv, float error_rate, float rsa, int32_t low_memory,
                   int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size, hs_sr_result** out, SrSparseLabels* sparse, SrWorkspace* keep) {
    if (c0 < 0 || c1 > b.n_contigs || c0 > c1 || cv->n_contigs != c1 - c0) { set_error("sr_run_from_cv: contig range does not match the stage-3 result"); return HS_EINVAL; }
    const int C = c1 - c0;
    const double t_prep0 = now_ms();
    // SNP columns that stayed on the device (stage 3 ran with `resident`): the result carries their offsets only. If SNPs have to
    // be dropped below, the columns are needed here after all: fetched, and stage 4 uploads what is left
    std::vector<int32_t> fetched_idx; std::vector<uint8_t> fetched_code;
    const int32_t* cv_idx = cv->col_idx; const uint8_t* cv_code = cv->col_code;
    // (a device interface that holds the columns is used whether or not the result ALSO carries their entries: a caller that keeps
    // .col's payload on the host still hands stage 3 -> 4 over on the device)
    const bool have_snps = cv->snp_off[C] > 0 && cv->col_off[cv->snp_off[C]] > 0;
    const bool resident_in = have_snps && (cv->col_idx == nullptr || dev.columns_resident());
    if (resident_in && !dev.columns_resident()) { set_error("sr_run_from_cv: the stage-3 result has no columns and the device has none either"); return HS_EINVAL; }
    if (resident_in) {
        bool all = true;
        for (int64_t s = 0; s < cv->snp_off[C] && all; ++s)
            if (!((float)cv->snp_n_alt[s] >= rsa * (float)(cv->snp_n_ref[s] + cv->snp_n_alt[s]))) all = false;
        if (!all) {
            if (int rc = dev.fetch_columns(fetched_idx, fetched_code)) return rc;
            dev.drop_resident_columns();
            cv_idx = fetched_idx.data(); cv_code = fetched_code.data();
        }
    } else if (dev.columns_resident()) dev.drop_resident_columns();
    std::vector<hs_sr_contig> hc((size_t)C);
    std::vector<std::vector<int32_t>> rs((size_t)C), re((size_t)C), spos((size_t)C);
    std::vector<std::vector<uint8_t>> sref((size_t)C), salt((size_t)C);
    std::vector<std::vector<int64_t>> coff((size_t)C);
    if (n_threads <= 0) n_threads = host_threads();
    parallel_for(C, n_threads, [&](int c) {
        const int gc = c0 + c;   // index in the batch
        const int r0 = b.contig_rec_off[(size_t)gc], r1 = b.contig_rec_off[(size_t)gc + 1];
        rs[(size_t)c].resize((size_t)(r1 - r0)); re[(size_t)c].resize((size_t)(r1 - r0));
        for (int r = r0; r < r1; ++r) {
            // READ line limits = (position_2_1, position_2_2) = (POS-1, POS + reference span) (input_output.cpp:503-511)
            rs[(size_t)c][(size_t)(r - r0)] = b.rec_pos[(size_t)r];
            re[(size_t)c][(size_t)(r - r0)] = (int32_t)(b.rec_pos[(size_t)r] + 1 + b.rec_refspan[(size_t)r]);
        }
        // SNPs whose second base is rarer than the threshold are dropped (parse_column_file, separate_reads.cpp:167).
        // Offsets stay global: the columns themselves are not copied.
        const int64_t s0 = cv->snp_off[c], s1 = cv->snp_off[c + 1];
        bool all_kept = true;
        std::vector<char> keep((size_t)(s1 - s0), 1);
        for (int64_t s = s0; s < s1; ++s) {
            int maj = 0, sec = 0;
            if (cv->snp_n_ref && cv->snp_n_alt) { maj = cv->snp_n_ref[s]; sec = cv->snp_n_alt[s]; }   // counted by stage 3 already
            else for (int64_t e = cv->col_off[s]; e < cv->col_off[s + 1]; ++e) { if (cv_code[e] == cv->snp_ref[s]) maj++; else if (cv_code[e] == cv->snp_alt[s]) sec++; }
            if (!((float)sec >= rsa * (float)(maj + sec))) { keep[(size_t)(s - s0)] = 0; all_kept = false; }
        }
        hs_sr_contig& h = hc[(size_t)c];
        h.length = b.contig_off[(size_t)gc + 1] - b.contig_off[(size_t)gc];
        h.n_reads = r1 - r0; h.read_start = rs[(size_t)c].data(); h.read_end = re[(size_t)c].data();
        h.col_idx = cv_idx; h.col_code = cv_code; h.ploidy = b.ploidy.empty() ? 0 : b.ploidy[(size_t)gc];
        if (all_kept) {
            h.n_snps = (int32_t)(s1 - s0); h.snp_pos = cv->snp_pos + s0; h.snp_ref = cv->snp_ref + s0; h.snp_alt = cv->snp_alt + s0;
            h.col_off = cv->col_off + s0;
        } else {
            // a dropped column leaves a hole a single offset array cannot express: private copies (rare path, below)
            coff[(size_t)c].assign(1, 0);
            for (int64_t s = s0; s < s1; ++s) {
                if (!keep[(size_t)(s - s0)]) continue;
                spos[(size_t)c].push_back(cv->snp_pos[s]); sref[(size_t)c].push_back(cv->snp_ref[s]); salt[(size_t)c].push_back(cv->snp_alt[s]);
            }
            h.n_snps = -1;   // marker: filled after the parallel loop (needs private entry storage)
        }
    });
    // rare path: contigs that lost SNPs get private column storage
    std::vector<std::vector<int32_t>> cidx((size_t)C);
    std::vector<std::vector<uint8_t>> ccode((size_t)C);
    for (int c = 0; c < C; ++c) {
        hs_sr_contig& h = hc[(size_t)c];
        if (h.n_snps != -1) continue;
        size_t k = 0;
        for (int64_t s = cv->snp_off[c]; s < cv->snp_off[c + 1]; ++s) {
            if (k < spos[(size_t)c].size() && cv->snp_pos[s] == spos[(size_t)c][k]) {
                cidx[(size_t)c].insert(cidx[(size_t)c].end(), cv_idx + cv->col_off[s], cv_idx + cv->col_off[s + 1]);
                ccode[(size_t)c].insert(ccode[(size_t)c].end(), cv_code + cv->col_off[s], cv_code + cv->col_off[s + 1]);
                coff[(size_t)c].push_back((int64_t)cidx[(size_t)c].size());
                k++;
            }
        }
        h.n_snps = (int32_t)spos[(size_t)c].size(); h.snp_pos = spos[(size_t)c].data(); h.snp_ref = sref[(size_t)c].data(); h.snp_alt = salt[(size_t)c].data();
        h.col_off = coff[(size_t)c].data(); h.col_idx = cidx[(size_t)c].data(); h.col_code = ccode[(size_t)c].data();
    }
    if (std::getenv("HS_TIMING")) std::fprintf(stderr, "[hs timing] sr hand-over from stage 3: %.2f ms\n", now_ms() - t_prep0);
    const int32_t w = window_size > 0 ? window_size : sr_window_size(hc.data(), C, amplicon != 0);
    return sr_run(dev, hc.data(), C, w, error_rate, low_memory, seed, n_threads, out, sparse, keep);
}

// ---------------------------------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------------------------------
}  // namespace hs
