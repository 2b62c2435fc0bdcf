// kmer-db-amd — command line front-end for the MI355X engine.
//
// Drop-in for the three kmer-db modes on the hot path:
//     kmer-db-amd all2all    [-sparse [-min [m:]v] [-max [m:]v]] <db> <out.csv>
//     kmer-db-amd all2all-sp [-min [m:]v] [-max [m:]v]           <db> <out.csv>
//     kmer-db-amd new2all    [-multisample-fasta] [-sparse ...]  <db> <sample-list> <out.csv>
//     kmer-db-amd one2all    <db> <sample> <out.csv>
//     kmer-db-amd all2all-parts [-min ...] [-max ...] <db-list> <out.csv>
// mirroring the reference consoles (reference src/console_all2all.cpp, console_all2all_sparse.cpp,
// console_new2all.cpp) around the calls that the C ABI replaces.  Options that only tune the
// reference's CPU engine (-t, -rt, -buffer, -bubble-size) are accepted; -t also sizes the
// query-parsing thread pool.  Output files are byte-identical to the reference's.
// Everything below the three kmdb_* compute calls runs on the GPU; there is no CPU engine here.
#include "kmdb_amd.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <exception>
#include <map>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <memory>
#include <ctime>
#include <unistd.h>
#include <thread>
#include <vector>

#include <zlib.h>

namespace {

using clk = std::chrono::high_resolution_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

struct usage_error : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- option helpers (same removal semantics as Params::findSwitch/findOption, params.h) -------
bool take_switch(std::vector<std::string>& a, const std::string& name) {
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i] == name) { a.erase(a.begin() + i); return true; }
    return false;
}
bool take_option(std::vector<std::string>& a, const std::string& name, std::string& value) {
    for (size_t i = 0; i + 1 < a.size(); ++i)
        if (a[i] == name) { value = a[i + 1]; a.erase(a.begin() + i, a.begin() + i + 2); return true; }
    return false;
}

// ---- -min / -max filters (params.cpp:14-42, 418-455; sparse_filters.h) ---------------------------
using metric_fn = std::function<double(uint32_t, uint32_t, uint32_t, int)>;

// the nine measures by name: ONE implementation, the library's kmdbh_metric (csrc/host_metrics.cpp: the reference's arithmetic,
// params.cpp:14-42) — the front-end's filters and its `distance` mode must agree with it bit for bit
std::map<std::string, metric_fn> metrics() {
    std::map<std::string, metric_fn> m;
    for (const char* name : {"jaccard", "min", "max", "cosine", "mash", "ani", "ani-shorter", "mash-query", "num-kmers"}) {
        const int id = kmdbh_metric_id(name);
        if (id < 0) throw std::runtime_error(std::string("internal: the library does not know the measure ") + name);
        m[name] = [id](uint32_t c, uint32_t a, uint32_t b, int k) { return kmdbh_metric(id, c, a, b, k); };
    }
    return m;
}

struct Filters {
    struct Bound { metric_fn fn; double lo = std::numeric_limits<double>::lowest(), hi = std::numeric_limits<double>::max(); };
    std::map<std::string, Bound> metric;
    uint32_t kmer_lo = 0, kmer_hi = std::numeric_limits<uint32_t>::max();

    void parse(std::vector<std::string>& args) {
        auto avail = metrics();
        const char* names[2] = {"-min", "-max"};
        for (int which = 0; which < 2; ++which) {
            std::string v;
            while (take_option(args, names[which], v)) {
                std::string metric_name = "num-kmers", num = v;
                auto sep = v.rfind(':');
                if (sep != std::string::npos) { metric_name = v.substr(0, sep); num = v.substr(sep + 1); }
                std::istringstream iss(num);
                double value;
                if (!(iss >> value)) throw std::runtime_error("Filtering error - unable to parse numerical value: " + v);
                if (metric_name == "num-kmers") {
                    (which == 0 ? kmer_lo : kmer_hi) = (uint32_t)std::lrint(value);
                } else if (avail.count(metric_name)) {
                    auto& b = metric[metric_name];
                    b.fn = avail[metric_name];
                    (which == 0 ? b.lo : b.hi) = value;
                } else {
                    throw std::runtime_error("Filtering error - unknown metric: " + metric_name);
                }
            }
        }
    }
    bool pass(uint32_t common, uint32_t row_cnt, uint32_t col_cnt, int k) const {
        for (auto& kv : metric) {
            double x = kv.second.fn(common, row_cnt, col_cnt, k);
            if (!(x >= kv.second.lo && x <= kv.second.hi)) return false;
        }
        return common >= kmer_lo && common <= kmer_hi;
    }
};

// -sample-rows [<criterion>:]<count> (all2all-sp, all2all-parts; reference src/sampler.h, src/params.cpp:533-557, src/array.h:450-540): every
// sample keeps at most `count` of its neighbours — every pair (i, j) that passes the filters is offered to row i AND to row j, so the sampled
// rows are symmetric rows, not rows of the triangle.  With a criterion: the `count` best by score, ties towards the smaller sample id — what
// the reference's heap (sampler.h:44-67: score descending, item ascending) leaves whatever the order of its insertions, so the output is
// byte-identical.  Without one the reference keeps a RANDOM subset (sampler.h:69-79: mt19937_64 with the default seed per sample, drawn in the
// order in which its hash tables happen to list a row): here the same draws, over the pairs in ascending order of the other sample — the same
// kind of subset, not the same bytes.
struct RowSampler {
    struct Item { uint32_t item, value; double score; };
    size_t cap = 0;
    bool best = false;
    metric_fn criterion = nullptr;
    std::vector<std::vector<Item>> rows;
    std::vector<size_t> seen;
    std::vector<std::mt19937_64> rng;
    bool on() const { return cap != 0; }
    void parse(std::vector<std::string>& args) {
        std::string v;
        if (!take_option(args, "-sample-rows", v)) return;
        std::string num = v;
        auto sep = v.rfind(':');
        if (sep != std::string::npos) {
            const std::string name = v.substr(0, sep);
            auto avail = metrics();
            if (!avail.count(name)) throw std::runtime_error("Sampling parameters error - unknown measure: " + name);
            criterion = avail[name]; best = true;
            num = v.substr(sep + 1);
        }
        std::istringstream iss(num);
        if (!(iss >> cap)) throw std::runtime_error("Sampling parameters error - unable to parse numerical value: " + v);
    }
    void init(size_t n) { rows.assign(n, {}); if (!best) { seen.assign(n, 0); rng.assign(n, std::mt19937_64()); } }
    static bool better(const Item& x, const Item& y) { return x.score != y.score ? x.score > y.score : x.item < y.item; }
    void add(size_t sample, uint32_t item, uint32_t value, double score) {
        auto& r = rows[sample];
        r.push_back(Item{item, value, score});
        if (best) {
            // (all candidates of a row are kept until the row is written: the `cap` best are chosen then)
            return;
        }
        ++seen[sample];
        if (r.size() <= cap) return;
        if (rng[sample]() % seen[sample] != 0) { const size_t id = rng[sample]() % cap; r[id] = r.back(); }      // sampler.h:69-79
        r.pop_back();
    }
    // the row as the reference writes it: ascending sample ids (sampler.h:123-139)
    void finish_row(size_t sample, std::vector<uint32_t>& cols, std::vector<uint32_t>& vals) {
        auto& r = rows[sample];
        if (best && r.size() > cap) { std::partial_sort(r.begin(), r.begin() + (std::ptrdiff_t)cap, r.end(), better); r.resize(cap); }
        std::sort(r.begin(), r.end(), [](const Item& x, const Item& y) { return x.item < y.item; });
        cols.clear(); vals.clear();
        for (auto& x : r) { cols.push_back(x.item); vals.push_back(x.value); }
        std::vector<Item>().swap(r);
    }
    double score(uint32_t common, uint32_t row_cnt, uint32_t col_cnt, int k) const { return criterion ? criterion(common, row_cnt, col_cnt, k) : 1.0; }
};

void check(int rc) {
    if (rc) throw std::runtime_error(kmdb_last_error());
}

struct Db {
    kmdbh_db* h = nullptr;
    kmdb_db* d = nullptr;
    kmdb_node* node = nullptr;               // -gpus N: the database prefix-sharded over the devices of the node
    std::thread dropper;                     // gives the host image's pages back while the run goes on (uploaded())
    // the database is on the device: from here on only names and k-mer counts are read from the host image
    void uploaded() { if (h && !dropper.joinable()) dropper = std::thread([hh = h]() { kmdbh_db_release_patterns(hh); }); }
    ~Db() { if (dropper.joinable()) dropper.join(); if (node) kmdb_node_free(node); if (d) kmdb_db_free(d); if (h) kmdbh_db_free(h); }
};

struct Common {
    int threads = 0;
    int device = 0;
    int gpus = 0;                            // -gpus N: N prefix-bucket shards over the node's devices (0: one device, no sharding)
    bool sparse = false;
    Filters filters;
    RowSampler sampler;
};

// -gpus N (all2all, all2all-sp): the database is read WITH its hashtables (the shard weights come from the items' prefix buckets),
// shard s of N goes to device s % D of the D devices the node has (from -gpu on), and the partial matrices meet in one RCCL
// reduce-scatter (kmdb_node_*).  On a node with fewer devices than shards the shards of a device run one after the other.
void node_upload(Db& db, const std::string& path, const Common& c) {
    check(kmdbh_db_load(path.c_str(), c.gpus > 1 ? 0 : 2, &db.h));
    const int have = kmdb_device_count();
    if (have <= c.device) throw std::runtime_error("no usable GPU (device " + std::to_string(c.device) + ")");
    std::vector<int32_t> devs;
    for (int d = c.device; d < have && (int)devs.size() < c.gpus; ++d) devs.push_back(d);
    check(kmdb_node_upload(kmdbh_db_view(db.h), (uint32_t)c.gpus, devs.data(), (uint32_t)devs.size(), &db.node));
    kmdb_node_stats st{};
    check(kmdb_node_stats_get(db.node, &st));
    std::cerr << "Database sharded by k-mer prefix bucket: " << st.n_shards << " shards on " << st.n_devices << " GPU(s)" << std::endl;
}
void node_report(const Db& db) {
    kmdb_node_stats st{};
    if (kmdb_node_stats_get(db.node, &st)) return;
    std::cerr << "  per device: compute " << st.call_ms << " ms, RCCL reduce-scatter " << st.collective_ms << " ms";
    if (st.rccl_version) std::cerr << " (RCCL " << st.rccl_version << ")";
    std::cerr << ", result to host " << st.d2h_ms << " ms" << std::endl;
    // every device by itself (an imbalanced shard shows here, not in the maxima above)
    for (uint32_t slot = 0; slot < st.n_devices; ++slot) {
        kmdb_node_device_stats ds{};
        if (kmdb_node_device_stats_get(db.node, slot, &ds)) break;
        std::cerr << "  GPU " << ds.device << ": " << ds.n_shards << " shard(s), " << ds.n_patterns << " nodes, " << ds.h2d_bytes / 1000000 << " MB over PCIe in " << ds.upload_s
                  << " s; " << ds.n_records << " block records, compute " << ds.call_ms << " ms, reduce-scatter " << ds.collective_ms << " ms, result " << ds.d2h_ms << " ms"
                  << std::endl;
    }
}

// seconds since the kernel started this process (exec, dynamic loading and static initialisers included; 10 ms resolution)
double since_process_start() {
    std::ifstream st("/proc/self/stat");
    std::string line;
    std::getline(st, line);
    const size_t rp = line.rfind(')');                          // the command name may hold spaces
    if (rp == std::string::npos) return -1;
    std::istringstream is(line.substr(rp + 2));
    std::string tok;
    for (int f = 3; f <= 22 && (is >> tok); ++f) {}            // field 22: starttime in clock ticks since boot
    timespec ts{};
    clock_gettime(CLOCK_BOOTTIME, &ts);
    return (double)ts.tv_sec + ts.tv_nsec * 1e-9 - std::strtod(tok.c_str(), nullptr) / (double)sysconf(_SC_CLK_TCK);
}

// End of a one-database run: the table is on disk.  Taking the process down piece by piece (the host image of the database — 9 GB at 100 M
// patterns —, the device pools allocation by allocation, the runtime's own teardown) only adds to the wall clock of the command: the OS and
// the driver reclaim everything at once when the process ends.  KMDB_FULL_TEARDOWN=1 keeps the ordinary exit (profilers that write their
// files from exit handlers need it).
// What the kernel would have to free at the end is given back first, on many threads (the host image: Db::uploaded; the upload's staging
// buffers: kmdb_db_settle): the end of the process frees pages on one thread, at 0.07 s per GB.
int finish(Db& db, std::ofstream& ofs, const std::string& path) {
    ofs.close();
    if (!ofs) throw std::runtime_error("Cannot write the output file " + path);
    if (db.dropper.joinable()) db.dropper.join();              // (its threads free the host image many times faster than the end of the process would)
    if (db.d) kmdb_db_settle(db.d);                            // the same for the upload's staging buffers
    std::cerr << "Process up for " << since_process_start() << " s" << std::endl;
    std::cout.flush();
    std::cerr.flush();
    const char* e = std::getenv("KMDB_FULL_TEARDOWN");
    if (!(e && e[0] == '1')) std::_Exit(0);
    return 0;
}

void write_header(const Db& db, std::ofstream& ofs) {
    uint64_t n = kmdbh_db_n_samples(db.h);
    std::vector<char> buf(10000 + n * 100);
    for (uint64_t i = 0; i < n; ++i) buf.resize(buf.size() + std::strlen(kmdbh_db_sample_name(db.h, i)));
    size_t len = kmdbh_format_header(db.h, buf.data(), buf.size());
    ofs.write(buf.data(), (std::streamsize)len);
}

// ---- all2all (console_all2all.cpp:7-89) -----------------------------------------------------------
int run_all2all(std::vector<std::string>& args, Common& c) {
    std::string v;
    take_option(args, "-buffer", v);
    take_option(args, "-bubble-size", v);
    c.sparse = take_switch(args, "-sparse");
    if (c.sparse) c.filters.parse(args);
    if (args.size() != 2) throw usage_error("all2all");
    std::cerr << "All versus all comparison" << std::endl;
    Db db;
    std::cerr << "Loading k-mer database " << args[0] << "..." << std::endl;
    std::ofstream ofs(args[1]);
    kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = c.device; o.shard_count = 1; o.flags = KMDB_FLAG_ONE_SHOT;
    if (c.gpus > 0) node_upload(db, args[0], c);
    else {
        const auto tl = clk::now();
        std::thread warm([&]() { (void)kmdb_device_prepare(c.device); });      // the device's first use, next to the read of the file (an error shows at the upload)
        const int load_rc = kmdbh_db_load(args[0].c_str(), 2, &db.h);
        const std::string load_err = load_rc ? kmdb_last_error() : "";
        warm.join();
        if (load_rc) throw std::runtime_error(load_err);
        const double load_s = since(tl);
        const auto tu = clk::now();
        check(kmdb_db_upload(kmdbh_db_view(db.h), &o, 0, &db.d));
        std::cerr << "Database loaded in " << load_s << " s, uploaded in " << since(tu) << " s" << std::endl;
    }
    const uint64_t n = kmdbh_db_n_samples(db.h);
    const int k = (int)kmdbh_db_kmer_length(db.h);
    std::cerr << "Calculating matrix of common k-mers..." << std::endl;
    auto t0 = clk::now();
    std::vector<uint32_t> m(n ? n * (n - 1) / 2 + 1 : 1);
    if (db.node) { check(kmdb_node_all2all_dense(db.node, m.data(), nullptr)); node_report(db); }
    else {
        check(kmdb_all2all_dense(db.d, m.data(), &o));
        kmdb_stats st{};
        if (!kmdb_db_stats(db.d, &st) && st.path == KMDB_PATH_GLOBAL)
            std::cerr << "WARNING: the fast pipeline could not take this database (" << kmdb_db_fallback_reason(db.d) << "); the slow HBM-atomics kernel ran" << std::endl;
    }
    std::cerr << "OK (" << since(t0) << " seconds)" << std::endl;
    db.uploaded();                                             // (after the call: the page drop and the call's first allocations get in each other's way)
    std::cerr << "Storing matrix of common k-mers in " << args[1] << "...";
    t0 = clk::now();
    write_header(db, ofs);
    // rows formatted by a pool of threads, a stripe of rows at a time (about 32 M cells: the text of a 10 000-sample table is 300 MB), and
    // written in order (console_all2all.cpp:53-75 formats and writes row by row on one thread)
    const int nthr = (int)std::max<size_t>(1, std::min<size_t>(c.threads > 0 ? (size_t)c.threads : 16, std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1));
    size_t name_max = 0;
    for (uint64_t i = 0; i < n; ++i) name_max = std::max(name_max, std::strlen(kmdbh_db_sample_name(db.h, i)));
    for (uint64_t i0 = 0; i0 < n;) {
        uint64_t i1 = i0, cells_in = 0;
        while (i1 < n && (i1 == i0 || cells_in + i1 <= (32u << 20))) { cells_in += i1; ++i1; }
        std::vector<std::vector<char>> text(nthr);
        std::atomic<int> failed{0};
        auto work = [&](int t) {
            // thread t: the rows [a, b) of the stripe holding its share of the cells
            auto cut = [&](int q) -> uint64_t {
                const double lo = (double)i0 * (double)i0, hi = (double)i1 * (double)i1;
                uint64_t r = (uint64_t)std::sqrt(lo + (hi - lo) * q / nthr);
                return std::min<uint64_t>(i1, std::max<uint64_t>(i0, r));
            };
            const uint64_t a = t == 0 ? i0 : cut(t), b = t + 1 == nthr ? i1 : cut(t + 1);
            std::vector<char>& out = text[t];
            std::vector<uint32_t> cols, vals;
            size_t used = 0;
            for (uint64_t i = a; i < b; ++i) {
                const uint32_t* r = m.data() + i * (i - 1) / 2;
                const char* name = kmdbh_db_sample_name(db.h, i);
                const size_t need = 64 + name_max + i * (c.sparse ? 22 : 11);
                if (out.size() < used + need) out.resize(std::max(out.size() * 2, used + need));
                if (c.sparse) {
                    cols.clear(); vals.clear();
                    for (uint64_t j = 0; j < i; ++j)     // LowerTriangularMatrix::compact + saveRowSparse (array.h:169-181,259-262)
                        if (r[j] && c.filters.pass(r[j], (uint32_t)kmdbh_db_sample_kmers(db.h, i), (uint32_t)kmdbh_db_sample_kmers(db.h, j), k)) {
                            cols.push_back((uint32_t)j); vals.push_back(r[j]);
                        }
                    used += kmdbh_format_sparse_row(name, kmdbh_db_sample_kmers(db.h, i), cols.data(), vals.data(), cols.size(), out.data() + used);
                } else {
                    used += kmdbh_format_dense_row(name, kmdbh_db_sample_kmers(db.h, i), r, i, out.data() + used);
                }
            }
            out.resize(used);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthr; ++t) pool.emplace_back([&, t]() { try { work(t); } catch (...) { failed = 1; } });
        work(0);
        for (auto& th : pool) th.join();
        if (failed) throw std::runtime_error("out of memory while formatting the table");
        for (auto& tx : text) ofs.write(tx.data(), (std::streamsize)tx.size());
        i0 = i1;
    }
    std::cerr << "OK (" << since(t0) << " seconds)" << std::endl;
    return finish(db, ofs, args[1]);
}

// ---- all2all-sp (console_all2all_sparse.cpp:13-111) -------------------------------------------------
int run_all2all_sp(std::vector<std::string>& args, Common& c) {
    std::string v;
    take_option(args, "-buffer", v);
    uint32_t bubble = 8000;
    if (take_option(args, "-bubble-size", v)) bubble = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    take_switch(args, "-sparse");
    c.filters.parse(args);
    c.sampler.parse(args);
    if (args.size() != 2) throw usage_error("all2all-sp");
    std::cerr << "All versus all comparison (sparse computation)" << std::endl;
    Db db;
    std::cerr << "Loading k-mer database " << args[0] << "..." << std::endl;
    std::ofstream ofs(args[1], std::ios::binary);
    kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = c.device; o.shard_count = 1; o.bubble_size = bubble; o.flags = KMDB_FLAG_ONE_SHOT;
    if (c.gpus > 0) node_upload(db, args[0], c);
    else {
        const auto tl = clk::now();
        std::thread warm([&]() { (void)kmdb_device_prepare(c.device); });      // the device's first use, next to the read of the file (an error shows at the upload)
        const int load_rc = kmdbh_db_load(args[0].c_str(), 2, &db.h);
        const std::string load_err = load_rc ? kmdb_last_error() : "";
        warm.join();
        if (load_rc) throw std::runtime_error(load_err);
        const double load_s = since(tl);
        const auto tu = clk::now();
        check(kmdb_db_upload(kmdbh_db_view(db.h), &o, 0, &db.d));
        std::cerr << "Database loaded in " << load_s << " s, uploaded in " << since(tu) << " s" << std::endl;
    }
    const uint64_t n = kmdbh_db_n_samples(db.h);
    const int k = (int)kmdbh_db_kmer_length(db.h);
    std::cerr << "Calculating matrix of common k-mers...";
    auto t0 = clk::now();
    kmdb_sparse_rows sp{};
    {
        // the -min / -max filters go with the call: cells that miss a bound never leave the device
        std::vector<kmdb_cell_filter> fl;
        for (auto& kv : c.filters.metric) fl.push_back(kmdb_cell_filter{kmdbh_metric_id(kv.first.c_str()), 0, kv.second.lo, kv.second.hi});
        if (c.filters.kmer_lo != 0 || c.filters.kmer_hi != std::numeric_limits<uint32_t>::max())
            fl.push_back(kmdb_cell_filter{KMDB_METRIC_NUM_KMERS, 0, (double)c.filters.kmer_lo, (double)c.filters.kmer_hi});
        std::vector<uint32_t> counts(n);
        for (uint64_t i = 0; i < n; ++i) counts[i] = (uint32_t)kmdbh_db_sample_kmers(db.h, i);
        if (fl.size() > 8) fl.clear();                          // (the host-side pass below applies every bound anyway)
        if (db.node) { check(kmdb_node_all2all_sparse(db.node, fl.data(), fl.size(), counts.data(), -1, &sp, nullptr)); node_report(db); }
        else if (fl.empty()) check(kmdb_all2all_sparse(db.d, &sp, &o));
        else check(kmdb_all2all_sparse_filtered(db.d, fl.data(), fl.size(), counts.data(), -1, &sp, &o));
    }
    std::cerr << "OK (" << since(t0) << " seconds)" << std::endl;
    db.uploaded();                                             // (after the call: the page drop and the call's first allocations get in each other's way)
    std::cerr << "Storing matrix of common k-mers in " << args[1] << "...";
    t0 = clk::now();
    write_header(db, ofs);
    std::vector<char> row(10000 + n * 100);
    std::vector<uint32_t> cols, vals;
    size_t saved = 0;
    if (c.sampler.on()) {
        // -sample-rows (console_all2all_sparse.cpp:70-76, array.h:450-540): every pair that passes the filters is offered to both its samples
        c.sampler.init(n);
        for (uint64_t i = 0; i < n; ++i)
            for (uint64_t e = sp.row_ptr[i]; e < sp.row_ptr[i + 1]; ++e) {
                const uint32_t ci = (uint32_t)kmdbh_db_sample_kmers(db.h, i), cj = (uint32_t)kmdbh_db_sample_kmers(db.h, sp.col[e]);
                if (!c.filters.pass(sp.val[e], ci, cj, k)) continue;
                const double sc = c.sampler.score(sp.val[e], ci, cj, k);
                c.sampler.add(i, sp.col[e], sp.val[e], sc);
                c.sampler.add(sp.col[e], (uint32_t)i, sp.val[e], sc);
            }
    }
    for (uint64_t i = 0; i < n; ++i) {
        cols.clear(); vals.clear();
        if (c.sampler.on()) c.sampler.finish_row(i, cols, vals);
        else
        for (uint64_t e = sp.row_ptr[i]; e < sp.row_ptr[i + 1]; ++e)   // compact2's filter (array.h:424-427)
            if (c.filters.pass(sp.val[e], (uint32_t)kmdbh_db_sample_kmers(db.h, i), (uint32_t)kmdbh_db_sample_kmers(db.h, sp.col[e]), k)) {
                cols.push_back(sp.col[e]); vals.push_back(sp.val[e]);
            }
        const char* name = kmdbh_db_sample_name(db.h, i);
        if (row.size() < 10000 + n * 100 + std::strlen(name)) row.resize(10000 + n * 100 + std::strlen(name));
        size_t len = kmdbh_format_sparse_row(name, kmdbh_db_sample_kmers(db.h, i), cols.data(), vals.data(), cols.size(), row.data());
        ofs.write(row.data(), (std::streamsize)len);
        saved += cols.size();
    }
    kmdb_sparse_free(&sp);
    std::cerr << "OK (" << since(t0) << " seconds)" << std::endl;
    std::cerr << "No. saved pairs: " << saved << std::endl;
    return finish(db, ofs, args[1]);
}

// ---- all2all-parts (console_all2all_parts.cpp:21-399): a collection split into several databases --------------
// Grid of cells (row part, column part <= row part): the diagonal cells are all2all_sp of one database, the others
// db2db_sp of two (similarity_calculator.cpp:1225-1540 -> kmdb_db2db_dense).  Row k of row part i is written as the
// concatenation of its cells' sparse rows with the column index shifted by the samples of the earlier parts
// (:292-310), which is the sparse lower-triangular matrix of the whole collection.
int run_all2all_parts(std::vector<std::string>& args, Common& c) {
    std::string v;
    take_option(args, "-buffer", v);
    take_option(args, "-bubble-size", v);
    take_switch(args, "-sparse");
    c.filters.parse(args);
    c.sampler.parse(args);
    if (args.size() != 2) throw usage_error("all2all-parts");
    std::cerr << "All versus all comparison (parts)" << std::endl;
    std::ifstream lst(args[0]);
    if (!lst) throw std::runtime_error("Cannot open file with list of database files " + args[0]);
    std::vector<std::string> files;
    for (std::string ln; std::getline(lst, ln);) {
        while (!ln.empty() && (ln.back() == '\r' || ln.back() == ' ')) ln.pop_back();
        if (!ln.empty()) files.push_back(ln);
    }
    if (files.empty()) throw std::runtime_error("Empty list of database files");
    kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = c.device; o.shard_count = 1;
    // names and k-mer counts of the whole collection (:60-100): every database is read once up front
    std::vector<std::string> names;
    std::vector<uint64_t> counts, part_n;
    uint32_t k = 0;
    double fraction = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        Db d;
        if (kmdbh_db_load(files[i].c_str(), 2, &d.h)) throw std::runtime_error("Cannot open k-mer database " + files[i]);
        if (i == 0) { k = kmdbh_db_kmer_length(d.h); fraction = kmdbh_db_fraction(d.h); }
        else if (k != kmdbh_db_kmer_length(d.h) || fraction != kmdbh_db_fraction(d.h))
            throw std::runtime_error("Databases have different k-mer lengths or fractions");
        const uint64_t n = kmdbh_db_n_samples(d.h);
        part_n.push_back(n);
        for (uint64_t s = 0; s < n; ++s) { names.push_back(kmdbh_db_sample_name(d.h, s)); counts.push_back(kmdbh_db_sample_kmers(d.h, s)); }
    }
    std::ofstream ofs(args[1], std::ios::binary);
    {
        char head[128];
        std::snprintf(head, sizeof head, "kmer-length: %u fraction: %g ,db-samples ,", k, fraction);
        ofs << head;
        for (auto& nm : names) ofs << nm << ",";
        ofs << "\nquery-samples,total-kmers,";
        for (auto cnt : counts) ofs << cnt << ",";
        ofs << "\n";
    }
    // Every part serves as the row database once and as a column database for all later rows: uploaded parts stay resident while HBM
    // lasts (an upload that fails evicts the others and is tried again).
    // -gpus W ("the grid maps to a multi-GPU grid of cells"; reference console_all2all_parts.cpp:143-331 walks the cells on
    // one CPU): the block rows of the grid are dealt to W workers — worker t takes the rows t, t + W, ... on device `-gpu` + t % (devices of
    // the node), its parts resident on ITS device — and the rows' text is written in order as the blocks complete.  Cell (i, j) needs
    // both parts on one device, so a part lives on every device whose rows reach it; nothing crosses between devices.
    std::vector<uint64_t> row_start(files.size() + 1, 0);
    for (size_t i = 0; i < files.size(); ++i) row_start[i + 1] = row_start[i] + part_n[i];
    const int have = std::max(1, kmdb_device_count());
    const size_t W = c.gpus > 0 ? std::min<size_t>((size_t)c.gpus, files.size()) : 1;
    struct Worker {
        kmdb_opts o{};
        std::vector<std::unique_ptr<Db>> resident;
    };
    std::vector<Worker> workers(W);
    for (size_t t = 0; t < W; ++t) {
        workers[t].o = o;
        workers[t].o.device = c.device + (int)(t % (size_t)std::max(1, have - c.device));
        workers[t].resident.resize(files.size());
    }
    if (W > 1) std::cerr << "Block rows of the grid dealt to " << W << " workers on " << std::min<size_t>(W, (size_t)std::max(1, have - c.device)) << " GPU(s)" << std::endl;
    std::vector<std::string> block_text(files.size());
    std::vector<size_t> block_saved(files.size(), 0);
    std::vector<char> block_done(files.size(), 0);
    std::mutex mu;
    std::condition_variable cv;
    std::exception_ptr failure;
    auto block_row = [&](Worker& wk, size_t i) {
        auto get = [&](size_t p, size_t keep) -> Db& {
            if (wk.resident[p]) return *wk.resident[p];
            auto d = std::make_unique<Db>();
            check(kmdbh_db_load(files[p].c_str(), 0, &d->h));
            if (kmdb_db_upload(kmdbh_db_view(d->h), &wk.o, 1, &d->d)) {
                for (size_t j = 0; j < wk.resident.size(); ++j) if (j != keep) wk.resident[j].reset();
                check(kmdb_db_upload(kmdbh_db_view(d->h), &wk.o, 1, &d->d));
            }
            kmdbh_db_free(d->h); d->h = nullptr;                   // the host copy is not needed once the part is in HBM
            wk.resident[p] = std::move(d);
            return *wk.resident[p];
        };
        std::vector<char> row(10000 + names.size() * 100);
        std::vector<uint32_t> cols, vals;
        std::string text;
        size_t saved = 0;
        const uint64_t row_shift = row_start[i];
        Db& drow = get(i, i);
        const uint64_t nr = part_n[i];
        std::vector<std::vector<uint32_t>> cross(i);              // cross[j]: nr x part_n[j]
        for (size_t j = 0; j < i; ++j) {
            { std::lock_guard<std::mutex> g(mu); std::cerr << "Processing cell (" << i + 1 << "," << j + 1 << ")" << std::endl; }
            Db& dcol = get(j, i);
            cross[j].resize(nr * part_n[j] + 1);
            if (kmdb_db2db_dense(drow.d, dcol.d, cross[j].data(), &wk.o)) {
                // out of HBM inside the call (its scratch, lazily made working sets): the other resident parts go, one more try
                for (size_t q = 0; q < wk.resident.size(); ++q) if (q != i && q != j) wk.resident[q].reset();
                check(kmdb_db2db_dense(drow.d, dcol.d, cross[j].data(), &wk.o));
            }
        }
        { std::lock_guard<std::mutex> g(mu); std::cerr << "Processing cell (" << i + 1 << "," << i + 1 << ")" << std::endl; }
        kmdb_sparse_rows sp{};
        if (kmdb_all2all_sparse(drow.d, &sp, &wk.o)) {
            for (size_t q = 0; q < wk.resident.size(); ++q) if (q != i) wk.resident[q].reset();
            check(kmdb_all2all_sparse(drow.d, &sp, &wk.o));
        }
        for (uint64_t r = 0; r < nr; ++r) {
            cols.clear(); vals.clear();
            const uint32_t cr = (uint32_t)counts[row_shift + r];
            uint64_t shift = 0;
            for (size_t j = 0; j < i; ++j) {
                const uint32_t* m = cross[j].data() + r * part_n[j];
                for (uint64_t cidx = 0; cidx < part_n[j]; ++cidx)
                    if (m[cidx] && c.filters.pass(m[cidx], cr, (uint32_t)counts[shift + cidx], (int)k)) {
                        cols.push_back((uint32_t)(shift + cidx)); vals.push_back(m[cidx]);
                    }
                shift += part_n[j];
            }
            for (uint64_t e = sp.row_ptr[r]; e < sp.row_ptr[r + 1]; ++e)
                if (c.filters.pass(sp.val[e], cr, (uint32_t)counts[shift + sp.col[e]], (int)k)) {
                    cols.push_back((uint32_t)(shift + sp.col[e])); vals.push_back(sp.val[e]);
                }
            if (c.sampler.on()) {
                // -sample-rows (console_all2all_parts.cpp:137,191,237,275): the pairs go to the sampler (both samples of a pair), the rows are
                // written when every cell is done — one lock for all workers: the pairs of one sample arrive from several block rows
                std::lock_guard<std::mutex> g(mu);
                for (size_t e = 0; e < cols.size(); ++e) {
                    const double sc = c.sampler.score(vals[e], cr, (uint32_t)counts[cols[e]], (int)k);
                    c.sampler.add(row_shift + r, cols[e], vals[e], sc);
                    c.sampler.add(cols[e], (uint32_t)(row_shift + r), vals[e], sc);
                }
                continue;
            }
            const std::string& name = names[row_shift + r];
            if (row.size() < 10000 + names.size() * 100 + name.size()) row.resize(10000 + names.size() * 100 + name.size());
            size_t len = kmdbh_format_sparse_row(name.c_str(), counts[row_shift + r], cols.data(), vals.data(), cols.size(), row.data());
            text.append(row.data(), len);
            saved += cols.size();
        }
        kmdb_sparse_free(&sp);
        std::lock_guard<std::mutex> g(mu);
        block_text[i] = std::move(text); block_saved[i] = saved; block_done[i] = 1;
        cv.notify_all();
    };
    if (c.sampler.on()) c.sampler.init(names.size());
    std::vector<std::thread> pool;
    for (size_t t = 0; t < W; ++t)
        pool.emplace_back([&, t]() {
            try {
                for (size_t i = t; i < files.size(); i += W) {
                    { std::lock_guard<std::mutex> g(mu); if (failure) return; }
                    block_row(workers[t], i);
                }
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!failure) failure = std::current_exception();
                cv.notify_all();
            }
        });
    size_t saved = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return block_done[i] || failure; });
        if (!block_done[i]) break;
        std::string text = std::move(block_text[i]);
        saved += block_saved[i];
        g.unlock();
        ofs.write(text.data(), (std::streamsize)text.size());
    }
    for (auto& th : pool) th.join();
    workers.clear();                                              // (the parts' handles go before the process ends)
    if (failure) std::rethrow_exception(failure);
    if (c.sampler.on()) {
        // the sampled rows, every cell being in (console_all2all_parts.cpp:333-345).  (Random strategy: the subset of a row depends on the order
        // its pairs arrived in — block rows in the workers' order; with a criterion the rows do not depend on it.)
        std::vector<char> row(10000 + names.size() * 100);
        std::vector<uint32_t> cols, vals;
        for (size_t sidx = 0; sidx < names.size(); ++sidx) {
            c.sampler.finish_row(sidx, cols, vals);
            if (row.size() < 10000 + names.size() * 100 + names[sidx].size()) row.resize(10000 + names.size() * 100 + names[sidx].size());
            const size_t len = kmdbh_format_sparse_row(names[sidx].c_str(), counts[sidx], cols.data(), vals.data(), cols.size(), row.data());
            ofs.write(row.data(), (std::streamsize)len);
            saved += cols.size();
        }
    }
    std::cerr << "No. saved pairs: " << saved << std::endl;
    return 0;
}

// ---- query loading (genome_input_file.h:60-137, 287-337; loader_ex.cpp:22-124,168) ------------------
bool slurp(const std::string& base, std::string& data) {
    static const char* exts[] = {"", ".fa", ".fna", ".fasta", ".gz", ".fa.gz", ".fna.gz", ".fasta.gz"};
    for (const char* e : exts) {
        std::string p = base + e;
        if (FILE* f = std::fopen(p.c_str(), "rb")) {
            std::fclose(f);
            gzFile g = gzopen(p.c_str(), "rb");          // transparent for plain text
            if (!g) return false;
            data.clear();
            char buf[1 << 16];
            int n;
            while ((n = gzread(g, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
            gzclose(g);
            return true;
        }
    }
    return false;
}

struct Record { std::string header, seq; };

void split_fasta(const std::string& data, std::vector<Record>& recs) {
    size_t pos = data.find('>');
    while (pos != std::string::npos) {
        size_t eol = data.find('\n', pos);
        if (eol == std::string::npos) eol = data.size();
        Record r;
        r.header = data.substr(pos + 1, eol - pos - 1);
        if (!r.header.empty() && r.header.back() == '\r') r.header.pop_back();
        size_t sp = r.header.find(' ');
        if (sp != std::string::npos) r.header.resize(sp);        // header up to the first space
        size_t next = data.find('>', eol);
        size_t end = next == std::string::npos ? data.size() : next;
        r.seq.reserve(end > eol ? end - eol : 0);
        for (size_t i = eol + 1; i < end; ++i) {
            char ch = data[i];
            if (ch != '\n' && ch != '\r') r.seq.push_back(ch);
        }
        recs.push_back(std::move(r));
        pos = next;
    }
}

struct Query {
    std::string name;
    std::vector<uint64_t> kmers;             // from_kmers: sorted unique k-mers from the host loader
    std::string text;                        // else: records joined by '\n' (device-side extraction)
    bool from_kmers = false;
};
// the device loader indexes positions with 32 bits: longer genomes go through the host loader
constexpr size_t LONG_QUERY_BASES = 1800u << 20;

std::string basename_of(const std::string& p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}

// ---- new2all (console_new2all.cpp:12-174) ---------------------------------------------------------
int run_new2all(std::vector<std::string>& args, Common& c) {
    if (take_switch(args, "-from-kmers") || take_switch(args, "-from-minhash"))
        throw std::runtime_error("only genome (FASTA) query input is supported by the GPU front-end");
    const bool multi = take_switch(args, "-multisample-fasta");
    const bool host_extract = take_switch(args, "-host-extract");   // k-mer extraction on the host instead of the device
    c.sparse = take_switch(args, "-sparse");
    if (c.sparse) c.filters.parse(args);
    if (args.size() != 3) throw usage_error("new2all");
    std::cerr << "Set of new samples  (from genomes) versus entire database comparison" << std::endl;
    Db db;
    std::cerr << "Loading k-mer database " << args[0] << "..." << std::endl;
    auto t0 = clk::now();
    check(kmdbh_db_load(args[0].c_str(), 0, &db.h));
    kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = c.device; o.shard_count = 1;
    check(kmdb_db_upload(kmdbh_db_view(db.h), &o, 1, &db.d));
    std::cerr << "OK (" << since(t0) << " seconds)" << std::endl;
    const uint64_t n = kmdbh_db_n_samples(db.h);
    const uint32_t k = kmdbh_db_kmer_length(db.h);
    const double fraction = kmdbh_db_fraction(db.h), fstart = kmdbh_db_start_fraction(db.h);
    const int32_t alphabet = kmdbh_db_alphabet(db.h);              // AlphabetType as the file stores it (alphabet.h:10-18): nt, nt-preserve, four protein alphabets
    if (alphabet < 0 || alphabet >= KMDB_ALPHABET_COUNT) throw std::runtime_error("Invalid alphabet type");

    std::ifstream lst(args[1]);
    if (!lst) throw std::runtime_error("Unable to open sample list " + args[1]);
    std::vector<std::string> entries;
    for (std::string ln; std::getline(lst, ln);) {
        while (!ln.empty() && (ln.back() == '\r' || ln.back() == ' ')) ln.pop_back();
        if (!ln.empty()) entries.push_back(ln);
    }
    std::cerr << "Processing queries..." << std::endl;
    auto total0 = clk::now();
    std::ofstream ofs(args[2]);
    write_header(db, ofs);
    std::vector<char> row(10000 + n * 100);
    int nthreads = c.threads > 0 ? c.threads : (int)std::max(1u, std::thread::hardware_concurrency());

    // queries are produced in input order; similarities are computed in batches on the GPU and rows
    // written in the same order (the reference re-orders through a priority queue, :60,114-117)
    std::vector<Query> batch;
    size_t batch_bases = 0;                    // flushed by size as well as by count: the engine's scratch grows with the bases
    auto flush = [&]() {
        if (batch.empty()) return;
        batch_bases = 0;
        std::vector<size_t> cnts(batch.size());
        std::vector<uint32_t> out(batch.size() * n + 1);
        // a query is either sequence text (loader + KmerHelper::unique + one2all on the device, kmdb_new2all_batch_seq) or,
        // with -host-extract and for genomes beyond the device loader's 2^31 positions, a k-mer list (kmdb_new2all_batch)
        std::vector<size_t> by_text, by_kmers;
        for (size_t q = 0; q < batch.size(); ++q) (batch[q].from_kmers ? by_kmers : by_text).push_back(q);
        if (!by_kmers.empty()) {
            std::vector<const uint64_t*> ptrs(by_kmers.size());
            std::vector<size_t> kc(by_kmers.size());
            std::vector<uint32_t> part(by_kmers.size() * n + 1);
            for (size_t t = 0; t < by_kmers.size(); ++t) { ptrs[t] = batch[by_kmers[t]].kmers.data(); kc[t] = batch[by_kmers[t]].kmers.size(); }
            check(kmdb_new2all_batch(db.d, ptrs.data(), kc.data(), by_kmers.size(), part.data(), &o));
            for (size_t t = 0; t < by_kmers.size(); ++t) {
                cnts[by_kmers[t]] = kc[t];
                std::copy(part.begin() + t * n, part.begin() + (t + 1) * n, out.begin() + by_kmers[t] * n);
            }
        }
        if (!by_text.empty()) {
            std::vector<const char*> ptrs(by_text.size());
            std::vector<size_t> lens(by_text.size());
            std::vector<uint64_t> uniq(by_text.size());
            std::vector<uint32_t> part(by_text.size() * n + 1);
            for (size_t t = 0; t < by_text.size(); ++t) { ptrs[t] = batch[by_text[t]].text.data(); lens[t] = batch[by_text[t]].text.size(); }
            check(kmdb_new2all_batch_seq_alphabet(db.d, ptrs.data(), lens.data(), by_text.size(), fraction, fstart, alphabet, part.data(), uniq.data(), &o));
            for (size_t t = 0; t < by_text.size(); ++t) {
                cnts[by_text[t]] = (size_t)uniq[t];
                std::copy(part.begin() + t * n, part.begin() + (t + 1) * n, out.begin() + by_text[t] * n);
            }
        }
        std::vector<uint32_t> cols, vals;
        for (size_t q = 0; q < batch.size(); ++q) {
            const uint32_t* r = out.data() + q * n;
            if (row.size() < 10000 + n * 100 + batch[q].name.size()) row.resize(10000 + n * 100 + batch[q].name.size());
            size_t len;
            if (c.sparse) {
                cols.clear(); vals.clear();
                for (uint64_t j = 0; j < n; ++j)           // one2all_sp keeps count>0 (:1040-1047), then the filter (:131-148)
                    if (r[j] && c.filters.pass(r[j], (uint32_t)cnts[q], (uint32_t)kmdbh_db_sample_kmers(db.h, j), (int)k)) {
                        cols.push_back((uint32_t)j); vals.push_back(r[j]);
                    }
                len = kmdbh_format_sparse_row(batch[q].name.c_str(), cnts[q], cols.data(), vals.data(), cols.size(), row.data());
            } else {
                len = kmdbh_format_dense_row(batch[q].name.c_str(), cnts[q], r, n, row.data());
            }
            ofs.write(row.data(), (std::streamsize)len);
        }
        batch.clear();
    };

    auto make_query = [&](const std::string& name, const std::vector<const std::string*>& seqs) {
        Query q;
        q.name = name;
        size_t bytes = 0;
        for (auto* s : seqs) bytes += s->size() + 1;
        q.from_kmers = host_extract || bytes >= LONG_QUERY_BASES;
        if (!q.from_kmers) {
            q.text.reserve(bytes);
            for (auto* s : seqs) { q.text += *s; q.text += '\n'; }
            return q;
        }
        size_t total = 0;
        for (auto* s : seqs) total += s->size();
        q.kmers.resize(total + 1);
        size_t cnt = 0;
        for (auto* s : seqs) cnt += kmdbh_extract_kmers_alphabet(s->data(), s->size(), k, alphabet, fraction, fstart, q.kmers.data() + cnt);
        cnt = kmdbh_sort_unique(q.kmers.data(), cnt);              // KmerHelper::unique (console_new2all.cpp:73)
        q.kmers.resize(cnt);
        return q;
    };

    const size_t BATCH = 64, BATCH_BASES = 512u << 20;
    auto add = [&](Query&& q) {
        batch_bases += q.text.size() + q.kmers.size();
        batch.push_back(std::move(q));
        if (batch.size() == BATCH || batch_bases >= BATCH_BASES) flush();
    };
    if (multi) {
        for (auto& e : entries) {
            std::string data;
            if (!slurp(e, data)) { std::cerr << "failed:" << e << std::endl; continue; }
            std::vector<Record> recs;
            split_fasta(data, recs);
            for (auto& r : recs) {
                add(make_query(r.header, {&r.seq}));
            }
        }
    } else {
        for (size_t base = 0; base < entries.size(); base += BATCH) {
            size_t cntq = std::min(BATCH, entries.size() - base);
            std::vector<Query> qs(cntq);
            std::vector<char> okv(cntq, 0);
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < cntq;) {
                    std::string data;
                    if (!slurp(entries[base + i], data)) continue;
                    std::vector<Record> recs;
                    split_fasta(data, recs);
                    std::vector<const std::string*> seqs;
                    for (auto& r : recs) seqs.push_back(&r.seq);
                    qs[i] = make_query(basename_of(entries[base + i]), seqs);
                    okv[i] = 1;
                }
            };
            std::vector<std::thread> pool;
            for (int t = 0; t < std::min<int>(nthreads, (int)cntq); ++t) pool.emplace_back(worker);
            for (auto& t : pool) t.join();
            for (size_t i = 0; i < cntq; ++i) {
                if (!okv[i]) { std::cerr << "failed:" << entries[base + i] << std::endl; continue; }
                add(std::move(qs[i]));
            }
        }
    }
    flush();
    std::cerr << std::endl << std::endl << "EXECUTION TIMES" << std::endl << "Total: " << since(total0) << std::endl;
    return 0;
}

// ---- one2all (console_one2all.cpp:12-96): one sample against the database --------------------------
// Same engine call as new2all with a batch of one; the row is labelled with the path as given on the
// command line and the file does not end with a newline (console_one2all.cpp:88-93).
int run_one2all(std::vector<std::string>& args, Common& c) {
    if (take_switch(args, "-from-kmers") || take_switch(args, "-from-minhash"))
        throw std::runtime_error("only genome (FASTA) query input is supported by the GPU front-end");
    if (args.size() != 3) throw usage_error("one2all");
    std::cerr << "One new sample  (from genomes) versus entire database comparison" << std::endl;
    Db db;
    std::cerr << "Loading k-mer database " << args[0] << ":" << std::endl;
    auto t0 = clk::now();
    check(kmdbh_db_load(args[0].c_str(), 0, &db.h));
    kmdb_opts o{}; o.abi_version = KMDB_ABI_VERSION; o.device = c.device; o.shard_count = 1;
    check(kmdb_db_upload(kmdbh_db_view(db.h), &o, 1, &db.d));
    std::cerr << "OK (" << since(t0) << " seconds)" << std::endl;
    const uint64_t n = kmdbh_db_n_samples(db.h);
    const uint32_t k = kmdbh_db_kmer_length(db.h);
    const int32_t alphabet = kmdbh_db_alphabet(db.h);
    if (alphabet < 0 || alphabet >= KMDB_ALPHABET_COUNT) throw std::runtime_error("Invalid alphabet type");
    std::string data;
    if (!slurp(args[1], data)) throw std::runtime_error("Cannot open sample file: " + args[1]);
    std::vector<Record> recs;
    split_fasta(data, recs);
    // loader (kmer_extract.h, filter.h), KmerHelper::sortAndUnique (console_one2all.cpp:64-66) and one2all on the device;
    // genomes beyond the device loader's 32-bit positions take the host loader
    size_t bases = 0;
    for (auto& r : recs) bases += r.seq.size() + 1;
    uint64_t cnt = 0;
    std::vector<uint32_t> sims(n + 1);
    std::cerr << "Calculating similarity vector..." << std::endl;
    if (bases < LONG_QUERY_BASES) {
        std::string text;
        text.reserve(bases);
        for (auto& r : recs) { text += r.seq; text += '\n'; }
        const char* tp = text.data();
        size_t tl = text.size();
        check(kmdb_new2all_batch_seq_alphabet(db.d, &tp, &tl, 1, kmdbh_db_fraction(db.h), kmdbh_db_start_fraction(db.h), alphabet, sims.data(), &cnt, &o));
    } else {
        std::vector<uint64_t> kmers(bases + 1);
        size_t kc = 0;
        for (auto& r : recs)
            kc += kmdbh_extract_kmers_alphabet(r.seq.data(), r.seq.size(), k, alphabet, kmdbh_db_fraction(db.h), kmdbh_db_start_fraction(db.h), kmers.data() + kc);
        kc = kmdbh_sort_unique(kmers.data(), kc);
        const uint64_t* kp = kmers.data();
        check(kmdb_new2all_batch(db.d, &kp, &kc, 1, sims.data(), &o));
        cnt = kc;
    }
    std::cerr << "Number of k-mers: " << cnt << std::endl;
    std::ofstream ofs(args[2]);
    write_header(db, ofs);
    std::vector<char> row(10000 + n * 100 + args[1].size());
    size_t len = kmdbh_format_dense_row(args[1].c_str(), cnt, sims.data(), n, row.data());
    if (len && row[len - 1] == '\n') --len;
    ofs.write(row.data(), (std::streamsize)len);
    std::cerr << "OK" << std::endl;
    return 0;
}

// ---- distance (console_distance.cpp:7-213, params.cpp:603-668) ----------------------------------------
// Text in, text out: the table of common k-mer counts that all2all / all2all-sp / new2all wrote -> one similarity / distance
// measure per cell.  Same row semantics as the reference: a dense triangle (first row named like the first sample, no counts
// in it) keeps i values in row i, any other dense table whole rows; sparse input or -sparse give "column:value" pairs of the
// non-zero counts that pass the filters; a bound without a criterion filters the chosen measure.

// a measure in the reference's fixed notation (conversion.h:167-219, 262-268): exactly 0 is "0", everything else has six
// decimals, rounded half up on the magnitude
size_t put_measure(double v, char* out) {
    if (v == 0) { *out = '0'; return 1; }
    char* p = out;
    if (v < 0) { *p++ = '-'; v = -v; }
    const unsigned long long x = (unsigned long long)(v * 1000000.0 + 0.5);
    p += std::snprintf(p, 40, "%llu.%06llu", x / 1000000ull, x % 1000000ull);
    return (size_t)(p - out);
}
size_t put_uint(unsigned long long v, char* out) { return (size_t)std::snprintf(out, 24, "%llu", v); }

int run_distance(std::vector<std::string>& args) {
    bool sparse_out = take_switch(args, "-sparse");
    const bool phylip = take_switch(args, "-phylip-out");
    if (phylip) sparse_out = false;
    auto avail = metrics();
    struct Bound { metric_fn fn; double lo = std::numeric_limits<double>::lowest(), hi = std::numeric_limits<double>::max(); };
    std::map<std::string, Bound> bounds;
    long kmer_lo = 0, kmer_hi = std::numeric_limits<uint32_t>::max();
    const char* names[2] = {"-min", "-max"};
    for (int which = 0; which < 2; ++which) {
        std::string v;
        while (take_option(args, names[which], v)) {
            std::string crit = "?", num = v;
            const auto sep = v.rfind(':');
            if (sep != std::string::npos) { crit = v.substr(0, sep); num = v.substr(sep + 1); }
            std::istringstream iss(num);
            double value;
            if (!(iss >> value)) throw std::runtime_error("Filtering error - unable to parse numerical value: " + v);
            if (crit == "num-kmers") (which == 0 ? kmer_lo : kmer_hi) = std::lrint(value);
            else if (crit == "?" || avail.count(crit)) (which == 0 ? bounds[crit].lo : bounds[crit].hi) = value;
            else throw std::runtime_error("Filtering error - unknown metric: " + crit);
        }
    }
    if (args.empty()) throw std::runtime_error("No distance/similarity metric specified");
    const std::string measure_name = args.front();
    args.erase(args.begin());
    if (args.size() < 2) throw usage_error("distance");
    if (!avail.count(measure_name)) throw std::runtime_error("unknown measure: " + measure_name);
    if (bounds.count("?")) {                                   // a bare bound belongs to the chosen measure
        const Bound b = bounds["?"];
        bounds.erase("?");
        bounds[measure_name] = b;
    }
    for (auto& kv : bounds) kv.second.fn = avail[kv.first];
    const metric_fn measure = avail[measure_name];

    std::ifstream in(args[0]);
    if (!in) throw std::runtime_error("Cannot open common k-mers table: " + args[0]);
    std::ofstream out(args[1]);
    std::string tok, rest;
    uint32_t k = 0;
    double fraction = 0;
    in >> tok >> k >> tok >> fraction >> tok;                  // "kmer-length: K fraction: F db-samples"
    std::getline(in, rest);                                    // ",name,name,..."
    if (!phylip) out << "kmer-length: " << k << " fraction: " << fraction << rest << std::endl;
    std::vector<std::string> sample_names;
    {
        std::string t = rest;
        std::replace(t.begin(), t.end(), ',', ' ');
        std::istringstream iss(t);
        for (std::string w; iss >> w;) sample_names.push_back(w);
    }
    std::vector<uint32_t> db_counts;
    {
        std::getline(in, rest);                                // "query-samples,total-kmers,c0,c1,..."
        std::replace(rest.begin(), rest.end(), ',', ' ');
        std::istringstream iss(rest);
        iss >> tok >> tok;
        for (unsigned long long c; iss >> c;) db_counts.push_back((uint32_t)c);
    }
    const size_t n = db_counts.size();
    if (phylip) out << n << std::endl;

    auto pass = [&](uint32_t common, uint32_t qcnt, size_t col) {
        for (auto& kv : bounds) {
            const double x = kv.second.fn(common, qcnt, db_counts[col], (int)k);
            if (!(x >= kv.second.lo && x <= kv.second.hi)) return false;
        }
        return (long)common >= kmer_lo && (long)common <= kmer_hi;
    };
    auto read_uint = [](const char*& p) { unsigned long long v = 0; while (*p >= '0' && *p <= '9') v = v * 10 + (unsigned)(*p++ - '0'); return v; };

    std::vector<uint32_t> dense(n, 0);
    std::vector<std::pair<size_t, uint32_t>> hits;
    std::vector<char> obuf;
    bool triangle = false;
    std::string line;
    for (size_t row = 0; std::getline(in, line); ++row) {
        const char* p = line.c_str();
        const char* end = p + line.size();
        const char* comma = std::find(p, end, ',');
        const std::string qname(p, comma);
        p = comma < end ? comma + 1 : end;
        const uint32_t qcnt = (uint32_t)read_uint(p);
        if (p < end) ++p;
        size_t n_read = 0;
        for (; end - p > 1; ++n_read) {
            const unsigned long long v = read_uint(p);
            if (*p == ':') {                                   // sparse input: 1-based column, count
                ++p;
                const uint32_t common = (uint32_t)read_uint(p);
                if (phylip) { if (v >= 1 && v <= n) dense[v - 1] = common; }
                else {
                    sparse_out = true;                         // sparse input always gives sparse output
                    if (common > 0 && v >= 1 && v <= n && pass(common, qcnt, v - 1)) hits.emplace_back(v - 1, common);
                }
            } else if (sparse_out) {
                if (v > 0 && n_read < n && pass((uint32_t)v, qcnt, n_read)) hits.emplace_back(n_read, (uint32_t)v);
            } else if (n_read < n) dense[n_read] = (uint32_t)v;
            if (p < end) ++p;
        }
        const bool empty_first = sparse_out ? hits.empty() : (n == 0 || dense[0] == 0);
        if (row == 0 && !sample_names.empty() && qname == sample_names[0] && empty_first) triangle = true;
        const size_t n_proc = sparse_out ? hits.size() : (triangle ? std::min(row, n) : n);
        obuf.resize(qname.size() + 64 + (phylip ? n_read : n_proc) * 48);
        char* o = obuf.data();
        std::memcpy(o, qname.data(), qname.size());
        o += qname.size();
        if (phylip) {
            *o++ = ' ';
            for (size_t c = 0; c < n_read && c < n; ++c) { o += put_measure(c < n_proc ? measure(dense[c], qcnt, db_counts[c], (int)k) : 0.0, o); *o++ = ' '; }
        } else {
            *o++ = ',';
            if (sparse_out)
                for (auto& h : hits) { o += put_uint(h.first + 1, o); *o++ = ':'; o += put_measure(measure(h.second, qcnt, db_counts[h.first], (int)k), o); *o++ = ','; }
            else
                for (size_t c = 0; c < n_proc; ++c) { o += put_measure(measure(dense[c], qcnt, db_counts[c], (int)k), o); *o++ = ','; }
        }
        out.write(obuf.data(), o - obuf.data());
        out << std::endl;
        if (sparse_out && !phylip) hits.clear(); else std::fill(dense.begin(), dense.end(), 0u);
    }
    return 0;
}

void usage() {
    std::cerr << "kmer-db-amd (MI355X engine for kmer-db's all2all / all2all-sp / new2all)\n"
                 "USAGE\n"
                 "    kmer-db-amd all2all [-sparse [-min [<criterion>:]<v>] [-max [<criterion>:]<v>]] <database> <common_table>\n"
                 "    kmer-db-amd all2all-sp [-min ...] [-max ...] <database> <common_table>\n"
                 "    kmer-db-amd new2all [-multisample-fasta] [-sparse [-min ...] [-max ...]] <database> <sample_list> <common_table>\n"
                 "    kmer-db-amd one2all <database> <sample> <similarity_vector>\n"
                 "    kmer-db-amd all2all-parts [-min ...] [-max ...] <db_list> <common_table>\n"
                 "    kmer-db-amd distance [-sparse] [-phylip-out] [-min [<criterion>:]<v>]* [-max [<criterion>:]<v>]* <measure> <common_table> <output>\n"
                 "Common options: -t <threads>, -gpu <device>\n"
                 "all2all / all2all-sp: -gpus <N>  the k-mer space in N prefix-bucket shards over the node's GPUs (from -gpu on), partial matrices\n"
                 "                                  summed by one RCCL reduce-scatter; more shards than devices: a device runs its shards in turn\n"
                 "all2all-parts: -gpus <W>         the block rows of the grid dealt to W workers over the node's GPUs (parts resident per device)\n";
}

}  // namespace

int main(int argc, char** argv) {
    const double up_at_main = since_process_start();
    if (std::getenv("KMDB_VERBOSE")) std::cerr << "[kmdb] main() entered " << up_at_main << " s after the process started" << std::endl;
    std::vector<std::string> args(argv + 1, argv + argc);
    try {
        if (args.empty()) { usage(); return 0; }
        std::string mode = args[0];
        args.erase(args.begin());
        Common c;
        std::string v;
        if (take_option(args, "-t", v)) c.threads = std::atoi(v.c_str());
        take_option(args, "-rt", v);
        if (take_option(args, "-gpu", v)) c.device = std::atoi(v.c_str());
        if (take_option(args, "-gpus", v)) {
            c.gpus = std::atoi(v.c_str());
            if (c.gpus < 1 || c.gpus > 4096) throw std::runtime_error("-gpus expects a number of prefix-bucket shards (1 or more)");
            if (mode != "all2all" && mode != "all2all-sp" && mode != "all2all-parts") throw std::runtime_error("-gpus applies to all2all, all2all-sp and all2all-parts");
        }
        take_switch(args, "-v");
        take_switch(args, "-vv");
        if (mode == "all2all") return run_all2all(args, c);
        if (mode == "all2all-sp") return run_all2all_sp(args, c);
        if (mode == "new2all") return run_new2all(args, c);
        if (mode == "one2all") return run_one2all(args, c);
        if (mode == "all2all-parts") return run_all2all_parts(args, c);
        if (mode == "distance") return run_distance(args);
        usage();
        return -1;
    } catch (usage_error&) {
        usage();
        return -1;
    } catch (std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;       // main.cpp:56-59
        return -1;
    }
}
