// host_db.cpp — front-end side: parse a kmer-db ".db" file into the flat arrays of
// kmdb_db_view (include/kmdb_amd.h).
//
// Plays the role of PrefixKmerDb::deserialize (reference src/prefix_kmer_db.cpp:578-748) for
// the two modes the hot path uses: Everything (new2all, console_new2all.cpp:32) and
// SkipHashtables (all2all / all2all-sp, console_all2all.cpp:26).  File layout (SURVEY §8a
// a13): header | sample table | raw hashtables (hashmap_lp.h:481-528) | pattern blocks
// (pattern.cpp:15-46).  The file is mapped once; the hashtables and the 64 MB pattern blocks are found header to header
// by one thread and then restored / parsed side by side on up to 16 threads (KMDB_LOAD_THREADS); patterns land in
// struct-of-arrays form and all gamma streams in one contiguous uint64 array.
#include "kmdb_amd.h"
#include "kmdb_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// a flat array that is NOT zero-filled when it is made: every element is written by the loader (the arrays of a 100 M-pattern database
// are 4 GB of pattern fields + 5 GB of streams: zero-filling them first costs as much as reading the file)
template <class T>
struct Buf {
    T* p = nullptr;
    size_t n = 0;
    Buf() = default;
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    ~Buf() { std::free(p); }
    bool alloc(size_t k) { std::free(p); p = (T*)std::malloc(std::max<size_t>(k, 1) * sizeof(T)); n = p ? k : 0; return p != nullptr; }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
};

}  // namespace

struct kmdbh_db {
    uint64_t format_word = 0;
    uint32_t kmer_length = 0;
    double fraction = 0, start_fraction = 0;
    int32_t alphabet = 0;
    uint8_t is_initialized = 0;
    uint64_t kmers_count = 0;
    std::vector<std::string> names;
    std::vector<uint64_t> sample_kmers;
    Buf<int64_t> num_kmers, parent_id;
    Buf<uint32_t> num_samples, num_local, last_id, num_bits;
    Buf<uint64_t> data_offset, data;
    Buf<uint64_t> bucket_offset, slots;
    uint64_t pattern_section_bytes = 0;
    kmdb_db_view view{};
};

namespace {

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    template <class T> T get() {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; }
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const uint8_t* take(size_t n) {
        if ((size_t)(end - p) < n) { ok = false; return nullptr; }
        const uint8_t* r = p;
        p += n;
        return r;
    }
};

constexpr uint64_t EMPTY_SLOT = (uint64_t)0x7fffffffu << 32;   // key 0, val INT32_MAX (hashmap_lp.h:78)

// fn(item) for item in [0, n) on up to `threads` host threads (items are dealt by an atomic counter: they differ in size)
template <class F>
void parallel_items(size_t n, unsigned threads, F&& fn) {
    if (n == 0) return;
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n));
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(i); };
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (unsigned t = 1; t < threads; ++t) {
        try { pool.emplace_back(work); }
        catch (const std::system_error&) { break; }          // no more threads to be had: the ones there are (this one at least) do the items
    }
    work();
    for (auto& th : pool) th.join();
}

unsigned loader_threads() {
    if (const char* e = std::getenv("KMDB_LOAD_THREADS")) { const int v = std::atoi(e); if (v > 0) return (unsigned)v; }
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hc ? hc : 1u));
}

struct HtBucket { const uint8_t* bv; const uint8_t* items; uint64_t filled, allocated; };
struct PatBlock { const uint8_t* p; uint64_t bytes, patterns, words; bool bad; };
constexpr size_t PAT_HEADER = 40;     // num_kmers, parent_id (8 + 8), num_samples, num_local, last id, num_bits (4 x 4), is_parent (4 written, 8 advanced: pattern.cpp:35-37)

}  // namespace

// what a load holds until it has succeeded: freed by whichever way the function is left, an exception included (a damaged count that
// passes the plausibility checks can still make a container throw std::bad_alloc — nothing may cross the extern "C" boundary)
namespace {
struct LoadHold {
    void* map = nullptr;
    size_t len = 0;
    kmdbh_db* db = nullptr;
    ~LoadHold() { if (map) munmap(map, len); delete db; }
};
}  // namespace
static int db_load_impl(const char* path, int mode, kmdbh_db** out);
extern "C" int kmdbh_db_load(const char* path, int mode, kmdbh_db** out) {
    if (!path || !out) return kmdb_set_error("kmdbh_db_load: null argument");
    *out = nullptr;
    try {
        return db_load_impl(path, mode, out);
    } catch (const std::bad_alloc&) {
        *out = nullptr;
        return kmdb_set_error(std::string("Cannot open k-mer database ") + path + " (out of memory)");
    } catch (const std::exception& e) {
        *out = nullptr;
        return kmdb_set_error(std::string("Cannot open k-mer database ") + path + " (" + e.what() + ")");
    }
}
static int db_load_impl(const char* path, int mode, kmdbh_db** out) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return kmdb_set_error(std::string("Cannot open k-mer database ") + path);
    struct stat st{};
    if (fstat(fd, &st) != 0 || st.st_size < 49) { ::close(fd); return kmdb_set_error(std::string("Cannot open k-mer database ") + path); }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return kmdb_set_error(std::string("Cannot map k-mer database ") + path);

    LoadHold hold;
    hold.map = map; hold.len = (size_t)st.st_size;
    auto* db = new kmdbh_db();
    hold.db = db;
    Cursor c{(const uint8_t*)map, (const uint8_t*)map + st.st_size};
    auto fail = [&](const char* what) {
        return kmdb_set_error(std::string("Cannot open k-mer database ") + path + " (" + what + ")");
    };
    const unsigned T = loader_threads();
    const bool verbose = std::getenv("KMDB_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kmdb] load: %-44s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

    db->format_word = c.get<uint64_t>();
    db->kmer_length = c.get<uint32_t>();
    db->fraction = c.get<double>();
    db->start_fraction = c.get<double>();
    db->alphabet = c.get<int32_t>();
    db->is_initialized = c.get<uint8_t>();
    db->kmers_count = c.get<uint64_t>();
    uint64_t n = c.get<uint64_t>();
    if (!c.ok || n > (uint64_t)st.st_size) return fail("bad header");
    db->names.resize(n);
    db->sample_kmers.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        db->sample_kmers[i] = c.get<uint64_t>();
        uint64_t len = c.get<uint64_t>();
        const uint8_t* s = c.ok ? c.take(len) : nullptr;
        if (!s && len) return fail("bad sample table");
        db->names[i].assign((const char*)s, len);
    }
    uint64_t nb = c.get<uint64_t>();
    if (!c.ok || nb > (uint64_t)st.st_size / 64) return fail("bad bucket count");          // every serialised table takes 64 header bytes at least
    if (!(db->format_word & 1ull)) return fail("compressed hashtable serialisation is not supported");
    const bool want_ht = (mode == 0);
    // ---- hashtables, pass 1 (one thread: header to header): where every table's fill vector and items are, and the slot offsets
    std::vector<HtBucket> ht(want_ht ? nb : 0);
    if (want_ht && !db->bucket_offset.alloc(nb + 1)) return fail("out of memory");
    uint64_t total_slots = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        c.take(8);                                   // max_fill_factor
        uint64_t filled = c.get<uint64_t>();
        uint64_t allocated = c.get<uint64_t>();
        c.take(8 * 5);                               // size_when_restruct, mask, ht_memory, ht_total, ht_match
        if (!c.ok) return fail("bad hashtable header");
        // hashmap_lp invariants (reference src/hashmap_lp.h:150,427-437): power-of-two capacity with at least one empty slot —
        // the device probes rely on both
        if (allocated == 0 || (allocated & (allocated - 1)) || filled >= allocated || allocated > (1ull << 40)) return fail("bad hashtable header");
        const uint8_t* bv = c.take(8 * ((allocated + 63) / 64));
        const uint8_t* items = bv ? c.take(8 * filled) : nullptr;
        if (!bv || (!items && filled)) return fail("bad hashtable body");
        if (want_ht) { ht[b] = HtBucket{bv, items, filled, allocated}; db->bucket_offset[b] = total_slots; total_slots += allocated; }
    }
    uint64_t P = c.get<uint64_t>();
    if (!c.ok || P > (uint64_t)st.st_size) return fail("bad pattern count");
    phase("header, sample table, hashtable headers");
    if (want_ht) {
        // ---- pass 2 (the buckets side by side): slot-exact restore (hashmap_lp.h:580-600); every stored pattern id must exist (the
        // device indexes with it)
        db->bucket_offset[nb] = total_slots;
        if (!db->slots.alloc(total_slots)) return fail("out of memory");
        std::atomic<int> err{0};              // 1: fill vector overflow, 2: item count mismatch, 3: pattern id out of range
        parallel_items(nb, T, [&](size_t b) {
            const HtBucket& h = ht[b];
            uint64_t* dst = db->slots.data() + db->bucket_offset[b];
            std::fill(dst, dst + h.allocated, EMPTY_SLOT);
            uint64_t it = 0;
            for (uint64_t w = 0; w < (h.allocated + 63) / 64; ++w) {
                uint64_t word;
                std::memcpy(&word, h.bv + 8 * w, 8);
                while (word) {
                    int bit = __builtin_ctzll(word);
                    word &= word - 1;
                    if (it >= h.filled || w * 64 + (uint64_t)bit >= h.allocated) { err = 1; return; }
                    uint64_t sl;
                    std::memcpy(&sl, h.items + 8 * it, 8);
                    const int32_t val = (int32_t)(sl >> 32);
                    if (val != INT32_MAX && (val < 0 || (uint64_t)val >= P)) { err = 3; return; }
                    dst[w * 64 + bit] = sl;
                    ++it;
                }
            }
            if (it != h.filled) err = 2;
        });
        if (err == 1) return fail("hashtable fill vector overflow");
        if (err == 2) return fail("hashtable fill vector does not match the item count");
        if (err == 3) return fail("hashtable item points past the pattern table");
        phase("hashtables restored");
    }

    // ---- patterns: the file holds them in blocks of at most 64 MB, each behind its byte count (prefix_kmer_db.cpp:544-571).  The blocks are
    // found header to header, then taken side by side: one walk counts a block's patterns and stream words, a prefix sum places the blocks
    // in the arrays, a second walk fills them.
    std::vector<PatBlock> blocks;
    {
        uint64_t bytes_seen = 0;
        // a pattern is at least PAT_HEADER bytes: stop collecting once the blocks seen could already hold P of them and the next header is bad
        while (c.p < c.end) {
            Cursor probe = c;
            uint64_t bs = probe.get<uint64_t>();
            const uint8_t* blk = probe.ok ? probe.take(bs) : nullptr;
            if (!blk && bs) break;                   // not a block: the patterns must be complete by now (checked below)
            if (!probe.ok) break;
            blocks.push_back(PatBlock{blk, bs, 0, 0, false});
            bytes_seen += bs;
            c = probe;
        }
        (void)bytes_seen;
    }
    parallel_items(blocks.size(), T, [&](size_t b) {
        PatBlock& B = blocks[b];
        const uint8_t* q = B.p;
        const uint8_t* e = B.p + B.bytes;
        while (q < e) {
            if ((size_t)(e - q) < PAT_HEADER) { B.bad = true; return; }
            uint32_t bits;
            std::memcpy(&bits, q + 28, 4);
            const size_t words = (size_t)(((uint64_t)bits + 127) / 128) * 2;
            if ((size_t)(e - q) - PAT_HEADER < words * 8) { B.bad = true; return; }
            q += PAT_HEADER + words * 8;
            ++B.patterns; B.words += words;
        }
    });
    phase("pattern blocks counted");
    // the blocks that hold the P patterns (the reference reads block after block until it has them all: prefix_kmer_db.cpp:700-748)
    std::vector<uint64_t> pid0(blocks.size() + 1, 0), word0(blocks.size() + 1, 0);
    size_t used_blocks = 0;
    for (; used_blocks < blocks.size() && pid0[used_blocks] < P; ++used_blocks) {
        const PatBlock& B = blocks[used_blocks];
        if (B.bad) return fail("bad pattern");
        pid0[used_blocks + 1] = pid0[used_blocks] + B.patterns;
        word0[used_blocks + 1] = word0[used_blocks] + B.words;
        if (pid0[used_blocks + 1] > P) return fail("too many patterns");
    }
    if (pid0[used_blocks] < P) return fail("bad pattern block");
    const uint64_t n_words = word0[used_blocks];
    if (!db->num_kmers.alloc(P) || !db->parent_id.alloc(P) || !db->num_samples.alloc(P) || !db->num_local.alloc(P) || !db->last_id.alloc(P) ||
        !db->num_bits.alloc(P) || !db->data_offset.alloc(P) || !db->data.alloc(n_words + 2))
        return fail("out of memory");
    // the arrays are fresh memory: 2.3 M first-touch faults at 100 M patterns.  Every thread asks for the pages of its block's stretches in
    // bulk first (MADV_POPULATE_WRITE, Linux 5.14: one call instead of a trap per page; ignored where the kernel does not know it)
    const char* pe = std::getenv("KMDB_LOAD_POPULATE");
    const bool populate = !(pe && pe[0] == '0');
    auto prefault = [&](void* base, size_t lo_bytes, size_t hi_bytes) {
#ifdef MADV_POPULATE_WRITE
        const uintptr_t a = ((uintptr_t)base + lo_bytes + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)base + hi_bytes) & ~(uintptr_t)4095;
        if (populate && e > a) (void)madvise((void*)a, e - a, MADV_POPULATE_WRITE);
#else
        (void)base; (void)lo_bytes; (void)hi_bytes;
#endif
    };
    parallel_items(used_blocks, T, [&](size_t b) {
        const PatBlock& B = blocks[b];
        const uint8_t* q = B.p;
        uint64_t pid = pid0[b], o = word0[b];
        uint64_t* data = db->data.data();
        {
            const uint64_t p0 = pid0[b], p1 = pid0[b + 1];
            prefault(db->num_kmers.data(), p0 * 8, p1 * 8); prefault(db->parent_id.data(), p0 * 8, p1 * 8); prefault(db->data_offset.data(), p0 * 8, p1 * 8);
            prefault(db->num_samples.data(), p0 * 4, p1 * 4); prefault(db->num_local.data(), p0 * 4, p1 * 4);
            prefault(db->last_id.data(), p0 * 4, p1 * 4); prefault(db->num_bits.data(), p0 * 4, p1 * 4);
            prefault(data, word0[b] * 8, word0[b + 1] * 8);
        }
        for (uint64_t k = 0; k < B.patterns; ++k, ++pid) {
            uint32_t f[4];
            std::memcpy(&db->num_kmers[pid], q, 8);
            std::memcpy(&db->parent_id[pid], q + 8, 8);
            std::memcpy(f, q + 16, 16);
            db->num_samples[pid] = f[0]; db->num_local[pid] = f[1]; db->last_id[pid] = f[2]; db->num_bits[pid] = f[3];
            const size_t words = (size_t)(((uint64_t)f[3] + 127) / 128) * 2;
            db->data_offset[pid] = o;
            if (words) std::memcpy(data + o, q + PAT_HEADER, words * 8);
            o += words;
            q += PAT_HEADER + words * 8;
        }
    });
    phase("pattern blocks parsed");
    db->pattern_section_bytes = PAT_HEADER * P + 8 * n_words;
    munmap(map, (size_t)st.st_size);
    hold.map = nullptr;
    phase("file unmapped");
    db->data[n_words] = 0;                            // one padding pair for 2-word decode windows
    db->data[n_words + 1] = 0;

    kmdb_db_view& v = db->view;
    v.abi_version = KMDB_ABI_VERSION;
    v.kmer_length = db->kmer_length;
    v.n_samples = n;
    v.n_patterns = P;
    v.num_kmers = db->num_kmers.data();
    v.parent_id = db->parent_id.data();
    v.num_samples = db->num_samples.data();
    v.num_local = db->num_local.data();
    v.last_sample_id = db->last_id.data();
    v.num_bits = db->num_bits.data();
    v.data_offset = db->data_offset.data();
    v.data = db->data.data();
    v.n_data_words = db->data.size();
    v.n_buckets = want_ht ? nb : 0;
    v.bucket_offset = want_ht ? db->bucket_offset.data() : nullptr;
    v.slots = want_ht ? db->slots.data() : nullptr;
    hold.db = nullptr;
    *out = db;
    return 0;
}

void kmdb_drop_pages(const std::vector<std::pair<void*, size_t>>& regions, unsigned threads) {
    const size_t page = 4096, step = (size_t)32 << 20;
    std::vector<std::pair<char*, size_t>> chunks;
    for (const auto& r : regions) {
        if (!r.first || r.second < 2 * page) continue;
        char* a = (char*)(((uintptr_t)r.first + page - 1) & ~(uintptr_t)(page - 1));       // (a malloc'ed block keeps its header page)
        char* e = (char*)(((uintptr_t)r.first + r.second) & ~(uintptr_t)(page - 1));
        for (; a < e; a += step) chunks.emplace_back(a, std::min<size_t>(step, (size_t)(e - a)));
    }
    parallel_items(chunks.size(), threads, [&](size_t i) { (void)madvise(chunks[i].first, chunks[i].second, MADV_DONTNEED); });
}

extern "C" void kmdbh_db_release_patterns(kmdbh_db* db) {
    if (!db) return;
    std::vector<std::pair<void*, size_t>> regions;
    auto add = [&](auto& b) { regions.emplace_back((void*)b.data(), b.size() * sizeof(*b.data())); };
    add(db->num_kmers); add(db->parent_id); add(db->num_samples); add(db->num_local); add(db->last_id); add(db->num_bits);
    add(db->data_offset); add(db->data); add(db->slots);
    kmdb_drop_pages(regions, loader_threads());
}

extern "C" void kmdbh_db_free(kmdbh_db* db) { delete db; }
extern "C" const kmdb_db_view* kmdbh_db_view(const kmdbh_db* db) { return &db->view; }
extern "C" uint32_t kmdbh_db_kmer_length(const kmdbh_db* db) { return db->kmer_length; }
extern "C" double kmdbh_db_fraction(const kmdbh_db* db) { return db->fraction; }
extern "C" double kmdbh_db_start_fraction(const kmdbh_db* db) { return db->start_fraction; }
extern "C" int32_t kmdbh_db_alphabet(const kmdbh_db* db) { return db->alphabet; }
extern "C" uint64_t kmdbh_db_n_samples(const kmdbh_db* db) { return db->names.size(); }
extern "C" const char* kmdbh_db_sample_name(const kmdbh_db* db, uint64_t i) { return db->names[i].c_str(); }
extern "C" uint64_t kmdbh_db_sample_kmers(const kmdbh_db* db, uint64_t i) { return db->sample_kmers[i]; }
extern "C" uint64_t kmdbh_db_pattern_section_bytes(const kmdbh_db* db) { return db->pattern_section_bytes; }
