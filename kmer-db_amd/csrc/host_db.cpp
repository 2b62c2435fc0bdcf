// host_db.cpp — front-end side: parse a kmer-db ".db" file into the flat arrays of
// kmdb_db_view (include/kmdb_amd.h).
//
// Plays the role of PrefixKmerDb::deserialize (reference src/prefix_kmer_db.cpp:578-748) for
// the two modes the hot path uses: Everything (new2all, console_new2all.cpp:32) and
// SkipHashtables (all2all / all2all-sp, console_all2all.cpp:26).  File layout (SURVEY §8a
// a13): header | sample table | raw hashtables (hashmap_lp.h:481-528) | pattern blocks
// (pattern.cpp:15-46).  The file is mapped once and walked with a cursor; patterns land in
// struct-of-arrays form and all gamma streams in one contiguous uint64 array.
#include "kmdb_amd.h"
#include "kmdb_internal.h"

#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct kmdbh_db {
    uint64_t format_word = 0;
    uint32_t kmer_length = 0;
    double fraction = 0, start_fraction = 0;
    int32_t alphabet = 0;
    uint8_t is_initialized = 0;
    uint64_t kmers_count = 0;
    std::vector<std::string> names;
    std::vector<uint64_t> sample_kmers;
    std::vector<int64_t> num_kmers, parent_id;
    std::vector<uint32_t> num_samples, num_local, last_id, num_bits;
    std::vector<uint64_t> data_offset, data;
    std::vector<uint64_t> bucket_offset, slots;
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

}  // namespace

extern "C" int kmdbh_db_load(const char* path, int mode, kmdbh_db** out) {
    *out = nullptr;
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return kmdb_set_error(std::string("Cannot open k-mer database ") + path);
    struct stat st{};
    if (fstat(fd, &st) != 0 || st.st_size < 49) { ::close(fd); return kmdb_set_error(std::string("Cannot open k-mer database ") + path); }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return kmdb_set_error(std::string("Cannot map k-mer database ") + path);
    madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);

    auto* db = new kmdbh_db();
    Cursor c{(const uint8_t*)map, (const uint8_t*)map + st.st_size};
    auto fail = [&](const char* what) {
        munmap(map, (size_t)st.st_size);
        delete db;
        return kmdb_set_error(std::string("Cannot open k-mer database ") + path + " (" + what + ")");
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
    if (!c.ok) return fail("bad bucket count");
    if (!(db->format_word & 1ull)) return fail("compressed hashtable serialisation is not supported");
    const bool want_ht = (mode == 0);
    if (want_ht) db->bucket_offset.assign(nb + 1, 0);
    // pass 1 over hashtables: sizes (and skip when not wanted)
    const uint8_t* ht_begin = c.p;
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
        if (!c.take(8 * ((allocated + 63) / 64)) || !c.take(8 * filled)) return fail("bad hashtable body");
        if (want_ht) { db->bucket_offset[b] = total_slots; total_slots += allocated; }
    }
    if (want_ht) {
        db->bucket_offset[nb] = total_slots;
        db->slots.assign(total_slots, EMPTY_SLOT);
        Cursor h{ht_begin, c.p};
        for (uint64_t b = 0; b < nb; ++b) {
            h.take(8);
            uint64_t filled = h.get<uint64_t>();
            uint64_t allocated = h.get<uint64_t>();
            h.take(8 * 5);
            const uint8_t* bv = h.take(8 * ((allocated + 63) / 64));
            const uint8_t* items = h.take(8 * filled);
            uint64_t* dst = db->slots.data() + db->bucket_offset[b];
            uint64_t it = 0;
            for (uint64_t w = 0; w < (allocated + 63) / 64; ++w) {
                uint64_t word;
                std::memcpy(&word, bv + 8 * w, 8);
                while (word) {                        // slot-exact restore (hashmap_lp.h:580-600)
                    int bit = __builtin_ctzll(word);
                    word &= word - 1;
                    if (it >= filled || w * 64 + (uint64_t)bit >= allocated) return fail("hashtable fill vector overflow");
                    std::memcpy(&dst[w * 64 + bit], items + 8 * it, 8);
                    ++it;
                }
            }
            if (it != filled) return fail("hashtable fill vector does not match the item count");
        }
    }

    uint64_t P = c.get<uint64_t>();
    if (!c.ok || P > (uint64_t)st.st_size) return fail("bad pattern count");
    for (uint64_t sl : db->slots) {                     // every stored pattern id must exist (the device indexes with it)
        const int32_t val = (int32_t)(sl >> 32);
        if (val != INT32_MAX && (val < 0 || (uint64_t)val >= P)) return fail("hashtable item points past the pattern table");
    }
    db->num_kmers.resize(P); db->parent_id.resize(P);
    db->num_samples.resize(P); db->num_local.resize(P);
    db->last_id.resize(P); db->num_bits.resize(P); db->data_offset.resize(P);
    db->data.reserve((size_t)((c.end - c.p) / 8));
    uint64_t pid = 0;
    while (pid < P) {
        uint64_t bs = c.get<uint64_t>();
        const uint8_t* blk = c.ok ? c.take(bs) : nullptr;
        if (!blk && bs) return fail("bad pattern block");
        Cursor q{blk, blk + bs};
        while (q.p < q.end) {
            if (pid >= P) return fail("too many patterns");
            db->num_kmers[pid] = q.get<int64_t>();
            db->parent_id[pid] = q.get<int64_t>();
            db->num_samples[pid] = q.get<uint32_t>();
            db->num_local[pid] = q.get<uint32_t>();
            db->last_id[pid] = q.get<uint32_t>();
            uint32_t bits = q.get<uint32_t>();
            q.take(8);                               // is_parent: 4 bytes written, 8 advanced (pattern.cpp:35-37)
            db->num_bits[pid] = bits;
            size_t words = (size_t)((bits + 127) / 128) * 2;
            const uint8_t* d = q.take(words * 8);
            if (!q.ok) return fail("bad pattern");
            db->data_offset[pid] = db->data.size();
            size_t o = db->data.size();
            db->data.resize(o + words);
            if (words) std::memcpy(db->data.data() + o, d, words * 8);
            db->pattern_section_bytes += 40 + words * 8;
            ++pid;
        }
    }
    munmap(map, (size_t)st.st_size);
    db->data.push_back(0);                            // one padding word for 2-word decode windows
    db->data.push_back(0);

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
    *out = db;
    return 0;
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
