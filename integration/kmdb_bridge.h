// kmdb_bridge.h — the glue a kmer-db maintainer adds to the reference's src/ to route the all2all / new2all hot path
// through libkmdb_amd.so.  Compiles against the UNMODIFIED reference headers: the fields of
// pattern_t that have no getter (last_sample_id) are taken from pattern_t::pack, the reference's own serialiser
// (reference src/pattern.cpp:15-46), and the hashtables are walked slot by slot through hash_map_lp::cbegin/cend
// (src/hashmap_lp.h:125-128), which keeps the slot-exact layout the device probes expect.
//
// oracle/Makefile builds integration/bridge_driver.cpp against this header and the reference's translation units
// (oracle/_ref/bridge_driver); tests/test_gpu_parity.py runs it against the reference's golden outputs.
#pragma once
#include "kmdb_amd.h"
#include "prefix_kmer_db.h"
#include "sparse_filters.h"

#include <cstring>
#include <stdexcept>
#include <vector>

// Flatten getPatterns() / getHashtables() (prefix_kmer_db.h:108-113) into a kmdb_db_view.
struct KmdbFlatDb {
    std::vector<int64_t> num_kmers, parent;
    std::vector<uint32_t> n, l, last, bits;
    std::vector<uint64_t> off, data, bucket_off, slots;
    kmdb_db_view view{};

    KmdbFlatDb(const PrefixKmerDb& db, bool with_hashtables) {
        const auto& pats = db.getPatterns();
        const size_t P = pats.size();
        num_kmers.resize(P); parent.resize(P); n.resize(P); l.resize(P); last.resize(P); bits.resize(P); off.resize(P);
        size_t words = 0;
        for (const auto& p : pats) words += p.get_data_bytes() / 8;
        data.resize(words + 2);
        std::vector<char> buf;
        size_t w = 0;
        for (size_t i = 0; i < P; ++i) {
            const pattern_t& p = pats[i];
            const size_t db_ = p.get_data_bytes();
            buf.resize(40 + db_);
            p.pack(buf.data());                                   // on-disk image: 40-byte header + stream (pattern.cpp:15-46)
            std::memcpy(&num_kmers[i], buf.data() + 0, 8);
            std::memcpy(&parent[i], buf.data() + 8, 8);
            std::memcpy(&n[i], buf.data() + 16, 4);
            std::memcpy(&l[i], buf.data() + 20, 4);
            std::memcpy(&last[i], buf.data() + 24, 4);
            std::memcpy(&bits[i], buf.data() + 28, 4);
            off[i] = w;
            if (db_) std::memcpy(&data[w], buf.data() + 40, db_);
            w += db_ / 8;
        }
        if (with_hashtables) {
            const auto& hts = db.getHashtables();
            bucket_off.push_back(0);
            for (const auto& ht : hts) {                          // item_t{u32 key; i32 val}, empty = INT32_MAX (hashmap_lp.h:71-78)
                for (auto it = ht.cbegin(); it != ht.cend(); ++it)
                    slots.push_back((uint64_t)(uint32_t)it->key | ((uint64_t)(uint32_t)it->val << 32));
                bucket_off.push_back(slots.size());
            }
        }
        view.abi_version = KMDB_ABI_VERSION;  view.kmer_length = db.getKmerLength();
        view.n_samples = db.getSamplesCount(); view.n_patterns = P;
        view.num_kmers = num_kmers.data();    view.parent_id = parent.data();
        view.num_samples = n.data();          view.num_local = l.data();
        view.last_sample_id = last.data();    view.num_bits = bits.data();
        view.data_offset = off.data();        view.data = data.data();   view.n_data_words = w;
        view.n_buckets = with_hashtables ? bucket_off.size() - 1 : 0;
        view.bucket_offset = with_hashtables ? bucket_off.data() : nullptr;
        view.slots = with_hashtables ? slots.data() : nullptr;
    }
};

inline void kmdb_check(int rc) { if (rc) throw std::runtime_error(kmdb_last_error()); }

// drop-in for `calculator.all2all(*db, matrix)` (console_all2all.cpp:34): fills the caller-owned matrix
inline void kmdb_bridge_all2all(const PrefixKmerDb& db, LowerTriangularMatrix<uint32_t>& matrix, int device = 0) {
    matrix.resize(db.getSamplesCount());
    KmdbFlatDb flat(db, /*with_hashtables=*/false);
    kmdb_opts o{};
    o.abi_version = KMDB_ABI_VERSION; o.device = device; o.shard_count = 1;
    kmdb_db* gpu = nullptr;
    kmdb_check(kmdb_db_upload(&flat.view, &o, 0, &gpu));
    const int rc = kmdb_all2all_dense(gpu, matrix.getData().data(), &o);
    kmdb_db_free(gpu);
    kmdb_check(rc);
}

// The -min / -max filters of the reference's Params (params.h:99-100, filled by Params::parse, params.cpp:418-455) as the
// bounds kmdb_all2all_sparse_filtered takes: one entry per criterion, the k-mer count filter as KMDB_METRIC_NUM_KMERS.
inline std::vector<kmdb_cell_filter> kmdb_bridge_filters(const std::map<std::string, MetricFilter>& metricFilters, const KmerFilter& kmerFilter) {
    std::vector<kmdb_cell_filter> out;
    for (const auto& kv : metricFilters) {
        const int id = kmdbh_metric_id(kv.first.c_str());
        if (id < 0) throw std::runtime_error("kmdb bridge: unknown criterion " + kv.first);
        out.push_back(kmdb_cell_filter{id, 0, kv.second.bounds[0], kv.second.bounds[1]});
    }
    const KmerFilter all;
    if (kmerFilter.bounds[0] != all.bounds[0] || kmerFilter.bounds[1] != all.bounds[1])
        out.push_back(kmdb_cell_filter{KMDB_METRIC_NUM_KMERS, 0, (double)kmerFilter.bounds[0], (double)kmerFilter.bounds[1]});
    return out;
}

// drop-in for `calculator.all2all_sp(...)` + `matrix.compact2(filter)` (console_all2all_sparse.cpp:44-79): the filtered rows
inline void kmdb_bridge_all2all_sp(kmdb_db* gpu, const PrefixKmerDb& db, const std::map<std::string, MetricFilter>& metricFilters, const KmerFilter& kmerFilter,
                                   kmdb_sparse_rows* rows, const kmdb_opts* o) {
    const std::vector<kmdb_cell_filter> fl = kmdb_bridge_filters(metricFilters, kmerFilter);
    if (fl.empty()) kmdb_check(kmdb_all2all_sparse(gpu, rows, o));
    else kmdb_check(kmdb_all2all_sparse_filtered(gpu, fl.data(), fl.size(), db.getSampleKmersCount().data(), -1, rows, o));
}
