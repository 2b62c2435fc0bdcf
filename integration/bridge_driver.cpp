// bridge_driver.cpp — proof that integration/kmdb_bridge.h compiles and runs inside the reference: the database is
// loaded by the reference's own PrefixKmerDb::deserialize, flattened by the bridge and handed to libkmdb_amd.so.
//   bridge_driver all2all    <db> <out.u32>     raw lower-triangular matrix (compare: ref_driver all2all)
//   bridge_driver all2all_sp <db> <out.txt>     sparse rows "col1based:val," (compare: ref_driver all2all_sp)
//   bridge_driver sample_rows_ref <db> <out.txt> -sample-rows ...   the reference's own sampled rows (CPU only: expected output for the front-end)
//   bridge_driver new2all    <db> <out.u32>     every sample's own k-mers... not available from a .db: the first
//                                               query is the keys of bucket 0 instead (hash lookups through the
//                                               slot-exact tables), one row of N counts
#include "kmdb_bridge.h"
#include "params.h"
#include "similarity_calculator.h"

#include <cstdio>
#include <fstream>
#include <string>

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: bridge_driver all2all|all2all_sp|all2all_sp_filtered|new2all <db> <out> [-min/-max options]\n"); return 2; }
    const std::string cmd = argv[1];
    try {
        std::ifstream f(argv[2], std::ios::binary);
        if (!f) { fprintf(stderr, "Cannot open k-mer database %s\n", argv[2]); return 1; }
        PrefixKmerDb db(1);
        const bool ht = cmd == "new2all";
        if (!db.deserialize(f, ht ? AbstractKmerDb::DeserializationMode::Everything : AbstractKmerDb::DeserializationMode::SkipHashtables)) return 1;
        if (cmd == "all2all") {
            LowerTriangularMatrix<uint32_t> m;
            kmdb_bridge_all2all(db, m);                               // was: calculator.all2all(*db, matrix)  (console_all2all.cpp:34)
            FILE* o = fopen(argv[3], "wb");
            fwrite(m.getData().data(), 4, m.getData().size(), o);
            fclose(o);
            return 0;
        }
        if (cmd == "sample_rows_ref") {
            // bridge_driver sample_rows_ref <db> <out.txt> -sample-rows [crit:]count [-min ..]* [-max ..]*: the REFERENCE alone (no GPU) — its Params::parse,
            // all2all_sp, SparseMatrix::add_to_sampler and Sampler::saveRowSparse (console_all2all_sparse.cpp:44,70-89) — as the expected output of
            // `kmer-db-amd all2all-sp -sample-rows`
            std::vector<std::string> av = {"kmer-db", "all2all-sp"};
            for (int i = 4; i < argc; ++i) av.push_back(argv[i]);
            av.push_back(argv[2]); av.push_back(argv[3]);
            std::vector<char*> avp;
            for (auto& a : av) avp.push_back(a.data());
            Params params;
            if (!params.parse((int)avp.size(), avp.data())) { fprintf(stderr, "Params::parse rejected the options\n"); return 2; }
            SimilarityCalculator calc(2, 8);
            SparseMatrix<uint32_t> matrix;
            CBubbleHelper bubbles(params.bubbleSize);
            calc.all2all_sp(db, matrix, bubbles);
            const auto& counts = db.getSampleKmersCount();
            CombinedFilter<uint32_t> filter(params.metricFilters, params.kmerFilter, counts, counts, db.getKmerLength());
            using sampler_t = Sampler<uint32_t, uint32_t, double>;
            sampler_t sampler(db.getSamplesCount(), params.samplingSize, params.samplingCriterion ? sampler_t::strategy_t::best : sampler_t::strategy_t::random);
            matrix.add_to_sampler(filter, sampler, params.samplingCriterion, counts, counts, 0, 0, db.getKmerLength(), 2, bubbles);
            std::vector<char> row(10000 + db.getSamplesCount() * 100);
            FILE* out = fopen(argv[3], "wb");
            for (size_t sid = 0; sid < db.getSamplesCount(); ++sid) {
                const int len = sampler.saveRowSparse(sid, row.data(), 0);
                fwrite(row.data(), 1, (size_t)len, out);
                fputc('\n', out);
            }
            fclose(out);
            return 0;
        }
        KmdbFlatDb flat(db, ht);
        kmdb_opts o{};
        o.abi_version = KMDB_ABI_VERSION; o.shard_count = 1;
        kmdb_db* gpu = nullptr;
        kmdb_check(kmdb_db_upload(&flat.view, &o, ht ? 1 : 0, &gpu));
        if (cmd == "all2all_sp") {
            kmdb_sparse_rows sp{};
            kmdb_check(kmdb_all2all_sparse(gpu, &sp, &o));            // was: all2all_sp + compact2 (console_all2all_sparse.cpp:44,79)
            FILE* out = fopen(argv[3], "wb");
            for (uint64_t r = 0; r < sp.n_rows; ++r) {
                for (uint64_t e = sp.row_ptr[r]; e < sp.row_ptr[r + 1]; ++e) fprintf(out, "%u:%u,", sp.col[e] + 1, sp.val[e]);
                fputc('\n', out);
            }
            fclose(out);
            kmdb_sparse_free(&sp);
        } else if (cmd == "all2all_sp_filtered") {
            // bridge_driver all2all_sp_filtered <db> <out.txt> [-min [crit:]v]* [-max [crit:]v]*: the options go through the reference's
            // own Params::parse (params.cpp:59-139); the engine's filtered rows must equal the reference's CombinedFilter
            // (sparse_filters.h:38-61) applied to the unfiltered rows
            std::vector<std::string> av = {"kmer-db", "all2all-sp"};
            for (int i = 4; i < argc; ++i) av.push_back(argv[i]);
            av.push_back(argv[2]); av.push_back(argv[3]);
            std::vector<char*> avp;
            for (auto& a : av) avp.push_back(a.data());
            Params params;
            if (!params.parse((int)avp.size(), avp.data())) { fprintf(stderr, "Params::parse rejected the options\n"); kmdb_db_free(gpu); return 2; }
            kmdb_sparse_rows sp{}, all{};
            kmdb_bridge_all2all_sp(gpu, db, params.metricFilters, params.kmerFilter, &sp, &o);
            kmdb_check(kmdb_all2all_sparse(gpu, &all, &o));
            const auto& counts = db.getSampleKmersCount();
            CombinedFilter<num_kmers_t> filter(params.metricFilters, params.kmerFilter, counts, counts, (int)db.getKmerLength());
            uint64_t w = 0, kept = 0;
            bool same = true;
            for (uint64_t r = 0; r < all.n_rows && same; ++r) {
                for (uint64_t e = all.row_ptr[r]; e < all.row_ptr[r + 1]; ++e)
                    if (filter(all.val[e], (int)r, (int)all.col[e])) {
                        same = same && w < sp.nnz && sp.col[w] == all.col[e] && sp.val[w] == all.val[e];
                        ++w; ++kept;
                    }
                same = same && sp.row_ptr[r + 1] == w;
            }
            FILE* out = fopen(argv[3], "wb");
            for (uint64_t r = 0; r < sp.n_rows; ++r) {
                for (uint64_t e = sp.row_ptr[r]; e < sp.row_ptr[r + 1]; ++e) fprintf(out, "%u:%u,", sp.col[e] + 1, sp.val[e]);
                fputc('\n', out);
            }
            fclose(out);
            const uint64_t nnz = sp.nnz, total = all.nnz;
            kmdb_sparse_free(&sp); kmdb_sparse_free(&all);
            if (!same || kept != nnz) { fprintf(stderr, "all2all_sp_filtered: engine rows differ from CombinedFilter on the unfiltered rows\n"); kmdb_db_free(gpu); return 3; }
            printf("all2all_sp_filtered: %llu of %llu cells kept, identical to the reference's CombinedFilter\n", (unsigned long long)nnz, (unsigned long long)total);
        } else if (cmd == "new2all") {
            // query = every key stored in bucket 0's table, widened back to a k-mer (bucket << 32 | key): all of them hit
            std::vector<uint64_t> q;
            const auto& t0 = db.getHashtables()[0];
            for (auto it = t0.cbegin(); it != t0.cend(); ++it) if (!t0.is_free(*it)) q.push_back((uint64_t)(uint32_t)it->key);
            std::sort(q.begin(), q.end());
            const uint64_t* qp = q.data();
            size_t qn = q.size();
            std::vector<uint32_t> row(db.getSamplesCount() + 1);
            kmdb_check(kmdb_new2all_batch(gpu, &qp, &qn, 1, row.data(), &o));   // was: calculator.one2all<false>(...) (console_new2all.cpp:82)
            std::vector<uint32_t> ref;
            SimilarityCalculator calc(1, 8);
            calc.one2all<false>(db, q.data(), q.size(), ref);
            FILE* out = fopen(argv[3], "wb");
            fwrite(row.data(), 4, db.getSamplesCount(), out);
            fclose(out);
            if (ref.size() != db.getSamplesCount() || std::memcmp(ref.data(), row.data(), ref.size() * 4) != 0) {
                fprintf(stderr, "new2all: engine row differs from SimilarityCalculator::one2all\n");
                kmdb_db_free(gpu);
                return 3;
            }
            printf("new2all: %zu k-mers, row identical to SimilarityCalculator::one2all\n", q.size());
        } else { fprintf(stderr, "unknown command\n"); kmdb_db_free(gpu); return 2; }
        kmdb_db_free(gpu);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
}
