// ref_driver.cpp — ORACLE INFRASTRUCTURE (not product code).
//
// A thin command-line driver around the REAL reference hot path.  It is compiled together
// with the reference's own translation units, from the sources where they lie under
// /root/reference (see oracle/Makefile: similarity_calculator.cpp, prefix_kmer_db.cpp,
// pattern.cpp, log.cpp, parallel_sorter.cpp, simd/row_add_avx{,2}.cpp) into
// oracle/_ref/ref_driver.  No reference source is copied into this repository; this file
// only CALLS the reference's public C++ interface:
//   PrefixKmerDb::addKmers / serialize / deserialize      (prefix_kmer_db.h:89-129)
//   SimilarityCalculator::all2all / all2all_sp / one2all<false> / one2all_sp / db2db_sp
//                                                          (similarity_calculator.h:4-16)
//   SparseMatrix::compact2 / saveRowSparse                 (array.h:391-446, 625-637)
//
// Usage:
//   ref_driver build    <kmers.bin> <out.db> [threads] [alphabet: nt (default), aa, aa11_diamond, aa12_mmseqs, aa6_dayhoff]
//   ref_driver all2all  <db> <out.u32> [threads] [bufferMb]      raw lower-triangular matrix
//   ref_driver all2all_sp <db> <out.txt> [threads] [bufferMb] [bubbleSize]
//   ref_driver one2all  <db> <queries.bin> <out.u32> [threads]   nq x N dense rows
//   ref_driver one2all_sp <db> <queries.bin> <out.txt> [threads]
//   ref_driver new2all  <db> <queries.bin> <out.u32> <workers>   the reference's new2all: `workers` threads each running one2all<false>
//                                                                 on its next query (console_new2all.cpp:64-95); wall clock of all queries
//   ref_driver db2db_sp <db_row> <db_col> <out.txt> [threads]    sparse rows of the cell (row part, column part)
//
// kmers.bin / queries.bin (little endian): u32 magic 'KMRS', u32 k, f64 fraction,
//   u64 n_samples, then per sample: u64 name_len, name bytes, u64 count, count x u64 k-mers
//   (sorted + unique for build; any order for queries — the driver calls KmerHelper-like
//   sort+unique itself exactly as console_new2all.cpp:73 does).
// Every command prints one JSON line with the reference-timed compute interval.

#include "prefix_kmer_db.h"
#include "similarity_calculator.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <thread>
#include <atomic>
#include <vector>
#include <algorithm>

struct SampleBlob { std::string name; std::vector<uint64_t> kmers; };

static bool read_samples(const char* path, uint32_t& k, double& fraction, std::vector<SampleBlob>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint32_t magic = 0; uint64_t n = 0;
    bool ok = fread(&magic, 4, 1, f) == 1 && magic == 0x53524d4bu && fread(&k, 4, 1, f) == 1 &&
              fread(&fraction, 8, 1, f) == 1 && fread(&n, 8, 1, f) == 1;
    out.resize(ok ? n : 0);
    for (uint64_t i = 0; ok && i < n; ++i) {
        uint64_t len = 0, cnt = 0;
        ok = fread(&len, 8, 1, f) == 1;
        out[i].name.resize(len);
        if (ok && len) ok = fread(out[i].name.data(), 1, len, f) == len;
        ok = ok && fread(&cnt, 8, 1, f) == 1;
        out[i].kmers.resize(cnt);
        if (ok && cnt) ok = fread(out[i].kmers.data(), 8, cnt, f) == cnt;
    }
    fclose(f);
    return ok;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: see header of oracle/ref_driver.cpp\n"); return 2; }
    std::string cmd = argv[1];
    try {
        if (cmd == "build") {
            int threads = argc > 4 ? atoi(argv[4]) : 1;
            uint32_t k; double fraction; std::vector<SampleBlob> samples;
            if (!read_samples(argv[2], k, fraction, samples)) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
            refresh::active_thread_pool atp(4, 1024, std::chrono::milliseconds(2));
            PrefixKmerDb db(threads);
            double t0 = now_s();
            // (the alphabet decides how many bits of a k-mer are prefix: console_build.cpp passes the -alphabet option's type on)
            const AlphabetType alpha = argc > 5 ? AlphabetFactory::instance().str2type(argv[5]) : AlphabetType::nt;
            for (auto& s : samples)
                db.addKmers(s.name, s.kmers.data(), (uint32_t)s.kmers.size(), k, fraction, alpha, atp);
            double t1 = now_s();
            std::ofstream ofs(argv[3], std::ios::binary);
            db.serialize(ofs, true);                 // console_build.cpp:149
            ofs.close();
            printf("{\"cmd\":\"build\",\"samples\":%zu,\"patterns\":%zu,\"kmers\":%zu,\"seconds\":%.6f}\n",
                   samples.size(), db.getPatternsCount(), db.getKmersCount(), t1 - t0);
            return 0;
        }

        int threads = 1; size_t bufferMb = 8;
        std::ifstream dbFile(argv[2], std::ios::binary);
        if (!dbFile) { fprintf(stderr, "Cannot open k-mer database %s\n", argv[2]); return 1; }

        if (cmd == "all2all") {
            if (argc > 4) threads = atoi(argv[4]);
            if (argc > 5) bufferMb = (size_t)atoll(argv[5]);
            PrefixKmerDb db(threads);
            if (!db.deserialize(dbFile, AbstractKmerDb::DeserializationMode::SkipHashtables)) return 1;   // console_all2all.cpp:26
            SimilarityCalculator calc(threads, bufferMb);
            LowerTriangularMatrix<uint32_t> m;
            double t0 = now_s();
            calc.all2all(db, m);                     // console_all2all.cpp:34
            double t1 = now_s();
            FILE* o = fopen(argv[3], "wb");
            fwrite(m.getData().data(), 4, m.getData().size(), o);
            fclose(o);
            printf("{\"cmd\":\"all2all\",\"samples\":%zu,\"patterns\":%zu,\"threads\":%d,\"buffer_mb\":%zu,\"seconds\":%.6f}\n",
                   db.getSamplesCount(), db.getPatternsCount(), threads, bufferMb, t1 - t0);
            return 0;
        }
        if (cmd == "all2all_sp") {
            size_t bubble = 8000;
            if (argc > 4) threads = atoi(argv[4]);
            if (argc > 5) bufferMb = (size_t)atoll(argv[5]);
            if (argc > 6) bubble = (size_t)atoll(argv[6]);
            PrefixKmerDb db(threads);
            if (!db.deserialize(dbFile, AbstractKmerDb::DeserializationMode::SkipHashtables)) return 1;   // console_all2all_sparse.cpp:35
            SimilarityCalculator calc(threads, bufferMb);
            SparseMatrix<uint32_t> m;
            CBubbleHelper bubbles(bubble);
            double t0 = now_s();
            calc.all2all_sp(db, m, bubbles);         // console_all2all_sparse.cpp:44
            double t1 = now_s();
            std::map<std::string, MetricFilter> noMetric;
            KmerFilter kf;
            CombinedFilter<uint32_t> filter(noMetric, kf, db.getSampleKmersCount(), db.getSampleKmersCount(), db.getKmerLength());
            m.compact2(filter, threads, bubbles);    // console_all2all_sparse.cpp:79
            std::vector<char> row(10000 + db.getSamplesCount() * 100);
            FILE* o = fopen(argv[3], "wb");
            for (size_t sid = 0; sid < db.getSamplesCount(); ++sid) {
                int n = m.saveRowSparse(sid, row.data(), 0);
                fwrite(row.data(), 1, n, o);
                fputc('\n', o);
            }
            fclose(o);
            printf("{\"cmd\":\"all2all_sp\",\"samples\":%zu,\"threads\":%d,\"seconds\":%.6f}\n", db.getSamplesCount(), threads, t1 - t0);
            return 0;
        }
        if (cmd == "db2db_sp") {
            // ref_driver db2db_sp <db_row> <db_col> <out.txt> [threads]: the off-diagonal cell of all2all-parts
            // (console_all2all_parts.cpp:159-226): rows of db_row against the samples of db_col
            if (argc < 5) return 2;
            if (argc > 5) threads = atoi(argv[5]);
            PrefixKmerDb db_row(threads), db_col(threads);
            std::ifstream f2(argv[3], std::ios::binary);
            if (!db_row.deserialize(dbFile, AbstractKmerDb::DeserializationMode::CompactedHashtables)) return 1;
            if (!f2 || !db_col.deserialize(f2, AbstractKmerDb::DeserializationMode::CompactedHashtables)) return 1;
            SimilarityCalculator calc(threads, bufferMb);
            SparseMatrix<uint32_t> m;
            CBubbleHelper bubbles(8000);
            double t0 = now_s();
            calc.db2db_sp(db_row, db_col, m, bubbles);      // console_all2all_parts.cpp:180
            double t1 = now_s();
            std::map<std::string, MetricFilter> noMetric;
            KmerFilter kf;
            CombinedFilter<uint32_t> filter(noMetric, kf, db_row.getSampleKmersCount(), db_col.getSampleKmersCount(), db_row.getKmerLength());
            m.compact2(filter, threads, bubbles);           // console_all2all_parts.cpp:196
            std::vector<char> row(10000 + db_col.getSamplesCount() * 100);
            FILE* o = fopen(argv[4], "wb");
            for (size_t sid = 0; sid < db_row.getSamplesCount(); ++sid) {
                int n = m.saveRowSparse(sid, row.data(), 0);
                fwrite(row.data(), 1, n, o);
                fputc('\n', o);
            }
            fclose(o);
            printf("{\"cmd\":\"db2db_sp\",\"rows\":%zu,\"cols\":%zu,\"threads\":%d,\"seconds\":%.6f}\n", db_row.getSamplesCount(),
                   db_col.getSamplesCount(), threads, t1 - t0);
            return 0;
        }
        if (cmd == "new2all") {
            // console_new2all.cpp:64-95: numThreads workers share the database and the calculator; each takes the next query,
            // makes its k-mers unique (:73) and calls one2all<false> (:82).  The rows are kept in query order.
            if (argc < 6) return 2;
            const int workers = std::max(1, atoi(argv[5]));
            PrefixKmerDb db(workers);
            if (!db.deserialize(dbFile)) return 1;
            uint32_t k; double fraction; std::vector<SampleBlob> qs;
            if (!read_samples(argv[3], k, fraction, qs)) { fprintf(stderr, "cannot read %s\n", argv[3]); return 1; }
            SimilarityCalculator calc(workers, bufferMb);
            std::vector<std::vector<uint32_t>> rows(qs.size());
            std::atomic<size_t> next{0};
            double t0 = now_s();
            std::vector<std::thread> th;
            for (int t = 0; t < workers; ++t)
                th.emplace_back([&]() {
                    for (size_t i; (i = next.fetch_add(1)) < qs.size();) {
                        auto& q = qs[i];
                        std::sort(q.kmers.begin(), q.kmers.end());
                        q.kmers.erase(std::unique(q.kmers.begin(), q.kmers.end()), q.kmers.end());
                        calc.one2all<false>(db, q.kmers.data(), q.kmers.size(), rows[i]);
                    }
                });
            for (auto& x : th) x.join();
            double t1 = now_s();
            FILE* o = fopen(argv[4], "wb");
            for (auto& r : rows) fwrite(r.data(), 4, r.size(), o);
            fclose(o);
            printf("{\"cmd\":\"new2all\",\"queries\":%zu,\"threads\":%d,\"seconds\":%.6f}\n", qs.size(), workers, t1 - t0);
            return 0;
        }
        if (cmd == "one2all" || cmd == "one2all_sp") {
            if (argc < 5) return 2;
            if (argc > 5) threads = atoi(argv[5]);
            PrefixKmerDb db(threads);
            if (!db.deserialize(dbFile)) return 1;   // console_new2all.cpp:32
            uint32_t k; double fraction; std::vector<SampleBlob> qs;
            if (!read_samples(argv[3], k, fraction, qs)) { fprintf(stderr, "cannot read %s\n", argv[3]); return 1; }
            SimilarityCalculator calc(threads, bufferMb);
            FILE* o = fopen(argv[4], "wb");
            double total = 0;
            for (auto& q : qs) {
                std::sort(q.kmers.begin(), q.kmers.end());          // console_new2all.cpp:73 (KmerHelper::unique)
                q.kmers.erase(std::unique(q.kmers.begin(), q.kmers.end()), q.kmers.end());
                double t0 = now_s();
                if (cmd == "one2all") {
                    std::vector<uint32_t> sims;
                    calc.one2all<false>(db, q.kmers.data(), q.kmers.size(), sims);    // console_new2all.cpp:82
                    total += now_s() - t0;
                    fwrite(sims.data(), 4, sims.size(), o);
                } else {
                    std::vector<std::pair<sample_id_t, num_kmers_t>> sims;
                    calc.one2all_sp(db, q.kmers.data(), q.kmers.size(), sims);        // console_new2all.cpp:78
                    total += now_s() - t0;
                    for (auto& x : sims) fprintf(o, "%u:%u,", x.first + 1, x.second);
                    fputc('\n', o);
                }
            }
            fclose(o);
            printf("{\"cmd\":\"%s\",\"queries\":%zu,\"threads\":%d,\"seconds\":%.6f}\n", cmd.c_str(), qs.size(), threads, total);
            return 0;
        }
    } catch (std::exception& e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 2;
}
