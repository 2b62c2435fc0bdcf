// host_csv.cpp — byte-compatible CSV text of the all2all / all2all-sp / new2all consoles.
//
// Format contract (reference files):
//   line 1  "kmer-length: <k> fraction: <f> ,db-samples ,<name>,...,\n"   console_all2all.cpp:40-42
//           (<f> printed with default ostream formatting, i.e. like printf("%g"))
//   line 2  "query-samples,total-kmers,<cnt>,...,\n"                       console_all2all.cpp:48-51
//   rows    "<name>,<cnt>," then either every value followed by ','        array.h:254-257
//           or "<col+1>:<val>," for the kept entries                       conversion.h:286-298
//   numbers are plain decimal                                              conversion.h:99-165
#include "kmdb_amd.h"

#include <cstdio>
#include <cstring>

namespace {

// "00" "01" ... "99"
struct Pairs {
    char d[200];
    constexpr Pairs() : d() { for (int i = 0; i < 100; ++i) { d[2 * i] = (char)('0' + i / 10); d[2 * i + 1] = (char)('0' + i % 10); } }
};
constexpr Pairs PAIRS{};

// decimal digits of v, returns length (no terminator): the number of digits first, then two digits per division from the right
// (the cells of a 10 000-sample table are fifty million numbers)
inline size_t put_u64(uint64_t v, char* out) {
    if (v < 10) { out[0] = (char)('0' + v); return 1; }
    if (v <= 0xFFFFFFFFull) {
        uint32_t x = (uint32_t)v;
        const size_t n = x < 100u ? 2 : x < 1000u ? 3 : x < 10000u ? 4 : x < 100000u ? 5 : x < 1000000u ? 6 : x < 10000000u ? 7 : x < 100000000u ? 8 : x < 1000000000u ? 9 : 10;
        char* p = out + n;
        while (x >= 100u) { const uint32_t q = x / 100u, r = x - q * 100u; p -= 2; p[0] = PAIRS.d[2 * r]; p[1] = PAIRS.d[2 * r + 1]; x = q; }
        if (x >= 10u) { p -= 2; p[0] = PAIRS.d[2 * x]; p[1] = PAIRS.d[2 * x + 1]; }
        else *--p = (char)('0' + x);
        return n;
    }
    char tmp[24];
    size_t n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (size_t i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return n;
}

inline size_t put_row_head(const char* name, uint64_t kmers, char* out) {
    size_t len = std::strlen(name);
    std::memcpy(out, name, len);
    char* p = out + len;
    *p++ = ',';
    p += put_u64(kmers, p);
    *p++ = ',';
    return (size_t)(p - out);
}

}  // namespace

extern "C" size_t kmdbh_format_header(const kmdbh_db* db, char* out, size_t cap) {
    uint64_t n = kmdbh_db_n_samples(db);
    size_t need = 128;
    for (uint64_t i = 0; i < n; ++i) need += std::strlen(kmdbh_db_sample_name(db, i)) + 1 + 21;
    if (cap < need) return 0;
    char* p = out;
    p += std::snprintf(p, 96, "kmer-length: %u fraction: %g ,db-samples ,", kmdbh_db_kmer_length(db), kmdbh_db_fraction(db));
    for (uint64_t i = 0; i < n; ++i) {
        const char* s = kmdbh_db_sample_name(db, i);
        size_t len = std::strlen(s);
        std::memcpy(p, s, len);
        p += len;
        *p++ = ',';
    }
    *p++ = '\n';
    static const char second[] = "query-samples,total-kmers,";
    std::memcpy(p, second, sizeof(second) - 1);
    p += sizeof(second) - 1;
    for (uint64_t i = 0; i < n; ++i) {
        p += put_u64(kmdbh_db_sample_kmers(db, i), p);
        *p++ = ',';
    }
    *p++ = '\n';
    return (size_t)(p - out);
}

extern "C" size_t kmdbh_format_dense_row(const char* name, uint64_t kmers, const uint32_t* row, size_t n, char* out) {
    char* p = out + put_row_head(name, kmers, out);
    for (size_t j = 0; j < n; ++j) {
        p += put_u64(row[j], p);
        *p++ = ',';
    }
    *p++ = '\n';
    return (size_t)(p - out);
}

extern "C" size_t kmdbh_format_sparse_row(const char* name, uint64_t kmers, const uint32_t* cols, const uint32_t* vals,
                                          size_t n, char* out) {
    char* p = out + put_row_head(name, kmers, out);
    for (size_t j = 0; j < n; ++j) {
        p += put_u64((uint64_t)cols[j] + 1, p);
        *p++ = ':';
        p += put_u64(vals[j], p);
        *p++ = ',';
    }
    *p++ = '\n';
    return (size_t)(p - out);
}
