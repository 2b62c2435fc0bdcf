// host_kmers.cpp — query-side k-mer extraction for new2all (front-end, CPU).
//
// Produces exactly the 64-bit k-mer words the reference's loader feeds to one2all:
//   * 2 bits per base, A,C,G,T/U = 0..3, either case (reference src/alphabet.h:80, :52-56)
//   * canonical k-mer = min(forward, reverse complement) unless the alphabet preserves the
//     strand (src/kmer_extract.h:79-84)
//   * a window containing a non-ACGT symbol is dropped (src/kmer_extract.h:50-76)
//   * when 2k-32 < 8 the word is widened so that the prefix above the 32-bit suffix has at
//     least 8 bits: word = (x << s) | (x & (2^s - 1)), s = 8 - (2k - 32)
//     (src/kmer_extract.h:37-45,88) — k=18 gives 40-bit words and 256 prefix buckets
//   * optional minhash subsampling: keep the k-mer when its MurmurHash3-derived 64-bit hash
//     lies in [start*2^64, (start+fraction)*2^64) (src/filter.h:38-51,96-115)
#include "kmdb_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>

namespace {

struct BaseTable {
    int8_t code[256];
    BaseTable() {
        for (auto& c : code) c = -1;
        const char* sym = "ACGT";
        for (int i = 0; i < 4; ++i) {
            code[(unsigned char)sym[i]] = (int8_t)i;
            code[(unsigned char)(sym[i] | 0x20)] = (int8_t)i;
        }
        code[(unsigned char)'U'] = code[(unsigned char)'u'] = 3;
    }
};
const BaseTable kBases;

inline uint64_t mix64(uint64_t v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull;
    v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull;
    v ^= v >> 33;
    return v;
}

// 64-bit hash of a k-mer word as used by the reference's MinHashFilter
inline uint64_t minhash_value(uint64_t word, uint64_t quarter_len) {
    uint64_t a = word * 0x87c37b91114253d5ull;
    a = (a << 31) | (a >> 33);
    a *= 0x4cf5ad432745937full;
    uint64_t h1 = (42ull ^ a) ^ quarter_len;
    uint64_t h2 = 42ull ^ quarter_len;
    h1 += h2; h2 += h1;
    h1 = mix64(h1); h2 = mix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;
}

}  // namespace

extern "C" size_t kmdbh_extract_kmers(const char* seq, size_t len, uint32_t k, double fraction, double start_fraction,
                                      int preserve_strand, uint64_t* out) {
    if (k == 0 || k > 31 || len < k) return 0;
    const uint64_t word_mask = (1ull << (2 * k)) - 1;
    const unsigned top_shift = 2 * (k - 1);
    const int prefix_bits = 2 * (int)k - 32;
    const unsigned widen = prefix_bits < 8 ? (unsigned)(8 - prefix_bits) : 0u;
    const uint64_t tail = widen ? ((1ull << widen) - 1) : 0ull;
    const bool subsample = fraction < 1.0;
    const double u64max = (double)std::numeric_limits<uint64_t>::max();
    const uint64_t lo = (uint64_t)(u64max * start_fraction);
    const uint64_t hi = (uint64_t)(u64max * (start_fraction + fraction));
    const uint64_t quarter_len = (uint64_t)std::ceil((double)k / 4.0);

    uint64_t fwd = 0, rc = 0;
    uint32_t valid_run = 0;       // number of consecutive valid symbols ending here (capped)
    size_t n_out = 0;
    for (size_t i = 0; i < len; ++i) {
        int c = kBases.code[(unsigned char)seq[i]];
        if (c < 0) { c = 0; valid_run = 0; } else if (valid_run < k) ++valid_run;
        fwd = ((fwd << 2) | (uint64_t)c) & word_mask;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << top_shift);
        if (valid_run < k) continue;                 // window incomplete or crosses an invalid symbol
        uint64_t w = (preserve_strand || fwd < rc) ? fwd : rc;
        w = (w << widen) | (w & tail);
        if (subsample) {
            uint64_t h = minhash_value(w, quarter_len);
            if (h < lo || h >= hi) continue;
        }
        out[n_out++] = w;
    }
    return n_out;
}

// ---- every alphabet of the reference (src/alphabet.h:79-86): nt, nt-preserve and the four protein alphabets.  The groups of a description
// are the symbols, upper and lower case alike (alphabet.h:41-58); bits per symbol = ceil(log2(groups)) (:36); a k-mer of k symbols must leave
// the word's top bit free (:37: maxKmerLen = 64 / bits - 1).  Every protein alphabet preserves the strand; the reverse word exists for the
// two nucleotide alphabets only (kmer_extract.h:59,73: size - 1 - symbol).
namespace {
struct AlphabetDesc { const char* groups; int preserve; };
const AlphabetDesc kAlphabets[KMDB_ALPHABET_COUNT] = {
    {"A,C,G,TU", 0}, {"A,C,G,TU", 1}, {"K,R,E,D,Q,N,C,G,H,I,L,V,M,F,Y,W,P,S,T,A", 1}, {"KREDQN,C,G,H,ILV,M,F,Y,W,P,STA", 1},
    {"AST,C,DN,EQ,FY,G,H,IV,KR,LM,P,W", 1}, {"STPAG,NDEQ,HRK,MILV,FYW,C", 1}};
}  // namespace

extern "C" int kmdbh_alphabet_table(int32_t alphabet, int8_t* map256, uint32_t* n_symbols, uint32_t* bits_per_symbol, int* preserve_strand) {
    if (alphabet < 0 || alphabet >= KMDB_ALPHABET_COUNT || !map256) return 1;
    for (int i = 0; i < 256; ++i) map256[i] = -1;
    uint32_t size = 1;
    for (const char* g = kAlphabets[alphabet].groups; *g; ++g) {
        if (*g == ',') { ++size; continue; }
        const unsigned char c = (unsigned char)*g;
        map256[c | 0x20] = map256[c & ~0x20] = (int8_t)(size - 1);        // (letters only: either case)
    }
    uint32_t bits = 0;
    while ((1u << bits) < size) ++bits;
    if (n_symbols) *n_symbols = size;
    if (bits_per_symbol) *bits_per_symbol = bits;
    if (preserve_strand) *preserve_strand = kAlphabets[alphabet].preserve;
    return 0;
}

extern "C" size_t kmdbh_extract_kmers_alphabet(const char* seq, size_t len, uint32_t k, int32_t alphabet, double fraction, double start_fraction,
                                               uint64_t* out) {
    int8_t map[256];
    uint32_t size = 0, bits = 0;
    int preserve = 0;
    if (kmdbh_alphabet_table(alphabet, map, &size, &bits, &preserve)) return 0;
    if (k == 0 || k > 64u / bits - 1u || len < k) return 0;
    const uint64_t word_mask = (1ull << (bits * k)) - 1;
    const unsigned top_shift = bits * (k - 1);
    const int prefix_bits = (int)(bits * k) - 32;
    const unsigned widen = prefix_bits < 8 ? (unsigned)(8 - prefix_bits) : 0u;
    const uint64_t tail = widen ? ((1ull << widen) - 1) : 0ull;
    const bool subsample = fraction < 1.0;
    const double u64max = (double)std::numeric_limits<uint64_t>::max();
    const uint64_t lo = (uint64_t)(u64max * start_fraction);
    const uint64_t hi = (uint64_t)(u64max * (start_fraction + fraction));
    const uint64_t quarter_len = (uint64_t)std::ceil((double)k / 4.0);
    uint64_t fwd = 0, rc = 0;
    uint32_t valid_run = 0;
    size_t n_out = 0;
    for (size_t i = 0; i < len; ++i) {
        int c = map[(unsigned char)seq[i]];
        if (c < 0) { c = 0; valid_run = 0; } else if (valid_run < k) ++valid_run;
        fwd = ((fwd << bits) | (uint64_t)c) & word_mask;
        rc = (rc >> bits) | ((uint64_t)(size - 1 - (uint32_t)c) << top_shift);
        if (valid_run < k) continue;
        uint64_t w = (preserve || fwd < rc) ? fwd : rc;
        w = (w << widen) | (w & tail);
        if (subsample) {
            const uint64_t h = minhash_value(w, quarter_len);
            if (h < lo || h >= hi) continue;
        }
        out[n_out++] = w;
    }
    return n_out;
}

extern "C" size_t kmdbh_sort_unique(uint64_t* kmers, size_t n) {
    std::sort(kmers, kmers + n);
    return (size_t)(std::unique(kmers, kmers + n) - kmers);
}
