// host_metrics.cpp — the similarity / distance measures of a cell and their names (reference src/params.cpp:14-42).
// Host code: the measures use libm's log, and results must be bit-identical with the reference's CPU arithmetic, so the
// final decision of every -min / -max filter and every reported measure is computed here; the device only pre-filters.
#include "kmdb_amd.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {
// mash distance of a k-mer set similarity j (params.cpp:19-22): j == 0 -> 1
inline double mash_of(double j, int k) { return j == 0 ? 1.0 : (-1.0 / k) * std::log((2 * j) / (j + 1)); }
const char* const kNames[KMDB_METRIC_COUNT] = {"jaccard", "min", "max", "cosine", "mash", "ani", "ani-shorter", "mash-query", "num-kmers"};
}  // namespace

extern "C" double kmdbh_metric(int metric, uint32_t c, uint32_t a, uint32_t b, int k) {
    // the integer parts are num_kmers_t (uint32) expressions in the reference and wrap like them
    switch (metric) {
    case KMDB_METRIC_JACCARD:     return (double)c / (uint32_t)(a + b - c);
    case KMDB_METRIC_MIN:         return (double)c / std::min(a, b);
    case KMDB_METRIC_MAX:         return (double)c / std::max(a, b);
    case KMDB_METRIC_COSINE:      return (double)c / std::sqrt((uint32_t)(a * b));
    case KMDB_METRIC_MASH:        return mash_of((double)c / (uint32_t)(a + b - c), k);
    case KMDB_METRIC_ANI:         return 1.0 - mash_of((double)c / (uint32_t)(a + b - c), k);
    case KMDB_METRIC_ANI_SHORTER: return 1.0 - mash_of((double)c / std::min(a, b), k);
    case KMDB_METRIC_MASH_QUERY:  return mash_of((double)c / a, k);
    case KMDB_METRIC_NUM_KMERS:   return (double)c;
    default:                      return std::nan("");
    }
}

extern "C" int kmdbh_metric_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < KMDB_METRIC_COUNT; ++i)
        if (std::strcmp(name, kNames[i]) == 0) return i;
    return -1;
}
