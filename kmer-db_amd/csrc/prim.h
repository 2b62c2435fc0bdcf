// prim.h — the device-wide primitives the engine takes from the ROCm library (scans of small tables, the radix sorts of the upload and
// of the secondary modes, reductions): rocPRIM called directly, with the temporary-storage convention of its API (first call with
// a null pointer returns the size).  The sorts of the hot path (block records, chunk tables) are the engine's own kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <iterator>

namespace prim {

template <class In, class Out>
inline hipError_t exclusive_sum(void* tmp, size_t& bytes, In in, Out out, size_t n, hipStream_t st = 0) {
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}
template <class In, class Out>
inline hipError_t sum(void* tmp, size_t& bytes, In in, Out out, size_t n, hipStream_t st = 0) {
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::reduce(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}
template <class In, class Out>
inline hipError_t max(void* tmp, size_t& bytes, In in, Out out, size_t n, hipStream_t st = 0) {
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::reduce(tmp, bytes, in, out, T(0), n, rocprim::maximum<T>(), st);
}
template <class K, class V>
inline hipError_t sort_pairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st = 0) {
    return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
}
template <class K, class V>
inline hipError_t sort_pairs_desc(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st = 0) {
    return rocprim::radix_sort_pairs_desc(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
}
template <class K>
inline hipError_t sort_keys(void* tmp, size_t& bytes, const K* kin, K* kout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st = 0) {
    return rocprim::radix_sort_keys(tmp, bytes, kin, kout, n, begin_bit, end_bit, st);
}
template <class In, class Out, class Cnt, class Pred>
inline hipError_t select_if(void* tmp, size_t& bytes, In in, Out out, Cnt count_out, size_t n, Pred pred, hipStream_t st = 0) {
    return rocprim::select(tmp, bytes, in, out, count_out, n, pred, st);
}
template <class In, class Uniq, class Cnt, class Runs>
inline hipError_t run_length_encode(void* tmp, size_t& bytes, In in, Uniq uniq, Cnt counts, Runs n_runs, size_t n, hipStream_t st = 0) {
    return rocprim::run_length_encode(tmp, bytes, in, (unsigned int)n, uniq, counts, n_runs, st);
}

}  // namespace prim
