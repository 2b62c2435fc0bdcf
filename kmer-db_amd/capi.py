"""ctypes declarations for include/kmdb_amd.h and thin numpy-facing wrappers."""
import ctypes as C
import os
import sys

import numpy as np

ABI_VERSION = 7
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class KmdbError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libkmdb_amd.so")


class _View(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("kmer_length", C.c_uint32),
        ("n_samples", C.c_uint64), ("n_patterns", C.c_uint64),
        ("num_kmers", C.c_void_p), ("parent_id", C.c_void_p), ("num_samples", C.c_void_p),
        ("num_local", C.c_void_p), ("last_sample_id", C.c_void_p), ("num_bits", C.c_void_p),
        ("data_offset", C.c_void_p), ("data", C.c_void_p), ("n_data_words", C.c_uint64),
        ("n_buckets", C.c_uint64), ("bucket_offset", C.c_void_p), ("slots", C.c_void_p),
    ]


class _Opts(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device", C.c_int32), ("shard_index", C.c_uint32),
        ("shard_count", C.c_uint32), ("bubble_size", C.c_uint32), ("flags", C.c_uint32),
        ("stream", C.c_void_p),
    ]


class _Sparse(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("nnz", C.c_uint64), ("row_ptr", C.POINTER(C.c_uint64)),
                ("col", C.POINTER(C.c_uint32)), ("val", C.POINTER(C.c_uint32)), ("measure", C.POINTER(C.c_double))]


class _CellFilter(C.Structure):
    _fields_ = [("metric", C.c_int32), ("reserved", C.c_int32), ("lo", C.c_double), ("hi", C.c_double)]


METRICS = ["jaccard", "min", "max", "cosine", "mash", "ani", "ani-shorter", "mash-query", "num-kmers"]


class _Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double),
                ("algorithmic_bytes", C.c_uint64), ("tree_updates", C.c_uint64), ("sum_pairs", C.c_uint64),
                ("device_bytes", C.c_uint64), ("n_segments", C.c_uint64), ("tile_flushes", C.c_uint64),
                ("k1_ms", C.c_double), ("k2_ms", C.c_double), ("n_records", C.c_uint64), ("k0_ms", C.c_double),
                ("k1n_ms", C.c_double), ("k1g_ms", C.c_double), ("upload_ms", C.c_double), ("n_wide", C.c_uint64),
                ("n_chunks", C.c_uint64), ("path", C.c_uint32), ("width", C.c_uint32), ("sized_call", C.c_uint32),
                ("n_joined", C.c_uint32), ("n_patterns", C.c_uint64), ("h2d_bytes", C.c_uint64), ("n_direct", C.c_uint64)]


class _NodeStats(C.Structure):
    _fields_ = [("n_shards", C.c_uint32), ("n_devices", C.c_uint32), ("rccl_version", C.c_int32), ("reserved", C.c_uint32),
                ("upload_s", C.c_double), ("plan_s", C.c_double), ("call_ms", C.c_double), ("collective_ms", C.c_double), ("d2h_ms", C.c_double)]


class _NodeDeviceStats(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_shards", C.c_uint32), ("upload_s", C.c_double), ("call_ms", C.c_double), ("collective_ms", C.c_double),
                ("d2h_ms", C.c_double), ("h2d_bytes", C.c_uint64), ("n_patterns", C.c_uint64), ("n_records", C.c_uint64)]


FLAG_FORCE_GLOBAL_ATOMICS = 1
FLAG_FORCE_DIRECT = 2
FLAG_FORCE_TILE = 4
FLAG_NO_FALLBACK = 8
FLAG_ONE_SHOT = 16
PATH_NONE, PATH_RECORDS, PATH_TILE, PATH_GLOBAL = 0, 1, 2, 3

# every symbol include/kmdb_amd.h declares
EXPORTS = [
    "kmdb_last_error", "kmdb_abi_version", "kmdb_device_count", "kmdb_device_prepare", "kmdb_db_upload", "kmdb_db_upload_shard", "kmdb_db_free", "kmdb_db_settle", "kmdb_db_stats", "kmdb_db_fallback_reason",
    "kmdb_node_upload", "kmdb_node_free", "kmdb_node_stats_get", "kmdb_node_device_stats_get", "kmdb_node_all2all_dense", "kmdb_node_all2all_sparse",
    "kmdb_all2all_dense", "kmdb_all2all_dense_device", "kmdb_all2all_sparse", "kmdb_all2all_sparse_filtered", "kmdb_sparse_from_dense_device", "kmdbh_metric", "kmdbh_metric_id", "kmdb_sparse_free",
    "kmdb_new2all_batch", "kmdb_new2all_batch_sparse", "kmdb_new2all_batch_seq", "kmdb_new2all_batch_seq_alphabet", "kmdb_db2db_dense",
    "kmdbh_shard_plan_counts", "kmdbh_db_load", "kmdbh_db_free", "kmdbh_db_release_patterns", "kmdbh_db_view", "kmdbh_db_kmer_length", "kmdbh_db_fraction",
    "kmdbh_db_start_fraction", "kmdbh_db_alphabet", "kmdbh_db_n_samples", "kmdbh_db_sample_name",
    "kmdbh_db_sample_kmers", "kmdbh_db_pattern_section_bytes", "kmdbh_extract_kmers", "kmdbh_extract_kmers_alphabet", "kmdbh_alphabet_table", "kmdbh_sort_unique",
    "kmdbh_format_header", "kmdbh_format_dense_row", "kmdbh_format_sparse_row",
]


def lib():
    """Load libkmdb_amd.so; never falls back to anything else."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise KmdbError("libkmdb_amd.so is not built (run __graft_entry__.build() or `make -C kmer-db_amd`)")
    # torch ships its own copy of the HIP runtime; if the process is going to use torch as well (device
    # buffers for the RCCL reduce), it has to be the first one loaded or torch finds no device afterwards
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(p)
    L.kmdb_last_error.restype = C.c_char_p
    L.kmdb_db_upload.argtypes = [C.POINTER(_View), C.POINTER(_Opts), C.c_int, C.POINTER(C.c_void_p)]
    L.kmdb_db_upload_shard.argtypes = [C.POINTER(_View), C.POINTER(_Opts), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.kmdb_db_free.argtypes = [C.c_void_p]
    L.kmdb_db_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
    L.kmdb_db_fallback_reason.argtypes = [C.c_void_p]
    L.kmdb_db_fallback_reason.restype = C.c_char_p
    L.kmdb_node_upload.argtypes = [C.POINTER(_View), C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]
    L.kmdb_node_free.argtypes = [C.c_void_p]
    L.kmdb_node_stats_get.argtypes = [C.c_void_p, C.POINTER(_NodeStats)]
    L.kmdb_node_device_stats_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_NodeDeviceStats)]
    L.kmdb_node_all2all_dense.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Opts)]
    L.kmdb_node_all2all_sparse.argtypes = [C.c_void_p, C.POINTER(_CellFilter), C.c_size_t, C.c_void_p, C.c_int, C.POINTER(_Sparse), C.POINTER(_Opts)]
    L.kmdb_all2all_dense.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Opts)]
    L.kmdb_all2all_dense_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Opts)]
    L.kmdb_all2all_sparse.argtypes = [C.c_void_p, C.POINTER(_Sparse), C.POINTER(_Opts)]
    L.kmdb_all2all_sparse_filtered.argtypes = [C.c_void_p, C.POINTER(_CellFilter), C.c_size_t, C.c_void_p, C.c_int, C.POINTER(_Sparse), C.POINTER(_Opts)]
    L.kmdb_sparse_from_dense_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(_CellFilter), C.c_size_t, C.c_void_p, C.c_int,
                                                C.POINTER(_Sparse), C.POINTER(_Opts)]
    L.kmdbh_metric.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.kmdbh_metric.restype = C.c_double
    L.kmdbh_metric_id.argtypes = [C.c_char_p]
    L.kmdb_sparse_free.argtypes = [C.POINTER(_Sparse)]
    L.kmdb_new2all_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p, C.POINTER(_Opts)]
    L.kmdb_new2all_batch_sparse.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(_Sparse), C.POINTER(_Opts)]
    L.kmdb_db2db_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Opts)]
    L.kmdb_new2all_batch_seq.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_double, C.c_double, C.c_int,
                                         C.c_void_p, C.c_void_p, C.POINTER(_Opts)]
    L.kmdb_new2all_batch_seq_alphabet.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_double, C.c_double, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.POINTER(_Opts)]
    L.kmdbh_db_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.kmdbh_db_free.argtypes = [C.c_void_p]
    L.kmdbh_db_release_patterns.argtypes = [C.c_void_p]
    L.kmdbh_db_release_patterns.restype = None
    L.kmdbh_db_view.restype = C.POINTER(_View)
    L.kmdbh_db_view.argtypes = [C.c_void_p]
    L.kmdbh_db_kmer_length.restype = C.c_uint32
    L.kmdbh_db_kmer_length.argtypes = [C.c_void_p]
    L.kmdbh_db_fraction.restype = C.c_double
    L.kmdbh_db_fraction.argtypes = [C.c_void_p]
    L.kmdbh_db_start_fraction.restype = C.c_double
    L.kmdbh_db_start_fraction.argtypes = [C.c_void_p]
    L.kmdbh_db_alphabet.restype = C.c_int32
    L.kmdbh_db_alphabet.argtypes = [C.c_void_p]
    L.kmdbh_db_n_samples.restype = C.c_uint64
    L.kmdbh_db_n_samples.argtypes = [C.c_void_p]
    L.kmdbh_db_sample_name.restype = C.c_char_p
    L.kmdbh_db_sample_name.argtypes = [C.c_void_p, C.c_uint64]
    L.kmdbh_db_sample_kmers.restype = C.c_uint64
    L.kmdbh_db_sample_kmers.argtypes = [C.c_void_p, C.c_uint64]
    L.kmdbh_db_pattern_section_bytes.restype = C.c_uint64
    L.kmdbh_db_pattern_section_bytes.argtypes = [C.c_void_p]
    L.kmdbh_extract_kmers.restype = C.c_size_t
    L.kmdbh_extract_kmers.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_double, C.c_double, C.c_int, C.c_void_p]
    L.kmdbh_extract_kmers_alphabet.restype = C.c_size_t
    L.kmdbh_extract_kmers_alphabet.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_int32, C.c_double, C.c_double, C.c_void_p]
    L.kmdbh_alphabet_table.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.kmdbh_sort_unique.restype = C.c_size_t
    L.kmdbh_sort_unique.argtypes = [C.c_void_p, C.c_size_t]
    L.kmdbh_format_header.restype = C.c_size_t
    L.kmdbh_format_header.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.kmdbh_format_dense_row.restype = C.c_size_t
    L.kmdbh_format_dense_row.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_char_p]
    L.kmdbh_format_sparse_row.restype = C.c_size_t
    L.kmdbh_format_sparse_row.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise KmdbError(lib().kmdb_last_error().decode(errors="replace"))


def device_count():
    return int(lib().kmdb_device_count())


def _opts(device=0, shard=(0, 1), flags=0, stream=None, bubble=0):
    o = _Opts()
    o.abi_version = ABI_VERSION
    o.device = device
    o.shard_index, o.shard_count = shard
    o.bubble_size = bubble
    o.flags = flags
    o.stream = stream
    return o


# ------------------------------------------------------------------------------------------------
class HostDB:
    """A .db file parsed by the front-end's reader (kmdbh_db_load)."""

    def __init__(self, path, skip_hashtables=False):
        self._h = C.c_void_p()
        _check(lib().kmdbh_db_load(os.fsencode(path), 2 if skip_hashtables else 0, C.byref(self._h)))
        L = lib()
        self.N = int(L.kmdbh_db_n_samples(self._h))
        self.k = int(L.kmdbh_db_kmer_length(self._h))
        self.fraction = float(L.kmdbh_db_fraction(self._h))
        self.start_fraction = float(L.kmdbh_db_start_fraction(self._h))
        self.alphabet = int(L.kmdbh_db_alphabet(self._h))
        self.names = [L.kmdbh_db_sample_name(self._h, i).decode() for i in range(self.N)]
        self.sample_kmers = np.array([L.kmdbh_db_sample_kmers(self._h, i) for i in range(self.N)], dtype=np.uint64)
        self.pattern_section_bytes = int(L.kmdbh_db_pattern_section_bytes(self._h))

    @property
    def view(self):
        return lib().kmdbh_db_view(self._h)

    def shard_plan_counts(self, n_shards):
        """kmdbh_shard_plan_counts: (nodes kept, k-mers owned) per prefix shard, planned on the host"""
        kept = np.zeros(n_shards, np.uint64)
        kmers = np.zeros(n_shards, np.uint64)
        L = lib()
        L.kmdbh_shard_plan_counts.argtypes = [C.POINTER(_View), C.c_uint32, C.c_void_p, C.c_void_p]
        _check(L.kmdbh_shard_plan_counts(self.view, n_shards, kept.ctypes.data, kmers.ctypes.data))
        return kept, kmers

    def release_patterns(self):
        """kmdbh_db_release_patterns: the pattern arrays' pages go back to the kernel (after the upload); names and counts stay"""
        lib().kmdbh_db_release_patterns(self._h)

    def view_arrays(self):
        """numpy copies of the flat view (for tests of the reader)."""
        v = self.view.contents
        P = int(v.n_patterns)

        def arr(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()

        out = {
            "num_kmers": arr(v.num_kmers, P, np.int64), "parent_id": arr(v.parent_id, P, np.int64),
            "num_samples": arr(v.num_samples, P, np.uint32), "num_local": arr(v.num_local, P, np.uint32),
            "last_sample_id": arr(v.last_sample_id, P, np.uint32), "num_bits": arr(v.num_bits, P, np.uint32),
            "data_offset": arr(v.data_offset, P, np.uint64), "data": arr(v.data, int(v.n_data_words), np.uint64),
            "n_buckets": int(v.n_buckets),
        }
        if v.n_buckets:
            out["bucket_offset"] = arr(v.bucket_offset, int(v.n_buckets) + 1, np.uint64)
            out["slots"] = arr(v.slots, int(out["bucket_offset"][-1]), np.uint64)
        return out

    def header_bytes(self):
        buf = C.create_string_buffer(20000 + 200 * self.N + sum(len(n) for n in self.names))
        n = lib().kmdbh_format_header(self._h, buf, len(buf))
        return buf.raw[:n]

    def close(self):
        if self._h:
            lib().kmdbh_db_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_view(kmer_length, n_samples, num_kmers, parent_id, num_samples, num_local, last_sample_id, num_bits,
              data_offset, data, bucket_offset=None, slots=None):
    """Build a kmdb_db_view over caller-owned numpy arrays; returns (view, keepalive)."""
    keep = [np.ascontiguousarray(num_kmers, np.int64), np.ascontiguousarray(parent_id, np.int64),
            np.ascontiguousarray(num_samples, np.uint32), np.ascontiguousarray(num_local, np.uint32),
            np.ascontiguousarray(last_sample_id, np.uint32), np.ascontiguousarray(num_bits, np.uint32),
            np.ascontiguousarray(data_offset, np.uint64), np.ascontiguousarray(data, np.uint64)]
    v = _View()
    v.abi_version = ABI_VERSION
    v.kmer_length = kmer_length
    v.n_samples = n_samples
    v.n_patterns = keep[0].size
    (v.num_kmers, v.parent_id, v.num_samples, v.num_local, v.last_sample_id, v.num_bits, v.data_offset, v.data) = [
        a.ctypes.data for a in keep]
    v.n_data_words = keep[7].size
    if bucket_offset is not None:
        bo = np.ascontiguousarray(bucket_offset, np.uint64)
        sl = np.ascontiguousarray(slots, np.uint64)
        keep += [bo, sl]
        v.n_buckets = bo.size - 1
        v.bucket_offset = bo.ctypes.data
        v.slots = sl.ctypes.data
    return v, keep


class SparseRows:
    def __init__(self, raw):
        n, nnz = int(raw.n_rows), int(raw.nnz)
        self.row_ptr = np.ctypeslib.as_array(raw.row_ptr, shape=(n + 1,)).copy()
        self.col = np.ctypeslib.as_array(raw.col, shape=(max(nnz, 1),))[:nnz].copy()
        self.val = np.ctypeslib.as_array(raw.val, shape=(max(nnz, 1),))[:nnz].copy()
        self.n_rows, self.nnz = n, nnz
        self.measure = np.ctypeslib.as_array(raw.measure, shape=(max(nnz, 1),))[:nnz].copy() if raw.measure else None

    def row(self, i):
        a, b = int(self.row_ptr[i]), int(self.row_ptr[i + 1])
        return self.col[a:b], self.val[a:b]


class DeviceDB:
    """A database resident in HBM (kmdb_db_upload)."""

    def __init__(self, src, device=0, with_hashtables=False, flags=0, prefix_shard=None):
        """prefix_shard=(index, count): keep only the k-mers of the prefix buckets b with b % count == index
        (kmdb_db_upload_shard; the source must carry the hashtables)."""
        self._keep = None
        if isinstance(src, HostDB):
            view = src.view
            self._keep = src
        elif isinstance(src, tuple):
            view, self._keep = C.pointer(src[0]), src
        else:
            raise TypeError("DeviceDB expects a HostDB or the (view, keepalive) pair from make_view()")
        self.device = device
        self._d = C.c_void_p()
        o = _opts(device, (0, 1), flags)
        if prefix_shard is None:
            _check(lib().kmdb_db_upload(view, C.byref(o), int(with_hashtables), C.byref(self._d)))
        else:
            _check(lib().kmdb_db_upload_shard(view, C.byref(o), int(with_hashtables), int(prefix_shard[0]), int(prefix_shard[1]),
                                              C.byref(self._d)))
        self.N = int(view.contents.n_samples)
        self.P = int(view.contents.n_patterns)

    def tri_size(self):
        return self.N * (self.N - 1) // 2 if self.N else 0

    def all2all_dense(self, shard=(0, 1), flags=0):
        out = np.zeros(max(1, self.tri_size()), dtype=np.uint32)
        o = _opts(self.device, shard, flags)
        _check(lib().kmdb_all2all_dense(self._d, out.ctypes.data, C.byref(o)))
        return out[: self.tri_size()]

    def all2all_dense_device(self, dev_ptr, stream=None, shard=(0, 1), flags=0):
        """Result stays in device memory at dev_ptr (e.g. torch tensor .data_ptr())."""
        o = _opts(self.device, shard, flags, stream)
        _check(lib().kmdb_all2all_dense_device(self._d, C.c_void_p(dev_ptr), C.byref(o)))

    def all2all_sparse(self, shard=(0, 1)):
        raw = _Sparse()
        o = _opts(self.device, shard)
        _check(lib().kmdb_all2all_sparse(self._d, C.byref(raw), C.byref(o)))
        try:
            return SparseRows(raw)
        finally:
            lib().kmdb_sparse_free(C.byref(raw))

    def all2all_sparse_filtered(self, filters, sample_kmers, measure=None):
        """filters: [(criterion name, lo, hi)], None = unbounded; measure: a criterion name whose value is returned for every kept cell"""
        raw = _Sparse()
        o = _opts(self.device)
        fs = (_CellFilter * max(1, len(filters)))()
        for i, (name, lo, hi) in enumerate(filters):
            fs[i].metric = METRICS.index(name)
            fs[i].lo = -np.finfo(np.float64).max if lo is None else lo
            fs[i].hi = np.finfo(np.float64).max if hi is None else hi
        cnt = np.ascontiguousarray(sample_kmers, np.uint32)
        _check(lib().kmdb_all2all_sparse_filtered(self._d, fs, len(filters), cnt.ctypes.data, -1 if measure is None else METRICS.index(measure),
                                                  C.byref(raw), C.byref(o)))
        try:
            return SparseRows(raw)
        finally:
            lib().kmdb_sparse_free(C.byref(raw))

    def sparse_from_dense_device(self, dev_ptr, cell_lo=0, cell_hi=None, filters=(), sample_kmers=None, measure=None, stream=None):
        """Sparse rows of caller-accumulated cells [cell_lo, cell_hi) of the lower triangle, dev_ptr = device address of cell_lo
        (kmdb_sparse_from_dense_device): the compaction stage of all2all-sp after a multi-GPU reduce of the partial matrices."""
        raw = _Sparse()
        o = _opts(self.device, stream=stream)
        fs = (_CellFilter * max(1, len(filters)))()
        for i, (name, lo, hi) in enumerate(filters):
            fs[i].metric = METRICS.index(name)
            fs[i].lo = -np.finfo(np.float64).max if lo is None else lo
            fs[i].hi = np.finfo(np.float64).max if hi is None else hi
        cnt = None if sample_kmers is None else np.ascontiguousarray(sample_kmers, np.uint32)
        _check(lib().kmdb_sparse_from_dense_device(self._d, C.c_void_p(dev_ptr), int(cell_lo), self.tri_size() if cell_hi is None else int(cell_hi),
                                                   fs, len(filters), None if cnt is None else cnt.ctypes.data,
                                                   -1 if measure is None else METRICS.index(measure), C.byref(raw), C.byref(o)))
        try:
            return SparseRows(raw)
        finally:
            lib().kmdb_sparse_free(C.byref(raw))

    def new2all(self, queries):
        qs = [np.ascontiguousarray(q, np.uint64) for q in queries]
        nq = len(qs)
        ptrs = (C.c_void_p * max(nq, 1))(*[q.ctypes.data for q in qs])
        cnts = (C.c_size_t * max(nq, 1))(*[q.size for q in qs])
        out = np.zeros((nq, self.N), dtype=np.uint32)
        o = _opts(self.device)
        _check(lib().kmdb_new2all_batch(self._d, ptrs, cnts, nq, out.ctypes.data if out.size else None, C.byref(o)))
        return out

    def db2db(self, col):
        """shared k-mers between every sample of this database (rows) and every sample of `col` (columns)"""
        out = np.zeros((self.N, col.N), dtype=np.uint32)
        buf = out if out.size else np.zeros(1, np.uint32)
        o = _opts(self.device)
        _check(lib().kmdb_db2db_dense(self._d, col._d, buf.ctypes.data, C.byref(o)))
        return out

    def new2all_seq(self, seqs, fraction=1.0, start_fraction=0.0, preserve_strand=False, alphabet=None):
        """queries given as sequence text (bytes / str); k-mer extraction, minhash filter, sort + unique on the device.
        alphabet: the database's AlphabetType (HostDB.alphabet; ALPHABETS lists the names) — None: nt / nt-preserve by preserve_strand.
        Returns (similarities nq x N, unique k-mer count per query)."""
        if alphabet is None:
            alphabet = 1 if preserve_strand else 0
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        nq = len(bs)
        ptrs = (C.c_char_p * max(nq, 1))(*bs)
        lens = (C.c_size_t * max(nq, 1))(*[len(b) for b in bs])
        out = np.zeros((nq, self.N), dtype=np.uint32)
        cnt = np.zeros(max(nq, 1), dtype=np.uint64)
        o = _opts(self.device)
        _check(lib().kmdb_new2all_batch_seq_alphabet(self._d, ptrs, lens, nq, float(fraction), float(start_fraction), int(alphabet),
                                                     out.ctypes.data if out.size else None, cnt.ctypes.data, C.byref(o)))
        return out, cnt[:nq]

    def new2all_sparse(self, queries):
        qs = [np.ascontiguousarray(q, np.uint64) for q in queries]
        nq = len(qs)
        ptrs = (C.c_void_p * max(nq, 1))(*[q.ctypes.data for q in qs])
        cnts = (C.c_size_t * max(nq, 1))(*[q.size for q in qs])
        raw = _Sparse()
        o = _opts(self.device)
        _check(lib().kmdb_new2all_batch_sparse(self._d, ptrs, cnts, nq, C.byref(raw), C.byref(o)))
        try:
            return SparseRows(raw)
        finally:
            lib().kmdb_sparse_free(C.byref(raw))

    def stats(self):
        s = _Stats()
        _check(lib().kmdb_db_stats(self._d, C.byref(s)))
        return {f: getattr(s, f) for f, _ in _Stats._fields_}

    def fallback_reason(self):
        """why the last all2all call could not take the block-record pipeline ("" when it did)"""
        return lib().kmdb_db_fallback_reason(self._d).decode(errors="replace")

    def close(self):
        if self._d:
            lib().kmdb_db_free(self._d)
            self._d = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NodeDB:
    """One database prefix-sharded over the devices of the node (kmdb_node_upload): n_shards shards, shard s on devices[s % D]."""

    def __init__(self, src, n_shards, devices=(0,)):
        self._keep = src
        view = src.view if isinstance(src, HostDB) else C.pointer(src[0])
        self._n = C.c_void_p()
        devs = (C.c_int32 * len(devices))(*devices)
        _check(lib().kmdb_node_upload(view, int(n_shards), devs, len(devices), C.byref(self._n)))
        self.N = int(view.contents.n_samples)

    def tri_size(self):
        return self.N * (self.N - 1) // 2 if self.N else 0

    def all2all_dense(self):
        out = np.zeros(max(1, self.tri_size()), dtype=np.uint32)
        _check(lib().kmdb_node_all2all_dense(self._n, out.ctypes.data, None))
        return out[: self.tri_size()]

    def all2all_sparse(self, filters=(), sample_kmers=None, measure=None):
        raw = _Sparse()
        fs = (_CellFilter * max(1, len(filters)))()
        for i, (name, lo, hi) in enumerate(filters):
            fs[i].metric = METRICS.index(name)
            fs[i].lo = -np.finfo(np.float64).max if lo is None else lo
            fs[i].hi = np.finfo(np.float64).max if hi is None else hi
        cnt = None if sample_kmers is None else np.ascontiguousarray(sample_kmers, np.uint32)
        _check(lib().kmdb_node_all2all_sparse(self._n, fs, len(filters), None if cnt is None else cnt.ctypes.data,
                                              -1 if measure is None else METRICS.index(measure), C.byref(raw), None))
        try:
            return SparseRows(raw)
        finally:
            lib().kmdb_sparse_free(C.byref(raw))

    def stats(self):
        s = _NodeStats()
        _check(lib().kmdb_node_stats_get(self._n, C.byref(s)))
        out = {f: getattr(s, f) for f, _ in _NodeStats._fields_}
        out["devices"] = []
        for slot in range(s.n_devices):
            ds = _NodeDeviceStats()
            _check(lib().kmdb_node_device_stats_get(self._n, slot, C.byref(ds)))
            out["devices"].append({f: getattr(ds, f) for f, _ in _NodeDeviceStats._fields_})
        return out

    def close(self):
        if self._n:
            lib().kmdb_node_free(self._n)
            self._n = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
def extract_kmers(seq, k, fraction=1.0, start_fraction=0.0, preserve_strand=False):
    if isinstance(seq, str):
        seq = seq.encode()
    out = np.zeros(max(1, len(seq)), dtype=np.uint64)
    n = lib().kmdbh_extract_kmers(seq, len(seq), k, fraction, start_fraction, int(preserve_strand), out.ctypes.data)
    return out[:n]


ALPHABETS = ("nt", "nt-preserve", "aa", "aa11_diamond", "aa12_mmseqs", "aa6_dayhoff")      # AlphabetType order (reference src/alphabet.h:10-18)


def extract_kmers_alphabet(seq, k, alphabet, fraction=1.0, start_fraction=0.0):
    """k-mer words of a sequence over any alphabet of the reference (alphabet = index into ALPHABETS or its name)"""
    if isinstance(seq, str):
        seq = seq.encode()
    a = ALPHABETS.index(alphabet) if isinstance(alphabet, str) else int(alphabet)
    out = np.zeros(max(1, len(seq)), dtype=np.uint64)
    n = lib().kmdbh_extract_kmers_alphabet(seq, len(seq), k, a, fraction, start_fraction, out.ctypes.data)
    return out[:n]


def sort_unique(kmers):
    a = np.ascontiguousarray(kmers, np.uint64).copy()
    n = lib().kmdbh_sort_unique(a.ctypes.data, a.size)
    return a[:n]


def format_header(hostdb):
    return hostdb.header_bytes()


def format_dense_row(name, kmers, row):
    r = np.ascontiguousarray(row, np.uint32)
    buf = C.create_string_buffer(len(name) + 64 + 11 * r.size)
    n = lib().kmdbh_format_dense_row(name.encode(), int(kmers), r.ctypes.data, r.size, buf)
    return buf.raw[:n]


def format_sparse_row(name, kmers, cols, vals):
    c = np.ascontiguousarray(cols, np.uint32)
    v = np.ascontiguousarray(vals, np.uint32)
    buf = C.create_string_buffer(len(name) + 64 + 22 * c.size)
    n = lib().kmdbh_format_sparse_row(name.encode(), int(kmers), c.ctypes.data, v.ctypes.data, c.size, buf)
    return buf.raw[:n]
