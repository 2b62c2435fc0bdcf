"""kmerdb_amd — Python binding (ctypes) of the MI355X common-k-mer counting engine.

The product is the C-ABI shared library ``libkmdb_amd.so`` (include/kmdb_amd.h); this package
only marshals numpy arrays / raw device pointers into it for tests, bench.py and
multi-GPU orchestration with torch.distributed.  There is no Python or CPU compute path:
every call ends up in the HIP kernels and fails loudly when the library or a GPU is missing.

The directory is called ``kmer-db_amd`` (not importable by name); load it with
``import_kmerdb_amd()`` from the repo-root helper ``_kmerdb_loader.py``.
"""
from .capi import (  # noqa: F401
    ABI_VERSION,
    DeviceDB,
    HostDB,
    KmdbError,
    NodeDB,
    SparseRows,
    device_count,
    ALPHABETS,
    extract_kmers,
    extract_kmers_alphabet,
    format_dense_row,
    format_header,
    format_sparse_row,
    lib,
    lib_path,
    make_view,
    sort_unique,
)
