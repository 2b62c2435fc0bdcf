"""ctypes front for the CPU ORACLE (test infrastructure, NOT product code).

Wraps oracle/libkmdb_oracle.so (C restatement, see kmdb_oracle.h) and, when present, the
oracle/_ref/ref_driver binary (the real reference hot path compiled from /root/reference).
Also restates the reference consoles' CSV text format so whole-pipeline outputs can be pinned
byte-for-byte against the reference's golden files:

  header / rows ............ console_all2all.cpp:40-78, console_new2all.cpp:99-160
  dense row ................ array.h:254-257 + conversion.h:274-284
  sparse row ............... conversion.h:286-298 (1-based col:val,)
  FASTA -> sample .......... genome_input_file.h:60-100,142-213,287-337; loader_ex.cpp:168

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import gzip
import json
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkmdb_oracle.so")
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")


class _Pattern(C.Structure):
    _fields_ = [("num_kmers", C.c_int64), ("parent_id", C.c_int64), ("num_samples", C.c_uint32),
                ("num_local", C.c_uint32), ("last_sample_id", C.c_uint32), ("num_bits", C.c_uint32),
                ("is_parent", C.c_uint32), ("data", C.POINTER(C.c_uint64))]


class _Table(C.Structure):
    _fields_ = [("max_fill", C.c_double), ("filled", C.c_uint64), ("allocated", C.c_uint64),
                ("size_when_restruct", C.c_uint64), ("mask", C.c_uint64), ("ht_memory", C.c_uint64),
                ("ht_total", C.c_uint64), ("ht_match", C.c_uint64), ("slots", C.POINTER(C.c_uint64))]


class _Db(C.Structure):
    _fields_ = [("format_word", C.c_uint64), ("kmer_length", C.c_uint32), ("fraction", C.c_double),
                ("start_fraction", C.c_double), ("alphabet", C.c_int32), ("is_initialized", C.c_uint8),
                ("kmers_count", C.c_uint64), ("n_samples", C.c_uint64),
                ("sample_names", C.POINTER(C.c_char_p)), ("sample_kmers", C.POINTER(C.c_uint64)),
                ("n_buckets", C.c_uint64), ("tables", C.POINTER(_Table)),
                ("n_patterns", C.c_uint64), ("patterns", C.POINTER(_Pattern)),
                ("pattern_section_bytes", C.c_uint64), ("blob_", C.c_void_p), ("blob_bytes_", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", HERE, "libkmdb_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(LIB_PATH)
        L.kmo_db_load.restype = C.POINTER(_Db)
        L.kmo_db_load.argtypes = [C.c_char_p, C.c_int]
        L.kmo_db_free.argtypes = [C.POINTER(_Db)]
        L.kmo_last_error.restype = C.c_char_p
        L.kmo_gamma_decode.restype = C.c_uint32
        L.kmo_gamma_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.kmo_gamma_encode.restype = C.c_uint32
        L.kmo_gamma_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.kmo_decode_chain.restype = C.c_uint32
        L.kmo_decode_chain.argtypes = [C.POINTER(_Db), C.c_int64, C.c_void_p]
        L.kmo_all2all_dense.argtypes = [C.POINTER(_Db), C.c_void_p]
        L.kmo_all2all_flat.argtypes = [C.POINTER(_Db), C.c_void_p]
        L.kmo_update_counts.argtypes = [C.POINTER(_Db), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.kmo_one2all.argtypes = [C.POINTER(_Db), C.c_void_p, C.c_size_t, C.c_void_p]
        L.kmo_db2db_dense.argtypes = [C.POINTER(_Db), C.POINTER(_Db), C.c_void_p]
        L.kmo_extract_kmers.restype = C.c_size_t
        L.kmo_extract_kmers.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.kmo_extract_kmers_alphabet.restype = C.c_size_t
        L.kmo_extract_kmers_alphabet.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.kmo_sort_unique.restype = C.c_size_t
        L.kmo_sort_unique.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


# Which comparisons against the REAL reference (oracle/_ref, a git-ignored binary that ships to the GPU box) actually ran: every call of
# have_ref() is recorded with its caller, so that a test log says whether a test compared with the reference or with the oracle alone
# (tests/conftest.py prints the list).  KMDB_REQUIRE_REF=1: a missing reference build is an error, not a silently weaker test.
REF_BRANCHES = {}          # "file:function" -> True (reference compared) / False (reference absent)


def have_ref():
    import inspect
    ok = os.path.exists(REF_DRIVER)
    try:
        fr = inspect.stack()[1]
        REF_BRANCHES[os.path.basename(fr.filename) + ":" + fr.function] = ok
    except Exception:                            # (never let the bookkeeping break a comparison)
        pass
    if not ok and os.environ.get("KMDB_REQUIRE_REF", "") == "1":
        raise AssertionError("KMDB_REQUIRE_REF=1: oracle/_ref/ref_driver (the compiled reference) is missing — build it where /root/reference exists (make -C oracle)")
    return ok


# ----------------------------------------------------------------------------------------
# gamma helpers
# ----------------------------------------------------------------------------------------
def gamma_encode(values):
    v = np.ascontiguousarray(values, dtype=np.uint32)
    words = np.zeros(max(2, (int(v.size) * 64 + 127) // 128 * 2), dtype=np.uint64)
    nbits = lib().kmo_gamma_encode(v.ctypes.data, v.size, words.ctypes.data)
    return words[: max(2, (nbits + 127) // 128 * 2)], nbits


def gamma_decode(words, nbits, max_out):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    out = np.zeros(max_out, dtype=np.uint32)
    n = lib().kmo_gamma_decode(w.ctypes.data, nbits, out.ctypes.data)
    return out[:n]


# ----------------------------------------------------------------------------------------
# database
# ----------------------------------------------------------------------------------------
class OracleDB:
    """A .db file parsed by the oracle's own reader (prefix_kmer_db.cpp:578-748 restated)."""

    def __init__(self, path, skip_hashtables=False):
        self._p = lib().kmo_db_load(os.fsencode(path), 2 if skip_hashtables else 0)
        if not self._p:
            raise RuntimeError(lib().kmo_last_error().decode())
        d = self._p.contents
        self.N = int(d.n_samples)
        self.P = int(d.n_patterns)
        self.k = int(d.kmer_length)
        self.fraction = float(d.fraction)
        self.start_fraction = float(d.start_fraction)
        self.names = [d.sample_names[i].decode() for i in range(self.N)]
        self.sample_kmers = np.array([d.sample_kmers[i] for i in range(self.N)], dtype=np.uint64)
        self.n_buckets = int(d.n_buckets)
        self.pattern_section_bytes = int(d.pattern_section_bytes)

    def close(self):
        if self._p:
            lib().kmo_db_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tri_size(self):
        return self.N * (self.N - 1) // 2 if self.N else 0

    def all2all_dense(self):
        out = np.zeros(max(1, self.tri_size()), dtype=np.uint32)
        lib().kmo_all2all_dense(self._p, out.ctypes.data)
        return out[: self.tri_size()]

    def all2all_flat(self):
        out = np.zeros(max(1, self.tri_size()), dtype=np.uint32)
        lib().kmo_all2all_flat(self._p, out.ctypes.data)
        return out[: self.tri_size()]

    def update_counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().kmo_update_counts(self._p, C.byref(a), C.byref(b), C.byref(c))
        return {"tree_updates": a.value, "flat_updates": b.value, "sum_matrix": c.value}

    def one2all(self, kmers):
        q = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(max(1, self.N), dtype=np.uint32)
        lib().kmo_one2all(self._p, q.ctypes.data, q.size, out.ctypes.data)
        return out[: self.N]

    def db2db(self, col):
        """shared k-mers between every sample of this database (rows) and every sample of `col` (columns)"""
        out = np.zeros((self.N, col.N), dtype=np.uint32)
        buf = out if out.size else np.zeros(1, np.uint32)
        rc = lib().kmo_db2db_dense(self._p, col._p, buf.ctypes.data)
        assert rc == 0, "k-mer lengths differ"
        return out

    def decode_chain(self, pid):
        n = self._p.contents.patterns[pid].num_samples
        out = np.zeros(max(1, n), dtype=np.uint32)
        lib().kmo_decode_chain(self._p, pid, out.ctypes.data)
        return out[:n]

    def pattern_headers(self):
        """(num_kmers, parent, n, l, last, bits) arrays — convenience for tests."""
        d = self._p.contents
        P = self.P
        arr = np.zeros((P, 6), dtype=np.int64)
        for i in range(P):
            p = d.patterns[i]
            arr[i] = (p.num_kmers, p.parent_id, p.num_samples, p.num_local, p.last_sample_id, p.num_bits)
        return arr


# ----------------------------------------------------------------------------------------
# FASTA -> k-mers (genome_input_file.h restated in Python + kmer_extract.h restated in C)
# ----------------------------------------------------------------------------------------
_EXTS = ["", ".fa", ".fna", ".fasta", ".gz", ".fa.gz", ".fna.gz", ".fasta.gz"]   # genome_input_file.h:73-75


def _read_fasta_text(path_no_ext):
    for e in _EXTS:
        p = path_no_ext + e
        if os.path.exists(p):
            with open(p, "rb") as f:
                raw = f.read()
            if raw[:2] == b"\x1f\x8b":
                raw = gzip.decompress(raw)
            return raw
    raise FileNotFoundError(path_no_ext)


def _split_records(raw):
    """extractSubsequences (genome_input_file.h:287-337): header up to first space, newlines removed."""
    recs = []
    for chunk in raw.split(b">")[1:]:
        nl = chunk.find(b"\n")
        header = chunk[:nl].rstrip(b"\r") if nl >= 0 else chunk
        sp = header.find(b" ")
        if sp >= 0:
            header = header[:sp]
        seq = chunk[nl + 1:].replace(b"\n", b"").replace(b"\r", b"") if nl >= 0 else b""
        recs.append((header.decode(), seq))
    return recs


def extract_seq(seq, k, fraction=1.0, start_fraction=0.0, preserve_strand=False):
    out = np.zeros(max(1, len(seq)), dtype=np.uint64)
    n = lib().kmo_extract_kmers(seq, len(seq), k, fraction, start_fraction, int(preserve_strand), out.ctypes.data)
    return out[:n]


# the reference's alphabets (src/alphabet.h:79-86): name -> (groups, preserve strand)
ALPHABETS = {"nt": ("A,C,G,TU", False), "nt-preserve": ("A,C,G,TU", True),
             "aa": ("K,R,E,D,Q,N,C,G,H,I,L,V,M,F,Y,W,P,S,T,A", True), "aa11_diamond": ("KREDQN,C,G,H,ILV,M,F,Y,W,P,STA", True),
             "aa12_mmseqs": ("AST,C,DN,EQ,FY,G,H,IV,KR,LM,P,W", True), "aa6_dayhoff": ("STPAG,NDEQ,HRK,MILV,FYW,C", True)}


def extract_seq_alphabet(seq, k, alphabet, fraction=1.0, start_fraction=0.0):
    """KmerHelper::extract over one of the reference's alphabets (`build -alphabet <name>`, src/params.cpp / src/alphabet.h)"""
    groups, preserve = ALPHABETS[alphabet]
    out = np.zeros(max(1, len(seq)), dtype=np.uint64)
    n = lib().kmo_extract_kmers_alphabet(seq, len(seq), k, groups.encode(), fraction, start_fraction, int(preserve), out.ctypes.data)
    return out[:n]


def sort_unique(kmers):
    a = np.ascontiguousarray(kmers, dtype=np.uint64).copy()
    n = lib().kmo_sort_unique(a.ctypes.data, a.size)
    return a[:n]


def load_samples(list_file, k, fraction=1.0, multisample=False, unique=True):
    """The samples a `build` / `new2all` run would see: [(name, kmers)].
    One sample per listed file (name = basename, loader_ex.cpp:168) or, with
    -multisample-fasta, one per FASTA record (name = header, genome_input_file.h:253)."""
    base = os.path.dirname(os.path.abspath(list_file))
    with open(list_file) as f:
        entries = [ln.strip() for ln in f if ln.strip()]
    samples = []
    for e in entries:
        p = e if os.path.isabs(e) else os.path.normpath(os.path.join(os.getcwd(), e))
        if not any(os.path.exists(p + x) for x in _EXTS):
            p = os.path.join(base, os.path.basename(e))
        recs = _split_records(_read_fasta_text(p))
        if multisample:
            for h, s in recs:
                km = extract_seq(s, k, fraction)
                samples.append((h, sort_unique(km) if unique else km))
        else:
            parts = [extract_seq(s, k, fraction) for _, s in recs]
            km = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
            samples.append((os.path.basename(e), sort_unique(km) if unique else km))
    return samples


def write_kmers_bin(path, k, fraction, samples):
    with open(path, "wb") as f:
        f.write(struct.pack("<IId", 0x53524d4b, k, fraction))
        f.write(struct.pack("<Q", len(samples)))
        for name, km in samples:
            nb = name.encode()
            f.write(struct.pack("<Q", len(nb)))
            f.write(nb)
            km = np.ascontiguousarray(km, dtype=np.uint64)
            f.write(struct.pack("<Q", km.size))
            f.write(km.tobytes())


# ----------------------------------------------------------------------------------------
# the real reference (oracle/_ref/ref_driver)
# ----------------------------------------------------------------------------------------
def _run_ref(args):
    r = subprocess.run([REF_DRIVER] + [str(a) for a in args], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("ref_driver failed: " + r.stderr[-2000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def ref_build(kmers_bin, out_db, threads=1, alphabet="nt"):
    return _run_ref(["build", kmers_bin, out_db, threads, alphabet])


def ref_all2all(db_path, out_path, threads=1, buffer_mb=8):
    info = _run_ref(["all2all", db_path, out_path, threads, buffer_mb])
    return np.fromfile(out_path, dtype=np.uint32), info


def ref_all2all_sp(db_path, out_path, threads=1, buffer_mb=8, bubble=8000):
    info = _run_ref(["all2all_sp", db_path, out_path, threads, buffer_mb, bubble])
    with open(out_path, "rb") as f:
        return f.read(), info


def ref_db2db_sp(db_row, db_col, out_path, threads=1):
    """the reference's db2db_sp + compact2 on two databases: sparse rows 'col1based:val,' of the cell"""
    info = _run_ref(["db2db_sp", db_row, db_col, out_path, threads])
    with open(out_path, "rb") as f:
        return f.read(), info


def ref_one2all(db_path, queries_bin, out_path, threads=1):
    info = _run_ref(["one2all", db_path, queries_bin, out_path, threads])
    return np.fromfile(out_path, dtype=np.uint32), info


def ref_new2all(db_path, queries_bin, out_path, workers):
    """the reference's new2all compute: `workers` threads, one one2all<false> per query (console_new2all.cpp:64-95); wall clock"""
    info = _run_ref(["new2all", db_path, queries_bin, out_path, workers])
    return np.fromfile(out_path, dtype=np.uint32), info


def ref_one2all_sp(db_path, queries_bin, out_path, threads=1):
    info = _run_ref(["one2all_sp", db_path, queries_bin, out_path, threads])
    with open(out_path, "rb") as f:
        return f.read(), info


# ----------------------------------------------------------------------------------------
# CSV text (byte-compatible restatement of the consoles)
# ----------------------------------------------------------------------------------------
def _fmt_fraction(x):
    return "%g" % x        # ostream default formatting (console_all2all.cpp:40)


def csv_header(k, fraction, names, sample_kmers):
    s = "kmer-length: %d fraction: %s ,db-samples ," % (k, _fmt_fraction(fraction))
    s += "".join(n + "," for n in names) + "\n"
    s += "query-samples,total-kmers," + "".join("%d," % int(c) for c in sample_kmers) + "\n"
    return s


def tri_row(matrix, i):
    o = i * (i - 1) // 2
    return matrix[o: o + i]


def format_all2all(k, fraction, names, sample_kmers, matrix, sparse=False):
    out = [csv_header(k, fraction, names, sample_kmers)]
    for i, name in enumerate(names):
        row = tri_row(matrix, i)
        if sparse:
            nz = np.nonzero(row)[0]
            body = "".join("%d:%d," % (j + 1, row[j]) for j in nz)
        else:
            body = "".join("%d," % v for v in row)
        out.append("%s,%d,%s\n" % (name, int(sample_kmers[i]), body))
    return "".join(out).encode()


def format_new2all(k, fraction, names, sample_kmers, queries, rows, sparse=False):
    """queries: [(name, n_unique_kmers)], rows: list of dense uint32[N] results."""
    out = [csv_header(k, fraction, names, sample_kmers)]
    for (qn, qc), row in zip(queries, rows):
        if sparse:
            nz = np.nonzero(row)[0]
            body = "".join("%d:%d," % (j + 1, row[j]) for j in nz)
        else:
            body = "".join("%d," % v for v in row)
        out.append("%s,%d,%s\n" % (qn, qc, body))
    return "".join(out).encode()
