"""CPU-side tests of the product: the C-ABI library loads and exports what include/kmdb_amd.h
declares, the front-end's .db reader / k-mer extractor / CSV writer agree with the oracle, and
the GPU entry points fail loudly (no CPU fallback) when there is no device."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import DBS, ROOT


def test_library_exports_every_declared_symbol(K):
    hdr = open(os.path.join(ROOT, "include", "kmdb_amd.h")).read()
    declared = set(re.findall(r"\b(kmdbh?_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(K.capi.EXPORTS)
    L = ctypes.CDLL(K.lib_path())
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert K.lib().kmdb_abi_version() == K.ABI_VERSION


def test_shard_plan_on_the_host(K, golden_dir):
    """host_shards.cpp: the plan kmdb_node_upload works from — per prefix shard the k-mers it owns (hashtable items of the buckets
    b % S == s, reference src/hashmap_lp.h:71-78, bucket = kmer >> 32 src/types.h:25-27) and the nodes it keeps (a node whose subtree
    holds one of them) — against the same computed with numpy from the view."""
    for stem in ("virus_k18", "clade64", "clade64_k25_f01", "synth_k21"):
        h = K.HostDB(os.path.join(golden_dir, stem + ".db"))
        a = h.view_arrays()
        par, bo, sl = a["parent_id"], a["bucket_offset"], a["slots"]
        P = len(par)
        val = (sl >> np.uint64(32)).astype(np.int64)
        bucket = np.repeat(np.arange(len(bo) - 1), np.diff(bo).astype(np.int64))
        ok = val != 0x7FFFFFFF
        for S in (1, 2, 3, 8, 11):
            kept, kmers = h.shard_plan_counts(S)
            for s in range(S):
                w = np.bincount(val[ok & (bucket % S == s)], minlength=P)
                keep = w > 0
                for q in range(P - 1, 0, -1):
                    if keep[q] and par[q] >= 0:
                        keep[par[q]] = True
                assert int(kept[s]) == int(keep.sum()) and int(kmers[s]) == int(w.sum()), (stem, S, s)
            assert int(kmers.sum()) == int(ok.sum())
    with pytest.raises(K.KmdbError, match="no hashtables"):
        K.HostDB(os.path.join(golden_dir, "clade64.db"), skip_hashtables=True).shard_plan_counts(2)


def test_product_does_not_touch_the_oracle():
    # the product path must never import / link / call anything under oracle/
    for base, _, files in os.walk(os.path.join(ROOT, "kmer-db_amd")):
        if os.path.basename(base) in ("build", "bin", "__pycache__"):
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(base, fn), errors="replace").read()
                assert "kmdb_oracle" not in txt and "oracle." not in txt and "from oracle" not in txt, fn
    out = subprocess.run(["ldd", os.path.join(ROOT, "kmer-db_amd", "libkmdb_amd.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


@pytest.mark.parametrize("stem", DBS)
def test_db_reader_matches_oracle_reader(K, O, golden_dir, stem):
    path = os.path.join(golden_dir, stem + ".db")
    h = K.HostDB(path)
    o = O.OracleDB(path)
    assert (h.N, h.k, h.fraction, h.names) == (o.N, o.k, o.fraction, o.names)
    assert np.array_equal(h.sample_kmers, o.sample_kmers)
    assert h.pattern_section_bytes == o.pattern_section_bytes
    v = h.view_arrays()
    hd = o.pattern_headers()
    assert np.array_equal(v["num_kmers"], hd[:, 0]) and np.array_equal(v["parent_id"], hd[:, 1])
    assert np.array_equal(v["num_samples"], hd[:, 2]) and np.array_equal(v["num_local"], hd[:, 3])
    assert np.array_equal(v["last_sample_id"], hd[:, 4]) and np.array_equal(v["num_bits"], hd[:, 5])
    assert v["n_buckets"] == o.n_buckets
    # gamma streams: decode every pattern's local ids from the product's flat view with the oracle decoder
    for pid in range(0, o.P, max(1, o.P // 400)):
        l, nb = int(hd[pid, 3]), int(hd[pid, 5])
        if l > 1:
            off = int(v["data_offset"][pid])
            deltas = O.gamma_decode(v["data"][off: off + (nb + 127) // 128 * 2], nb, l)
            chain = o.decode_chain(pid)
            local = chain[len(chain) - l:]
            assert np.array_equal(np.diff(local.astype(np.int64)), deltas.astype(np.int64))
    # hashtables: every stored key is found by the oracle's find at the same pattern id
    if stem in ("virus_k18_part1", "clade64"):
        bo, sl = v["bucket_offset"], v["slots"]
        tested = 0
        for b in range(v["n_buckets"]):
            seg = sl[int(bo[b]): int(bo[b + 1])]
            vals = (seg >> np.uint64(32)).astype(np.int64)
            for it in seg[vals != 0x7fffffff][:50]:
                kmer = (np.uint64(b) << np.uint64(32)) | (it & np.uint64(0xffffffff))
                pid = int(it >> np.uint64(32))
                row = o.one2all(np.array([kmer], dtype=np.uint64))
                exp = np.zeros(o.N, np.uint32)
                if hd[pid, 0] != 0:
                    exp[o.decode_chain(pid)] = 1
                assert np.array_equal(row, exp)
                tested += 1
        assert tested > 100
    h2 = K.HostDB(path, skip_hashtables=True)
    assert h2.view_arrays()["n_buckets"] == 0
    assert np.array_equal(h2.view_arrays()["data"], v["data"])


def test_host_image_pages_given_back(K, golden_dir):
    """kmdbh_db_release_patterns (the front-end calls it once the database is on the device): the big arrays' pages are dropped — they
    read as zeros where whole pages went —, names and k-mer counts stay, and the handle is freed as usual."""
    h = K.HostDB(os.path.join(golden_dir, "clade64.db"))
    names, counts, before = list(h.names), h.sample_kmers.copy(), h.view_arrays()
    assert before["data"].nbytes > 3 * 4096 and before["data"].any()
    h.release_patterns()
    after = h.view_arrays()
    assert not after["data"][1024:-1024].any()                         # (the first and last partial pages of a block keep their bytes)
    L = K.lib()
    assert [L.kmdbh_db_sample_name(h._h, i).decode() for i in range(h.N)] == names
    assert [int(L.kmdbh_db_sample_kmers(h._h, i)) for i in range(h.N)] == [int(c) for c in counts]
    h.release_patterns()                                               # harmless twice
    h.close()


def test_db_reader_survives_damaged_files(golden_dir, tmp_path):
    """400 damaged images of a valid database (truncated anywhere, random bytes, wild 32 / 64-bit fields), each read with and without
    the hashtables on 1 - 8 threads: the reader refuses or accepts, never walks off its mapping (the fuzz runs as a process of its own and
    must end normally), and what it accepts is consistent (every stream inside the data array)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "reader_fuzz.py"), os.path.join(golden_dir, "clade64.db"), str(tmp_path / "fz.db"), "7", "200"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    acc, ref = (int(x) for x in re.findall(r"accepted (\d+) refused (\d+)", r.stdout)[0])
    assert acc + ref == 400 and ref > 50 and acc > 50


def test_db_reader_errors(K, tmp_path):
    with pytest.raises(K.KmdbError, match="Cannot open k-mer database"):
        K.HostDB(str(tmp_path / "missing.db"))
    p = tmp_path / "trunc.db"
    p.write_bytes(b"\x01" + b"\x00" * 60)
    with pytest.raises(K.KmdbError):
        K.HostDB(str(p))


def _pattern_section(raw):
    """(offset of the pattern count, P, [(offset, bytes) of every pattern]) of a .db image (layout: host_db.cpp / prefix_kmer_db.cpp:438-571)"""
    import struct
    off = 8 + 4 + 8 + 8 + 4 + 1 + 8
    n, = struct.unpack_from("<Q", raw, off)
    off += 8
    for _ in range(n):
        off += 8
        ln, = struct.unpack_from("<Q", raw, off)
        off += 8 + ln
    nb, = struct.unpack_from("<Q", raw, off)
    off += 8
    for _ in range(nb):
        filled, alloc = struct.unpack_from("<QQ", raw, off + 8)
        off += 64 + 8 * ((alloc + 63) // 64) + 8 * filled
    p_off = off
    P, = struct.unpack_from("<Q", raw, off)
    off += 8
    pats = []
    while off < len(raw) and len(pats) < P:
        bs, = struct.unpack_from("<Q", raw, off)
        off += 8
        end = off + bs
        while off < end:
            bits, = struct.unpack_from("<I", raw, off + 28)
            size = 40 + 16 * ((bits + 127) // 128)
            pats.append((off, size))
            off += size
    return p_off, P, pats


@pytest.mark.parametrize("stem", ["virus_k18", "clade64", "virus_k18_f01"])
def test_db_reader_takes_the_pattern_blocks_side_by_side(K, golden_dir, tmp_path, monkeypatch, stem):
    """The reader finds the pattern blocks header to header and parses them on several threads.  The same database re-written with
    many small blocks (3, 1 and 7 patterns per block in turn, an empty block in between: any split is a valid file,
    prefix_kmer_db.cpp:544-571 / 700-748) must give the same arrays on 1, 2 and 16 threads, with and without the hashtables."""
    import struct
    raw = open(os.path.join(golden_dir, stem + ".db"), "rb").read()
    p_off, P, pats = _pattern_section(raw)
    assert len(pats) == P and P > 50
    out = bytearray(raw[:p_off + 8])
    i, sizes, k = 0, (3, 1, 7), 0
    while i < P:
        take = pats[i: i + sizes[k % 3]]
        blk = b"".join(raw[o: o + s] for o, s in take)
        out += struct.pack("<Q", len(blk)) + blk
        if k == 4:
            out += struct.pack("<Q", 0)                       # an empty block
        i += len(take)
        k += 1
    p2 = str(tmp_path / "reblocked.db")
    open(p2, "wb").write(bytes(out))
    ref = K.HostDB(os.path.join(golden_dir, stem + ".db")).view_arrays()
    for threads in ("1", "2", "16"):
        monkeypatch.setenv("KMDB_LOAD_THREADS", threads)
        for skip in (False, True):
            h = K.HostDB(p2, skip_hashtables=skip)
            v = h.view_arrays()
            for key, a in ref.items():
                if skip and key in ("bucket_offset", "slots", "n_buckets"):
                    continue
                assert np.array_equal(v[key], a), (key, threads)
            assert h.pattern_section_bytes == sum(s for _, s in pats)
            h.close()
    # damaged pattern sections are refused, wherever the damage is
    good = bytes(out)
    last_off = len(good) - pats[-1][1]
    cases = {
        "truncated inside the last pattern": good[:-9],
        "the last block is missing": good[:len(raw[:p_off + 8]) + 8 + sum(s for _, s in pats[:3])],
        # (announcing one pattern less is only noticed when the surplus pattern shares a block with announced ones: the reader, like the
        # reference, stops taking blocks once it has P patterns)
        "more patterns in a block than announced": good[:p_off] + struct.pack("<Q", 2) + good[p_off + 8:],
        "a stream longer than its block": good[:last_off + 28] + struct.pack("<I", 1 << 20) + good[last_off + 32:],
    }
    for what, img in cases.items():
        p3 = str(tmp_path / "bad.db")
        open(p3, "wb").write(img)
        for threads in ("1", "16"):
            monkeypatch.setenv("KMDB_LOAD_THREADS", threads)
            with pytest.raises(K.KmdbError, match="Cannot open k-mer database"):
                K.HostDB(p3, skip_hashtables=True)
                pytest.fail(what)
    # a hashtable item that points past the pattern table
    if stem == "clade64":
        img = bytearray(raw)
        off = 8 + 4 + 8 + 8 + 4 + 1 + 8
        n, = struct.unpack_from("<Q", raw, off)
        off += 8
        for _ in range(n):
            ln, = struct.unpack_from("<Q", raw, off + 8)
            off += 16 + ln
        off += 8
        while True:                                            # first non-empty table: its first item's value
            filled, alloc = struct.unpack_from("<QQ", raw, off + 8)
            items = off + 64 + 8 * ((alloc + 63) // 64)
            if filled:
                struct.pack_into("<i", img, items + 4, P + 5)
                break
            off = items
        p4 = str(tmp_path / "badht.db")
        open(p4, "wb").write(bytes(img))
        with pytest.raises(K.KmdbError, match="points past the pattern table"):
            K.HostDB(p4)
        K.HostDB(p4, skip_hashtables=True).close()            # the all2all modes never look at the tables


def test_kmer_extraction_matches_oracle(K, O, golden_dir):
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGTacgtNnUuRYX-", dtype=np.uint8)
    for k, frac in [(18, 1.0), (18, 0.1), (21, 1.0), (24, 1.0), (25, 0.1), (16, 1.0), (12, 1.0), (31, 0.5)]:
        for _ in range(6):
            n = int(rng.integers(0, 3000))
            p = np.array([10] * 8 + [1] * 8, dtype=float)
            seq = bytes(alphabet[rng.choice(16, size=n, p=p / p.sum())])
            for preserve in (False, True):
                a = K.extract_kmers(seq, k, frac, 0.0, preserve)
                b = O.extract_seq(seq, k, frac, 0.0, preserve)
                assert np.array_equal(a, b)
            assert np.array_equal(K.sort_unique(a), O.sort_unique(a))
    # short and empty inputs
    assert K.extract_kmers(b"ACGT", 18).size == 0 and K.extract_kmers(b"", 18).size == 0
    # a real genome from the reference's fixtures
    raw = open(os.path.join(golden_dir, "test", "virus", "data", "MN908947.fasta"), "rb").read()
    seq = b"".join(raw.split(b"\n")[1:])
    assert np.array_equal(K.extract_kmers(seq, 18), O.extract_seq(seq, 18))
    # k=18 words are 40 bits wide: 256 prefix buckets (kmer_extract.h:37-45)
    assert int(K.extract_kmers(seq, 18).max()) >> 32 < 256


def test_kmer_extraction_over_every_alphabet_matches_oracle(K, O):
    """kmdbh_extract_kmers_alphabet (the front-end's query loader for databases over the protein alphabets, reference src/alphabet.h:79-126,
    src/kmer_extract.h:13-97) == the oracle's restatement: the records of test/protein/aa_100x1000.fasta and random text with invalid
    letters, every alphabet, several k (n-bit symbols; k beyond 64 / bits - 1 is refused, alphabet.h:37), minhash fractions."""
    import lzma
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with lzma.open(os.path.join(here, "protein.aa_100x1000.fasta.xz")) as f:
        recs = O._split_records(f.read())
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdefghiklmnpqrstvwyBJOUXZ.*-", np.uint8)
    rnd = [bytes(letters[rng.integers(0, letters.size, size=int(n))]) for n in (0, 5, 9, 300, 4000)]
    bits = {"nt": 2, "nt-preserve": 2, "aa": 5, "aa11_diamond": 4, "aa12_mmseqs": 4, "aa6_dayhoff": 3}
    for a, name in enumerate(K.ALPHABETS):
        assert name in O.ALPHABETS
        for k, frac in ((3, 1.0), (7, 1.0), (8, 1.0), (8, 0.25), (64 // bits[name] - 1, 1.0)):
            for s in [r for _, r in recs[:12]] + rnd:
                got = K.extract_kmers_alphabet(s, k, a, frac)
                assert np.array_equal(got, O.extract_seq_alphabet(s, k, name, frac)), (name, k, frac, len(s))
                assert np.array_equal(got, K.extract_kmers_alphabet(s, k, name, frac))
        # a k the alphabet's symbols do not fit 63 bits with
        assert K.extract_kmers_alphabet(recs[0][1], 64 // bits[name], a).size == 0
    # the nucleotide entry points are the alphabets 0 and 1
    seq = bytes(np.frombuffer(b"ACGTacgtNU", np.uint8)[rng.integers(0, 10, size=3000)])
    assert np.array_equal(K.extract_kmers(seq, 18), K.extract_kmers_alphabet(seq, 18, "nt"))
    assert np.array_equal(K.extract_kmers(seq, 18, preserve_strand=True), K.extract_kmers_alphabet(seq, 18, "nt-preserve"))
    assert K.extract_kmers_alphabet(seq, 18, 6).size == 0 and K.extract_kmers_alphabet(seq, 18, -1).size == 0      # unknown alphabet


def test_csv_formatting_matches_oracle(K, O, golden_dir):
    path = os.path.join(golden_dir, "virus_k18.db")
    h = K.HostDB(path, skip_hashtables=True)
    o = O.OracleDB(path, skip_hashtables=True)
    m = o.all2all_dense()
    txt = K.format_header(h)
    for i in range(h.N):
        txt += K.format_dense_row(h.names[i], h.sample_kmers[i], O.tri_row(m, i))
    assert txt == open(os.path.join(golden_dir, "virus.k18.csv"), "rb").read()
    txt = K.format_header(h)
    for i in range(h.N):
        row = O.tri_row(m, i)
        nz = np.nonzero(row)[0]
        txt += K.format_sparse_row(h.names[i], h.sample_kmers[i], nz, row[nz])
    assert txt == open(os.path.join(golden_dir, "virus.k18.sparse.csv"), "rb").read()
    assert K.format_dense_row("x", 4294967295, np.array([0, 4294967295, 10], np.uint32)) == b"x,4294967295,0,4294967295,10,\n"


def test_gpu_entry_points_fail_loudly_without_a_device(K, golden_dir):
    if K.device_count() > 0:
        pytest.skip("a GPU is present")
    h = K.HostDB(os.path.join(golden_dir, "synth_k21.db"))
    with pytest.raises(K.KmdbError, match="no HIP device|no CPU fallback"):
        K.DeviceDB(h)
    exe = os.path.join(ROOT, "kmer-db_amd", "bin", "kmer-db-amd")
    r = subprocess.run([exe, "all2all", os.path.join(golden_dir, "synth_k21.db"), os.path.join(golden_dir, "o.csv")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR:" in r.stderr


def test_unknown_flag_bits_are_rejected(K, golden_dir):
    """kmdb_opts.flags: only the documented bits are accepted (the library used to read timing experiments from the high
    bits); the check comes before any device work, so it holds on a box without a GPU too."""
    h = K.HostDB(os.path.join(golden_dir, "synth_k21.db"))
    for bad in (1 << 8, 1 << 9, 1 << 13, 32, 0x80000000):
        with pytest.raises(K.KmdbError, match="unknown bits"):
            K.DeviceDB(h, flags=bad)
    # the shard arguments of the prefix-shard upload are checked up front as well
    with pytest.raises(K.KmdbError, match="shard_index"):
        K.DeviceDB(h, prefix_shard=(2, 2))
    # and so is the sample count: ids take 20 bits in the device layout (the reference has 32-bit ids, src/types.h:15-18; 2^20 samples
    # would be a 2 TB matrix)
    z = np.zeros(1, np.int64)
    for n, ok in ((1 << 20, False), ((1 << 20) - 1, True)):
        view = K.make_view(18, n, z, z - 1, z.astype(np.uint32), z.astype(np.uint32), z.astype(np.uint32), z.astype(np.uint32), z.astype(np.uint64), z.astype(np.uint64))
        if ok and K.device_count() > 0:
            continue                                            # (with a GPU the upload would go on and build a database of 2^20 - 1 empty samples)
        with pytest.raises(K.KmdbError, match="no HIP device" if ok else "1048576 samples or more"):
            K.DeviceDB(view)


def test_hashtable_headers_are_validated(K, golden_dir, tmp_path):
    """a capacity that is not a power of two, or a table without an empty slot, would hang or misdirect the device probes
    (reference src/hashmap_lp.h:150,308-333,427-437): the reader refuses such files"""
    import struct
    raw = bytearray(open(os.path.join(golden_dir, "synth_k21.db"), "rb").read())
    # header: formatWord u64, k u32, fraction f64, startFraction f64, alphabet i32, isInit u8, kmersCount u64 | N u64, samples | buckets
    off = 8 + 4 + 8 + 8 + 4 + 1 + 8
    n = struct.unpack_from("<Q", raw, off)[0]
    off += 8
    for _ in range(n):
        off += 8
        ln = struct.unpack_from("<Q", raw, off)[0]
        off += 8 + ln
    off += 8                                             # bucket count; first table: f64 maxFill, u64 filled, u64 allocated
    filled, allocated = struct.unpack_from("<QQ", raw, off + 8)
    assert allocated and allocated & (allocated - 1) == 0 and filled < allocated
    for bad_alloc in (allocated - 1, 0):
        b = bytearray(raw)
        struct.pack_into("<Q", b, off + 16, bad_alloc)
        p = str(tmp_path / "bad.db")
        open(p, "wb").write(b)
        with pytest.raises(K.KmdbError, match="hashtable"):
            K.HostDB(p)


def test_cli_usage_and_open_errors(golden_dir):
    exe = os.path.join(ROOT, "kmer-db_amd", "bin", "kmer-db-amd")
    r = subprocess.run([exe, "all2all", "/nonexistent.db", os.path.join(golden_dir, "o2.csv")], capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR: Cannot open k-mer database /nonexistent.db" in r.stderr
    r = subprocess.run([exe, "all2all", "onlyone"], capture_output=True, text=True)
    assert r.returncode != 0 and "USAGE" in r.stderr


DISTANCE_CASES = [
    # the reference's own distance tests (.github/workflows/self-hosted.yml:146-234, main.yml:99-110): arguments, input, golden
    (["mash"], "synth.a2a", "synth.a2a.mash"),
    (["ani"], "synth.a2a", "synth.a2a.ani"),
    (["-sparse", "ani"], "synth.a2a", "synth.a2a-sparse.ani"),
    (["-sparse", "-max", "1.0", "-min", "-1.0", "mash"], "synth.a2a", "synth.a2a-sparse.mash"),
    (["mash"], "synth.a2a-sparse", "synth.a2a-sparse.mash"),
    (["ani"], "synth.a2a-sparse", "synth.a2a-sparse.ani"),
    (["-sparse", "mash", "-min", "0.03", "-max", "mash:1.0"], "synth.a2a-sparse", "synth.a2a.mash.above-below"),
    (["-sparse", "-min", "0.03", "-max", "mash:1.0", "-min", "num-kmers:36", "mash"], "synth.a2a", "synth.a2a.mash-sparse-min2max"),
    (["mash"], "synth.n2a", "synth.n2a.mash"),
    (["ani"], "synth.n2a", "synth.n2a.ani"),
    (["-sparse", "ani"], "synth.n2a", "synth.n2a-sparse.ani"),
    (["-sparse", "-max", "1.0", "-min", "-1.0", "mash"], "synth.n2a", "synth.n2a-sparse.mash"),
    (["mash"], "synth.n2a-sparse", "synth.n2a-sparse.mash"),
    (["ani"], "synth.n2a-sparse", "synth.n2a-sparse.ani"),
    (["cosine"], "virus.k18.csv", "virus.k18.csv.cosine"),
    (["jaccard"], "virus.k18.csv", "virus.k18.csv.jaccard"),
    (["mash"], "virus.k18.csv", "virus.k18.csv.mash"),
    (["max"], "virus.k18.csv", "virus.k18.csv.max"),
    (["min"], "virus.k18.csv", "virus.k18.csv.min"),
]


@pytest.mark.parametrize("case", DISTANCE_CASES, ids=lambda c: c[2] + ":" + "_".join(c[0]))
def test_cli_distance_matches_reference_goldens(golden_dir, tmp_path, case):
    """`distance` (console_distance.cpp:7-213): byte-identical with every golden the reference's workflows compare."""
    opts, src, want = case
    out = tmp_path / "out"
    r = subprocess.run([os.path.join(ROOT, "kmer-db_amd", "bin", "kmer-db-amd"), "distance"] + opts + [os.path.join(golden_dir, src), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == open(os.path.join(golden_dir, want), "rb").read()


def test_cli_distance_errors(golden_dir, tmp_path):
    exe = os.path.join(ROOT, "kmer-db_amd", "bin", "kmer-db-amd")
    src = os.path.join(golden_dir, "synth.a2a")
    for argv in (["distance", src, str(tmp_path / "o")],                                   # no measure: the first file name is taken for one
                 ["distance", "-min", "nosuch:1", "mash", src, str(tmp_path / "o")],
                 ["distance", "mash", str(tmp_path / "missing"), str(tmp_path / "o")]):
        r = subprocess.run([exe] + argv, capture_output=True, text=True)
        assert r.returncode != 0 and ("ERROR" in r.stderr or "USAGE" in r.stderr)


def test_metrics_match_the_reference_formulas(K):
    """kmdbh_metric / kmdbh_metric_id (params.cpp:14-42): names, uint32 wrap-around integer parts, libm log."""
    import math
    L = K.capi.lib()
    names = K.capi.METRICS
    assert [L.kmdbh_metric_id(n.encode()) for n in names] == list(range(len(names))) and L.kmdbh_metric_id(b"nosuch") == -1
    rng = np.random.default_rng(3)
    u32 = lambda x: x & 0xFFFFFFFF        # noqa: E731

    def mash(j, k):
        return 1.0 if j == 0 else (-1.0 / k) * math.log((2 * j) / (j + 1))
    for _ in range(300):
        a, b = int(rng.integers(1, 1 << 22)), int(rng.integers(1, 1 << 22))
        c = int(rng.integers(1, min(a, b) + 1))
        k = int(rng.integers(10, 31))
        want = {"jaccard": c / u32(a + b - c), "min": c / min(a, b), "max": c / max(a, b), "cosine": c / math.sqrt(u32(a * b)),
                "mash": mash(c / u32(a + b - c), k), "ani": 1.0 - mash(c / u32(a + b - c), k), "ani-shorter": 1.0 - mash(c / min(a, b), k),
                "mash-query": mash(c / a, k), "num-kmers": float(c)}
        for i, n in enumerate(names):
            assert L.kmdbh_metric(i, c, a, b, k) == want[n], (n, c, a, b, k)
    # products that wrap in uint32, as in the reference's num_kmers_t arithmetic
    assert L.kmdbh_metric(names.index("cosine"), 5, 70000, 70000, 18) == 5 / math.sqrt(u32(70000 * 70000))


def test_filtered_sparse_call_validates_its_arguments(K):
    """kmdb_all2all_sparse_filtered rejects bad filter lists before it touches a device."""
    import ctypes as C
    L = K.capi.lib()
    F = K.capi._CellFilter
    raw = K.capi._Sparse()
    cnt = np.ones(4, np.uint32)
    one = (F * 1)(); one[0].metric = 0; one[0].lo = 0.0; one[0].hi = 1.0
    bad = (F * 1)(); bad[0].metric = 99
    many = (F * 13)()
    cases = [((None, 1, cnt.ctypes.data, -1), "null argument"), ((one, 1, None, -1), "null argument"), ((None, 0, None, 3), "null argument"),
             ((one, 1, cnt.ctypes.data, 99), "unknown measure"), ((many, 13, cnt.ctypes.data, -1), "more than 12"),
             ((bad, 1, cnt.ctypes.data, -1), "unknown metric")]
    for (fl, n, counts, measure), msg in cases:
        rc = L.kmdb_all2all_sparse_filtered(None, fl, n, counts, measure, C.byref(raw), None)
        assert rc != 0 and msg in L.kmdb_last_error().decode(), (msg, L.kmdb_last_error())
    # every criterion of the reference plus num-kmers (ten bounds) is accepted by the argument check (the call then fails on the null handle)
    ten = (F * 10)()
    for i in range(10):
        ten[i].metric = i % 9; ten[i].lo = 0.0; ten[i].hi = 1.0
    rc = L.kmdb_all2all_sparse_filtered(None, ten, 10, cnt.ctypes.data, -1, C.byref(raw), None)
    assert rc != 0 and "null argument" in L.kmdb_last_error().decode()
    # the compaction of caller-accumulated cells validates the same way, before it touches a device
    rc = L.kmdb_sparse_from_dense_device(None, None, 0, 0, None, 0, None, -1, C.byref(raw), None)
    assert rc != 0 and "null argument" in L.kmdb_last_error().decode()
