"""Pins the CPU oracle (oracle/kmdb_oracle.c) against the reference: its own golden CSVs
(test/virus, test/synth) and raw outputs of the real reference hot path (tests/golden/*.ref.*,
produced by oracle/_ref/ref_driver in tests/golden/make_fixtures.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import DBS


def _read(golden_dir, name):
    with open(os.path.join(golden_dir, name), "rb") as f:
        return f.read()


@pytest.mark.parametrize("stem", [d for d in DBS if d != "virus_k18_parts"])
def test_dense_matches_reference_raw(O, golden_dir, stem):
    db = O.OracleDB(os.path.join(golden_dir, stem + ".db"))
    ref = np.fromfile(os.path.join(golden_dir, stem + ".a2a.ref.u32"), dtype=np.uint32)
    m = db.all2all_dense()
    assert np.array_equal(m, ref)
    # flat form (all2all_sp semantics) gives the same cells, and the checksum identity holds
    assert np.array_equal(db.all2all_flat(), ref)
    assert int(m.astype(np.uint64).sum()) == db.update_counts()["sum_matrix"]


@pytest.mark.parametrize("stem,golden,sparse", [
    ("virus_k18", "virus.k18.csv", False), ("virus_k18_parts", "virus.k18.csv", False),
    ("virus_k18", "virus.k18.sparse.csv", True), ("virus_k24", "virus.k24.csv", False),
    ("virus_k18_f01", "virus.k18.frac.csv", False), ("synth_k21", "synth.a2a", False),
    ("synth_k21", "synth.a2a-sparse", True),
    # test/protein/{dna,dna-preserve}.a2a (self-hosted.yml:393-403): k = 24, the records of one FASTA as samples; canonical k-mers / -preserve-strand
    ("protein_dna_k24", "protein.dna.a2a", False), ("protein_dna_k24_preserve", "protein.dna-preserve.a2a", False),
    # test/protein/aa*.a2a (self-hosted.yml:404-427): k = 8 over the amino-acid alphabets of src/alphabet.h:79-86 (5 / 4 / 4 / 3 bits per symbol);
    # aa_k7.a2a: the aa alphabet at k = 7 (35 bits: the 8-bit-prefix rule of kmer_extract.h:37-45 applies)
    ("protein_aa", "protein.aa.a2a", False), ("protein_aa11_diamond", "protein.aa11_diamond.a2a", False),
    ("protein_aa12_mmseqs", "protein.aa12_mmseqs.a2a", False), ("protein_aa6_dayhoff", "protein.aa6_dayhoff.a2a", False),
    ("protein_aa_k7", "protein.aa_k7.a2a", False)])
def test_all2all_csv_matches_reference_golden(O, golden_dir, stem, golden, sparse):
    db = O.OracleDB(os.path.join(golden_dir, stem + ".db"))
    csv = O.format_all2all(db.k, db.fraction, db.names, db.sample_kmers, db.all2all_dense(), sparse=sparse)
    assert csv == _read(golden_dir, golden)


@pytest.mark.parametrize("stem", ["virus_k18", "clade64", "clade64_k25_f01", "synth_k21"])
def test_sparse_rows_match_reference_all2all_sp(O, golden_dir, stem):
    db = O.OracleDB(os.path.join(golden_dir, stem + ".db"))
    m = db.all2all_flat()
    lines = _read(golden_dir, stem + ".a2a_sp.ref.txt").split(b"\n")
    for i in range(db.N):
        row = O.tri_row(m, i)
        mine = "".join("%d:%d," % (j + 1, row[j]) for j in np.nonzero(row)[0]).encode()
        assert mine == lines[i]


def _virus_queries(O, golden_dir, lst):
    cwd = os.getcwd()
    os.chdir(golden_dir)
    try:
        return O.load_samples(os.path.join(golden_dir, lst), 18, unique=False)
    finally:
        os.chdir(cwd)


def test_new2all_matches_reference(O, golden_dir):
    db = O.OracleDB(os.path.join(golden_dir, "virus_k18_part1.db"))
    qs = _virus_queries(O, golden_dir, "virus.seqs.part2.list")
    rows, meta = [], []
    for name, km in qs:
        u = O.sort_unique(km)
        rows.append(db.one2all(u))
        meta.append((name, len(u)))
    ref = np.fromfile(os.path.join(golden_dir, "virus_k18_part1.n2a_part2.ref.u32"), dtype=np.uint32).reshape(len(qs), db.N)
    assert np.array_equal(np.stack(rows), ref)
    assert O.format_new2all(db.k, db.fraction, db.names, db.sample_kmers, meta, rows) == _read(golden_dir, "virus.k18.n2a.csv")
    assert O.format_new2all(db.k, db.fraction, db.names, db.sample_kmers, meta, rows, sparse=True) == _read(golden_dir, "virus.k18.n2a.sparse.csv")
    sp = _read(golden_dir, "virus_k18_part1.n2a_part2_sp.ref.txt").split(b"\n")
    for r, line in zip(rows, sp):
        assert "".join("%d:%d," % (j + 1, r[j]) for j in np.nonzero(r)[0]).encode() == line


def test_new2all_itself_golden(O, golden_dir):
    db = O.OracleDB(os.path.join(golden_dir, "virus_k18.db"))
    qs = _virus_queries(O, golden_dir, "virus.seqs.list")
    rows, meta = [], []
    for name, km in qs:
        u = O.sort_unique(km)
        rows.append(db.one2all(u))
        meta.append((name, len(u)))
    assert O.format_new2all(db.k, db.fraction, db.names, db.sample_kmers, meta, rows) == _read(golden_dir, "virus.k18.n2a.itself.csv")


def test_new2all_clade_queries(O, golden_dir):
    db = O.OracleDB(os.path.join(golden_dir, "clade64.db"))
    q = np.load(os.path.join(golden_dir, "clade64.queries.npz"))
    rows = np.stack([db.one2all(O.sort_unique(q[k])) for k in sorted(q.files, key=lambda s: int(s[1:]))])
    assert np.array_equal(rows, np.fromfile(os.path.join(golden_dir, "clade64.n2a.ref.u32"), dtype=np.uint32).reshape(-1, db.N))


def test_gamma_roundtrip_and_known_answers(O):
    # code shape (elias_gamma.h:104-128): 1 -> "0", 2 -> "100", 3 -> "101", 4 -> "11000", 5 -> "11001"
    words, nbits = O.gamma_encode([1, 2, 3, 4, 5])
    assert nbits == 1 + 3 + 3 + 5 + 5
    assert int(words[0]) >> (64 - nbits) == int("0" "100" "101" "11000" "11001", 2)
    rng = np.random.default_rng(7)
    for _ in range(50):
        vals = rng.integers(1, 2 ** rng.integers(1, 31), size=rng.integers(1, 200)).astype(np.uint32)
        w, nb = O.gamma_encode(vals)
        assert np.array_equal(O.gamma_decode(w, nb, vals.size + 1), vals)
    # a value whose code straddles a 64-bit word boundary
    vals = np.array([1] * 60 + [1000, 7, 1], dtype=np.uint32)
    w, nb = O.gamma_encode(vals)
    assert np.array_equal(O.gamma_decode(w, nb, vals.size), vals)


def test_chain_lists_strictly_increasing(O, golden_dir):
    db = O.OracleDB(os.path.join(golden_dir, "clade64.db"), skip_hashtables=True)
    for pid in range(0, db.P, 97):
        ids = db.decode_chain(pid).astype(np.int64)
        assert np.all(np.diff(ids) > 0)


def test_db2db_restatement_pinned_to_the_reference_matrix(O, golden_dir):
    """oracle db2db (db2db_sp restated) of part 2 x part 1 == the corresponding block of the reference's own all2all
    matrix of the union database (samples of seqs.list = part 1 followed by part 2)."""
    import os
    import numpy as np
    p1 = O.OracleDB(os.path.join(golden_dir, "virus_k18_part1.db"))
    p2 = O.OracleDB(os.path.join(golden_dir, "virus_k18_part2.db"))
    ref = np.fromfile(os.path.join(golden_dir, "virus_k18.a2a.ref.u32"), dtype=np.uint32)
    cross = p2.db2db(p1)
    for r in range(p2.N):
        assert np.array_equal(cross[r], O.tri_row(ref, p1.N + r)[: p1.N]), r
    assert np.array_equal(p1.db2db(p2), cross.T)


def test_db2db_restatement_equals_the_real_reference(O, golden_dir, tmp_path):
    """when oracle/_ref is built: the reference's own db2db_sp + compact2 (ref_driver db2db_sp) on the two virus parts"""
    import os
    import pytest
    if not O.have_ref():                        # (raises under KMDB_REQUIRE_REF=1)
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    a, b = os.path.join(golden_dir, "virus_k18_part2.db"), os.path.join(golden_dir, "virus_k18_part1.db")
    txt, _ = O.ref_db2db_sp(a, b, str(tmp_path / "o.txt"), threads=2)
    m = O.OracleDB(a).db2db(O.OracleDB(b))
    lines = txt.split(b"\n")
    for r in range(m.shape[0]):
        assert "".join("%d:%d," % (c + 1, v) for c, v in enumerate(m[r]) if v).encode() == lines[r]


@pytest.mark.parametrize("stem,k,alphabet", [("aa", 8, "aa"), ("aa11_diamond", 8, "aa11_diamond"), ("aa12_mmseqs", 8, "aa12_mmseqs"),
                                              ("aa6_dayhoff", 8, "aa6_dayhoff"), ("aa_k7", 7, "aa")])
def test_protein_goldens_from_the_definition(O, stem, k, alphabet):
    """test/protein/aa*.a2a (reference .github/workflows/self-hosted.yml:404-427) straight from the definition: the records of
    aa_100x1000.fasta as samples, k-mers by the oracle's restatement of KmerHelper::extract over the alphabet (src/kmer_extract.h:13-97,
    src/alphabet.h:22-86: n-bit symbols, '.' and every letter outside the groups invalidate the k-mers that hold them), cell (i, j) =
    |K_i ∩ K_j| — no database, pattern or tree involved.  Pins the extractor the protein fixtures were built with."""
    import lzma
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with lzma.open(os.path.join(here, "protein.aa_100x1000.fasta.xz")) as f:
        recs = O._split_records(f.read())
    sets = [O.sort_unique(O.extract_seq_alphabet(s, k, alphabet)) for _, s in recs]
    with open(os.path.join(here, "protein.%s.a2a" % stem)) as f:
        lines = f.read().split("\n")
    head = lines[0].split(",")
    assert head[0].startswith("kmer-length: %d " % k) and [h for h in head[2:] if h] == [h for h, _ in recs]
    assert [int(x) for x in lines[1].split(",")[2:] if x] == [int(s.size) for s in sets]
    for i, (name, _) in enumerate(recs):
        cells = lines[2 + i].split(",")
        assert cells[0] == name and int(cells[1]) == sets[i].size
        want = [int(x) for x in cells[2:] if x]
        assert want == [int(np.intersect1d(sets[i], sets[j], assume_unique=True).size) for j in range(i)], (stem, i)


def test_alphabet_extractor_is_the_nucleotide_one_on_nt(O):
    """kmo_extract_kmers_alphabet with the nt groups == kmo_extract_kmers (canonical and strand-preserving), invalid letters included"""
    rng = np.random.default_rng(3)
    seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNUu.", np.uint8), 5000, p=[.2, .2, .2, .2, .04, .04, .04, .04, .01, .01, .01, .01]))
    for k in (7, 16, 18, 24, 31):
        assert np.array_equal(O.extract_seq(seq, k), O.extract_seq_alphabet(seq, k, "nt"))
        assert np.array_equal(O.extract_seq(seq, k, preserve_strand=True), O.extract_seq_alphabet(seq, k, "nt-preserve"))
        assert np.array_equal(O.extract_seq(seq, k, 0.3), O.extract_seq_alphabet(seq, k, "nt", 0.3))
