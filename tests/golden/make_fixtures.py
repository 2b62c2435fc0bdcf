#!/usr/bin/env python3
"""Regenerates tests/golden/ — run in the build container only (needs /root/reference).

What it does (everything it writes is DATA: inputs and expected outputs, never source):
  1. copies the reference's own golden outputs for the hot path (test/virus/k18*.csv,
     test/synth/{a2a,a2a-sparse,n2a,n2a-sparse}, ...) and its tiny test inputs
     (test/synth/synth.fa; the 165 test/virus/data FASTA files packed as virus_data.tar.xz);
  2. extracts k-mers with the ORACLE's restatement of kmer_extract.h, builds .db files with
     the REAL reference (oracle/_ref/ref_driver build -> PrefixKmerDb::addKmers + serialize),
     and stores them (virus_k18.db, virus_k18_part1.db, virus_k24.db, virus_k18_f01.db,
     synth_k21.db, clade64.db);
  3. runs the REAL reference all2all / all2all_sp / one2all on them and stores the raw
     outputs as *.ref.* vectors.
tests/test_oracle_golden.py then pins the oracle against all of these without needing
/root/reference; the -m gpu tests pin the HIP path against the same files.
"""
import io
import os
import shutil
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference"


def copy(src, dst):
    shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
    os.chmod(os.path.join(HERE, dst), 0o644)


def build_db(samples, k, fraction, name, threads=4, alphabet="nt"):
    with tempfile.TemporaryDirectory() as td:
        kb = os.path.join(td, "k.bin")
        O.write_kmers_bin(kb, k, fraction, samples)
        out = os.path.join(HERE, name)
        info = O.ref_build(kb, out, threads, alphabet)
        print(name, info)
    return out


def ref_outputs(db, stem):
    with tempfile.TemporaryDirectory() as td:
        m, info = O.ref_all2all(db, os.path.join(td, "m.u32"), threads=4)
        m.tofile(os.path.join(HERE, stem + ".a2a.ref.u32"))
        txt, _ = O.ref_all2all_sp(db, os.path.join(td, "s.txt"), threads=4)
        with open(os.path.join(HERE, stem + ".a2a_sp.ref.txt"), "wb") as f:
            f.write(txt)
        # a low bubble threshold exercises CBubbleHelper (bubble_helper.h:79-152)
        txt2, _ = O.ref_all2all_sp(db, os.path.join(td, "s2.txt"), threads=4, bubble=20)
        assert txt2 == txt, "bubble path changed the reference's own output"


def clade_genomes(n_clades, per_clade, length, r1, r2, seed):
    rng = np.random.default_rng(seed)
    root = rng.integers(0, 4, length, dtype=np.uint8)
    out = []
    for c in range(n_clades):
        anc = root.copy()
        m = rng.random(length) < r1
        anc[m] = (anc[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
        for s in range(per_clade):
            g = anc.copy()
            m = rng.random(length) < r2
            g[m] = (g[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            out.append(("g%05d" % (c * per_clade + s), bytes(np.frombuffer(b"ACGT", np.uint8)[g])))
    return out


def protein_dna():
    """test/protein/{dna,dna-preserve}.a2a (self-hosted.yml:393-403): k = 24 all2all of the 100 records of dna_100x1000.fasta as samples
    (-multisample-fasta), canonical k-mers and -preserve-strand.  The k-mers come from the oracle's extractor, the databases from the
    real reference's addKmers + serialize; the goldens are the reference's own files."""
    copy("test/protein/dna.a2a", "protein.dna.a2a")
    copy("test/protein/dna-preserve.a2a", "protein.dna-preserve.a2a")
    recs = O._split_records(O._read_fasta_text(os.path.join(REF, "test/protein/dna_100x1000")))
    for preserve, name in ((False, "protein_dna_k24.db"), (True, "protein_dna_k24_preserve.db")):
        samples = [(h, O.sort_unique(O.extract_seq(s, 24, 1.0, 0.0, preserve))) for h, s in recs]
        db = build_db(samples, 24, 1.0, name)
        ref_outputs(db, name[:-3])


# test/protein/aa*.a2a (self-hosted.yml:404-427): `build -k 8 -multisample-fasta -alphabet <name>` of aa_100x1000.fasta + all2all; aa_k7.a2a is
# the same at k = 7 with the aa alphabet (a golden the workflow does not run)
PROTEIN_AA = (("aa", 8, "aa"), ("aa11_diamond", 8, "aa11_diamond"), ("aa12_mmseqs", 8, "aa12_mmseqs"), ("aa6_dayhoff", 8, "aa6_dayhoff"), ("aa_k7", 7, "aa"))


def protein_aa():
    """The protein goldens of test/protein: the 100 records of aa_100x1000.fasta as samples, k-mers over the amino-acid alphabets by the
    oracle's restatement of KmerHelper::extract (n-bit symbols, alphabet.h), the databases from the real reference's addKmers (told the
    alphabet: it sets the prefix bits) + serialize; the goldens are the reference's own files."""
    recs = O._split_records(O._read_fasta_text(os.path.join(REF, "test/protein/aa_100x1000")))
    for stem, k, alphabet in PROTEIN_AA:
        copy("test/protein/%s.a2a" % stem, "protein.%s.a2a" % stem)
        samples = [(h, O.sort_unique(O.extract_seq_alphabet(s, k, alphabet))) for h, s in recs]
        db = build_db(samples, k, 1.0, "protein_%s.db" % stem, alphabet=alphabet)
        ref_outputs(db, "protein_%s" % stem)


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
    if sys.argv[1:] == ["protein"]:
        protein_dna()
        protein_aa()
        print("done (protein only)")
        return
    if sys.argv[1:] == ["protein-aa"]:
        protein_aa()
        print("done (protein aa only)")
        return
    # ---- 1. reference goldens + inputs -------------------------------------------------
    for f in ["k18.csv", "k18.sparse.csv", "k18.frac.csv", "k24.csv", "k18.n2a.csv",
              "k18.n2a.sparse.csv", "k18.n2a.itself.csv"]:
        copy("test/virus/" + f, "virus." + f)
    for f in ["a2a", "a2a-sparse", "n2a", "n2a-sparse", "a2a.sparse.above-below", "n2a.sparse.above-below", "synth.fa"]:
        copy("test/synth/" + f, "synth." + f)
    for f in ["seqs.list", "seqs.part1.list", "seqs.part2.list"]:
        copy("test/virus/" + f, "virus." + f)
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w:xz", preset=9) as tf:
        d = os.path.join(REF, "test/virus/data")
        for fn in sorted(os.listdir(d)):
            if fn.endswith(".fasta") and fn != "seqs.fasta":
                tf.add(os.path.join(d, fn), arcname="data/" + fn)
    with open(os.path.join(HERE, "virus_data.tar.xz"), "wb") as f:
        f.write(buf.getvalue())
    print("virus_data.tar.xz", len(buf.getvalue()), "bytes")

    # ---- 2. databases built by the real reference --------------------------------------
    os.chdir(REF)      # list entries are relative to the reference root (./test/virus/data/...)
    full = O.load_samples("test/virus/seqs.list", 18)
    p1 = O.load_samples("test/virus/seqs.part1.list", 18)
    p2 = O.load_samples("test/virus/seqs.part2.list", 18)
    db_full = build_db(full, 18, 1.0, "virus_k18.db")
    db_p1 = build_db(p1, 18, 1.0, "virus_k18_part1.db")
    build_db(p1 + p2, 18, 1.0, "virus_k18_parts.db")          # build + extend order (main.yml:128-133)
    db_k24 = build_db(O.load_samples("test/virus/seqs.list", 24), 24, 1.0, "virus_k24.db")
    db_f01 = build_db(O.load_samples("test/virus/seqs.list", 18, 0.1), 18, 0.1, "virus_k18_f01.db")
    with tempfile.TemporaryDirectory() as td:
        lst = os.path.join(td, "synth.list")
        shutil.copyfile(os.path.join(REF, "test/synth/synth.fa"), os.path.join(td, "synth.fa"))
        with open(lst, "w") as f:
            f.write(os.path.join(td, "synth") + "\n")
        synth = O.load_samples(lst, 21, multisample=True)
    db_synth = build_db(synth, 21, 1.0, "synth_k21.db")
    # a small clade-structured synthetic set (the bench's genome model at toy size)
    cl_all = clade_genomes(4, 18, 20000, 0.10, 0.01, 20260928)
    cl = [g for i, g in enumerate(cl_all) if i % 18 < 16]
    cl = [("g%05d" % i, s) for i, (_, s) in enumerate(cl)]
    cl_q = [g for i, g in enumerate(cl_all) if i % 18 >= 16]
    cl_s = [(n, O.sort_unique(O.extract_seq(s, 18))) for n, s in cl]
    db_cl = build_db(cl_s, 18, 1.0, "clade64.db")
    # k=25 f=0.1 minhash DB (config-4 shaped)
    cl25 = [(n, O.sort_unique(O.extract_seq(s, 25, 0.1))) for n, s in cl]
    db_cl25 = build_db(cl25, 25, 0.1, "clade64_k25_f01.db")

    # ---- 3. reference outputs ----------------------------------------------------------
    for db, stem in [(db_full, "virus_k18"), (db_p1, "virus_k18_part1"), (db_k24, "virus_k24"),
                     (db_f01, "virus_k18_f01"), (db_synth, "synth_k21"), (db_cl, "clade64"), (db_cl25, "clade64_k25_f01")]:
        ref_outputs(db, stem)
    # new2all: part2 queries vs the part1 db (main.yml:73-81)
    with tempfile.TemporaryDirectory() as td:
        qb = os.path.join(td, "q.bin")
        O.write_kmers_bin(qb, 18, 1.0, p2)
        rows, _ = O.ref_one2all(db_p1, qb, os.path.join(td, "o.u32"), threads=1)
        rows.tofile(os.path.join(HERE, "virus_k18_part1.n2a_part2.ref.u32"))
        txt, _ = O.ref_one2all_sp(db_p1, qb, os.path.join(td, "o.txt"), threads=1)
        with open(os.path.join(HERE, "virus_k18_part1.n2a_part2_sp.ref.txt"), "wb") as f:
            f.write(txt)
        # clade queries: fresh strains vs clade64
        qs = [("q%d" % i, O.extract_seq(s, 18)) for i, (_, s) in enumerate(cl_q)]   # strains 16,17 of each clade: not in the db
        O.write_kmers_bin(qb, 18, 1.0, qs)
        np.savez_compressed(os.path.join(HERE, "clade64.queries.npz"), **{n: km for n, km in qs})
        rows, _ = O.ref_one2all(db_cl, qb, os.path.join(td, "o2.u32"), threads=1)
        rows.tofile(os.path.join(HERE, "clade64.n2a.ref.u32"))
    protein_dna()
    protein_aa()
    print("done")


if __name__ == "__main__":
    main()
