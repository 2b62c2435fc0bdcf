#!/usr/bin/env python3
"""Expected outputs of `all2all-sp -sample-rows` (reference src/sampler.h, src/array.h:450-540, src/console_all2all_sparse.cpp:70-89), made by the
REAL reference: oracle/_ref/bridge_driver sample_rows_ref runs the reference's own Params::parse, all2all_sp, SparseMatrix::add_to_sampler and
Sampler::saveRowSparse on the golden databases (the rows' sparse parts, one line per sample).  Only the criterion ("best") strategy is kept as a
fixture: the random strategy's subsets depend on the order in which the reference's hash tables list a row.

    python tests/golden/make_fixture_sample_rows.py        (where /root/reference exists and `make -C oracle` has run)
"""
import lzma
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "bridge_driver")
CASES = [("virus_k18", "sr_jaccard5", ["-sample-rows", "jaccard:5"]),
         ("virus_k18", "sr_numkmers3_min", ["-sample-rows", "num-kmers:3", "-min", "jaccard:0.02"]),
         ("clade64", "sr_ani7", ["-sample-rows", "ani:7"]),
         ("clade64", "sr_max2", ["-sample-rows", "max:2", "-max", "num-kmers:3000"])]


def main():
    assert os.path.exists(EXE), "build oracle/_ref first (make -C oracle)"
    with tempfile.TemporaryDirectory() as td:
        for stem, tag, opts in CASES:
            db = os.path.join(td, stem + ".db")
            if not os.path.exists(db):
                with lzma.open(os.path.join(HERE, stem + ".db.xz")) as f, open(db, "wb") as o:
                    o.write(f.read())
            out = os.path.join(HERE, "%s.%s.ref.txt" % (stem, tag))
            subprocess.check_call([EXE, "sample_rows_ref", db, out] + opts)
            print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
