#!/usr/bin/env python3
"""Fixture for all2all-parts / db2db (reference .github/workflows/self-hosted.yml:355-362): the database of
test/virus/seqs.part2.list built by the REAL reference (oracle/_ref); part 1 is virus_k18_part1.db, the expected
output of the grid is the reference's own golden virus.k18.sparse.csv.
Run in the build container (needs /root/reference): python tests/golden/make_fixture_parts.py"""
import lzma
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_fixtures as MF          # noqa: E402
from oracle import oracle as O      # noqa: E402

assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
os.chdir(MF.REF)
p2 = O.load_samples("test/virus/seqs.part2.list", 18)
db = MF.build_db(p2, 18, 1.0, "virus_k18_part2.db")
with open(db, "rb") as f, lzma.open(db + ".xz", "wb", preset=9) as g:
    g.write(f.read())
os.remove(db)
print("virus_k18_part2.db.xz", os.path.getsize(db + ".xz"), "bytes")
