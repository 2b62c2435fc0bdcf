#!/usr/bin/env python3
"""Fixture for the one2all mode (reference .github/workflows/main.yml:156-160): the k=25 f=0.1 database of
test/virus/seqs.part1.list built by the REAL reference (oracle/_ref) and the reference's own golden
test/virus/MT159713.csv.  Run in the build container (needs /root/reference): python tests/golden/make_fixture_one2all.py"""
import lzma
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_fixtures as MF          # noqa: E402
from oracle import oracle as O      # noqa: E402

assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
MF.copy("test/virus/MT159713.csv", "virus.MT159713.csv")
os.chdir(MF.REF)
p1 = O.load_samples("test/virus/seqs.part1.list", 25, 0.1)
db = MF.build_db(p1, 25, 0.1, "virus_k25_f01_part1.db")
with open(db, "rb") as f, lzma.open(db + ".xz", "wb", preset=9) as g:
    g.write(f.read())
os.remove(db)
print("virus_k25_f01_part1.db.xz", os.path.getsize(db + ".xz"), "bytes")
