import lzma
import os
import sys
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The parity tests against the oracle, the reference's raw outputs and its golden files run FIRST; the long tests — tens of thousands of
# samples (property checks: checksum identity, rows from the definition) and the bench.py contract tests (subprocesses) — run LAST, so that a
# box slow enough to hit the driver's step limit loses the least informative tests, not the parity proper.
LONG_LAST = ("test_baseline_sample_counts", "test_more_than_65535_samples", "test_db2db_large_parts", "test_new2all_thousand_queries",
             "test_randomised_stress", "test_bench_")


def pytest_collection_modifyitems(config, items):
    rank = lambda it: next((k + 1 for k, name in enumerate(LONG_LAST) if it.name.startswith(name)), 0)      # noqa: E731
    items.sort(key=rank)                                        # stable: file order inside every class


@pytest.fixture(scope="session")
def K():
    from _kmerdb_loader import import_kmerdb_amd
    return import_kmerdb_amd()


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def golden_dir(tmp_path_factory):
    """tests/golden with the .db.xz fixtures and the virus FASTA tarball unpacked into a temp dir."""
    d = tmp_path_factory.mktemp("golden")
    for fn in os.listdir(GOLDEN):
        src = os.path.join(GOLDEN, fn)
        if fn.endswith(".db.xz"):
            with lzma.open(src) as f, open(os.path.join(d, fn[:-3]), "wb") as o:
                o.write(f.read())
        elif fn == "virus_data.tar.xz":
            os.makedirs(os.path.join(d, "test", "virus"), exist_ok=True)
            with tarfile.open(src) as tf:
                tf.extractall(os.path.join(d, "test", "virus"))
        else:
            os.symlink(src, os.path.join(d, fn))
    return str(d)


DBS = ["virus_k18", "virus_k18_part1", "virus_k18_parts", "virus_k24", "virus_k18_f01", "synth_k21", "clade64",
       "clade64_k25_f01", "protein_dna_k24", "protein_dna_k24_preserve",
       "protein_aa", "protein_aa11_diamond", "protein_aa12_mmseqs", "protein_aa6_dayhoff", "protein_aa_k7"]
