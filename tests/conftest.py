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


# The compiled reference (oracle/_ref: ref_driver = the reference's own similarity_calculator etc., bridge_driver = INTEGRATION.md's glue inside
# it) is a git-ignored binary that travels to the GPU box with the snapshot.  Whether it is there decides how strong some tests are, so the log
# says so up front, names at the end every test branch that compared with it (or could not), and KMDB_REQUIRE_REF=1 — set by the job scripts
# under profiles/ — turns a missing reference build into a failure instead of a skip / an oracle-only comparison.
def _ref_state():
    ref = os.path.join(ROOT, "oracle", "_ref")
    return os.path.exists(os.path.join(ref, "ref_driver")), os.path.exists(os.path.join(ref, "bridge_driver"))


def pytest_report_header(config):
    r, b = _ref_state()
    return ["reference driver: %s; bridge driver: %s; KMDB_REQUIRE_REF=%s" %
            ("present" if r else "ABSENT", "present" if b else "ABSENT", os.environ.get("KMDB_REQUIRE_REF", "0"))]


def pytest_terminal_summary(terminalreporter):
    try:
        from oracle import oracle
    except Exception:
        return
    r, b = _ref_state()
    terminalreporter.write_line("reference driver: %s; bridge driver: %s" % ("present" if r else "ABSENT", "present" if b else "ABSENT"))
    ran = sorted(k for k, v in oracle.REF_BRANCHES.items() if v)
    missed = sorted(k for k, v in oracle.REF_BRANCHES.items() if not v)
    if ran:
        terminalreporter.write_line("compared with the real reference in: " + ", ".join(ran))
    if missed:
        terminalreporter.write_line("REFERENCE ABSENT, oracle only in: " + ", ".join(missed))


def require_ref_or_skip(path, what):
    """a test that cannot run without a reference build: skipped where the build is absent, FAILED under KMDB_REQUIRE_REF=1"""
    if os.path.exists(path):
        return
    if os.environ.get("KMDB_REQUIRE_REF", "") == "1":
        pytest.fail("KMDB_REQUIRE_REF=1: %s is missing (%s)" % (path, what))
    pytest.skip(what)


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
