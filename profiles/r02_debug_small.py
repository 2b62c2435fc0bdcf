# debugging aid: one small database through the pipeline with a sync after every stage
import lzma, os, sys, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["KMDB_VERBOSE"] = "1"; os.environ["KMDB_SYNC_DEBUG"] = "1"
from _kmerdb_loader import import_kmerdb_amd
from oracle import oracle as O
K = import_kmerdb_amd()
stem = sys.argv[1] if len(sys.argv) > 1 else "clade64"
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, stem + ".db")
    with lzma.open(os.path.join("tests", "golden", stem + ".db.xz")) as f, open(path, "wb") as o:
        o.write(f.read())
    h = K.HostDB(path, skip_hashtables=True)
    d = K.DeviceDB(h, device=0)
    print("uploaded", d.P, d.N, flush=True)
    ref = d.all2all_dense(flags=K.capi.FLAG_FORCE_GLOBAL_ATOMICS)
    exp = O.OracleDB(path, skip_hashtables=True).all2all_dense()
    print("v1 global == oracle:", np.array_equal(ref, exp), flush=True)
    got = d.all2all_dense()
    print("records == oracle:", np.array_equal(got, exp), "diff cells", int((got != exp).sum()), d.stats(), flush=True)
