"""Synthetic k-mer databases for bench.py and the tests (fixture generator, not the hot path).

Genome model "clade-mutation" (SURVEY.md §8d): a random root genome of length L; every clade
ancestor is the root with i.i.d. substitutions at rate r1; every strain is its clade ancestor
with substitutions at rate r2; samples are emitted clade by clade.  Randomness comes from a
counter-based hash (splitmix64 of seed/stream/position) so that a sample can be regenerated
anywhere — on the CPU for the small parity cases or on the GPU for the full-size bench — with
identical results, without storing genomes.

From the genomes the module derives exactly the database `kmer-db build` would write:
  * canonical 2-bit k-mers with the >=8-bit-prefix widening (reference src/kmer_extract.h:13-97)
    and optional minhash subsampling (src/filter.h:38-51,96-115);
  * the pattern tree of PrefixKmerDb::addKmers (src/prefix_kmer_db.cpp:181-240,244-434): samples are
    added in order; the k-mers of a sample are grouped by their current pattern; a group that
    takes ALL k-mers of a childless pattern extends that pattern in place, any other group
    becomes a new child pattern;
  * Elias-gamma coded local id lists (src/pattern.h:195-203, src/elias_gamma.h:104-128).
All steps are vectorised torch ops (sort / unique / searchsorted / scatter), so the same code runs
on CPU tensors and on an MI355X; tests/test_synth.py checks the result against a database built
by the real reference from the same k-mers.  The generator is plumbing around the engine: the
engine itself only ever sees the resulting kmdb_db_view.
"""
import math
import struct
import sys

import numpy as np
import torch

_M64 = (1 << 64) - 1


def _i64(x):
    """python int (0..2^64) -> the same bit pattern as a signed 64-bit python int"""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(x, s):
    """logical shift right on int64 tensors"""
    if s == 0:
        return x
    return (x >> s) & ((1 << (64 - s)) - 1)


def _splitmix(x):
    x = x + _i64(0x9E3779B97F4A7C15)
    x = (x ^ _lsr(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _i64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _stream(seed, stream_id, L, device):
    pos = torch.arange(L, dtype=torch.int64, device=device)
    return _splitmix(pos ^ _i64((seed * 0x100000001B3 + stream_id * 0x9E3779B1) << 20))


class CladeGenomes:
    """Regenerable genomes: sample(i) -> uint8 tensor of base codes (A,C,G,T = 0..3)."""

    def __init__(self, n_samples, clade_size, length, r1=0.10, r2=0.01, seed=20260928, device="cpu"):
        self.n_samples, self.clade_size, self.L = n_samples, clade_size, length
        self.r1, self.r2, self.seed, self.device = r1, r2, seed, torch.device(device)
        self.root = (_lsr(_stream(seed, 0, length, self.device), 13) & 3).to(torch.uint8)
        self._clade_cache = (-1, None)

    def _mutate(self, base, stream_id, rate):
        h = _stream(self.seed, stream_id, self.L, self.device)
        hit = (_lsr(h, 40) & 0xFFFFFF) < int(rate * (1 << 24))
        shift = (_lsr(h, 8) % 3 + 1).to(torch.uint8)
        return torch.where(hit, (base + shift) & 3, base)

    def clade_ancestor(self, c):
        if self._clade_cache[0] != c:
            self._clade_cache = (c, self._mutate(self.root, 1 + c, self.r1))
        return self._clade_cache[1]

    def sample(self, i):
        c = i // self.clade_size
        return self._mutate(self.clade_ancestor(c), 1_000_003 + i, self.r2)

    def strain(self, c, stream_id):
        """a fresh strain of clade c (a query genome that is not in the collection): stream_id >= n_samples"""
        return self._mutate(self.clade_ancestor(c), 1_000_003 + stream_id, self.r2)

    def name(self, i):
        return "g%05d" % i

    def fasta(self, i):
        codes = self.sample(i).cpu().numpy()
        return ">%s\n%s\n" % (self.name(i), np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode())


_WINDOWS_AT_ONCE = 1 << 16       # kmers_of: genomes up to this many k-mer windows take the (n, k) form
_WEIGHTS = {}                    # (k, device) -> 4^j, j < k


def kmers_of(codes, k, fraction=1.0, start_fraction=0.0, prefix_shard=None):
    """Sorted, duplicate-free k-mer words of one genome (torch int64 holding the uint64 bit pattern;
    all words are < 2^62 so signed order == unsigned order).  prefix_shard=(index, count): only the k-mers whose prefix
    bucket (kmer >> 32) is congruent to index modulo count (dropped before the sort)."""
    L = codes.numel()
    if L < k:
        return torch.zeros(0, dtype=torch.int64, device=codes.device)
    b = codes.to(torch.int64)
    n = L - k + 1
    fwd = rc = None
    if n <= _WINDOWS_AT_ONCE and k <= 31:
        # short genomes (the tests' collections of 10 000 - 70 000 samples): all windows as one (n, k) view, two products and two sums
        # instead of 6 k launches — the generator's time there is launch overhead (profiles/r05_close_test_phases.txt)
        try:
            w = _WEIGHTS.get((k, codes.device))
            if w is None:
                w = _WEIGHTS[(k, codes.device)] = torch.tensor([1 << (2 * j) for j in range(k)], dtype=torch.int64, device=codes.device)
            U = b.unfold(0, k, 1)
            fwd = (U * w.flip(0)).sum(1)
            rc = ((3 - U) * w).sum(1)
        except RuntimeError:                      # (an operator the device lacks: the loop form below)
            fwd = rc = None
    if fwd is None:
        fwd = torch.zeros(n, dtype=torch.int64, device=codes.device)
        rc = torch.zeros(n, dtype=torch.int64, device=codes.device)
        for j in range(k):
            seg = b[j: j + n]
            fwd = (fwd << 2) | seg
            rc = rc | ((3 - seg) << (2 * j))
    can = torch.minimum(fwd, rc)
    prefix_bits = 2 * k - 32
    if prefix_bits < 8:
        s = 8 - prefix_bits
        can = (can << s) | (can & ((1 << s) - 1))
    if fraction < 1.0:
        kd4 = int(math.ceil(k / 4))
        h = can * _i64(0x87c37b91114253d5)
        h = (h << 31) | _lsr(h, 33)
        h = h * _i64(0x4cf5ad432745937f)
        h1 = (h ^ 42) ^ kd4
        h2 = torch.full_like(h1, 42 ^ kd4)
        h1 = h1 + h2
        h2 = h2 + h1

        def fmix(x):
            x = (x ^ _lsr(x, 33)) * _i64(0xff51afd7ed558ccd)
            x = (x ^ _lsr(x, 33)) * _i64(0xc4ceb9fe1a85ec53)
            return x ^ _lsr(x, 33)
        h1, h2 = fmix(h1), fmix(h2)
        h1 = h1 + h2
        h2 = h2 + h1
        hv = h1 ^ h2
        lo = int(float(_M64) * start_fraction)
        hi = int(float(_M64) * (start_fraction + fraction))
        sign = _i64(1 << 63)
        keep = ((hv ^ sign) >= _i64(lo ^ (1 << 63))) & ((hv ^ sign) < _i64(min(hi, _M64) ^ (1 << 63)))
        can = can[keep]
    if prefix_shard is not None and prefix_shard[1] > 1:
        can = can[((can >> 32) % prefix_shard[1]) == prefix_shard[0]]
    return torch.unique(can)           # sorted


# ------------------------------------------------------------------------------------------------
class _Grow:
    """append-only 1-D tensor with amortised doubling"""

    def __init__(self, dtype, device, cap=1 << 16, fill=0):
        self.t = torch.full((cap,), fill, dtype=dtype, device=device)
        self.n = 0
        self.fill = fill

    def ensure(self, n):
        if n > self.t.numel():
            cap = max(n, self.t.numel() * 2)
            t = torch.full((cap,), self.fill, dtype=self.t.dtype, device=self.t.device)
            t[: self.n] = self.t[: self.n]
            self.t = t

    def extend_to(self, n):
        self.ensure(n)
        self.n = n


_KEEP_KMERS = 1 << 26            # build_patterns: k-mers of all samples together (512 MB) up to which phase A's sets are kept for phase B


def build_patterns(sample_kmers_iter, n_samples, device, dictionary=None, progress=None):
    """Emulates `kmer-db build` over samples 0..n-1.

    sample_kmers_iter(i) -> sorted unique int64 k-mer tensor of sample i (called once per phase).
    Returns dict with per-pattern arrays (creation order, parent < child):
      num_kmers, parent, num_samples, num_local, local ids CSR (local_ptr, local_ids),
      plus kmer dictionary (sorted) and kmer_pid (pattern id of every k-mer) and sample_counts.
    """
    dev = torch.device(device)
    # ---- phase A: the set of distinct k-mers --------------------------------------------------
    kept, kept_n = None, 0              # the samples' k-mer sets of phase A, kept for phase B while they are few (collections of short genomes:
                                        # deriving every sample twice was half of the generator's time there)
    if dictionary is None:
        parts, acc, acc_n = [], [], 0
        kept = []
        for i in range(n_samples):
            km = sample_kmers_iter(i)
            if kept is not None:
                kept_n += km.numel()
                if kept_n > _KEEP_KMERS:
                    kept = None
                else:
                    kept.append(km)
            acc.append(km)
            acc_n += km.numel()
            if acc_n >= (1 << 28) or i == n_samples - 1:
                parts.append(torch.unique(torch.cat(acc)))
                acc, acc_n = [], 0
                if len(parts) > 1 and sum(p.numel() for p in parts[:-1]) < parts[-1].numel() * 4:
                    parts = [torch.unique(torch.cat(parts))]
        dictionary = torch.unique(torch.cat(parts)) if len(parts) > 1 else parts[0]
        del parts
    D = dictionary
    cur = torch.zeros(D.numel(), dtype=torch.int32, device=dev)          # pattern id of every k-mer (0 = none yet)

    here = _Grow(torch.int64, dev)      # k-mers currently AT the pattern (pattern_t::num_kmers)
    nsam = _Grow(torch.int32, dev)      # num_samples
    par = _Grow(torch.int32, dev, fill=-1)
    isp = _Grow(torch.bool, dev, fill=False)
    for g in (here, nsam, par, isp):
        g.extend_to(1)                  # pattern 0 = the empty pattern (prefix_kmer_db.cpp:24)
    ev_pid, ev_sid = [], []             # chunks of patterns / their sample: "sample appended to the pattern's local list"
    sample_counts = []

    for s in range(n_samples):
        km = kept[s] if kept is not None else sample_kmers_iter(s)
        sample_counts.append(int(km.numel()))
        if km.numel() == 0:
            continue
        slot = torch.searchsorted(D, km)
        pid = cur[slot].to(torch.int64)
        spid, perm = torch.sort(pid)
        uq, cnt = torch.unique_consecutive(spid, return_counts=True)
        P = here.n
        # prefix_kmer_db.cpp:199-204: extend in place iff the group takes every k-mer of a childless pattern
        ext = (here.t[uq] == cnt) & (~isp.t[uq]) & (uq != 0)
        new = ~ext
        n_new = int(new.sum())
        # extension
        uq_ext = uq[ext]
        if uq_ext.numel():
            nsam.t[uq_ext] += 1
            ev_pid.append(uq_ext.to(torch.int32))
            ev_sid.append(s)
        # new child patterns (:206-222, pattern.h:104-113)
        if n_new:
            uq_new, cnt_new = uq[new], cnt[new]
            ids = torch.arange(P, P + n_new, dtype=torch.int64, device=dev)
            for g in (here, nsam, par, isp):
                g.extend_to(P + n_new)
            has_parent = nsam.t[uq_new] > 0
            par.t[ids] = torch.where(has_parent, uq_new, torch.full_like(uq_new, -1)).to(torch.int32)
            nsam.t[ids] = nsam.t[uq_new] + 1
            here.t[ids] = cnt_new
            isp.t[ids] = False
            isp.t[uq_new[has_parent]] = True
            nz = uq_new != 0
            here.t[uq_new[nz]] -= cnt_new[nz]
            ev_pid.append(ids.to(torch.int32))
            ev_sid.append(s)
            # k-mers of the new groups now point at the new pattern
            target = uq.clone()
            target[new] = ids
            cur[slot[perm]] = torch.repeat_interleave(target, cnt).to(torch.int32)
        if progress and (s + 1) % progress == 0:
            print("  synth build: %d/%d samples, %d patterns" % (s + 1, n_samples, here.n), file=sys.stderr, flush=True)

    P = here.n
    # The events were appended in sample order, one chunk per sample and at most one event per pattern in a chunk: every pattern's ids
    # are placed in ascending order by a running fill count — no sort over all events (a collection of 10 000 genomes of 625 kbp has
    # more than 2^31 of them, beyond what the device sorts take).
    num_local = torch.zeros(P, dtype=torch.int64, device=dev)
    one = None
    for ch in ev_pid:
        if one is None or one.numel() < ch.numel():
            one = torch.ones(ch.numel(), dtype=torch.int64, device=dev)
        num_local.index_add_(0, ch.to(torch.int64), one[: ch.numel()])
    local_ptr = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    local_ptr[1:] = torch.cumsum(num_local, 0)
    E = int(local_ptr[-1])
    local_ids = torch.zeros(E, dtype=torch.int64, device=dev)
    fill = local_ptr[:-1].clone()
    for ch, sid in zip(ev_pid, ev_sid):
        idx = ch.to(torch.int64)
        pos = fill[idx]
        local_ids[pos] = sid
        fill[idx] = pos + 1
    del fill
    return {
        "num_kmers": here.t[:P].clone(), "parent": par.t[:P].to(torch.int64), "num_samples": nsam.t[:P].to(torch.int64),
        "num_local": num_local, "local_ptr": local_ptr, "local_ids": local_ids,
        "dictionary": D, "kmer_pid": cur, "sample_counts": sample_counts,
    }


def gamma_encode_patterns(pat, max_events=1 << 27):
    """-> (last_sample_id, num_bits, data_offset (uint64 words), data words) in the on-disk layout:
    l-1 gamma-coded deltas per pattern, stream padded to 128 bits (pattern.h:79-81).  Worked through in ranges of patterns
    with at most `max_events` local ids each (every tensor stays far below 2^31 elements, whatever the collection)."""
    dev = pat["num_local"].device
    P = pat["num_local"].numel()
    lp, ids, l = pat["local_ptr"], pat["local_ids"], pat["num_local"]
    last = torch.zeros(P, dtype=torch.int64, device=dev)
    has = l > 0
    last[has] = ids[lp[1:][has] - 1]
    nbits = torch.zeros(P, dtype=torch.int64, device=dev)
    # ranges of patterns: cut where the running number of events passes a multiple of max_events
    cuts = [0]
    if P:
        marks = torch.searchsorted(lp, torch.arange(max_events, int(lp[-1]) + max_events, max_events, dtype=torch.int64, device=dev), right=True) - 1
        for m in marks.tolist():
            m = min(max(m, cuts[-1] + 1), P)
            if m > cuts[-1]:
                cuts.append(m)
        if cuts[-1] != P:
            cuts.append(P)

    def codes(p0, p1):
        """(owner pattern, code length, code) of every delta of the patterns [p0, p1)"""
        e0, e1 = int(lp[p0]), int(lp[p1])
        seg = ids[e0:e1]
        ll = l[p0:p1]
        first_mask = torch.zeros(e1 - e0, dtype=torch.bool, device=dev)
        hh = ll > 0
        first_mask[(lp[p0:p1][hh] - e0)] = True
        delta = torch.zeros(e1 - e0, dtype=torch.int64, device=dev)
        if e1 - e0 > 1:
            delta[1:] = seg[1:] - seg[:-1]
        dmask = ~first_mask
        d = delta[dmask]
        owner = torch.repeat_interleave(torch.arange(p0, p1, device=dev), ll)[dmask]
        Lb = torch.frexp(d.to(torch.float64))[1].to(torch.int64)             # bit length
        clen = 2 * Lb - 1
        code = (((1 << (Lb - 1)) - 1) << Lb) | (d - (1 << (Lb - 1)))
        return owner, clen, code

    for p0, p1 in zip(cuts[:-1], cuts[1:]):
        owner, clen, _ = codes(p0, p1)
        nbits.index_add_(0, owner, clen)
    words = ((nbits + 127) // 128) * 2
    data_off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    data_off[1:] = torch.cumsum(words, 0)
    total_words = int(data_off[-1])
    data = torch.zeros(total_words + 2, dtype=torch.int64, device=dev)
    for p0, p1 in zip(cuts[:-1], cuts[1:]):
        owner, clen, code = codes(p0, p1)
        if owner.numel() == 0:
            continue
        # bit offset of every code inside its pattern's stream
        csum = torch.cumsum(clen, 0) - clen
        cnt_codes = torch.bincount(owner - p0, minlength=p1 - p0)
        code_ptr = torch.cumsum(cnt_codes, 0) - cnt_codes
        pat_first = torch.zeros(p1 - p0, dtype=torch.int64, device=dev)
        nz = cnt_codes > 0
        pat_first[nz] = csum[code_ptr[nz]]
        boff = csum - pat_first[owner - p0] + data_off[owner] * 64
        w = boff >> 6
        sft = boff & 63
        room = 64 - sft
        fits = clen <= room
        hi_part = torch.where(fits, code << (room - clen).clamp(min=0), code >> (clen - room).clamp(min=0))
        data.index_add_(0, w, hi_part)
        spill = ~fits
        if bool(spill.any()):
            rem = (clen - room)[spill]
            lo_part = (code[spill] & ((1 << rem) - 1)) << (64 - rem)
            data.index_add_(0, w[spill] + 1, lo_part)
    return last, nbits, data_off[:-1], data[: max(total_words, 0) + 2]


def to_view_arrays(pat):
    """numpy arrays in the field order of kmdb_db_view (include/kmdb_amd.h)."""
    last, nbits, doff, data = gamma_encode_patterns(pat)
    c = lambda t, dt: t.cpu().numpy().astype(dt)      # noqa: E731
    return {
        "num_kmers": c(pat["num_kmers"], np.int64), "parent_id": c(pat["parent"], np.int64),
        "num_samples": c(pat["num_samples"], np.uint32), "num_local": c(pat["num_local"], np.uint32),
        "last_sample_id": c(last, np.uint32), "num_bits": c(nbits, np.uint32),
        "data_offset": c(doff, np.uint64), "data": data.cpu().numpy().view(np.uint64),
    }


def build_hashtables(dictionary, kmer_pid, k):
    """Prefix-bucketed linear-probing tables in the reference's layout (src/hashmap_lp.h): bucket =
    kmer >> 32, key = low 32 bits, home slot = murmur3 fmix32(key) & (capacity - 1), capacity a power of two
    with fill <= 0.8 (:150,427-437), empty slots = {0, INT32_MAX}.  Slot positions differ from a table the
    reference grew incrementally (insertion order), but `find` (:308-333) only needs a valid probe sequence.
    Returns (bucket_offset uint64[nb+1], slots uint64[total]) with item = key | val << 32."""
    kmers = dictionary.cpu().numpy().view(np.uint64)
    pids = kmer_pid.cpu().numpy().astype(np.int64)
    nb = 1 << max(8, 2 * k - 32)
    bucket = (kmers >> np.uint64(32)).astype(np.int64)
    counts = np.bincount(bucket, minlength=nb)
    caps = np.maximum(16, 1 << np.ceil(np.log2(np.maximum(counts, 1) / 0.8)).astype(np.int64))
    caps = np.where(counts > caps * 0.8, caps * 2, caps)
    boff = np.zeros(nb + 1, dtype=np.uint64)
    boff[1:] = np.cumsum(caps).astype(np.uint64)
    slots = np.full(int(boff[-1]), np.uint64(0x7fffffff) << np.uint64(32), dtype=np.uint64)
    key = (kmers & np.uint64(0xffffffff)).astype(np.uint64)
    h = key.astype(np.uint32)
    h ^= h >> np.uint32(16); h *= np.uint32(0x85ebca6b); h ^= h >> np.uint32(13); h *= np.uint32(0xc2b2ae35); h ^= h >> np.uint32(16)
    order = np.argsort(bucket, kind="stable")
    start = 0
    for b in range(nb):
        n = int(counts[b])
        if not n:
            continue
        idx = order[start: start + n]
        start += n
        cap = int(caps[b])
        home = (h[idx].astype(np.int64)) & (cap - 1)
        o2 = np.argsort(home, kind="stable")
        hs = home[o2]
        # sorted homes: position = max(home, previous + 1) = i + running max of (home - i)
        pos = np.arange(n) + np.maximum.accumulate(hs - np.arange(n))
        items = key[idx][o2] | (pids[idx][o2].astype(np.uint64) << np.uint64(32))
        base = int(boff[b])
        fit = pos < cap
        slots[base + pos[fit]] = items[fit]
        for it, hm in zip(items[~fit], hs[~fit]):              # the few that wrap past the end of the table
            p_ = int(hm)
            while (slots[base + p_] >> np.uint64(32)) != np.uint64(0x7fffffff):
                p_ = (p_ + 1) & (cap - 1)
            slots[base + p_] = it
    return boff, slots


def write_db(path, k, fraction, names, sample_counts, arrays, n_buckets=None, kmers_count=0, tables=None):
    """Serialise in the reference's .db format (prefix_kmer_db.cpp:438-574).  Without `tables` the raw
    hashtables are EMPTY — enough for all2all / all2all-sp, which skip them (console_all2all.cpp:26); with
    tables = build_hashtables(...) the file also serves new2all."""
    if n_buckets is None:
        n_buckets = 1 << max(8, 2 * k - 32)
    with open(path, "wb") as f:
        f.write(struct.pack("<QIddiBQ", 1, k, fraction, 0.0, 0, 1, kmers_count))
        f.write(struct.pack("<Q", len(names)))
        for nm, c in zip(names, sample_counts):
            b = nm.encode()
            f.write(struct.pack("<QQ", int(c), len(b)))
            f.write(b)
        f.write(struct.pack("<Q", n_buckets))
        if tables is None:
            empty = struct.pack("<dQQQQQQQ", 0.8, 0, 16, 12, 15, 128, 0, 0) + struct.pack("<Q", 0)   # header + 1 bit-vector word
            f.write(empty * n_buckets)
        else:
            boff, slots = tables                                   # hashmap_lp.h:481-528
            for b in range(n_buckets):
                seg = slots[int(boff[b]): int(boff[b + 1])]
                cap = seg.size
                used = (seg >> np.uint64(32)) != np.uint64(0x7fffffff)
                filled = int(used.sum())
                f.write(struct.pack("<dQQQQQQQ", 0.8, filled, cap, int(cap * 0.8), cap - 1, cap * 8, 0, 0))
                bv = np.zeros((cap + 63) // 64 * 64, dtype=np.uint8)
                bv[:cap] = used
                f.write(np.packbits(bv, bitorder="little").tobytes())      # bit i of word i/64 = slot i filled
                f.write(seg[used].tobytes())
        P = arrays["num_kmers"].size
        f.write(struct.pack("<Q", P))
        nb = arrays["num_bits"].astype(np.int64)
        words = ((nb + 127) // 128) * 2
        sizes = 40 + words * 8
        hdr = np.zeros((P, 5), dtype=np.uint64)
        hdr[:, 0] = arrays["num_kmers"].view(np.uint64)
        hdr[:, 1] = arrays["parent_id"].view(np.uint64)
        hdr[:, 2] = arrays["num_samples"].astype(np.uint64) | (arrays["num_local"].astype(np.uint64) << np.uint64(32))
        hdr[:, 3] = arrays["last_sample_id"].astype(np.uint64) | (arrays["num_bits"].astype(np.uint64) << np.uint64(32))
        doff = arrays["data_offset"].astype(np.int64)
        limit = 64 << 20
        start = 0
        while start < P:
            # greedily fill a 64 MB block (:544-551)
            csz = np.cumsum(sizes[start: start + (1 << 20)])
            cnt = int(np.searchsorted(csz, limit, side="right"))
            cnt = max(cnt, 1)
            end = min(P, start + cnt)
            block = bytearray()
            for p in range(start, end):
                block += hdr[p].tobytes()
                if words[p]:
                    block += arrays["data"][doff[p]: doff[p] + words[p]].tobytes()
            f.write(struct.pack("<Q", len(block)))
            f.write(block)
            start = end


def write_db_fast(path, k, fraction, names, sample_counts, arrays, kmers_count=0, device="cpu", n_buckets=None):
    """write_db for databases of 10^8 patterns: the pattern section (prefix_kmer_db.cpp:438-574, pattern.cpp:15-46) is
    assembled as one uint64 image with vectorised torch ops on `device` and written block by block; the raw
    hashtables are EMPTY (all2all / all2all-sp skip them, console_all2all.cpp:26)."""
    dev = torch.device(device)
    if n_buckets is None:
        n_buckets = 1 << max(8, 2 * k - 32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    P = arrays["num_kmers"].size
    nb = t(arrays["num_bits"].astype(np.int64))
    words = ((nb + 127) // 128) * 2
    sizes = 5 + words                                                    # uint64 words per pattern
    off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(sizes, 0)
    total = int(off[-1])
    img = torch.zeros(total, dtype=torch.int64, device=dev)
    o = off[:-1]
    img[o] = t(arrays["num_kmers"].astype(np.int64))
    img[o + 1] = t(arrays["parent_id"].astype(np.int64))
    img[o + 2] = t(arrays["num_samples"].astype(np.int64)) | (t(arrays["num_local"].astype(np.int64)) << 32)
    img[o + 3] = t(arrays["last_sample_id"].astype(np.int64)) | (nb << 32)
    data = t(arrays["data"].view(np.int64))
    doff = t(arrays["data_offset"].astype(np.int64))
    for j in range(int(words.max()) if P else 0):
        m = words > j
        img[o[m] + 5 + j] = data[doff[m] + j]
    del data, doff
    cum = (off[1:] * 8).cpu().numpy()                                    # bytes up to and including pattern p
    with open(path, "wb") as f:
        f.write(struct.pack("<QIddiBQ", 1, k, fraction, 0.0, 0, 1, kmers_count))
        f.write(struct.pack("<Q", len(names)))
        for nm, c in zip(names, sample_counts):
            b = nm.encode()
            f.write(struct.pack("<QQ", int(c), len(b)))
            f.write(b)
        f.write(struct.pack("<Q", n_buckets))
        empty = struct.pack("<dQQQQQQQ", 0.8, 0, 16, 12, 15, 128, 0, 0) + struct.pack("<Q", 0)
        f.write(empty * n_buckets)
        f.write(struct.pack("<Q", P))
        limit = 64 << 20
        start, done_bytes = 0, 0
        while start < P:
            end = int(np.searchsorted(cum, done_bytes + limit, side="right"))        # greedily fill a 64 MB block (:544-551)
            end = min(P, max(end, start + 1))
            a, b = done_bytes // 8, int(cum[end - 1]) // 8
            f.write(struct.pack("<Q", (b - a) * 8))
            f.write(img[a:b].cpu().numpy().tobytes())
            start, done_bytes = end, int(cum[end - 1])


def shard_item_lists(dictionary, kmer_pid, k):
    """(bucket_offset, items) of the k-mer -> pattern map grouped by prefix bucket (kmer >> 32): items = key | pid << 32
    like hashmap_lp's item_t (src/hashmap_lp.h:71-74), but WITHOUT empty slots and in sorted order — enough for
    kmdb_db_upload_shard, which only counts the items of a shard's buckets per pattern; not a probe-able table."""
    nb = 1 << max(8, 2 * k - 32)
    bucket = _lsr(dictionary, 32)
    counts = torch.bincount(bucket, minlength=nb)
    boff = torch.zeros(nb + 1, dtype=torch.int64, device=dictionary.device)
    boff[1:] = torch.cumsum(counts, 0)
    items = (dictionary & 0xFFFFFFFF) | (kmer_pid.to(torch.int64) << 32)       # the dictionary is sorted: buckets are contiguous
    return boff.cpu().numpy().astype(np.uint64), items.cpu().numpy().view(np.uint64)


def synth_database(n_samples, clade_size, length, k=18, fraction=1.0, seed=20260928, r1=0.10, r2=0.01,
                   device="cpu", progress=None):
    """Genomes -> k-mers -> patterns.  Returns (genomes, pat) with pat as in build_patterns()."""
    g = CladeGenomes(n_samples, clade_size, length, r1, r2, seed, device)
    fn = lambda i: kmers_of(g.sample(i), k, fraction)         # noqa: E731
    pat = build_patterns(fn, n_samples, device, progress=progress)
    return g, pat
