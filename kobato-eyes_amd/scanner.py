"""Duplicate scanning on the MI355X: drop-in for the reference's ``dup.scanner``.

Same public surface as src/dup/scanner.py:430-436 -- ``DuplicateFile``, ``DuplicateCluster``,
``DuplicateClusterEntry``, ``DuplicateScanConfig``, ``DuplicateScanner`` -- so it plugs into
``DupViewModel(scanner_factory=...)`` (src/ui/viewmodels/dup_view_model.py:31,41).

Candidate generation (src/dup/scanner.py:227-299) runs as one all-pairs popcount scan on the
GPU (csrc/ke_scan.hip) with the closed form of the reference's bucket loop as predicate;
connected components come from the library's host union-find; keeper choice and every sort
key follow src/dup/scanner.py:320-356, 402-415.
"""
from __future__ import annotations

import gc
import logging
import math
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Iterable, Mapping, Optional, Sequence

import numpy as np

from . import _native

logger = logging.getLogger("dup.scanner")

_MASK64 = (1 << 64) - 1
_EXT_RANK = {"png": 4, "apng": 4, "webp": 3, "tiff": 2, "tif": 2, "bmp": 1, "gif": 1,
             "jpeg": 0, "jpg": 0, "jpe": 0, "jfif": 0}

PHASH_KEYS = ("phash_u64", "phash", "phash64", "phash_hex", "phash_bytes", "signature", "sig")


def _row_get(row, key, default=None):
    """Field of a dict / sqlite3.Row / attribute bag; malformed access yields the default."""
    try:
        if isinstance(row, dict):
            return row.get(key, default)
        keys = getattr(row, "keys", None)
        if keys is not None and key in keys():
            return row[key]
        return getattr(row, key, default)
    except (AttributeError, KeyError, TypeError):
        return default


def _parse_phash_any(raw) -> Optional[int]:
    """Any stored representation -> unsigned 64-bit int, or None (src/dup/scanner.py:44-81)."""
    if raw is None:
        return None
    if isinstance(raw, (bytes, bytearray, memoryview)):
        return int.from_bytes(bytes(raw), "big", signed=False) & _MASK64
    if isinstance(raw, str):
        text = raw.strip()
        if not text:
            return None
        for base in (0, 16):
            try:
                return int(text, base) & _MASK64
            except ValueError:
                continue
        return None
    try:
        return int(raw) & _MASK64
    except (OverflowError, TypeError, ValueError):
        return None


def _int_or_none(value):
    return int(value) if isinstance(value, (int, float)) else None


@dataclass(frozen=True)
class DuplicateFile:
    """What the scan needs to know about one file (src/dup/scanner.py:87-128)."""

    file_id: int
    path: Path
    size: Optional[int]
    width: Optional[int]
    height: Optional[int]
    phash: int
    embedding: Optional[tuple] = None

    @classmethod
    def from_row(cls, row: Mapping[str, object]) -> "DuplicateFile":
        raw = None
        for key in PHASH_KEYS:
            raw = _row_get(row, key)
            if raw is not None:
                break
        value = _parse_phash_any(raw)
        if value is None:
            raise ValueError("Row is missing perceptual hash information")
        return cls(
            file_id=int(_row_get(row, "file_id", _row_get(row, "id", -1))),
            path=Path(str(_row_get(row, "path", _row_get(row, "file_path", "")))),
            size=_int_or_none(_row_get(row, "size")),
            width=_int_or_none(_row_get(row, "width")),
            height=_int_or_none(_row_get(row, "height")),
            phash=value,
        )

    @property
    def resolution(self) -> int:
        return (self.width or 0) * (self.height or 0)

    @property
    def extension_priority(self) -> int:
        return _EXT_RANK.get(self.path.suffix.lower().lstrip("."), 0)


@dataclass(frozen=True)
class DuplicateClusterEntry:
    file: DuplicateFile
    best_hamming: Optional[int]


@dataclass(frozen=True)
class DuplicateCluster:
    files: list
    keeper_id: int


@dataclass(frozen=True)
class DuplicateScanConfig:
    """Thresholds of the scan; validation as src/dup/scanner.py:147-166."""

    hamming_threshold: int = 8
    size_ratio: Optional[float] = None
    band_bits: int = 16
    band_count: int = 4
    cosine_threshold: Optional[float] = None

    def __post_init__(self) -> None:
        if self.band_bits <= 0:
            raise ValueError("band_bits must be positive")
        if self.band_count <= 0:
            raise ValueError("band_count must be positive")
        if not 0 <= self.hamming_threshold <= 64:
            raise ValueError("hamming_threshold must be in [0, 64]")
        if self.cosine_threshold is not None and not -1.0 <= self.cosine_threshold <= 1.0:
            raise ValueError("cosine_threshold must be between -1.0 and 1.0")


@dataclass
class DuplicateEdge:
    file_id_a: int
    file_id_b: int
    hamming: Optional[int]


def _safe_positive_int(value: Optional[str]) -> Optional[int]:
    if value is None or not value.strip():
        return None
    try:
        parsed = int(value)
    except ValueError:
        return None
    return parsed if parsed > 0 else None


def _size_ok(size_a: int, size_b: int, ratio: float) -> bool:
    """src/dup/scanner.py:358-370."""
    if not ratio or ratio <= 0 or size_a <= 0 or size_b <= 0:
        return True
    smaller, larger = (size_a, size_b) if size_a < size_b else (size_b, size_a)
    return smaller / larger >= ratio


def _cosine(left: DuplicateFile, right: DuplicateFile) -> Optional[float]:
    """float64 cosine of two embeddings; None = "cannot tell, let it pass" (src/dup/scanner.py:383-400)."""
    u, v = left.embedding, right.embedding
    if u is None or v is None or len(u) == 0 or len(v) == 0 or len(u) != len(v):
        return None
    dot = sum(p * q for p, q in zip(u, v))
    nu, nv = math.sqrt(sum(p * p for p in u)), math.sqrt(sum(q * q for q in v))
    if nu == 0.0 or nv == 0.0:
        return None
    return dot / (nu * nv)


class DuplicateScanner:
    """GPU-backed replacement of the reference scanner (same constructor, same method)."""

    def __init__(self, config: DuplicateScanConfig, *, device: int = 0, part_index: int = 0, part_count: int = 1) -> None:
        assert config.band_bits * config.band_count <= 64, "band config too large"
        self._config = config
        self._device = device
        self._part = (part_index, part_count)
        self._counters: Optional[dict] = None
        self._funnel: Optional[tuple] = None

    # -- candidate edges ---------------------------------------------------------------------
    def candidate_edges(self, candidates: Sequence[DuplicateFile]) -> dict:
        """{(id_lo, id_hi): DuplicateEdge} exactly as the reference's ``edges`` dict ends up."""
        cfg = self._config
        n = len(candidates)
        hashes = np.fromiter((f.phash & _MASK64 for f in candidates), dtype=np.uint64, count=n)
        ids = np.fromiter((f.file_id for f in candidates), dtype=np.int64, count=n)
        sizes = np.fromiter(((f.size or 0) for f in candidates), dtype=np.int64, count=n)
        cap = _safe_positive_int(os.environ.get("KE_DUP_BUCKET_PAIR_CAP"))
        self._log_bucket_stats(hashes, cap)
        ratio = cfg.size_ratio if (cfg.size_ratio is not None and cfg.size_ratio > 0) else 0.0
        ctx = _native.get_context(self._device)
        raw, counters = ctx.hamming_scan(
            hashes, n, ids=ids, sizes=sizes if ratio > 0 else None, threshold=cfg.hamming_threshold,
            band_bits=cfg.band_bits, band_count=cfg.band_count, size_ratio=ratio, bucket_pair_cap=cap or 0,
            part_index=self._part[0], part_count=self._part[1])
        return self._edges_from_raw(candidates, hashes, ids, raw, counters, sizes=sizes, ratio=ratio, cap=cap)

    # -- the reference's funnel counters (src/dup/scanner.py:258-299) ---------------------------------------------
    def _band_values(self, hashes: np.ndarray, band: int) -> np.ndarray:
        cfg = self._config
        mask = np.uint64((1 << cfg.band_bits) - 1) if cfg.band_bits < 64 else np.uint64(_MASK64)
        return (hashes >> np.uint64(band * cfg.band_bits)) & mask

    def _same_id_bucket_pairs(self, hashes, ids, sizes, ratio, cap) -> tuple[int, int]:
        """Bucket pairs of equal file id (the reference skips them before counting, :266): (all, those passing the
        size filter).  Only ids that occur more than once are looked at."""
        uniq, inverse, counts = np.unique(ids, return_inverse=True, return_counts=True)
        if not (counts > 1).any():
            return 0, 0
        cfg = self._config
        total = sized = 0
        bucket_len = []
        for band in range(cfg.band_count):
            vals = self._band_values(hashes, band)
            _, inv, cnt = np.unique(vals, return_inverse=True, return_counts=True)
            bucket_len.append((vals, cnt[inv]))
        for g in np.nonzero(counts > 1)[0]:
            pos = np.nonzero(inverse == g)[0]
            for x in range(len(pos) - 1):
                for y in range(x + 1, len(pos)):
                    i, j = int(pos[x]), int(pos[y])
                    for vals, blen in bucket_len:
                        if vals[i] != vals[j]:
                            continue
                        length = int(blen[i])
                        if cap is not None and length * (length - 1) // 2 > cap:
                            continue
                        total += 1
                        sized += _size_ok(int(sizes[i]), int(sizes[j]), ratio)
        return total, sized

    def _edges_from_raw(self, candidates, hashes, ids, raw, counters, *, sizes=None, ratio=0.0, cap=None) -> dict:
        cfg = self._config
        order = np.lexsort((raw["b"], raw["a"]))
        raw = raw[order]
        after_cos = 0
        keyed: dict[tuple[int, int], list] = {}
        for a, b, h, bands in zip(raw["a"].tolist(), raw["b"].tolist(), raw["h"].tolist(), raw["bands"].tolist()):
            if cfg.cosine_threshold is not None:
                cos = _cosine(candidates[a], candidates[b])
                if cos is not None and cos < cfg.cosine_threshold:
                    continue
            after_cos += bin(bands & 0xFFFFFFFF).count("1")
            ia, ib = int(ids[a]), int(ids[b])
            keyed.setdefault((ia, ib) if ia < ib else (ib, ia), []).append((a, b, h, bands))
        edges: dict[tuple[int, int], DuplicateEdge] = {}
        first_idx_cache: dict[tuple[int, int], int] = {}
        mask = (1 << cfg.band_bits) - 1

        def bucket_rank(pos: int, bands: int) -> int:
            # dict insertion order of the reference's buckets: (first position holding the value, band)
            best = None
            for band in range(min(cfg.band_count, 32)):
                if not (bands >> band) & 1:
                    continue
                val = (int(hashes[pos]) >> (band * cfg.band_bits)) & mask
                key = (band, val)
                if key not in first_idx_cache:
                    col = (hashes >> np.uint64(band * cfg.band_bits)) & np.uint64(mask)
                    first_idx_cache[key] = int(np.argmax(col == np.uint64(val)))
                rank = first_idx_cache[key] * cfg.band_count + band
                best = rank if best is None or rank < best else best
            return best if best is not None else 0

        for key, hits in keyed.items():
            if len(hits) > 1:  # duplicate file ids: the first writer wins (src/dup/scanner.py:287-290)
                hits.sort(key=lambda t: (bucket_rank(t[0], t[3]), t[0], t[1]))
            a, b, h, _ = hits[0]
            edges[key] = DuplicateEdge(int(ids[a]), int(ids[b]), int(h))
        # The reference's funnel figures (src/dup/scanner.py:292-299) are a log line, not part of the result: they are worked
        # out when somebody looks -- the logger at INFO, or a reader of ``last_counters`` -- and cost nothing otherwise.
        self._counters = None
        self._funnel = (hashes, ids, sizes, ratio, cap, [int(c) for c in counters], after_cos, len(edges))
        if logger.isEnabledFor(logging.INFO):
            c = self.last_counters
            logger.info("dup: pairs total=%d -> size=%d -> ham=%d -> cosine=%d -> edges=%d", c["pair_total"], c["after_size"],
                        c["after_ham"], c["after_cosine"], c["edges"])
        return edges

    @property
    def last_counters(self) -> Optional[dict]:
        """The funnel of the last scan: "pairs total" = the device's bucket-pair count (band histograms) minus bucket pairs of
        equal file id; "size" = the same with the size filter (``ke_band_pairs_after_size``).  A shard reports the whole
        table's figures for these two (they do not depend on the tiles it was dealt), its own share for the rest."""
        if self._counters is None and self._funnel is not None:
            hashes, ids, sizes, ratio, cap, counters, after_cos, n_edges = self._funnel
            same_total, same_sized = self._same_id_bucket_pairs(hashes, ids, sizes if sizes is not None else np.zeros(len(ids), np.int64),
                                                                ratio, cap)
            pair_total = counters[3] - same_total
            if ratio > 0 and sizes is not None:
                cfg = self._config
                after_size = _native.get_context(self._device).band_pairs_after_size(
                    hashes, sizes, len(hashes), band_bits=cfg.band_bits, band_count=cfg.band_count, size_ratio=ratio,
                    bucket_pair_cap=cap or 0) - same_sized
            else:
                after_size = pair_total
            self._counters = {"pairs_evaluated": counters[0], "pair_total": pair_total, "after_size": after_size,
                              "after_ham": counters[1], "after_cosine": after_cos, "edges": n_edges}
            self._funnel = None
        return self._counters

    @last_counters.setter
    def last_counters(self, value: Optional[dict]) -> None:
        self._counters, self._funnel = value, None

    def _log_bucket_stats(self, hashes: np.ndarray, cap: Optional[int]) -> None:
        if not logger.isEnabledFor(logging.INFO) and cap is None:
            return
        cfg = self._config
        mask = np.uint64((1 << cfg.band_bits) - 1)
        n_buckets = ge2 = max_bucket = 0
        for band in range(cfg.band_count):
            _, counts = np.unique((hashes >> np.uint64(band * cfg.band_bits)) & mask, return_counts=True)
            n_buckets += len(counts)
            ge2 += int((counts >= 2).sum())
            max_bucket = max(max_bucket, int(counts.max()) if len(counts) else 0)
        max_pairs = max_bucket * (max_bucket - 1) // 2
        logger.info("dup: buckets=%d (>=2:%d) max_bucket=%d max_bucket_pairs=%d pair_cap=%s", n_buckets, ge2, max_bucket,
                    max_pairs, cap)
        if cap is not None and max_pairs > cap:
            logger.warning("dup: largest bucket has %d pair(s), above KE_DUP_BUCKET_PAIR_CAP=%d; large buckets will be skipped",
                           max_pairs, cap)

    # -- the seam ----------------------------------------------------------------------------
    def build_clusters(self, files: Iterable[DuplicateFile]) -> list:
        cfg = self._config
        candidates = [f for f in files if f.phash is not None]
        logger.info("dup: candidates=%d band_bits=%d band_count=%d ham_th=%d size_ratio=%s cosine_th=%s", len(candidates),
                    cfg.band_bits, cfg.band_count, cfg.hamming_threshold, cfg.size_ratio, cfg.cosine_threshold)
        if len(candidates) < 2:
            return []
        # The host part allocates a few objects per edge and per cluster and none of them is part of a reference cycle; with a
        # million DuplicateFile objects alive every full pass of the cycle collector it sets off walks all of them (1.3 of the
        # 1.7 s of a million-file call went there).  The collector is paused for the duration of the call.
        paused = gc.isenabled()
        gc.disable()
        try:
            edges = self.candidate_edges(candidates)
            if not edges:
                return []
            return assemble_clusters(candidates, edges.values())
        finally:
            if paused:
                gc.enable()

    @staticmethod
    def _choose_keeper(entries: Sequence[DuplicateClusterEntry]) -> int:
        return min(entries, key=_keeper_key).file.file_id


def _keeper_key(entry: DuplicateClusterEntry) -> tuple:
    f = entry.file
    return (-(f.size or 0), -f.resolution, -f.extension_priority, f.path.suffix.lower(), f.path.name.lower(), f.file_id)


def _file_keys(f: DuplicateFile) -> tuple:
    """(keeper key, entry sort key without the keeper flag) of src/dup/scanner.py:402-415 and :338-347."""
    suffix = f.path.suffix.lower()
    name = f.path.name.lower()
    head = (-(f.size or 0), -f.resolution, -_EXT_RANK.get(suffix.lstrip("."), 0))
    return head + (suffix, name, f.file_id), head + (name, f.file_id)


def assemble_clusters(candidates: Sequence[DuplicateFile], edges: Iterable[DuplicateEdge]) -> list:
    """Edges -> ordered clusters (src/dup/scanner.py:304-356).  Components come from the library's host union-find over
    compacted file ids; best_hamming and the grouping are array work, Python only touches the members of each cluster once
    (a million-file table leaves some 60 000 clusters: their keys, not the union-find, are the time)."""
    edges = list(edges)
    if not edges:
        return []
    m = len(edges)
    by_id = {f.file_id: f for f in candidates}          # later duplicates of an id win, as in the reference
    ends = np.empty(2 * m, np.int64)
    ends[:m] = np.fromiter((e.file_id_a for e in edges), np.int64, m)
    ends[m:] = np.fromiter((e.file_id_b for e in edges), np.int64, m)
    ham = np.fromiter((-1 if e.hamming is None else e.hamming for e in edges), np.int64, m)
    node_ids, inverse = np.unique(ends, return_inverse=True)         # ascending ids -> members ascending
    raw = np.zeros(m, _native.EDGE_DTYPE)
    raw["a"], raw["b"] = inverse[:m], inverse[m:]
    labels = _native.cluster_labels(raw, len(node_ids))
    none = np.iinfo(np.int64).max
    best = np.full(len(node_ids), none, np.int64)
    known = ham >= 0
    np.minimum.at(best, inverse[:m][known], ham[known])
    np.minimum.at(best, inverse[m:][known], ham[known])
    order = np.argsort(labels, kind="stable")
    sorted_labels = labels[order]
    cuts = np.nonzero(np.concatenate(([True], sorted_labels[1:] != sorted_labels[:-1], [True])))[0].tolist()
    member_ids = node_ids[order].tolist()
    member_best = best[order].tolist()
    ranked = []                                          # (cluster sort key, cluster)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi - lo < 2:
            continue
        members = [(by_id.get(member_ids[k]), member_best[k]) for k in range(lo, hi)]
        members = [(f, None if b == none else b) for f, b in members if f is not None]
        if len(members) < 2:
            continue
        if len(members) == 2 and (members[0][0].size or 0) != (members[1][0].size or 0):
            # the common case needs no key at all: two files of different size, the larger is the keeper and comes first
            if (members[0][0].size or 0) < (members[1][0].size or 0):
                members.reverse()
            keeper = members[0][0].file_id
        else:
            keyed = [(_file_keys(f), f, b) for f, b in members]
            keeper = min(keyed, key=lambda t: t[0][0])[1].file_id
            keyed.sort(key=lambda t: (0 if t[1].file_id == keeper else 1,) + t[0][1])
            members = [(f, b) for _, f, b in keyed]
        entries = [DuplicateClusterEntry(f, b) for f, b in members]
        top = max((f.size or 0) for f, _ in members)
        ranked.append(((-top, members[0][0].path.as_posix().lower()), DuplicateCluster(files=entries, keeper_id=keeper)))
    ranked.sort(key=lambda t: t[0])
    return [c for _, c in ranked]


__all__ = ["DuplicateFile", "DuplicateCluster", "DuplicateClusterEntry", "DuplicateScanConfig", "DuplicateScanner"]
