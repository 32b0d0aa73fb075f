"""The two names the north star asks for -- ``compute_signature()`` / ``find_duplicates()`` --
as thin aliases over the drop-ins of the reference's real seams (SURVEY finding 2), plus a
headless restatement of the production call sequence ``DuplicateScanRunnable.run``
(src/ui/dup_workers.py:148-239): rows -> fill missing hashes -> from_row (bad rows skipped)
-> build_clusters.  Config surface: ``{hamming_threshold, ssim_threshold}`` with the defaults
and coercions of PipelineSettings (src/core/config/schema.py:116-117, 186-201).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Iterable, Mapping, Optional, Sequence

import importlib

from . import fastsig, refine as _refine

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash
from .cluster import ClusterBuilder
from .scanner import DuplicateCluster, DuplicateFile, DuplicateScanConfig, DuplicateScanner

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class DedupSettings:
    """``hamming_threshold`` / ``ssim_threshold`` as persisted by the reference's settings file."""

    hamming_threshold: int = 10
    ssim_threshold: float = 0.92

    @staticmethod
    def coerce(hamming_threshold=None, ssim_threshold=None) -> "DedupSettings":
        try:
            ham = max(0, int(hamming_threshold))
        except (TypeError, ValueError):
            ham = 10
        try:
            ssim = float(ssim_threshold)
        except (TypeError, ValueError):
            ssim = 0.92
        return DedupSettings(ham, ssim)


def compute_signature(image_or_path, *, device: int = 0) -> tuple[int, int]:
    """(signed pHash, signed dHash) of a PIL image, ndarray or file path."""
    if isinstance(image_or_path, (str, Path)):
        from PIL import Image

        with Image.open(image_or_path) as im:
            return _phash.phash_dhash(im, device=device)
    return _phash.phash_dhash(image_or_path, device=device)


def find_duplicates(files: Iterable, *, hamming_threshold: int = 8, ssim_threshold: Optional[float] = None,
                    size_ratio: Optional[float] = None, band_bits: int = 16, band_count: int = 4,
                    scanner_factory: Callable[[DuplicateScanConfig], DuplicateScanner] = DuplicateScanner,
                    device: int = 0) -> list:
    """Clusters of near-duplicates.

    ``files``: DuplicateFile objects or row mappings (anything ``DuplicateFile.from_row`` takes;
    malformed rows are skipped as at src/ui/dup_workers.py:205-209).  With ``ssim_threshold``
    set, every candidate edge is re-checked with the SSIM kernel on the files' pixels and only
    edges with ``ssim >= ssim_threshold`` survive into the clusters.
    """
    parsed: list[DuplicateFile] = []
    for item in files:
        if isinstance(item, DuplicateFile):
            parsed.append(item)
            continue
        try:
            parsed.append(DuplicateFile.from_row(item))
        except ValueError:
            logger.debug("skipping row without a usable hash")
    config = DuplicateScanConfig(hamming_threshold=hamming_threshold, size_ratio=size_ratio, band_bits=band_bits,
                                 band_count=band_count)
    try:
        scanner = scanner_factory(config, device=device)      # one process per GPU: scan on the caller's device
    except TypeError:
        scanner = scanner_factory(config)                     # a foreign factory with the reference's one-argument signature
    if ssim_threshold is None:
        return scanner.build_clusters(parsed)
    from .scanner import assemble_clusters

    candidates = [f for f in parsed if f.phash is not None]
    if len(candidates) < 2:
        return []
    edges = list(scanner.candidate_edges(candidates).values())
    by_id = {f.file_id: f for f in candidates}
    # every file decoded once, one fit + one SSIM launch per size group (refine_pairs), not one of each per edge
    matches = _refine.refine_pairs([(e.file_id_a, e.file_id_b, by_id[e.file_id_a].path, by_id[e.file_id_b].path) for e in edges],
                                   thresholds=_refine.RefinementThresholds(ssim=ssim_threshold), device=device)
    kept = [e for e, m in zip(edges, matches) if m is not None and m.is_duplicate]
    return assemble_clusters(candidates, kept) if kept else []


def run_duplicate_scan(rows: Sequence[Mapping], *, db_path: Optional[str] = None, config: Optional[DuplicateScanConfig] = None,
                       progress: Optional[Callable[[str, int, int], None]] = None,
                       cancel_fn: Optional[Callable[[], bool]] = None, device: int = 0) -> list:
    """Headless ``DuplicateScanRunnable.run``: stage names and order as the reference emits them."""
    def tick(stage, done, total):
        if progress:
            progress(stage, done, total)

    rows = [dict(r) for r in rows]
    tick("Loading files", len(rows), len(rows))
    missing = [(int(r.get("file_id", r.get("id"))), str(r["path"])) for r in rows
               if r.get("phash_u64") is None and r.get("path")]
    if missing:
        filled = fastsig.fast_fill_missing_signatures(
            db_path or ":memory:", missing, max_workers=8, chunksize=64, apply_to_db=db_path is not None,
            progress=lambda d, t: tick("Computing signatures", d, t), cancel_fn=cancel_fn, device=device)
        by_id = {fid: ph for fid, ph, _ in filled}
        for r in rows:
            fid = int(r.get("file_id", r.get("id")))
            if r.get("phash_u64") is None and fid in by_id:
                r["phash_u64"] = by_id[fid]
    if cancel_fn and cancel_fn():
        return []
    tick("Building groups", 0, len(rows))
    files = []
    for r in rows:
        try:
            files.append(DuplicateFile.from_row(r))
        except ValueError:
            continue
    tick("Clustering duplicates", 0, len(files))
    return DuplicateScanner(config or DuplicateScanConfig(), device=device).build_clusters(files)


__all__ = ["DedupSettings", "compute_signature", "find_duplicates", "run_duplicate_scan", "ClusterBuilder"]
