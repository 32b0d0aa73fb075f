"""Headless duplicate scan over a kobato-eyes SQLite database.

    python -m kobato_eyes_amd.cli scan-dups --db kobato.db --hamming 8 [--like '%/photos/%'] [--size-ratio 0.5]
                                            [--refine] [--csv out.csv]

Reproduces the call sequence of DuplicateScanRunnable.run (src/ui/dup_workers.py:148-239) without Qt: rows in
the shape of db.repository.iter_files_for_dup (src/db/repository.py:416-454), missing signatures filled and
upserted as core.fastsig does (src/core/fastsig.py:102-126), DuplicateFile.from_row with bad rows skipped,
build_clusters, optionally the shipped refine pipeline (tile aHash max_bits, then pixel MAE 0.004 as
src/ui/dup_tab.py:302-311 wires it), and the CSV layout of export_duplicate_clusters_csv
(src/ui/file_actions.py:61-80).
"""
from __future__ import annotations

import argparse
import csv
import logging
import os
import sqlite3
import sys
from typing import Iterator, Optional, Sequence

CSV_HEADER = ["group", "file_id", "path", "size", "width", "height", "keeper", "hamming"]


def iter_files_for_dup(conn: sqlite3.Connection, path_like: Optional[str]) -> Iterator[dict]:
    """Plain dicts with file_id, path, size, width, height, phash_u64 for every present file, by id."""
    sql = ("SELECT f.id AS file_id, f.path AS path, COALESCE(f.size, 0) AS size, f.width AS width, f.height AS height, "
           "s.phash_u64 AS phash_u64 FROM files f LEFT JOIN signatures s ON s.file_id = f.id WHERE f.is_present = 1")
    params: list = []
    if path_like:
        sql += " AND f.path LIKE ? ESCAPE '\\'"
        params.append(path_like)
    sql += " ORDER BY f.id"
    for row in conn.execute(sql, params):
        yield {"file_id": row[0], "path": row[1], "size": row[2], "width": row[3], "height": row[4], "phash_u64": row[5]}


def export_duplicate_clusters_csv(clusters: Sequence, file_path) -> None:
    with open(file_path, "w", encoding="utf-8", newline="") as handle:
        writer = csv.writer(handle)
        writer.writerow(CSV_HEADER)
        for group, cluster in enumerate(clusters, start=1):
            for entry in cluster.files:
                f = entry.file
                writer.writerow([group, f.file_id, f.path.as_posix(), f.size or 0, f.width or 0, f.height or 0,
                                 1 if f.file_id == cluster.keeper_id else 0,
                                 entry.best_hamming if entry.best_hamming is not None else ""])


def scan_database(db_path: str, *, hamming_threshold: int = 8, size_ratio: Optional[float] = None, path_like: Optional[str] = None,
                  refine: bool = False, tile_max_bits: int = 32, mae_thr: float = 0.004, device: int = 0, progress=None) -> list:
    from . import DuplicateFile, DuplicateScanConfig, DuplicateScanner, fast_fill_missing_signatures
    from .refine_parallel import refine_by_pixels_parallel, refine_by_tilehash_parallel

    def tick(stage, done, total):
        if progress:
            progress(stage, done, total)

    conn = sqlite3.connect(db_path)
    try:
        rows = list(iter_files_for_dup(conn, path_like))
    finally:
        conn.close()
    tick("Loading files", len(rows), len(rows))
    missing = [(r["file_id"], r["path"]) for r in rows if r["phash_u64"] is None and r["path"]]
    if missing:
        workers = int(os.environ.get("KE_SIG_WORKERS", "8"))
        chunk = int(os.environ.get("KE_SIG_CHUNK", "64"))
        filled = fast_fill_missing_signatures(db_path, missing, max_workers=workers, chunksize=chunk,
                                              progress=lambda d, t: tick("Computing signatures", d, t), device=device)
        by_id = {fid: ph for fid, ph, _ in filled}
        for r in rows:
            if r["phash_u64"] is None:
                r["phash_u64"] = by_id.get(r["file_id"])
    tick("Building groups", 0, len(rows))
    files = []
    for r in rows:
        try:
            files.append(DuplicateFile.from_row(r))
        except ValueError:
            continue                                   # rows still without a hash are skipped, as the reference does
    tick("Clustering duplicates", 0, len(files))
    clusters = DuplicateScanner(DuplicateScanConfig(hamming_threshold=hamming_threshold, size_ratio=size_ratio),
                                device=device).build_clusters(files)
    if refine and clusters:
        clusters = refine_by_tilehash_parallel(clusters, max_bits=tile_max_bits, device=device,
                                               tick=lambda d, t, phase: tick(f"Refining (tile hash {phase}/2)", d, t))
        clusters = refine_by_pixels_parallel(clusters, mae_thr=mae_thr, device=device,
                                             tick=lambda d, t: tick("Refining (pixels)", d, t))
    return clusters


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kobato_eyes_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    sp = sub.add_parser("scan-dups", help="cluster near-duplicate files of a kobato-eyes database")
    sp.add_argument("--db", required=True)
    sp.add_argument("--hamming", type=int, default=8)
    sp.add_argument("--size-ratio", type=float, default=None)
    sp.add_argument("--like", default=None, help="SQL LIKE pattern on files.path")
    sp.add_argument("--refine", action="store_true", help="run the tile-aHash + pixel-MAE refine pipeline on the clusters")
    sp.add_argument("--tile-max-bits", type=int, default=32)
    sp.add_argument("--mae-thr", type=float, default=0.004)
    sp.add_argument("--csv", default=None)
    sp.add_argument("--device", type=int, default=0)
    sp.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO if args.verbose else logging.WARNING)
    clusters = scan_database(args.db, hamming_threshold=args.hamming, size_ratio=args.size_ratio, path_like=args.like,
                             refine=args.refine, tile_max_bits=args.tile_max_bits, mae_thr=args.mae_thr, device=args.device)
    if args.csv:
        export_duplicate_clusters_csv(clusters, args.csv)
    members = sum(len(c.files) for c in clusters)
    print(f"{len(clusters)} duplicate group(s), {members} file(s)" + (f" -> {args.csv}" if args.csv else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
