"""Batch signature fill on the MI355X: drop-in for the reference's ``core.fastsig``.

Public names and contracts follow src/core/fastsig.py -- ``fast_fill_missing_signatures`` (:102-126),
``compute_signatures_mp`` (:65-99), ``bulk_upsert_signatures`` (:48-62), ``_to_signed64`` (:19-21): results come back in
input order with failed files left out (:36-37), progress fires every 200 files and at the end (:93), a true
``cancel_fn`` stops the run and hands back what is finished (:86-90), values are stored as signed 64-bit, and the
optional "unsafe fast" PRAGMAs are the reference's (:40-45).

The machinery is different: where the reference fans single files out to a spawn-context process pool, this is a
two-stage pipeline -- Pillow decodes chunk k+1 on a thread pool (it releases the GIL) while the GPU hashes chunk k
with one ``ke_hash_images`` call per channel count.
"""
from __future__ import annotations

import importlib
import os
import sqlite3
from concurrent.futures import Future, ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash

U64MASK = (1 << 64) - 1
PROGRESS_STRIDE = 200
FAST_PRAGMAS = ("journal_mode=WAL", "synchronous=OFF", "temp_store=MEMORY", "mmap_size=30000000000")
UPSERT_SQL = ("INSERT INTO signatures (file_id, phash_u64, dhash_u64) VALUES (?, ?, ?) "
              "ON CONFLICT(file_id) DO UPDATE SET phash_u64 = excluded.phash_u64, dhash_u64 = excluded.dhash_u64")

Task = Tuple[int, str]
Row = Tuple[int, int, int]


def _to_signed64(x: int) -> int:
    v = int(x) & U64MASK
    return v - (1 << 64) if v >> 63 else v


def _read_pixels(path_text: str):
    """Pixels of one file, or None for anything that cannot be opened (speed first: no reason kept)."""
    try:
        from PIL import Image

        path = Path(path_text)
        if not path.is_file():
            return None
        with Image.open(path) as im:
            return _phash.image_to_array(im)
    except Exception:
        return None


class _Pipeline:
    """decode (threads) -> hash (GPU), one chunk in flight on each side."""

    def __init__(self, tasks: Sequence[Task], workers: int, chunk: int, device: int) -> None:
        self.tasks, self.chunk, self.device = tasks, max(1, int(chunk)), device
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))

    def _submit(self, start: int) -> List[Future]:
        return [self.pool.submit(_read_pixels, p) for _, p in self.tasks[start:start + self.chunk]]

    def _hash(self, pixels: list) -> list:
        """Per decoded slot: (phash_s64, dhash_s64) or None."""
        live = [k for k, a in enumerate(pixels) if a is not None and a.size > 0]
        out: list = [None] * len(pixels)
        if live:
            ph, dh, ok = _phash.hash_batch([pixels[k] for k in live], want_dhash=True, device=self.device)
            for k, p, d, good in zip(live, ph.tolist(), dh.tolist(), ok.tolist()):
                if good:
                    out[k] = (_to_signed64(p), _to_signed64(d))
        return out

    def run(self) -> Iterator[Tuple[int, Optional[Tuple[int, int]]]]:
        """Yields (file_id, hashes | None) in task order."""
        try:
            pending = self._submit(0)
            for start in range(0, len(self.tasks), self.chunk):
                ahead = self._submit(start + self.chunk) if start + self.chunk < len(self.tasks) else []
                hashed = self._hash([f.result() for f in pending])     # next chunk decodes meanwhile
                for (fid, _), sig in zip(self.tasks[start:start + self.chunk], hashed):
                    yield int(fid), sig
                pending = ahead
        finally:
            self.pool.shutdown(wait=False, cancel_futures=True)


def compute_signatures_mp(tasks: List[Task], *, max_workers: Optional[int] = None, chunksize: int = 64,
                          progress: Optional[Callable[[int, int], None]] = None,
                          cancel_fn: Optional[Callable[[], bool]] = None, device: int = 0) -> List[Row]:
    """[(file_id, path)] -> [(file_id, phash_s64, dhash_s64)], input order, failures omitted."""
    total = len(tasks)
    rows: List[Row] = []
    if total == 0:
        return rows
    workers = max_workers or max(1, (os.cpu_count() or 4) - 1)
    for seen, (fid, sig) in enumerate(_Pipeline(tasks, workers, chunksize, device).run(), 1):
        if cancel_fn is not None and cancel_fn():
            break                                   # the generator's finally clause cancels what is queued
        if sig is not None:
            rows.append((fid, sig[0], sig[1]))
        if progress is not None and (seen % PROGRESS_STRIDE == 0 or seen == total):
            try:
                progress(seen, total)
            except Exception:
                pass
    return rows


def bulk_upsert_signatures(conn: sqlite3.Connection, rows: Iterable[Row]) -> int:
    """One executemany upsert into signatures(file_id, phash_u64, dhash_u64); returns the row count."""
    payload = [(int(fid), _to_signed64(ph), _to_signed64(dh)) for fid, ph, dh in rows]
    if not payload:
        return 0
    with conn:
        return conn.executemany(UPSERT_SQL, payload).rowcount or 0


def fast_fill_missing_signatures(db_path: str, items: List[Task], *, max_workers: Optional[int] = None, chunksize: int = 64,
                                 progress: Optional[Callable[[int, int], None]] = None, apply_to_db: bool = True,
                                 unsafe_fast: bool = True, cancel_fn: Optional[Callable[[], bool]] = None,
                                 device: int = 0) -> List[Row]:
    """Hash the files that have no signature yet and (optionally) upsert them; returns the computed rows."""
    rows = compute_signatures_mp(items, max_workers=max_workers, chunksize=chunksize, progress=progress, cancel_fn=cancel_fn,
                                 device=device)
    if rows and apply_to_db:
        with sqlite3.connect(db_path) as conn:
            if unsafe_fast:
                for pragma in FAST_PRAGMAS:
                    conn.execute(f"PRAGMA {pragma}")
            bulk_upsert_signatures(conn, rows)
    return rows
