"""Batch signature fill on the MI355X: drop-in for the reference's ``core.fastsig``.

Mirrors src/core/fastsig.py: ``fast_fill_missing_signatures`` / ``compute_signatures_mp`` /
``bulk_upsert_signatures`` / ``_to_signed64`` with the same arguments, progress cadence
(every 200 items and at the end, :93), cancel semantics (partial results, :86-90), silent
per-file failure (:36-37) and the same SQLite upsert with the "unsafe fast" PRAGMAs (:40-62).
Where the reference fans files out to a spawn-context process pool, this decodes on a small
thread pool (Pillow releases the GIL while decoding) and hashes each decoded chunk in one
ke_hash_images call on the GPU.
"""
from __future__ import annotations

import os
import sqlite3
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Iterable, List, Optional, Tuple

import importlib

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash

U64MASK = (1 << 64) - 1
_PROGRESS_EVERY = 200


def _to_signed64(x: int) -> int:
    v = int(x) & U64MASK
    return v - (1 << 64) if v >> 63 else v


def _decode(task: Tuple[int, str]):
    """(file_id, path) -> (file_id, ndarray) or None; what src/core/fastsig.py:28-33 opens."""
    fid, p = task
    try:
        from PIL import Image

        path = Path(p)
        if not path.exists() or not path.is_file():
            return None
        with Image.open(path) as im:
            return int(fid), _phash.image_to_array(im)
    except Exception:
        return None  # failures are dropped, speed first


def _fast_pragmas(conn: sqlite3.Connection) -> None:
    for pragma in ("journal_mode=WAL", "synchronous=OFF", "temp_store=MEMORY", "mmap_size=30000000000"):
        conn.execute(f"PRAGMA {pragma}")


def bulk_upsert_signatures(conn: sqlite3.Connection, rows: Iterable[Tuple[int, int, int]]) -> int:
    """executemany upsert into signatures(file_id, phash_u64, dhash_u64); values stored signed."""
    payload = [(int(fid), _to_signed64(ph), _to_signed64(dh)) for fid, ph, dh in rows]
    if not payload:
        return 0
    with conn:
        cur = conn.executemany(
            "INSERT INTO signatures (file_id, phash_u64, dhash_u64) VALUES (?, ?, ?) "
            "ON CONFLICT(file_id) DO UPDATE SET phash_u64 = excluded.phash_u64, dhash_u64 = excluded.dhash_u64",
            payload,
        )
    return cur.rowcount or 0


def compute_signatures_mp(
    tasks: List[Tuple[int, str]],
    *,
    max_workers: Optional[int] = None,
    chunksize: int = 64,
    progress: Optional[Callable[[int, int], None]] = None,
    cancel_fn: Optional[Callable[[], bool]] = None,
    device: int = 0,
) -> List[Tuple[int, int, int]]:
    """(file_id, path) list -> [(file_id, phash_s64, dhash_s64)] in input order, failures omitted."""
    if not tasks:
        return []
    total = len(tasks)
    done = 0
    results: List[Tuple[int, int, int]] = []
    workers = max_workers or max(1, (os.cpu_count() or 4) - 1)
    chunk = max(1, int(chunksize))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for start in range(0, total, chunk):
            decoded = list(pool.map(_decode, tasks[start:start + chunk]))
            good = [k for k, d in enumerate(decoded) if d is not None]
            hashed: dict[int, tuple[int, int]] = {}      # position in this chunk -> (phash, dhash)
            if good:
                ph, dh, ok = _phash.hash_batch([decoded[k][1] for k in good], want_dhash=True, device=device)
                for k, p, d, flag in zip(good, ph.tolist(), dh.tolist(), ok.tolist()):
                    if flag:
                        hashed[k] = (_to_signed64(p), _to_signed64(d))
            for k, d in enumerate(decoded):
                if cancel_fn and cancel_fn():
                    pool.shutdown(wait=False, cancel_futures=True)
                    return results
                done += 1
                if k in hashed:
                    results.append((d[0],) + hashed[k])
                if progress and (done % _PROGRESS_EVERY == 0 or done == total):
                    try:
                        progress(done, total)
                    except Exception:
                        pass
    return results


def fast_fill_missing_signatures(
    db_path: str,
    items: List[Tuple[int, str]],
    *,
    max_workers: Optional[int] = None,
    chunksize: int = 64,
    progress: Optional[Callable[[int, int], None]] = None,
    apply_to_db: bool = True,
    unsafe_fast: bool = True,
    cancel_fn: Optional[Callable[[], bool]] = None,
    device: int = 0,
) -> List[Tuple[int, int, int]]:
    computed = compute_signatures_mp(items, max_workers=max_workers, chunksize=chunksize, progress=progress,
                                     cancel_fn=cancel_fn, device=device)
    if apply_to_db and computed:
        with sqlite3.connect(db_path) as conn:
            if unsafe_fast:
                _fast_pragmas(conn)
            bulk_upsert_signatures(conn, computed)
    return computed
