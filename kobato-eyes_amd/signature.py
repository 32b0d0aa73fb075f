"""Inline single-image signatures: drop-in for the reference's ``core.signature`` (src/core/signature.py:17-62).

``compute_signatures_from_image`` returns the signed (pHash, dHash) pair of one image (both from a single upload to the
GPU); ``ensure_signatures`` stores it unless a row already exists, and reports success as a bool instead of raising.
"""
from __future__ import annotations

import importlib
import logging
from pathlib import Path
from typing import Optional

from .fastsig import UPSERT_SQL, _to_signed64
from .image_io import load_rgb

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash
log = logging.getLogger(__name__)


def compute_signatures_from_image(im, *, device: int = 0) -> tuple[int, int]:
    """Errors (no GPU, empty image ...) propagate: the callers decide what a failure means."""
    return tuple(_to_signed64(v) for v in _phash.phash_dhash(im, device=device))


def upsert_signatures(conn, *, file_id: int, phash_u64: int, dhash_u64: int) -> None:
    """The statement of src/db/repository.py:257-267, one row."""
    with conn:
        conn.execute(UPSERT_SQL, (int(file_id), int(phash_u64), int(dhash_u64)))


def _has_row(conn, file_id: int) -> bool:
    return conn.execute("SELECT 1 FROM signatures WHERE file_id=? LIMIT 1", (file_id,)).fetchone() is not None


def ensure_signatures(conn, file_id: int, *, image=None, path: Optional[str | Path] = None, force: bool = False,
                      device: int = 0) -> bool:
    """True when a signature row exists afterwards (kept or freshly written), False on any failure.
    ``force`` recomputes even when a row is present."""
    try:
        if not force and _has_row(conn, file_id):
            return True
        source = image if image is not None else (load_rgb(Path(path)) if path is not None else None)
        if source is None:
            return False
        p, d = compute_signatures_from_image(source, device=device)
        upsert_signatures(conn, file_id=file_id, phash_u64=p, dhash_u64=d)
        return True
    except Exception as exc:
        log.warning("ensure_signatures failed for %s: %s", path or f"file_id={file_id}", exc)
        return False
