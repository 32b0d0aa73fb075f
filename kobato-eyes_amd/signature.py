"""Inline single-image signatures: drop-in for the reference's ``core.signature``
(src/core/signature.py:17-62)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Optional

import importlib

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash
from .fastsig import _to_signed64
from .image_io import load_rgb

log = logging.getLogger(__name__)

_UPSERT = ("INSERT INTO signatures (file_id, phash_u64, dhash_u64) VALUES (?, ?, ?) "
           "ON CONFLICT(file_id) DO UPDATE SET phash_u64 = excluded.phash_u64, dhash_u64 = excluded.dhash_u64")


def compute_signatures_from_image(im, *, device: int = 0) -> tuple[int, int]:
    """(signed pHash, signed dHash); exceptions propagate to the caller (:24-28)."""
    p, d = _phash.phash_dhash(im, device=device)
    return _to_signed64(p), _to_signed64(d)


def upsert_signatures(conn, *, file_id: int, phash_u64: int, dhash_u64: int) -> None:
    """Same statement as src/db/repository.py:257-267."""
    with conn:
        conn.execute(_UPSERT, (int(file_id), int(phash_u64), int(dhash_u64)))


def ensure_signatures(conn, file_id: int, *, image=None, path: Optional[str | Path] = None, force: bool = False) -> bool:
    """Compute + store unless a row exists (``force`` recomputes).  Never raises: False on failure."""
    try:
        if not force and conn.execute("SELECT 1 FROM signatures WHERE file_id=? LIMIT 1", (file_id,)).fetchone() is not None:
            return True
        if image is None:
            if path is None:
                return False
            image = load_rgb(Path(path))
            if image is None:
                return False
        p, d = compute_signatures_from_image(image)
        upsert_signatures(conn, file_id=file_id, phash_u64=p, dhash_u64=d)
        return True
    except Exception as exc:
        log.warning("ensure_signatures failed for %s: %s", path or f"file_id={file_id}", exc)
        return False
