"""Minimal defensive image loader for the refine stage.

The reference's ``utils.image_io.safe_load_image`` (src/utils/image_io.py:60-138) is the step
*before* the hot path ("next"); this module keeps only what decides the pixels the
SSIM kernel sees: draft-decode hint, EXIF transpose, LANCZOS thumbnail to <= 4096 px, alpha
composited over white, RGB out, ``None`` for anything unreadable.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Optional

logger = logging.getLogger(__name__)
MAX_SIDE = 4096


def load_rgb(source, *, max_side: int = MAX_SIDE):
    try:
        from PIL import Image, ImageFile, ImageOps, UnidentifiedImageError
    except ModuleNotFoundError as exc:  # pragma: no cover
        raise RuntimeError("Pillow is required to decode image files") from exc
    try:
        img = Image.open(str(source))
        try:
            img.draft("RGB", (max_side, max_side))
        except Exception:
            pass
        ImageFile.LOAD_TRUNCATED_IMAGES = True
        img.load()
        try:
            img = ImageOps.exif_transpose(img)
        except (AttributeError, TypeError, ValueError):
            pass
        if max(img.size) > max_side:
            img.thumbnail((max_side, max_side), Image.Resampling.LANCZOS)
        if img.mode != "RGB":
            if img.mode in ("RGBA", "LA") or "transparency" in getattr(img, "info", {}):
                rgba = img.convert("RGBA")
                canvas = Image.new("RGBA", rgba.size, "WHITE")
                canvas.alpha_composite(rgba)
                img = canvas.convert("RGB")
            else:
                img = img.convert("RGB")
        return img
    except (UnidentifiedImageError, OSError, Image.DecompressionBombError) as exc:
        logger.warning("load_rgb failed for %s: %s", source, exc)
        return None
