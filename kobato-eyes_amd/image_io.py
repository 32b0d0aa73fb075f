"""Defensive image loader: the decode + normalisation step in front of the refine stage.

Mirrors what the reference's ``utils.image_io.safe_load_image`` (src/utils/image_io.py:60-151) decides about the
pixels a file contributes -- the step *before* the hot path:

* a pixel-count cap while opening (decompression-bomb guard), with the reference's two escape hatches: skip
  outright above ``hard_skip_pixels`` (judged from the header), and either skip or re-open without the cap
  when Pillow raises ``DecompressionBombError``;
* JPEG draft mode towards ``max_side`` so huge JPEGs decode reduced;
* truncated files are decoded as far as they go;
* EXIF orientation applied;
* anything larger than ``max_side`` is shrunk in place with ``thumbnail(..., LANCZOS)``;
* ``rgb=True``: alpha (RGBA, LA, or a palette ``transparency`` entry) is composited over white, everything
  else is converted to RGB;
* unreadable input returns ``None`` instead of raising.
"""
from __future__ import annotations

import logging
from contextlib import suppress
from pathlib import Path
from typing import Optional

logger = logging.getLogger(__name__)

DEFAULT_BOMB_CAP = 350_000_000
DEFAULT_MAX_SIDE = 4096
MAX_SIDE = DEFAULT_MAX_SIDE


def _to_rgb(image):
    from PIL import Image

    if image.mode == "RGB":
        return image
    has_alpha = image.mode in ("RGBA", "LA") or "transparency" in getattr(image, "info", {})
    if not has_alpha:
        return image.convert("RGB")
    rgba = image.convert("RGBA")
    canvas = Image.new("RGBA", rgba.size, "WHITE")
    canvas.alpha_composite(rgba)
    return canvas.convert("RGB")


def safe_load_image(source, *, max_side: int = DEFAULT_MAX_SIDE, bomb_pixel_cap: Optional[int] = DEFAULT_BOMB_CAP,
                    hard_skip_pixels: Optional[int] = None, rgb: bool = True, skip_on_bomb: bool = False):
    """PIL image (RGB when ``rgb``) or ``None``; never raises for unreadable or oversized files."""
    try:
        from PIL import Image, ImageFile, ImageOps, UnidentifiedImageError
        from PIL.Image import DecompressionBombError
    except ModuleNotFoundError as exc:  # pragma: no cover
        raise RuntimeError("Pillow is required to decode image files") from exc

    name = str(source)
    saved_cap = Image.MAX_IMAGE_PIXELS
    if bomb_pixel_cap is not None:
        Image.MAX_IMAGE_PIXELS = int(bomb_pixel_cap)
    try:
        img = Image.open(name)                     # header only so far
        width, height = img.size
        pixels = (width or 0) * (height or 0)
        if hard_skip_pixels is not None and pixels > hard_skip_pixels:
            logger.warning("Skip very large image (header %dx%d ~%d px): %s", width, height, pixels, name)
            with suppress(Exception):
                img.close()
            return None
        with suppress(Exception):
            img.draft("RGB", (max_side, max_side))
        ImageFile.LOAD_TRUNCATED_IMAGES = True
        try:
            img.load()
        except DecompressionBombError as exc:
            logger.warning("Huge image detected (bomb): %s (%s)", name, exc)
            with suppress(Exception):
                img.close()
            if skip_on_bomb:
                return None
            Image.MAX_IMAGE_PIXELS = None          # let it through: re-open without the cap
            img = Image.open(name)
            with suppress(Exception):
                img.draft("RGB", (max_side, max_side))
            img.load()
        except MemoryError:
            logger.error("MemoryError while decoding (header %dx%d ~%d px): %s", width, height, pixels, name)
            with suppress(Exception):
                img.close()
            return None
        try:
            turned = ImageOps.exif_transpose(img)
        except (AttributeError, TypeError, ValueError):
            turned = img
        if turned is not img:
            with suppress(Exception):
                img.close()
            img = turned
        if max(img.size) > max_side:
            img.thumbnail((max_side, max_side), Image.Resampling.LANCZOS)
        if rgb and img.mode != "RGB":
            converted = _to_rgb(img)
            if converted is not img:
                with suppress(Exception):
                    img.close()
            img = converted
        return img
    except (UnidentifiedImageError, OSError) as exc:
        logger.warning("safe_load_image failed for %s: %s", name, exc)
        return None
    finally:
        Image.MAX_IMAGE_PIXELS = saved_cap


def load_rgb(source, *, max_side: int = DEFAULT_MAX_SIDE):
    """RGB image ready for the refine kernels, or ``None``."""
    return safe_load_image(Path(source), max_side=max_side)
