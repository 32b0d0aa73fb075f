#!/usr/bin/env python3
"""Randomised damage to JPEG (and, with --png, PNG) files (test infrastructure): whatever the decoder still TAKES must carry
the pixels Pillow makes of the same bytes -- never other pixels, and never a file Pillow gives up on.  The kinds of damage:
  header : one byte anywhere from the first marker to just behind the first scan header (marker codes, segment lengths,
           sampling factors, table ids, Huffman / quantisation tables ...)
  fields : (JPEG) a sampling factor or a table number of the frame header set to another plausible value
  body   : in the entropy-coded data and the later scans: flipped bits, overwritten runs, deleted and inserted bytes,
           truncation (restart markers lost or renumbered, scans cut short ...)
The statuses may differ from Pillow's verdict in one direction only: refusing (-> the caller lets Pillow decide) is always right.
(PNG files: `header` = one byte anywhere in the file changed, `body` = the same edits anywhere behind the signature.)
    python tests/fuzz_jpeg_damage.py [variants per file] [seed] [--gpu] [--png]
Without --gpu the CPU build of the decoder's headers (oracle/libkeyes_jpeg_cpu.so) is exercised; with it the kernels, one batch
per kind.  Exits non-zero on the first mismatch."""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image, ImageFile

import _jpeg_cases as J
import _png_cases as P


def damaged(rng, files, variants, kind, fmt="jpeg"):
    """Yields (name, what was done, damaged bytes)."""
    for name, data, _ in files:
        sos = data.index(b"\xff\xda") if fmt == "jpeg" else 8
        for v in range(variants):
            d = bytearray(data)
            if kind == "fields":                            # the frame header's sampling factors and table numbers, plausible values
                sof = max(data.find(b"\xff\xc0"), data.find(b"\xff\xc2"))
                pos = sof + 11 + 3 * int(rng.integers(0, data[sof + 9])) + int(rng.integers(0, 2))
                d[pos] = int(rng.choice([0x11, 0x12, 0x21, 0x22, 0x14, 0x41, 0x44, 0x13, 0x31, 0x24, 0x42, 0, 1, 2, 3]))
                what = f"frame byte {pos - sof} = {d[pos]:#x}"
            elif kind == "header":
                pos = int(rng.integers(2, sos + 14)) if fmt == "jpeg" else int(rng.integers(8, len(d)))
                d[pos] = (d[pos] + int(rng.integers(1, 256))) & 255
                what = f"byte {pos}"
            else:
                how = v % 5
                pos = int(rng.integers(sos, len(d)))
                if how == 0:
                    d[pos] ^= 1 << int(rng.integers(0, 8))
                elif how == 1:
                    k = int(rng.integers(1, 24))
                    d[pos:pos + k] = rng.integers(0, 256, len(d[pos:pos + k]), dtype=np.uint8).tobytes()
                elif how == 2:
                    del d[pos:pos + int(rng.integers(1, 6))]
                elif how == 3:
                    d[pos:pos] = rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8).tobytes()
                else:
                    d = d[:pos]
                what = f"{('flip', 'overwrite', 'delete', 'insert', 'cut')[how]} at {pos} of {len(data)}"
            yield name, what, bytes(d)


def pillow_pixels(blob):
    """Pillow's strict decode (LOAD_TRUNCATED_IMAGES off, as in the batch hasher's worker processes) or None.  Palette, bilevel
    sub-byte gray and gray + alpha PNGs as the hashes see them: convert("L") (what ke_png_decode yields for those)."""
    saved, ImageFile.LOAD_TRUNCATED_IMAGES = ImageFile.LOAD_TRUNCATED_IMAGES, False
    try:
        im = Image.open(io.BytesIO(blob))
        im.load()
        if im.format == "PNG" and (im.mode in ("P", "1", "LA", "I;16") or (im.mode == "L" and blob[24] < 8)):
            return np.asarray(im.convert("L"))
        return np.asarray(im)
    except Exception:
        return None
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = saved


def check(decode_batch, variants, seed, files=40, fmt="jpeg"):
    """decode_batch([bytes]) -> ([pixels | None], [status]).  Returns (cases, taken, mismatches as text)."""
    rng = np.random.default_rng(seed)
    if fmt == "jpeg":
        pool = [c for c in J.supported() if c[2].shape[0] >= 16 and c[2].shape[1] >= 16]
    else:
        pool = [c for c in list(P.supported()) + list(P.handmade()) + list(P.mapped()) + list(P.interlaced()) + list(P.wide()) + [c for c in P.animated() if c[2] is not None] if c[2].shape[0] >= 8]
    pool = [pool[i] for i in rng.choice(len(pool), min(files, len(pool)), replace=False)]
    cases = taken = 0
    wrong = []
    for kind in ("header", "body") + (("fields",) if fmt == "jpeg" else ()):
        batch = list(damaged(rng, pool, variants, kind, fmt))
        out, status = decode_batch([b for _, _, b in batch])
        for (name, what, blob), px, st in zip(batch, out, status):
            cases += 1
            if st != 0:
                continue
            taken += 1
            ref = pillow_pixels(blob)
            if ref is None or ref.shape != px.shape or not np.array_equal(ref, px):
                wrong.append(f"{kind}: {name}, {what}: " + ("Pillow refuses the file" if ref is None else "other pixels than Pillow's"))
    return cases, taken, wrong


def cpu_decoder(fmt="jpeg"):
    if fmt == "jpeg":
        import test_jpeg_cpu as T
    else:
        import test_png_cpu as T
    lib = T._lib()

    def decode_batch(blobs):
        res = [T._decode(lib, b) for b in blobs]
        return [r[1] for r in res], [r[0] for r in res]
    return decode_batch


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    variants = int(args[0]) if args else 100
    seed = int(args[1]) if len(args) > 1 else 1
    fmt = "png" if "--png" in sys.argv else "jpeg"
    if "--gpu" in sys.argv:
        from kobato_eyes_amd import _native

        decode = getattr(_native.get_context(0), f"{fmt}_decode")
    else:
        decode = cpu_decoder(fmt)
    cases, taken, wrong = check(decode, variants, seed, fmt=fmt)
    print(f"{cases} damaged files, {taken} taken, {len(wrong)} mismatches")
    for line in wrong[:20]:
        print("  " + line)
    sys.exit(1 if wrong else 0)
