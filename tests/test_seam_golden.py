"""CPU tests: the oracle and the host-side normalisation against the reference-run END-TO-END fixtures
(tests/golden/make_golden.py --only-seams): BASELINE configs[0] whole (1 000 x 256^2 -> hashes -> clusters), the reference's
scanner on a 100 000-hash table (digests), and the reference's batch worker over a corpus of real files of every format
family (src/core/fastsig.py:24-37)."""
from __future__ import annotations

import io
import os
import warnings

import numpy as np
import pytest

import _golden as G
from oracle import oracle as O


def test_config0_whole_oracle_equals_reference():
    g = G.config0_golden()
    n, side = g["n"], g["side"]
    ph = np.empty(n, np.uint64)
    dh = np.empty(n, np.uint64)
    for lo in range(0, n, 250):
        p, d = O.hash_batch(O.synth_rgb_batch(lo, 250, side, side))
        ph[lo:lo + 250], dh[lo:lo + 250] = p, d
    assert ph.view(np.int64).tolist() == g["phash_s64"]          # src/sig/phash.py:33-46 (SciPy DCT stand-in), signed wrap
    assert dh.view(np.int64).tolist() == g["dhash_s64"]          # src/sig/phash.py:49-57
    ids = np.array([r["file_id"] for r in g["rows"]], np.int64)
    sizes = np.array([r["size"] for r in g["rows"]], np.int64)
    edges, counters = O.scan_banded(ph, ids, sizes, threshold=8)
    got = sorted([int(min(ids[e["a"]], ids[e["b"]])), int(max(ids[e["a"]], ids[e["b"]])), int(e["h"])] for e in edges)
    assert got == g["edges"]
    assert [int(c) for c in counters] == g["counters"][:3]
    clusters = O.assemble_clusters(g["rows"], got)
    assert clusters == [(c["keeper_id"], [tuple(e) for e in c["entries"]]) for c in g["clusters"]]


def test_scan_100k_oracle_equals_reference_digests():
    g = G.scan100k_golden()
    n = g["n"]
    hashes = O.synth_hashes(n)
    ids = np.arange(1, n + 1, dtype=np.int64)
    sizes = 1000 + (np.arange(n, dtype=np.int64) % 7)
    files = [{"file_id": i + 1, "path": f"img_{i:07d}.png", "size": 1000 + (i % 7), "width": 512, "height": 512} for i in range(n)]
    for run in g["runs"]:
        cfg = run["config"]
        edges, counters = O.scan_banded(hashes, ids, sizes, threshold=cfg["hamming_threshold"], size_ratio=cfg.get("size_ratio"))
        triples = [(int(ids[e["a"]]), int(ids[e["b"]]), int(e["h"])) for e in edges]
        assert len(triples) == run["n_edges"]
        assert [int(c) for c in counters] == run["counters"][:3]
        clusters = [[k, [list(e) for e in es]] for k, es in O.assemble_clusters(files, triples)]
        assert len(clusters) == run["n_clusters"] and clusters[:5] == run["first_clusters"]
        assert G.scan_listing_digests(triples, clusters) == (run["edges_sha256"], run["clusters_sha256"])


def test_worker_corpus_host_normalisation_plus_oracle_equals_reference(tmp_path):
    """What the build's Pillow route does on the host (fastsig._read_pixels: Image.open -> image_to_array -- L / RGB / RGBA
    untouched, every other mode through convert("L") -- strict about truncation whatever Pillow's process-wide switch says)
    followed by the oracle's hashes == the reference worker's row for every file of the corpus; files the reference drops fail
    to open or decode here as well.  Run once with the switch as it is and once with it on (safe_load_image leaves it on)."""
    pytest.importorskip("PIL.Image")
    from PIL import ImageFile

    from kobato_eyes_amd import fastsig

    tasks, expected = G.write_worker_corpus(tmp_path)
    rows = {r[0]: r for r in expected}
    saved = ImageFile.LOAD_TRUNCATED_IMAGES
    try:
        for switch in (False, True):
            ImageFile.LOAD_TRUNCATED_IMAGES = switch
            n_rows = 0
            for fid, path in tasks:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    arr = fastsig._read_pixels(path)
                got = None
                if arr is not None and arr.size:
                    p, d = O.hash_image(arr)
                    got = (fid, O.to_signed64(p), O.to_signed64(d))
                assert got == rows.get(fid), (os.path.basename(path), switch)
                n_rows += got is not None
            assert n_rows >= 60 and ImageFile.LOAD_TRUNCATED_IMAGES == switch      # the switch is put back as it was found
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = saved
