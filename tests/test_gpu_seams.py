"""GPU tests of the seams against END-TO-END outputs of the reference itself (tests/golden/make_golden.py --only-seams):

* BASELINE configs[0] whole -- 1 000 synthetic 256 x 256 files -> fast_fill_missing_signatures -> from_row ->
  DuplicateScanner.build_clusters (the call sequence of src/ui/dup_workers.py:148-239) == what the reference's phash / dhash
  (src/sig/phash.py:33-57) and build_clusters (src/dup/scanner.py:211-356) returned for the same images;
* the reference's scanner on a 100 000-hash table (BASELINE configs[1] size) by digest;
* compute_signatures_mp over a corpus of real files of every format family == the rows of the reference's
  _compute_worker (src/core/fastsig.py:24-37), by the GPU decoders and by the Pillow route.
"""
from __future__ import annotations

import os
import sqlite3

import numpy as np
import pytest

import _golden as G

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def K():
    import kobato_eyes_amd

    kobato_eyes_amd._native.get_context(0)
    return kobato_eyes_amd


def _listing(clusters):
    return [{"keeper_id": c.keeper_id, "entries": [[e.file.file_id, e.best_hamming] for e in c.files]} for c in clusters]


@pytest.mark.parametrize("dispatch", ["single_pass_forced", "default"])
def test_config0_arrays_hash_scan_cluster_equal_the_reference(K, monkeypatch, dispatch):
    g = G.config0_golden()
    ctx = K._native.get_context(0)
    n, side = g["n"], g["side"]
    if dispatch == "default":                       # the library's own choice of kernels (tests/conftest.py lifts it otherwise)
        monkeypatch.delenv("KE_FUSED_MIN_IMAGES", raising=False)
    px = ctx.synth_rgb(20260604, 0, n, side, side)
    ph, dh = ctx.hash_uniform(px, n, side, side, 3)
    assert np.asarray(ph, np.uint64).view(np.int64).tolist() == g["phash_s64"]
    assert np.asarray(dh, np.uint64).view(np.int64).tolist() == g["dhash_s64"]
    rows = [dict(r, phash_u64=p) for r, p in zip(g["rows"], g["phash_s64"])]
    files = [K.DuplicateFile.from_row(r) for r in rows]
    scanner = K.DuplicateScanner(K.DuplicateScanConfig(hamming_threshold=8))
    edges = scanner.candidate_edges(files)
    assert sorted([min(e.file_id_a, e.file_id_b), max(e.file_id_a, e.file_id_b), e.hamming] for e in edges.values()) == g["edges"]
    assert _listing(scanner.build_clusters(files)) == g["clusters"]


def test_config0_files_to_clusters_equal_the_reference(K, tmp_path):
    """The whole seam sequence on files: PNG (lossless) copies of the corpus images, no hashes in the rows."""
    from oracle import oracle as O

    g = G.config0_golden()
    n, side = g["n"], g["side"]
    rows = []
    for lo in range(0, n, 250):
        batch = O.synth_rgb_batch(lo, 250, side, side)
        for k, r in enumerate(g["rows"][lo:lo + 250]):
            p = tmp_path / r["path"]
            Image.fromarray(batch[k]).save(p, compress_level=1)
            rows.append(dict(r, path=str(p), phash_u64=None))
    db = tmp_path / "k.db"
    conn = sqlite3.connect(db)
    conn.execute("CREATE TABLE signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
    conn.commit()
    conn.close()
    stages = []
    clusters = K.run_duplicate_scan(rows, db_path=str(db), config=K.DuplicateScanConfig(hamming_threshold=8),
                                    progress=lambda s, d, t: stages.append(s))
    assert _listing(clusters) == g["clusters"]
    stored = sqlite3.connect(db).execute("SELECT file_id, phash_u64, dhash_u64 FROM signatures ORDER BY file_id").fetchall()
    assert stored == [(i + 1, p, d) for i, (p, d) in enumerate(zip(g["phash_s64"], g["dhash_s64"]))]
    assert [s for k, s in enumerate(stages) if k == 0 or stages[k - 1] != s] == \
        ["Loading files", "Computing signatures", "Building groups", "Clustering duplicates"]


def test_scan_100k_equals_the_reference_digests(K):
    g = G.scan100k_golden()
    n = g["n"]
    from oracle import oracle as O

    hashes = O.synth_hashes(n)                         # the inputs the fixture was made from (ke_synth_hashes == these: test_gpu_parity)
    from pathlib import Path

    files = [K.DuplicateFile(file_id=i + 1, path=Path(f"img_{i:07d}.png"), size=1000 + (i % 7), width=512, height=512,
                             phash=int(h), embedding=None) for i, h in enumerate(np.asarray(hashes).tolist())]
    for run in g["runs"]:
        scanner = K.DuplicateScanner(K.DuplicateScanConfig(**run["config"]))
        edges = [(e.file_id_a, e.file_id_b, e.hamming) for e in scanner.candidate_edges(files).values()]
        clusters = [[c.keeper_id, [[e.file.file_id, e.best_hamming] for e in c.files]] for c in scanner.build_clusters(files)]
        assert len(edges) == run["n_edges"] and len(clusters) == run["n_clusters"]
        c = scanner.last_counters                           # the reference's own funnel line (src/dup/scanner.py:292-299)
        assert [c["pair_total"], c["after_size"], c["after_ham"], c["after_cosine"]] == run["counters"]
        assert G.scan_listing_digests(edges, clusters) == (run["edges_sha256"], run["clusters_sha256"])


@pytest.mark.parametrize("route", ["gpu_decoders", "pillow_threads", "pillow_threads_truncation_switch_on"])
def test_worker_corpus_rows_equal_the_reference(K, tmp_path, monkeypatch, route):
    from PIL import ImageFile

    tasks, expected = G.write_worker_corpus(tmp_path)
    # the reference's workers are fresh processes: whatever the calling process did to Pillow's global truncation switch
    # (safe_load_image leaves it on, src/utils/image_io.py:86) does not reach them
    monkeypatch.setattr(ImageFile, "LOAD_TRUNCATED_IMAGES", route.endswith("switch_on"))
    monkeypatch.setenv("KE_DECODE_PROCESS_MIN", "100000")            # threads, also with the switch on
    if route.startswith("pillow_threads"):
        monkeypatch.setenv("KE_GPU_JPEG", "0")
        monkeypatch.setenv("KE_GPU_PNG", "0")
    seen = []
    rows = K.compute_signatures_mp(tasks, max_workers=4, chunksize=16, progress=lambda d, t: seen.append((d, t)))
    assert [r[0] for r in rows] == [r[0] for r in expected]                  # the same files dropped, input order kept
    by_name = {fid: os.path.basename(p) for fid, p in tasks}
    wrong = [by_name[a[0]] for a, b in zip(rows, expected) if tuple(a) != tuple(b)]
    assert not wrong, wrong
    assert seen[-1] == (len(tasks), len(tasks))
