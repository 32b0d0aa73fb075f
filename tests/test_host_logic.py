"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
drop-in dataclasses / row parsing / cluster assembly reproduce the reference's goldens, the
product refuses to run without a GPU, and the N>1 sharding logic works under gloo."""
from __future__ import annotations

import os
import re
import sqlite3
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def K():
    if not os.path.exists(os.path.join(ROOT, "kobato-eyes_amd", "libkeyes_hip.so")):
        subprocess.check_call(["bash", os.path.join(ROOT, "kobato-eyes_amd", "build.sh")])
    import kobato_eyes_amd

    return kobato_eyes_amd


def test_library_exports_every_declared_symbol(K):
    header = open(os.path.join(ROOT, "include", "keyes.h")).read()
    declared = set(re.findall(r"\b(ke_[a-z0-9_]+)\s*\(", header)) - {"ke_ctx"}
    lib = K._native.load_library()
    assert declared == set(K._native.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ke_abi_version() == 1


def test_host_side_file_batching_needs_no_gpu(K, tmp_path):
    """ke_host_read_files / ke_host_pack are plain host code (threads, read(2), memcpy): a batch of paths lands back to back
    in the caller's buffer, unreadable or empty files with size 0, and a buffer that is too small is reported, not overrun."""
    import ctypes as C

    lib = K._native.load_library()
    rng = np.random.default_rng(7)
    blobs = [rng.integers(0, 256, int(k), dtype=np.uint8).tobytes() for k in rng.integers(1, 5000, 700)]
    paths = []
    for k, b in enumerate(blobs):
        p = tmp_path / f"b{k}.bin"
        p.write_bytes(b)
        paths.append(str(p))
    (tmp_path / "empty.bin").write_bytes(b"")
    paths[100:100] = [str(tmp_path / "missing.bin"), str(tmp_path / "empty.bin"), str(tmp_path)]      # a directory, too
    blobs[100:100] = [b"", b"", b""]
    n = len(paths)
    names = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    off, size, needed = np.zeros(n, np.uint64), np.zeros(n, np.uint64), C.c_uint64(0)
    rc = lib.ke_host_read_files(names, n, None, 0, off.ctypes.data, size.ctypes.data, C.byref(needed))
    total = sum(len(b) for b in blobs)
    assert rc == -4 and needed.value == total + 64 and size.tolist() == [len(b) for b in blobs]
    small = np.full(total, 0xAB, np.uint8)
    assert lib.ke_host_read_files(names, n, small.ctypes.data, total, off.ctypes.data, size.ctypes.data, C.byref(needed)) == -4
    assert (small == 0xAB).all()
    buf = np.full(total + 64, 0xAB, np.uint8)
    import resource

    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    resource.setrlimit(resource.RLIMIT_NOFILE, (min(soft, 128), hard))      # far fewer descriptors than files in the batch
    try:
        assert lib.ke_host_read_files(names, n, buf.ctypes.data, total + 64, off.ctypes.data, size.ctypes.data, C.byref(needed)) == 0
    finally:
        resource.setrlimit(resource.RLIMIT_NOFILE, (soft, hard))
    assert bytes(buf[:total]) == b"".join(blobs) and not buf[total:].any()
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(b) for b in blobs])[:-1]]).tolist()
    # the same bytes from buffers already in memory
    keep = [b for b in blobs if b]
    sizes = np.array([len(b) for b in keep], np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    dst = np.zeros(total, np.uint8)
    srcs = (C.c_char_p * len(keep))(*keep)
    assert lib.ke_host_pack(dst.ctypes.data, srcs, offs.ctypes.data, sizes.ctypes.data, len(keep)) == 0
    assert bytes(dst) == b"".join(keep)


def test_decode_caveats_follow_the_files_markers(K):
    """ke_jpeg_caveats / ke_png_caveats (host code): the flags say when the reference's loader would turn the image by its EXIF
    orientation or composite its transparency -- checked against what Pillow itself reads from the same bytes."""
    import ctypes as C
    import io

    from PIL import Image, ImageOps

    lib = K._native.load_library()
    arr = (np.arange(24 * 16 * 3) % 251).astype(np.uint8).reshape(16, 24, 3)

    def save(fmt, img=None, **kw):
        b = io.BytesIO()
        (img or Image.fromarray(arr)).save(b, fmt, **kw)
        return b.getvalue()

    def exif(value):
        e = Image.Exif()
        e[0x0112] = value
        return e.tobytes()

    jpegs = [save("JPEG")] + [save("JPEG", exif=exif(v)) for v in range(0, 10)]
    app1 = b"\xff\xe1" + (2 + 6 + 5).to_bytes(2, "big") + b"Exif\0\0" + b"junk!"
    jpegs.append(jpegs[0][:2] + app1 + jpegs[0][2:])             # an EXIF block that cannot be followed
    pngs = [save("PNG"), save("PNG", exif=exif(6)), save("PNG", exif=exif(1)), save("PNG", Image.fromarray(arr).convert("P"), transparency=2),
            save("PNG", Image.fromarray(np.dstack([arr, arr[:, :, :1]]), "RGBA"))]
    for kind, blobs in (("jpeg", jpegs), ("png", pngs)):
        flat = np.frombuffer(b"".join(blobs) + bytes(64), np.uint8)
        sizes = np.array([len(b) for b in blobs], np.uint64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        flags = np.full(len(blobs), -1, np.int32)
        assert getattr(lib, f"ke_{kind}_caveats")(flat.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(blobs), flags.ctypes.data) == 0
        for k, blob in enumerate(blobs):
            try:
                with Image.open(io.BytesIO(blob)) as im:
                    turned = ImageOps.exif_transpose(im).size != im.size or im.getexif().get(0x0112, 1) in (2, 3, 4)
                    has_exif = "exif" in im.info or bool(im.getexif())
                    transparent = kind == "png" and "transparency" in im.info
            except Exception:
                turned, has_exif, transparent = True, True, False
            if turned:
                assert flags[k] & 1, (kind, k)
            if not has_exif:
                assert not flags[k] & 1, (kind, k)              # nothing to apply: the file may take the GPU decoder
            assert bool(flags[k] & 2) == transparent, (kind, k)


def test_no_cpu_fallback_without_gpu(K):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        K._native.Context(0)
    from PIL import Image

    with pytest.raises(RuntimeError):      # same contract as the reference when cv2 is missing (src/sig/phash.py:35-36)
        K.phash(Image.new("RGB", (64, 64)))


def test_product_never_imports_the_oracle():
    """Nothing under the product package imports, links or opens anything under oracle/."""
    pkg = os.path.join(ROOT, "kobato-eyes_amd")
    pattern = re.compile(r"(from|import)\s+oracle|libkeyes_oracle|keyes_oracle|[\"'/]oracle[/\"']")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")) and f != "dct_table.h":   # generated constants
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not pattern.search(text), os.path.join(dirpath, f)


def test_from_row_matches_reference(K):
    rows, expected, bad = G.rows_golden()
    for row, exp in zip(rows, expected):
        f = K.DuplicateFile.from_row(row)
        got = {"file_id": f.file_id, "path": f.path.as_posix(), "size": f.size, "width": f.width, "height": f.height,
               "phash": str(f.phash), "resolution": f.resolution, "extension_priority": f.extension_priority}
        assert got == exp
    for row in bad:
        with pytest.raises(ValueError, match="missing perceptual hash"):
            K.DuplicateFile.from_row(row)


def test_from_row_accepts_sqlite_rows(K):
    conn = sqlite3.connect(":memory:")
    conn.row_factory = sqlite3.Row
    conn.execute("CREATE TABLE t (file_id INTEGER, path TEXT, size INTEGER, width INTEGER, height INTEGER, phash_u64 INTEGER)")
    conn.execute("INSERT INTO t VALUES (7, 'a/b.PNG', 10, 3, 4, -2)")
    f = K.DuplicateFile.from_row(conn.execute("SELECT * FROM t").fetchone())
    assert (f.file_id, f.phash, f.extension_priority, f.resolution) == (7, (1 << 64) - 2, 4, 12)


def test_config_validation(K):
    for kw in ({"band_bits": 0}, {"band_count": 0}, {"hamming_threshold": -1}, {"hamming_threshold": 65},
               {"cosine_threshold": 1.5}):
        with pytest.raises(ValueError):
            K.DuplicateScanConfig(**kw)
    with pytest.raises(AssertionError):
        K.DuplicateScanner(K.DuplicateScanConfig(band_bits=32, band_count=3))
    s = K.DedupSettings.coerce("12", "0.5")
    assert (s.hamming_threshold, s.ssim_threshold) == (12, 0.5)
    s = K.DedupSettings.coerce("x", None)
    assert (s.hamming_threshold, s.ssim_threshold) == (10, 0.92)     # src/core/config/schema.py:186-201
    assert K.DedupSettings.coerce(-4, 1).hamming_threshold == 0


@pytest.mark.parametrize("name", sorted(G.scan_scenarios()))
def test_cluster_assembly_matches_reference(K, name):
    """Edges (from the golden file) -> clusters through the product's host code: union-find in
    libkeyes_hip.so + the reference's keeper / ordering rules."""
    sc = G.scan_scenarios()[name]
    files = [K.DuplicateFile(file_id=f["file_id"], path=Path(f["path"]), size=f["size"], width=f["width"],
                             height=f["height"], phash=f["phash"] & ((1 << 64) - 1)) for f in sc["files"]]
    from kobato_eyes_amd.scanner import DuplicateEdge

    edges = [DuplicateEdge(a, b, h) for a, b, h in sc["edges"]]
    clusters = K.assemble_clusters(files, edges) if edges else []
    got = [{"keeper_id": c.keeper_id, "entries": [[e.file.file_id, e.best_hamming] for e in c.files]} for c in clusters]
    assert got == sc["clusters"]


def test_first_writer_rule_with_duplicate_ids(K):
    """_edges_from_raw resolves duplicate file ids exactly as the reference's dict insertion order does."""
    sc = G.scan_scenarios()["dup_ids_signed"]
    from oracle import oracle as O

    files = [K.DuplicateFile(file_id=f["file_id"], path=Path(f["path"]), size=f["size"], width=f["width"],
                             height=f["height"], phash=f["phash"] & ((1 << 64) - 1)) for f in sc["files"]]
    hashes, ids, _ = G.files_to_arrays(sc["files"])
    raw = O.scan_bruteforce(hashes, threshold=10)          # every (i,j,h,bands) the GPU kernel would emit...
    raw = raw[ids[raw["a"]] != ids[raw["b"]]]              # ...after its same-id test
    scanner = K.DuplicateScanner(K.DuplicateScanConfig(**sc["config"]))
    edges = scanner._edges_from_raw(files, hashes, ids, raw, np.array([0, int(sum(bin(b).count("1") for b in raw["bands"])), len(raw), 0], np.uint64))
    assert sorted([a, b, e.hamming] for (a, b), e in edges.items()) == sc["edges"]


def test_cluster_builder(K):
    M = K.RefinedMatch
    out = K.ClusterBuilder().build([M(1, 2, 0.95, 0.2, True, "ssim"), M(2, 3, 0.93, 0.15, True, "ssim"),
                                    M(4, 5, 0.91, 0.16, True, "ssim"), M(3, 5, 0.5, 0.05, False, "below")])
    assert [c.members for c in out] == [[1, 2, 3], [4, 5]] and [c.representative for c in out] == [1, 4]
    assert [len(c.matches) for c in out] == [2, 1]
    assert K.ClusterBuilder().build([]) == []


def test_signed_wrap_and_upsert(K, tmp_path):
    from kobato_eyes_amd.fastsig import _to_signed64

    assert [_to_signed64(v) for v in (0, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, (1 << 64) + 7)] == [0, -1, -(1 << 63), (1 << 63) - 1, 7]
    db = tmp_path / "s.db"
    conn = sqlite3.connect(db)
    conn.execute("CREATE TABLE signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
    assert K.bulk_upsert_signatures(conn, [(1, (1 << 64) - 1, 5), (2, 2, 0)]) == 2
    assert K.bulk_upsert_signatures(conn, [(2, 1 << 63, 9)]) == 1
    assert conn.execute("SELECT * FROM signatures ORDER BY file_id").fetchall() == [(1, -1, 5), (2, -(1 << 63), 9)]
    assert K.bulk_upsert_signatures(conn, []) == 0


def test_fastsig_drops_missing_files_and_reports_progress(K, monkeypatch, tmp_path):
    """Progress cadence / cancel semantics of src/core/fastsig.py:86-98 with the GPU hasher stubbed out
    (this is host logic; the hashing itself is covered by the gpu tests)."""
    import kobato_eyes_amd.fastsig as fs
    from PIL import Image

    paths = []
    for k in range(5):
        p = tmp_path / f"i{k}.png"
        Image.new("RGB", (8, 8), (k, k, k)).save(p)
        paths.append((k + 1, str(p)))
    paths.insert(2, (99, str(tmp_path / "missing.png")))
    class FakeStage:                                       # the four calls of fastsig._GpuStage, no device behind them
        def __init__(self, device, stage_bytes, max_images):
            self.bufs, self.turn, self.submitted = [np.zeros(4096, np.uint8), np.zeros(4096, np.uint8)], 0, []

        def acquire(self):
            slot = self.turn
            self.turn ^= 1
            return slot, self.bufs[slot]

        def submit(self, slot, offsets, widths, heights, channels):
            assert all(o % 16 == 0 for o in offsets) and list(widths) == [8] * len(offsets) and list(channels) == [3] * len(offsets)
            for o in offsets:                              # the decode threads wrote the pixels where the allocator said
                px = self.bufs[slot][o:o + 192]
                assert len(set(px.tolist())) == 1
            self.submitted.append(len(offsets))
            n = len(offsets)
            return {"phash": np.arange(n, dtype=np.uint64) + np.uint64(1 << 63), "dhash": np.zeros(n, np.uint64), "status": np.zeros(n, np.int32)}

        def wait(self, slot):
            pass

        def jpeg_hash(self, blobs, kind="jpeg"):            # "the GPU decoder refuses every file": all take the Pillow route
            assert kind == "png" and all(b[:4] == b"\x89PNG" for b in blobs)
            n = len(blobs)
            return np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.ones(n, np.int32)

        def hash_one(self, arr):
            return (1 << 63, 0)

    monkeypatch.setattr(fs, "_make_stage", FakeStage)
    seen = []
    out = fs.compute_signatures_mp(paths, max_workers=2, chunksize=4, progress=lambda d, t: seen.append((d, t)))
    assert [r[0] for r in out] == [1, 2, 3, 4, 5] and seen == [(6, 6)]
    assert all(r[1] < 0 for r in out)                      # stored signed
    full = out
    calls = {"n": 0}

    def cancel():
        calls["n"] += 1
        return calls["n"] > 6

    # the callback is asked between the chunks of the Pillow route and before each GPU decode call (4 times for these six
    # files) and then once per result, as the reference does (src/core/fastsig.py:86-90): a prefix comes back
    out = fs.compute_signatures_mp(paths, max_workers=2, chunksize=4, cancel_fn=cancel)
    assert 0 < len(out) < len(full) and out == full[:len(out)]
    calls["n"] = 0
    stopped_early = fs.compute_signatures_mp(paths, max_workers=2, chunksize=4, cancel_fn=lambda: True)
    assert stopped_early == []                             # asked before any work: nothing decoded, nothing returned


def test_fastsig_progress_cadence_over_whole_batches(K, monkeypatch):
    """Without a cancel callback results are taken a GPU batch at a time; progress still fires at every 200th file and at
    the end (src/core/fastsig.py:95-98), failures are still omitted."""
    import kobato_eyes_amd.fastsig as fs

    class FakePipeline:
        def __init__(self, tasks, workers, chunk, device, cancel_fn=None):
            self.tasks = tasks

        def run_batches(self):
            for lo in range(0, len(self.tasks), 128):
                fids = np.array([t[0] for t in self.tasks[lo:lo + 128]], np.int64)
                yield fids, -fids, fids.copy(), fids % 50 != 0

    monkeypatch.setattr(fs, "_Pipeline", FakePipeline)
    for total, marks in ((450, [200, 400, 450]), (400, [200, 400]), (199, [199]), (128, [128])):
        seen = []
        rows = fs.compute_signatures_mp([(k, f"f{k}") for k in range(1, total + 1)], progress=lambda d, t: seen.append((d, t)))
        assert seen == [(m, total) for m in marks]
        assert rows == [(k, -k, k) for k in range(1, total + 1) if k % 50]


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from kobato_eyes_amd.distributed import allgather_hashes, allgather_edge_buffers, gather_edges, owned_indices, interleave_gathered
from kobato_eyes_amd import _native
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank, world, n = dist.get_rank(), 2, 1001
table = (np.arange(n, dtype=np.int64) * 7919) ^ 0x5555
mine = owned_indices(n, rank, world)
local = torch.from_numpy(table[mine].copy())
full = allgather_hashes(local, n)
assert np.array_equal(full.numpy(), table), "all-gather did not restore corpus order"
edges = np.zeros(3 + rank, _native.EDGE_DTYPE); edges["a"] = rank; edges["b"] = np.arange(len(edges)) + 10
merged = gather_edges(edges)
assert len(merged) == 7 and sorted(merged["a"].tolist()) == [0, 0, 0, 1, 1, 1, 1]
buf = torch.zeros(24 * 8, dtype=torch.uint8); buf[: edges.nbytes] = torch.from_numpy(edges.view(np.uint8).copy())
raw, counts = allgather_edge_buffers(buf, len(edges))
m2 = raw.view(_native.EDGE_DTYPE)
assert counts == [3, 4] and m2["a"].tolist() == [0, 0, 0, 1, 1, 1, 1] and m2["b"].tolist() == [10, 11, 12, 10, 11, 12, 13]
# one rank holds more edges than the fixed-width record carries: the second (full-width) all-gather, then the
# record grows and the next merge is a single collective again
big = np.zeros(1500 if rank == 1 else 2, _native.EDGE_DTYPE); big["a"] = rank; big["b"] = np.arange(len(big))
bbuf = torch.from_numpy(big.view(np.uint8).copy())
for _ in range(2):
    raw, counts = allgather_edge_buffers(bbuf, len(big))
    m3 = raw.view(_native.EDGE_DTYPE)
    assert counts == [2, 1500] and m3["a"].tolist() == [0, 0] + [1] * 1500 and m3["b"].tolist() == [0, 1] + list(range(1500))
raw, counts = allgather_edge_buffers(torch.zeros(0, dtype=torch.uint8), 0)
assert counts == [0, 0] and len(raw) == 0
parts = [table[owned_indices(n, r, world)] for r in range(world)]
assert np.array_equal(interleave_gathered(parts, n), table)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharding_helpers_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_decode_normalisation_matches_reference(K, tmp_path):
    """kobato_eyes_amd.image_io.safe_load_image against what the reference's utils.image_io.safe_load_image
    returned for the same files (tests/golden/image_io_golden.json): mode, size and every pixel."""
    import hashlib
    import warnings

    from kobato_eyes_amd.image_io import safe_load_image

    golden = G.image_io_golden()
    seen = 0
    for name, path, kwargs in G.write_image_io_files(tmp_path):
        exp = golden[name]
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                img = safe_load_image(path, **kwargs)
        except Exception as exc:
            assert exp == {"raises": type(exc).__name__}, name
            seen += 1
            continue
        if exp is None:
            assert img is None, name
        else:
            assert img is not None and "raises" not in exp, name
            got = {"mode": img.mode, "size": list(img.size), "sha256": hashlib.sha256(img.tobytes()).hexdigest()}
            assert got == exp, name
        seen += 1
    assert seen == len(golden) >= 20


def test_cluster_maintenance_matches_reference(K):
    """rebuild_clusters_after_removal / cluster_hamming_score / default_checked_entries against the reference's
    ui.dup_cluster_update and ui.dup_tree_state run on the same clusters (tests/golden/cluster_update_golden.json)."""
    import json

    from oracle import oracle as O

    with open(os.path.join(G.GOLDEN, "cluster_update_golden.json")) as fh:
        g = json.load(fh)
    ext = ("png", "jpg", "webp", "jpeg", "bmp", "tif")
    files = {i + 1: K.DuplicateFile(file_id=i + 1, path=Path(f"dir{i % 3}/img_{i:07d}.{ext[i % 6]}"), size=1000 + (i % 7),
                                    width=512 - (i % 5), height=512, phash=int(hv)) for i, hv in enumerate(O.synth_hashes(1000))}
    clusters = [K.DuplicateCluster(files=[K.DuplicateClusterEntry(files[fid], best) for fid, best in entries], keeper_id=keeper)
                for keeper, entries in g["clusters"]]
    enc = lambda cs: [[c.keeper_id, [[e.file.file_id, e.best_hamming] for e in c.files]] for c in cs]
    for case in g["removals"]:
        assert enc(K.rebuild_clusters_after_removal(clusters, set(case["removed"]))) == case["result"]
    assert [K.cluster_hamming_score(c) for c in clusters] == g["hamming_score"]
    assert [[e.file.file_id for e in K.default_checked_entries(c)] for c in clusters] == g["default_checked"]
    assert K.rebuild_cluster_after_removal(clusters[0], {e.file.file_id for e in clusters[0].files[1:]}) is None
    assert K.choose_keeper(clusters[0].files) == clusters[0].keeper_id


def test_batches_read_ahead_reach_the_decoders_in_their_own_order(tmp_path, monkeypatch):
    """fastsig._Pipeline with a stage that reads files itself (the shape of _GpuStage: read_ahead / hash_ahead / hash_files):
    the next batch is read while the current one is hashed, JPEG files first and PNG files after them in the buffer, each
    kind handed its own slice; a batch whose read-ahead found no buffer goes through hash_files; every buffer comes back, also
    when the run is given up half way.  No device: the 'hashes' are numbers taken from the file names."""
    from kobato_eyes_amd import fastsig as fs

    items = []
    for k in range(23):
        p = tmp_path / f"f{k:03d}.{'png' if k % 3 == 0 else 'jpg'}"
        p.write_bytes(b"x")
        items.append((1000 + k, str(p)))
    log = {"ahead": 0, "direct": 0, "released": 0, "taken": 0}

    class Held:
        def __init__(self, paths):
            self.paths = paths

        def release(self):
            log["released"] += 1

    def numbers(paths, kind):
        assert all(p.endswith("png" if kind == "png" else "jpg") for p in paths)
        n = np.array([int(os.path.basename(p)[1:4]) for p in paths], np.uint64)
        return n, n + np.uint64(500), np.where(n % 5 == 4, 1, 0).astype(np.int32)     # every fifth file is left to Pillow

    class Stage:
        def __init__(self, device, stage_bytes, max_images):
            self.buf = np.zeros(1 << 16, np.uint8)

        def acquire(self):
            return 0, self.buf

        def submit(self, slot, offsets, widths, heights, channels):
            raise AssertionError("nothing decodes here: the files are not images")

        def wait(self, slot):
            pass

        def hash_one(self, arr):
            return None

        def read_ahead(self, paths, spans=()):
            jpegs = sum(p.endswith("jpg") for p in paths)
            assert tuple(spans) == (("jpeg", 0, jpegs), ("png", jpegs, len(paths)), ("bmp", len(paths), len(paths)), ("gif", len(paths), len(paths)), ("tiff", len(paths), len(paths)))
            log["taken"] += 1
            if log["taken"] == 2:                              # "both buffers taken": this batch is read inside the call
                return None
            log["ahead"] += 1
            return Held(list(paths))

        def hash_ahead(self, held, lo, hi, kind="jpeg"):
            return numbers(held.paths[lo:hi], kind)

        def hash_files(self, paths, kind="jpeg"):
            log["direct"] += 1
            return numbers(paths, kind)

    monkeypatch.setattr(fs, "_make_stage", Stage)
    monkeypatch.setenv("KE_GPU_BATCH", "8")
    monkeypatch.setenv("KE_DECODE_PROCESSES", "0")
    rows = fs.compute_signatures_mp(items, max_workers=2, chunksize=8)
    want = [(1000 + k, k, k + 500) for k in range(23) if k % 5 != 4]          # the refused ones are not images: dropped
    assert rows == want
    assert log["ahead"] == 2 and log["released"] == 2 and log["direct"] == 2   # three batches: one of them jpeg + png direct
    batches = fs._Pipeline(items, 2, 8, 0).run_batches()
    next(batches)
    batches.close()
    assert log["released"] == log["ahead"]


def test_bench_self_launches_its_ranks_world_size_2_gloo():
    """`python bench.py --gpus 2` typed without a launcher (as the driver types it at N = 1) must start its ranks as fresh
    child processes under torch.distributed.run, relay rank 0's one JSON line on stdout and return the children's exit
    code.  Without a GPU the ranks only meet (gloo) and count each other."""
    import json

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--rendezvous-only"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_ranks_seen"] == 2 and out["world_size"] == 2 and out["self_launched"] is True
    # a failing rank is the launcher's failure too
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--no-such-flag"],
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
