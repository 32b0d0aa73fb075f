#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: images/s for "pHash every image, then
all-pairs Hamming scan + cluster membership" on synthetic 512x512 RGB (BASELINE.json configs[1]
at N=1, configs[2] -- the same 100 000 images split over the ranks -- at N>1).

    python bench.py [--gpus N --steps K --warmup W]            # N=1: plain python
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over the whole corpus, inputs already resident in HBM:
  1. ke_hash_uniform   : fused pHash kernel over this rank's images (device in, device out)
  2. all-gather (N>1)  : ONE ncclAllGather of the 64-bit hash shards + reorder kernel (ke_allgather_hashes, RCCL behind the C ABI)
  3. ke_hamming_scan   : this rank's share of the tile triangle, edges compacted on the device
  4. edge merge (N>1)  : ONE all-gather of fixed-width edge records + one D2H copy (ke_allgather_edges)
  5. ke_cluster_labels : host union-find -> cluster membership (rank 0)
PyTorch only provides device memory, the stream and the rendezvous (process group).
Rank 0 prints ONE JSON line (contract in the task statement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SEED = 20260604
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense MX-fp4 matrix peak (MI355X_MICROARCH.md: ~10 PF dense)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=100_000, help="corpus size (BASELINE configs[1]/[2]: 100000)")
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--threshold", type=int, default=8)
    ap.add_argument("--dhash", action="store_true", help="also compute dHash in the hash step")
    ap.add_argument("--ssim-threshold", type=float, default=None,
                    help="BASELINE configs[3] flavour: re-check every candidate edge with the SSIM kernel inside the step (pairs sharded over the ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive leg (pinned staging -> device -> hash)")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode-inclusive leg (JPEG files in host memory -> GPU decode -> hash)")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="exchange through torch.distributed (kobato_eyes_amd.distributed) instead of the library's own RCCL entry points")
    ap.add_argument("--phase-timing", action="store_true", help="after the timed run, time each phase of a step with syncs in between (stderr)")
    ap.add_argument("--cpu-sample", type=int, default=16000, help="images hashed by the CPU oracle for the baseline")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: --images is the corpus PER RANK (the job hashes images x N; --images 125000 --gpus 8 is BASELINE "
                         "configs[3]'s 1 000 000-image table)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks as child processes under torch.distributed.run even at --gpus 1 (the RCCL path on one GPU)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch check without a GPU: the ranks meet over gloo, count each other and rank 0 prints the line")
    ap.add_argument("--phase-steps", type=int, default=3, help="steps of the per-phase timing pass behind the timed region (0: none)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as FRESH child processes under
    torch.distributed.run -- before this process has imported torch or touched a GPU (a process that has initialised the GPU
    must never be replaced or forked) -- let rank 0 write its one JSON line to the inherited stdout, and hand back the
    children's exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:                       # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), KE_BENCH_SELF_LAUNCHED="1")
    print(f"[bench] starting {args.gpus} rank(s): {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(args) -> None:
    """The launch path without a GPU (CPU suite): gloo process group, every rank counts the others."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)                                       # gloo announces its connections on stdout
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo")
    one = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    if dist.get_rank() == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"rendezvous_only": True, "n_gpus": args.gpus, "n_ranks_seen": int(one.item()),
                          "world_size": dist.get_world_size(), "self_launched": os.environ.get("KE_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def cpu_baseline(ctx, args, table_host):
    """Oracle (CPU port of the reference path) timed on the host: a bounded sample of the same
    workload -- `cpu_sample` of the corpus images hashed on one core, plus the reference-shaped
    banded scan over the full hash table -- scaled to images/s for the whole corpus."""
    from oracle import oracle as O

    m = min(args.cpu_sample, args.images)
    t_hash, done = 0.0, 0
    while done < m:                                               # chunks keep the host copy of the pixels small
        k = min(2000, m - done)
        px = ctx.synth_rgb(SEED, done, k, args.side, args.side)   # same pixels the GPU hashed
        t0 = time.perf_counter()
        ph, _ = O.hash_batch(px, want_dhash=args.dhash)
        t_hash += time.perf_counter() - t0
        assert np.array_equal(ph, table_host[done:done + k]), "GPU pHash differs from the oracle on the baseline sample"
        done += k
    t0 = time.perf_counter()
    edges, _ = O.scan_banded(table_host, threshold=args.threshold)
    t_scan = time.perf_counter() - t0
    total = t_hash / m * args.images + t_scan
    # the reference's own parallelism for this step is 8 worker processes (KE_SIG_WORKERS, src/ui/dup_workers.py:180):
    # the same oracle on 8 threads (ctypes releases the GIL), reported beside the 1-core figure
    from concurrent.futures import ThreadPoolExecutor

    def threaded(threads, n_sample):
        px = ctx.synth_rgb(SEED, 0, n_sample, args.side, args.side)
        parts = [p for p in np.array_split(np.arange(n_sample), threads) if len(p)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda idx: O.hash_batch(px[idx[0]:idx[-1] + 1], want_dhash=args.dhash), parts))
        return n_sample / (time.perf_counter() - t0)

    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    try:                                               # a container's CPU share (cgroup v2 cpu.max: "<quota> <period>")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    t8 = min(8, usable)
    return {
        "cpu_model": model, "host_cores": cores, "usable_cores": usable,
        "threads_8": {"threads": t8, "hash_images_per_s": threaded(t8, 2000), "sample": "2000 images, one slice per thread"},
        # every core this process may run on (the reference's pool default is cpu_count - 1, src/core/fastsig.py:75)
        "all_cores": {"threads": usable, "hash_images_per_s": threaded(usable, max(2000, 120 * usable)),
                      "sample": f"{max(2000, 120 * usable)} images, one slice per thread"},
        "value": args.images / total, "unit": "images/s", "cores": 1, "kind": "port",
        "sample": f"{m} of the {args.images} corpus images hashed by oracle/keyes_oracle.c on 1 core ({t_hash:.2f} s, "
                  f"{m / t_hash:.0f} img/s) + reference-shaped banded scan over all {args.images} hashes ({t_scan:.2f} s, "
                  f"{len(edges)} edges); hash time scaled to the corpus",
        "hash_images_per_s": m / t_hash, "scan_seconds": t_scan,
    }


def main():
    args = parse_args()
    if "RANK" not in os.environ and (args.gpus > 1 or args.self_launch):
        raise SystemExit(self_launch(args))              # nothing below has run: no torch import, no GPU call in this process
    if args.rendezvous_only:
        return rendezvous_only(args)
    # stdout carries ONE JSON line: whatever libraries print while the job runs (RCCL announces its version on stdout when a
    # communicator is made) goes to stderr instead -- file descriptor 1 is pointed at 2 until the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from kobato_eyes_amd import _native
    from kobato_eyes_amd.distributed import RcclExchange, allgather_edge_buffers, allgather_hashes, ssim_refine_sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run the process group is used even at WORLD_SIZE=1, so the RCCL code
    # path can be rehearsed on a one-GPU box
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if distributed:
        dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    ctx = _native.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    # the exchange steps: ke_allgather_hashes / ke_allgather_edges on the library's own RCCL communicator (the process
    # group only carried the unique id)
    exchange = None
    if distributed and not args.torch_collectives:
        # made and rehearsed before anything is timed: a small table through ke_allgather_hashes against torch's own
        # all-gather.  Should the library's communicator fail on any rank (or disagree), every rank takes the
        # torch.distributed route instead -- both are RCCL, the JSON line says which one ran.
        ok = 1
        try:
            exchange = RcclExchange(ctx)
            n_try = 4099 * world
            per_try = (n_try + world - 1) // world
            mine_try = torch.arange(rank, n_try, world, dtype=torch.int64, device=dev) * 0x1E3779B97F4A7C15   # wraps; any values do
            local_try = torch.zeros(per_try, dtype=torch.int64, device=dev)
            local_try[: mine_try.numel()] = mine_try
            table_try = torch.empty(n_try, dtype=torch.int64, device=dev)
            exchange.hashes(local_try.data_ptr(), n_try, table_try.data_ptr())
            torch.cuda.synchronize()
            if not torch.equal(table_try, allgather_hashes(local_try, n_try)):
                raise RuntimeError("ke_allgather_hashes disagrees with torch.distributed.all_gather_into_tensor")
        except Exception as exc:   # noqa: BLE001 - any failure here only selects the other exchange
            print(f"[bench] rank {rank}: library RCCL exchange unavailable ({exc}); using torch.distributed", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and exchange is not None:
            try:
                exchange.close()
            except Exception:   # noqa: BLE001
                pass
            exchange = None
        elif int(flag.item()) == 0:
            exchange = None

    n_ranks_seen = 1
    if distributed:
        seen = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(seen)                            # over RCCL: the record shows how many ranks really met
        n_ranks_seen = int(seen.item())
    n_total, side = (args.images * world if args.weak else args.images), args.side
    img_bytes = side * side * 3
    # hash-partition of the corpus: image i lives on rank i mod world
    mine = np.arange(rank, n_total, world, dtype=np.int64)
    n_local = len(mine)
    per = (n_total + world - 1) // world

    # ---- setup (untimed): materialise this rank's images in HBM with the on-device generator
    pixels = torch.empty(n_local * img_bytes, dtype=torch.uint8, device=dev)
    if world == 1:
        ctx.synth_rgb(SEED, 0, n_local, side, side, out=pixels.data_ptr())
    else:
        ctx.synth_rgb_indexed(SEED, mine, side, side, pixels.data_ptr())   # this rank's strided positions, one launch
    local_margin = torch.zeros(per, dtype=torch.float32, device=dev)   # pHash tie margins (ke_hash_uniform_ex), telemetry
    local_hash = torch.zeros(per, dtype=torch.int64, device=dev)
    local_dhash = torch.zeros(per, dtype=torch.int64, device=dev) if args.dhash else None
    cap = max(1 << 16, n_total)
    edges_dev = torch.empty(cap * 24, dtype=torch.uint8, device=dev)
    table_dev = torch.empty(n_total, dtype=torch.int64, device=dev) if exchange else None
    torch.cuda.synchronize()

    state = {}

    def step():
        nonlocal edges_dev, cap
        # 1. hash this rank's shard (device -> device)
        ctx.hash_uniform(pixels.data_ptr(), n_local, side, side, 3, phash_out=local_hash.data_ptr(),
                         dhash_out=local_dhash.data_ptr() if args.dhash else None, want_dhash=args.dhash,
                         margin_out=local_margin.data_ptr())
        # 2. the one exchange on the data path
        if exchange:
            exchange.hashes(local_hash.data_ptr(), n_total, table_dev.data_ptr())
            table = table_dev
        else:
            table = allgather_hashes(local_hash, n_total) if distributed else local_hash[:n_total]
        # 3. sharded scan; edges stay on the device, the count comes back with the counters
        while True:
            edges, counters = _scan(ctx, table, n_total, args.threshold, rank, world, edges_dev, cap)
            if edges <= cap:
                break
            cap = int(edges)
            edges_dev = torch.empty(cap * 24, dtype=torch.uint8, device=dev)
        # 4. merge edge lists
        if exchange:
            all_edges, _ = exchange.edges(edges_dev.data_ptr(), edges)
        elif distributed:
            merged, _ = allgather_edge_buffers(edges_dev, edges)
            all_edges = merged.view(_native.EDGE_DTYPE)
        else:
            all_edges = edges_dev[: edges * 24].cpu().numpy().view(_native.EDGE_DTYPE)
        # 4b. optional SSIM refine of the candidate edges.  One GPU: every image is resident.  Several: the pairs are
        #     dealt round-robin and a rank regenerates the images of its pairs from (seed, position).
        if args.ssim_threshold is not None and len(all_edges):
            if world == 1:
                ssim = ctx.ssim_pairs_uniform(pixels.data_ptr(), n_total, side, side, 3, all_edges["a"], all_edges["b"])
            else:
                def fetch(ids):
                    need = len(ids) * img_bytes
                    if state.get("pair_px") is None or state["pair_px"].numel() < need:
                        state["pair_px"] = torch.empty(need, dtype=torch.uint8, device=dev)
                    ctx.synth_rgb_indexed(SEED, ids, side, side, state["pair_px"].data_ptr())
                    return state["pair_px"].data_ptr()
                ssim = ssim_refine_sharded(ctx, all_edges, fetch, side, side, 3)
            state["ssim_ms"] = ctx.last_kernel_ms(2)
            state["ssim_pairs"] = len(all_edges)
            state["ssim_quartiles"] = [float(q) for q in np.quantile(ssim, [0.0, 0.25, 0.5, 0.75, 1.0])]
            state["ssim_kept"] = int((ssim >= args.ssim_threshold).sum())
            # decisions that a 1e-4 disagreement with skimage could flip (the SSIM value is unpinned against the real library)
            state["ssim_near_threshold"] = int((np.abs(ssim - args.ssim_threshold) < 1e-4).sum())
            all_edges = all_edges[ssim >= args.ssim_threshold]
        # 5. cluster membership on the host: rank 0 (every rank holds the merged edges and could)
        labels = _native.cluster_labels(all_edges, n_total) if rank == 0 else None
        state.update(table=table, edges=all_edges, labels=labels, pairs=int(counters[0]))

    def _scan(ctx, table, n, thr, part, parts, edges_dev, cap):
        import ctypes as C

        lib = ctx._lib
        n_edges = C.c_int64(0)
        counters = np.zeros(4, np.uint64)
        rc = lib.ke_hamming_scan(ctx._h, table.data_ptr(), None, None, n, part, parts, thr, 16, 4, 0.0, 0,
                                 edges_dev.data_ptr(), cap, C.byref(n_edges), counters.ctypes.data)
        ctx._check(rc, "ke_hamming_scan")
        return n_edges.value, counters

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    hash_ms, scan_ms = [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        hash_ms.append(ctx.last_kernel_ms(0))   # hipEvents around the hash kernel on its stream
        scan_ms.append(ctx.last_kernel_ms(1))
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # what the batch-hasher seam launches per file is BOTH hashes (src/core/fastsig.py:31-34): the same kernel family with the
    # dHash leg, timed behind the timed region on the same resident images
    dual_ms = []
    dual_hash = torch.zeros(per, dtype=torch.int64, device=dev)
    dual_dhash = local_dhash if local_dhash is not None else torch.zeros(per, dtype=torch.int64, device=dev)
    for k in range(2 + max(3, min(args.steps, 10))):
        ctx.hash_uniform(pixels.data_ptr(), n_local, side, side, 3, phash_out=dual_hash.data_ptr(), dhash_out=dual_dhash.data_ptr(),
                         want_dhash=True)
        if k >= 2:
            dual_ms.append(ctx.last_kernel_ms(0))
    torch.cuda.synchronize()
    assert torch.equal(dual_hash[:n_local], local_hash[:n_local]), "pHash of the dual-hash launch differs from the single-hash launch"

    phase_ms = None
    if args.phase_timing or args.phase_steps > 0:
        # where a step's time goes: the same calls with a device sync (and, across ranks, the slowest rank) after each phase;
        # behind the timed region, so `value` is untouched
        acc = {}
        phase_steps = args.steps if args.phase_timing else args.phase_steps

        def lap(name, t0):
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

        for _ in range(phase_steps):
            t0 = time.perf_counter()
            ctx.hash_uniform(pixels.data_ptr(), n_local, side, side, 3, phash_out=local_hash.data_ptr(),
                             dhash_out=local_dhash.data_ptr() if args.dhash else None, want_dhash=args.dhash)
            lap("hash", t0)
            t0 = time.perf_counter()
            if exchange:
                exchange.hashes(local_hash.data_ptr(), n_total, table_dev.data_ptr())
                table = table_dev
            else:
                table = allgather_hashes(local_hash, n_total) if distributed else local_hash[:n_total]
            lap("allgather_hashes", t0)
            t0 = time.perf_counter()
            edges, counters = _scan(ctx, table, n_total, args.threshold, rank, world, edges_dev, cap)
            lap("scan+count", t0)
            t0 = time.perf_counter()
            if exchange:
                all_edges, _ = exchange.edges(edges_dev.data_ptr(), edges)
            elif distributed:
                merged, _ = allgather_edge_buffers(edges_dev, edges)
                all_edges = merged.view(_native.EDGE_DTYPE)
            else:
                all_edges = edges_dev[: edges * 24].cpu().numpy().view(_native.EDGE_DTYPE)
            lap("edge_merge", t0)
            t0 = time.perf_counter()
            if rank == 0:
                _native.cluster_labels(all_edges, n_total)
            lap("labels", t0)
        names = list(acc)
        per_phase = torch.tensor([acc[k] / phase_steps for k in names], dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(per_phase, op=dist.ReduceOp.MAX)
        phase_ms = {k: round(float(v), 4) for k, v in zip(names, per_phase.tolist())}
        if rank == 0 and args.phase_timing:
            print("phase ms/step:", phase_ms, file=sys.stderr, flush=True)

    h2d = None
    if rank == 0 and world == 1 and not args.no_h2d:
        # PCIe-inclusive rate of the hashing step (never `value`): the images start in the library's page-locked staging
        # buffers (where decoders put them, ke_stage_*), each batch is copied on the copy stream while the previous one is
        # hashed.  Two buffers of 512 images (403 MB each), 40 batches.
        per = 512
        ctx.stage_create(per * img_bytes, per, 2)
        try:
            filled = {}
            offs, dims, ch3 = np.arange(per, dtype=np.uint64) * np.uint64(img_bytes), [side] * per, [3] * per
            def submit():
                slot, view = ctx.stage_acquire()
                if slot not in filled:                         # same pixels every round: the copy is what is being timed
                    view[: per * img_bytes] = ctx.synth_rgb(SEED, slot * per, per, side, side).reshape(-1)
                    filled[slot] = True
                return ctx.stage_submit_hash(slot, offs, dims, dims, ch3, want_dhash=False)
            keep = [submit() for _ in range(4)]
            ctx.stage_wait(-1)
            t0 = time.perf_counter()
            keep = [submit() for _ in range(40)]
            ctx.stage_wait(-1)
            dt = time.perf_counter() - t0
            assert np.array_equal(keep[-2]["phash"], local_hash[:per].cpu().numpy().view(np.uint64)), "staged hashes differ from the resident path"
            h2d = {"images_per_s": 40 * per / dt, "gb_per_s": 40 * per * img_bytes / dt / 1e9,
                   "what": "pinned staging buffers -> H2D on the copy stream overlapped with the hash kernel of the previous batch "
                           f"(ke_stage_*; 40 batches of {per} images, 2 buffers); decode not included"}
        finally:
            ctx.stage_destroy()

    decode_leg = None
    if rank == 0 and world == 1 and not args.no_decode:
        # decode-inclusive rate (never `value`): the same corpus images as JPEG files in host memory (what a library scan starts
        # from) -> page-locked packing -> PCIe (compressed bytes) -> GPU JPEG decode -> the hash kernel on the decoded pixels.
        # 65 536 files in one call (one wave of the entropy kernel per SIMD); the hashes must be those of Pillow's decode of the
        # same files.
        try:
            import io

            from PIL import Image

            from concurrent.futures import ThreadPoolExecutor

            distinct, n_files = 4096, 65536                            # 16 copies of each file, spread out: lanes of a wave hold different images
            src_px = ctx.synth_rgb(SEED, 0, distinct, side, side)

            def encode(k):
                b = io.BytesIO()
                Image.fromarray(src_px[k]).save(b, "JPEG", quality=85, subsampling=2)
                return b.getvalue()

            with ThreadPoolExecutor(16) as ex:
                enc = list(ex.map(encode, range(distinct)))
            del src_px
            blobs = [enc[k % distinct] for k in range(n_files)]
            ctx.jpeg_hash(blobs, want_dhash=False)                      # buffers
            t0, k0 = time.perf_counter(), ctx.decode_kernel_ms
            ph_j, _, st_j = ctx.jpeg_hash(blobs, want_dhash=False)
            dt = time.perf_counter() - t0
            check = [np.asarray(Image.open(io.BytesIO(enc[k]))) for k in range(4)]
            ph_ref = ctx.hash_images(check, want_dhash=False)[0]
            ok = bool((st_j == 0).all()) and np.array_equal(np.asarray(ph_ref, np.uint64), ph_j[:4])
            decode_leg = {"images_per_s": n_files / dt, "files": n_files, "compressed_mb": sum(len(b) for b in blobs) / 1e6,
                          "decode_kernels_ms": ctx.decode_kernel_ms - k0, "matches_pillow_decode": ok,
                          "what": f"{side}x{side} JPEG (quality 85, 4:2:0) files in host memory -> ke_host_pack -> H2D -> ke_jpeg_decode -> "
                                  "ke_hash_images, one call, wall clock; file reading not included (benchmarks/bench_fastsig.py: from disk to SQLite rows)"}
            ctx.release_decode_buffers()
        except Exception as exc:   # noqa: BLE001 - an informational leg must not cost the bench line
            print(f"[bench] decode-inclusive leg skipped: {exc}", file=sys.stderr)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed
        hash_avg = float(np.mean(hash_ms))
        scan_avg = float(np.mean(scan_ms))
        alg_bytes = n_local * (img_bytes + 8 + (8 if args.dhash else 0))      # 3*W*H read + hash(es) written, per launch
        achieved = alg_bytes / (hash_avg * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hash_kernel_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("images_per_launch") == n_local and tj.get("side") == side:
                traffic = tj.get("hbm_bytes_per_launch")
        mg = local_margin[:n_local].cpu().numpy()                  # this rank's images (all of them at N=1)
        near_ties = {"images": int(n_local), "lt_1e-3": int((mg < 1e-3).sum()), "lt_1e-4": int((mg < 1e-4).sum()),
                     "exact_ties": int((mg == 0).sum()), "min_margin": float(mg.min())}
        pairs_rank = state["pairs"]
        pairs_s = pairs_rank / (scan_avg * 1e-3) if scan_avg > 0 else 0.0
        n_clusters = int(len(np.unique(state["labels"][np.unique(np.concatenate([state["edges"]["a"], state["edges"]["b"]]))]))) \
            if len(state["edges"]) else 0
        out = {
            "metric": "images/s (pHash + all-pairs Hamming scan + cluster membership)",
            "value": value, "unit": "images/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "u8 luma, int32 taps (i8 matrix cores), f64 DCT, 1-bit Hamming products (fp4 matrix cores, exact)", "data": "synthetic",
            "config": {
                "workload": f"{n_total} synthetic {side}x{side} RGB images resident in HBM, hamming_threshold={args.threshold}, "
                            f"band 16x4 (BASELINE configs[{1 if world == 1 else 2}]" + (f"; weak: {args.images} images per rank" if args.weak else "") + ")",
                "images": n_total, "side": side, "hamming_threshold": args.threshold, "dhash": bool(args.dhash),
                "partition": f"image i on rank i mod {world}; scan tiles dealt round-robin",
                "exchange": ("none (one GPU)" if not distributed else "torch.distributed all_gather_into_tensor" if not exchange
                             else "ke_allgather_hashes + ke_allgather_edges (RCCL behind the C ABI)"),
                "step": "hash kernel -> (all-gather) -> scan kernel -> edges to host -> "
                        + ("SSIM refine of the candidate edges -> " if args.ssim_threshold is not None else "") + "union-find labels",
            },
            "gpairs_per_s": pairs_s * world / 1e9,
            "hash_images_per_s": n_local * world / (hash_avg * 1e-3),
            "edges": int(len(state["edges"])), "clusters": n_clusters,
            "kernel_ms": {"hash": hash_avg, "scan": scan_avg, **({"ssim": state["ssim_ms"]} if "ssim_ms" in state else {})},
            "kernel_ms_median": {"hash": float(np.median(hash_ms)), "scan": float(np.median(scan_ms))},
            **({"phase_ms": phase_ms, "phase_ms_note": "max over ranks, device sync after every phase (not the pipelined step)"} if phase_ms else {}),
            "self_launched": os.environ.get("KE_BENCH_SELF_LAUNCHED") == "1",
            **({"ssim": {"threshold": args.ssim_threshold, "pairs": state["ssim_pairs"], "quartiles": state["ssim_quartiles"],
                         "kept": state["ssim_kept"], "within_1e-4_of_threshold": state["ssim_near_threshold"],
                         "pairs_per_s": state["ssim_pairs"] / (state["ssim_ms"] * 1e-3),
                         "kernel": "ke_ssim_fast (integer window sums in float32)",
                         "roofline": {"bound": "hbm", "achieved": state["ssim_pairs"] * (2 * img_bytes + 8) / (state["ssim_ms"] * 1e-3) / 1e9,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": state["ssim_pairs"] * (2 * img_bytes + 8) / (state["ssim_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}}}
               if "ssim_ms" in state else {}),
            # how exposed the corpus is to the one unpinnable step: images whose closest bit decision `coef > mean` sits
            # within 1e-3 / 1e-4 of a tie (another DCT implementation, e.g. OpenCV's float32 one, may flip such a bit)
            "phash_near_ties": near_ties,
            **({"h2d_inclusive": h2d} if h2d else {}),
            **({"decode_inclusive": decode_leg} if decode_leg else {}),
            "roofline": {"bound": "hbm", "kernel": "ke_phash_fused_mx", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/hash_kernel_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass of this kernel at this "
                                           "size, gfx950 corrections applied; counters cannot be read inside this process)" if traffic else None},
            # pHash + dHash in one launch: what kobato_eyes_amd.fastsig runs per batch (algorithmic bytes: 3*W*H read + 16 written)
            "roofline_dual": {"bound": "hbm", "kernel": "ke_phash_fused_mx (dHash leg on)", "ms": float(np.mean(dual_ms)),
                              "achieved": n_local * (img_bytes + 16) / (float(np.mean(dual_ms)) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": n_local * (img_bytes + 16) / (float(np.mean(dual_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "images_per_s": n_local * world / (float(np.mean(dual_ms)) * 1e-3)},
            # scan: one v_mfma_f32_16x16x128_f8f6f4 (fp4) = 256 pairs x 128 one-bit products -> 256 flop per pair
            # against the dense fp4 peak; the 16 B/pair HBM convention of SURVEY 8d is kept beside it (operands are
            # reused from registers/LDS, so that fraction exceeds 1)
            "roofline_scan": {"bound": "mfma", "kernel": "ke_scan_tiles", "achieved": pairs_s * 256 / 1e12, "peak": MFMA_FP4_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": pairs_s * 256 / 1e12 / MFMA_FP4_PEAK_TFLOPS,
                              "hbm_convention_gbs": pairs_s * 16 / 1e9, "hbm_convention_frac": pairs_s * 16 / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx, args, state["table"].cpu().numpy().view(np.uint64))
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if exchange:
        exchange.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
