#!/usr/bin/env bash
# Collects everything DESIGN.md section 5 quotes for the tree it is run from, on the GPU box:
#   gpurun --timeout 1150 -- 'bash profiles/collect.sh r03'        (round 3: part 2 with `bash profiles/collect.sh r03 more`)
# Outputs land under gpurun_out/<tag>_*; `python profiles/summarize.py <tag> ...` (see README.md) condenses them into profiles/.
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), never a wrapper; counters are collected in their
# own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc).
set -uo pipefail
tag="${1:-r03}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py"
run() { echo "== $*" >&2; "$@"; echo "rc=$?" >&2; }

if [ "${2:-}" = "more" ]; then
    # ---- second call of a round (the first one fills its time limit): everything DESIGN.md section 5 quotes beyond the headline
    # pHash + dHash in one launch (what the batch-hasher seam runs), its kernel statistics
    run python3 "$B" --steps 10 --warmup 3 --dhash --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_bench_dhash.json" 2> "$out/${tag}_bench_dhash.err"
    run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_dhash" -- python3 "$B" --steps 10 --warmup 3 --dhash --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_stats_dhash.log" 2>&1
    # the self-launched form of the bench (the parent starts the rank under torch.distributed.run; RCCL exchange on one rank)
    run python3 "$B" --gpus 1 --self-launch --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_bench_selflaunch.json" 2> "$out/${tag}_bench_selflaunch.err"
    # BASELINE configs[4] share on one GPU: 125 000 mixed-resolution images, pHash alone and both hashes, kernel statistics
    : > "$out/${tag}_mixed.jsonl"
    run python3 "$root/benchmarks/bench_mixed.py" --images 125000 >> "$out/${tag}_mixed.jsonl" 2>> "$out/${tag}_mixed.err"
    run python3 "$root/benchmarks/bench_mixed.py" --images 125000 --dhash >> "$out/${tag}_mixed.jsonl" 2>> "$out/${tag}_mixed.err"
    run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_mixed" -- python3 "$root/benchmarks/bench_mixed.py" --images 125000 --dhash --reps 2 > "$out/${tag}_stats_mixed.log" 2>&1
    # every (width, height) of that config on its own
    run python3 "$root/benchmarks/shape_grid.py" > "$out/${tag}_shape_grid_phash.txt" 2>> "$out/${tag}_mixed.err"
    run python3 "$root/benchmarks/shape_grid.py" dhash > "$out/${tag}_shape_grid_both.txt" 2>> "$out/${tag}_mixed.err"
    # per-kernel lines (hash shapes with and without dHash, scan at 100 000 / 1 000 000, SSIM pairs)
    run python3 "$root/benchmarks/bench_kernels.py" --cases hash,scan,ssim > "$out/${tag}_kernels.jsonl" 2> "$out/${tag}_kernels.err"
    # the seams: scanner (Python objects in, clusters out), decode call phases
    run python3 "$root/benchmarks/bench_scanner.py" > "$out/${tag}_scanner.jsonl" 2> "$out/${tag}_scanner.err"
    run python3 "$root/benchmarks/bench_scanner.py" --funnel --sizes 1000000 >> "$out/${tag}_scanner.jsonl" 2>> "$out/${tag}_scanner.err"
    run python3 "$root/benchmarks/decode_phases.py" 4096,16384,32768,65536 > "$out/${tag}_decode_phases.jsonl" 2> "$out/${tag}_decode_phases.err"
    # the Pillow route (formats outside the GPU decoders; decoder processes into shared page-locked buffers), small and camera-sized images
    : > "$out/${tag}_fastsig_pillow_route.jsonl"
    for spec in "webp 512 128 8192" "webp 2048 32 1024"; do
        set -- $spec
        run python3 "$root/benchmarks/bench_fastsig.py" --format "$1" --side "$2" --distinct "$3" --images "$4" --pillow-sample "$4" >> "$out/${tag}_fastsig_pillow_route.jsonl" 2>> "$out/${tag}_decode_phases.err"
    done
    # a collection of mixed formats in one call (70 % JPEG, 20 % PNG, 4 % BMP, 3 % WebP, 3 % TIFF): the GPU share and the Pillow share side by side
    : > "$out/${tag}_fastsig_collection.jsonl"
    run python3 "$root/benchmarks/bench_fastsig.py" --format collection --images 32768 --pillow-sample 8192 >> "$out/${tag}_fastsig_collection.jsonl" 2>> "$out/${tag}_decode_phases.err"
    run python3 "$root/benchmarks/bench_fastsig.py" --format collection --content drawing --images 65536 --pillow-sample 8192 >> "$out/${tag}_fastsig_collection.jsonl" 2>> "$out/${tag}_decode_phases.err"
    # GIF: first frame decoded on the GPU (ke_gif_decode) against the same files through the Pillow route
    : > "$out/${tag}_fastsig_gif.jsonl"
    for spec in "corpus 4096 4096" "corpus 32768 8192" "drawing 65536 8192"; do
        set -- $spec
        run python3 "$root/benchmarks/bench_fastsig.py" --format gif --content "$1" --images "$2" --pillow-sample "$3" >> "$out/${tag}_fastsig_gif.jsonl" 2>> "$out/${tag}_decode_phases.err"
    done
    : > "$out/${tag}_decode_gif.jsonl"
    for spec in "corpus 1024" "corpus 4096" "corpus 16384" "corpus 65536" "drawing 65536"; do
        set -- $spec
        run python3 "$root/benchmarks/bench_jpeg.py" --format gif --content "$1" --images "$2" >> "$out/${tag}_decode_gif.jsonl" 2>> "$out/${tag}_decode_phases.err"
    done
    # TIFF without compression: unpacked on the GPU (ke_tiff_decode) against the same files through the Pillow route
    : > "$out/${tag}_fastsig_tiff.jsonl"
    for spec in "512 256 16384 4096" "1024 64 4096 1024"; do
        set -- $spec
        run python3 "$root/benchmarks/bench_fastsig.py" --format tiff --side "$1" --distinct "$2" --images "$3" --pillow-sample "$4" >> "$out/${tag}_fastsig_tiff.jsonl" 2>> "$out/${tag}_decode_phases.err"
    done
    # BMP: unpacked on the GPU (ke_bmp_decode) against the same files through the Pillow route
    : > "$out/${tag}_fastsig_bmp.jsonl"
    for spec in "3500 16 256 256" "512 256 16384 4096" "1024 64 4096 1024"; do
        set -- $spec
        run python3 "$root/benchmarks/bench_fastsig.py" --format bmp --side "$1" --distinct "$2" --images "$3" --pillow-sample "$4" >> "$out/${tag}_fastsig_bmp.jsonl" 2>> "$out/${tag}_decode_phases.err"
    done
    ls "$out" | grep "^${tag}_" | head -60
    exit 0
fi
# 1. the bench lines themselves
run python3 "$B" --steps 10 --warmup 3 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
run python3 "$B" --steps 10 --warmup 3 --ssim-threshold 0.95 --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_bench_ssim095.json" 2> "$out/${tag}_bench_ssim095.err"
# 2. kernel statistics of the same commands
run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- python3 "$B" --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_stats.log" 2>&1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_ssim" -- python3 "$B" --steps 10 --warmup 3 --ssim-threshold 0.95 --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_stats_ssim.log" 2>&1
# 3. counters, one pass each (hash, scan and SSIM kernels all run in the --ssim-threshold step)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    name="$(echo "$c" | tr ' ' '+' | cut -c1-40)"
    run rocprofv3 --pmc $c --output-format csv -d "$out/${tag}_pmc_${name}" -- python3 "$B" --steps 2 --warmup 1 --ssim-threshold 0.95 --no-cpu-baseline --no-h2d --no-decode > "$out/${tag}_pmc_${name}.log" 2>&1
done
# 4. the scan alone at several table sizes (N = 1 000 000 is BASELINE configs[3]'s table), the streaming-read ceiling
run python3 "$root/benchmarks/scan_sizes.py" > "$out/${tag}_scan_sizes.jsonl" 2> "$out/${tag}_scan_sizes.err"
if [ -x "$root/benchmarks/micro/hbm_read.bin" ]; then run "$root/benchmarks/micro/hbm_read.bin" > "$out/${tag}_hbm_read.txt" 2>&1; fi
# 5. the decode step in front of the path: JPEG and PNG rates per batch size, kernel statistics of one batch of each
D="$root/benchmarks/bench_jpeg.py"
: > "$out/${tag}_decode.jsonl"
for spec in "jpeg corpus 4096" "jpeg corpus 16384" "jpeg corpus 65536" "png corpus 4096" "png corpus 16384" "png corpus 32768" "png drawing 4096" "png drawing 16384" "png drawing 65536"; do
    set -- $spec
    run python3 "$D" --format "$1" --content "$2" --images "$3" >> "$out/${tag}_decode.jsonl" 2>> "$out/${tag}_decode.err"
done
: > "$out/${tag}_fastsig.jsonl"
for spec in "jpeg corpus 16384" "jpeg corpus 65536" "jpeg corpus 131072" "mixed drawing 16384" "png drawing 65536" "png corpus 4096" "png corpus 16384"; do
    set -- $spec
    run python3 "$root/benchmarks/bench_fastsig.py" --format "$1" --content "$2" --images "$3" >> "$out/${tag}_fastsig.jsonl" 2>> "$out/${tag}_decode.err"
done
for n in 16384 65536; do
    run python3 "$D" --format jpeg --progressive --images "$n" >> "$out/${tag}_decode.jsonl" 2>> "$out/${tag}_decode.err"
done
run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_png" -- python3 "$D" --format png --images 4096 > "$out/${tag}_stats_png.log" 2>&1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_jpeg" -- python3 "$D" --format jpeg --images 65536 > "$out/${tag}_stats_jpeg.log" 2>&1
run rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_pngdrawing" -- python3 "$D" --format png --content drawing --images 16384 > "$out/${tag}_stats_pngdrawing.log" 2>&1
ls "$out" | grep "^${tag}_" | head -40
