"""kobato-eyes_amd: MI355X-native near-duplicate detection (pHash -> Hamming scan -> SSIM refine
-> clusters), a drop-in for the src/sig + src/dup hot path of srndpty/kobato-eyes.

Import name: ``kobato_eyes_amd`` (see kobato_eyes_amd.py at the repository root; the directory
name carries a hyphen).  All arithmetic runs in libkeyes_hip.so (csrc/, C ABI in
include/keyes.h); nothing here falls back to the CPU.
"""
from . import _native
from .api import DedupSettings, compute_signature, find_duplicates, run_duplicate_scan
from .cluster import Cluster, ClusterBuilder
from .cluster_update import (choose_keeper, cluster_hamming_score, default_checked_entries, rebuild_cluster_after_removal,
                             rebuild_clusters_after_removal, sort_entries_for_display)
from .fastsig import bulk_upsert_signatures, compute_signatures_mp, fast_fill_missing_signatures
from .phash import dhash, hamming64, hash_batch, phash, phash_dhash
from .refine_parallel import (refine_by_pixels_parallel, refine_by_tilehash_parallel, tile_ahash_bits,
                              tile_ahash_from_arrays, tile_hamming)
from .refine import RefinedMatch, RefinementThresholds, compute_ssim, refine_pair, refine_pairs, ssim_pairs
from .scanner import (DuplicateCluster, DuplicateClusterEntry, DuplicateFile, DuplicateScanConfig, DuplicateScanner,
                      assemble_clusters)
from .signature import compute_signatures_from_image, ensure_signatures

__all__ = [
    "phash", "dhash", "hamming64", "phash_dhash", "hash_batch", "compute_signature", "find_duplicates",
    "run_duplicate_scan", "DedupSettings", "DuplicateFile", "DuplicateCluster", "DuplicateClusterEntry",
    "DuplicateScanConfig", "DuplicateScanner", "assemble_clusters", "fast_fill_missing_signatures",
    "compute_signatures_mp", "bulk_upsert_signatures", "compute_signatures_from_image", "ensure_signatures",
    "tile_ahash_bits", "tile_hamming", "tile_ahash_from_arrays", "refine_by_tilehash_parallel", "refine_by_pixels_parallel",
    "choose_keeper", "sort_entries_for_display", "rebuild_cluster_after_removal", "rebuild_clusters_after_removal",
    "cluster_hamming_score", "default_checked_entries",
    "refine_pair", "refine_pairs", "compute_ssim", "ssim_pairs", "RefinementThresholds", "RefinedMatch", "Cluster", "ClusterBuilder",
]
