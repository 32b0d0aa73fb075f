"""Cluster maintenance after user actions: drop-ins for the reference's
``ui.dup_cluster_update`` (src/ui/dup_cluster_update.py:10-78) and the two cluster helpers of
``ui.dup_tree_state`` (src/ui/dup_tree_state.py:45-59).  Pure host code on the scanner's dataclasses; the
keeper and display orders are the ones build_clusters itself uses."""
from __future__ import annotations

from typing import Optional, Sequence

from .scanner import DuplicateCluster, DuplicateClusterEntry, _keeper_key


def choose_keeper(entries: Sequence[DuplicateClusterEntry]) -> int:
    return min(entries, key=_keeper_key).file.file_id


def sort_entries_for_display(entries: Sequence[DuplicateClusterEntry], keeper_id: int) -> list:
    def display_key(entry):
        f = entry.file
        return (f.file_id != keeper_id, -(f.size or 0), -f.resolution, -f.extension_priority, f.path.name.lower(), f.file_id)

    return sorted(entries, key=display_key)


def rebuild_cluster_after_removal(cluster: DuplicateCluster, removed_ids: set) -> Optional[DuplicateCluster]:
    """The cluster without the removed files, with keeper and order recomputed; None when fewer than two remain."""
    left = [e for e in cluster.files if e.file.file_id not in removed_ids]
    if len(left) < 2:
        return None
    keeper = choose_keeper(left)
    return DuplicateCluster(files=sort_entries_for_display(left, keeper), keeper_id=keeper)


def rebuild_clusters_after_removal(clusters: Sequence[DuplicateCluster], removed_ids: set) -> list:
    rebuilt = (rebuild_cluster_after_removal(c, removed_ids) for c in clusters)
    return [c for c in rebuilt if c is not None]


def cluster_hamming_score(cluster: DuplicateCluster) -> int:
    """Largest best_hamming among the non-keepers, -1 when none is known."""
    scores = [e.best_hamming for e in cluster.files if e.file.file_id != cluster.keeper_id and e.best_hamming is not None]
    return max(scores, default=-1)


def default_checked_entries(cluster: DuplicateCluster) -> list:
    return [e for e in cluster.files if e.file.file_id != cluster.keeper_id]


__all__ = ["choose_keeper", "sort_entries_for_display", "rebuild_cluster_after_removal", "rebuild_clusters_after_removal",
           "cluster_hamming_score", "default_checked_entries"]
