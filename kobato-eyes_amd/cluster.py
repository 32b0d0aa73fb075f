"""Connected components over refined matches: drop-in for the reference's ``dup.cluster``
(src/dup/cluster.py:12-70).  Components come from the library's host union-find."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable

import numpy as np

from . import _native
from .refine import RefinedMatch


@dataclass
class Cluster:
    representative: int
    members: list
    matches: list


class ClusterBuilder:
    def build(self, matches: Iterable[RefinedMatch]) -> list:
        kept = [m for m in matches if m.is_duplicate]
        if not kept:
            return []
        ids = sorted({fid for m in kept for fid in (m.file_id_a, m.file_id_b)})
        node = {fid: k for k, fid in enumerate(ids)}
        raw = np.zeros(len(kept), _native.EDGE_DTYPE)
        raw["a"] = [node[m.file_id_a] for m in kept]
        raw["b"] = [node[m.file_id_b] for m in kept]
        labels = _native.cluster_labels(raw, len(ids)).tolist()
        members: dict[int, list[int]] = {}
        for fid, lab in zip(ids, labels):
            members.setdefault(lab, []).append(fid)      # ids ascending -> members ascending
        by_label: dict[int, list[RefinedMatch]] = {}
        for m in kept:
            by_label.setdefault(labels[node[m.file_id_a]], []).append(m)
        clusters = [Cluster(representative=mem[0], members=mem, matches=by_label.get(lab, [])) for lab, mem in members.items()]
        clusters.sort(key=lambda c: c.representative)
        return clusters


__all__ = ["Cluster", "ClusterBuilder"]
