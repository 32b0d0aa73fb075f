// ke_cluster.cpp -- H1: connected components of the candidate graph on the host.
//
// Replaces DisjointSet + the union loop of DuplicateScanner.build_clusters
// (src/dup/scanner.py:176-200, 304-318) and ClusterBuilder's union-find
// (src/dup/cluster.py:30-46).  Cluster membership does not depend on union order, so an
// array-based union-find with path halving gives the same components; labels are made
// canonical (smallest member) so the host can group without a second pass.
#include <vector>

#include "ke_internal.h"

namespace {
inline int64_t find_root(int64_t *parent, int64_t x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}
}  // namespace

KE_API int ke_cluster_labels(const ke_edge *edges, int64_t n_edges, int64_t n_nodes, int64_t *label_out) {
    if (n_edges < 0 || n_nodes < 0 || (n_edges > 0 && !edges) || (n_nodes > 0 && !label_out)) return KE_EINVAL;
    // label_out doubles as the parent array (no allocation); only the end points of edges can end up under another root,
    // so the closing pass walks the edges, not the nodes: O(nodes) stores + O(edges) finds
    int64_t *parent = label_out;
    for (int64_t v = 0; v < n_nodes; ++v) parent[v] = v;
    for (int64_t e = 0; e < n_edges; ++e) {
        const int64_t a = edges[e].a, b = edges[e].b;
        if (a < 0 || b < 0 || a >= n_nodes || b >= n_nodes) return KE_EINVAL;
        const int64_t ra = find_root(parent, a), rb = find_root(parent, b);
        if (ra == rb) continue;
        if (ra < rb) parent[rb] = ra; else parent[ra] = rb;   // smaller index stays root
    }
    for (int64_t e = 0; e < n_edges; ++e) {                   // flatten: every touched node points at its root
        const int64_t ra = find_root(parent, edges[e].a);
        parent[edges[e].a] = ra;
        parent[edges[e].b] = ra;
    }
    return KE_OK;
}
