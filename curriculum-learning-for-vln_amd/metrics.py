"""Host-side navigation metrics for the evaluation path.

The greedy rollouts (`feedback="argmax"`, reference `BaseAgent.test`, src/agent/base.py:63-82) run on the HIP
modules; what is left is bookkeeping over the recorded trajectories, which stays on the host exactly like the
reference's `Evaluation.score` (src/engine/evaluator.py:101-146) with its DTW (src/utils/dtw.py:55-82) and CLS
(src/utils/cls.py:63-90) helpers.  Distances are plain nested dicts `dist[u][v]` (the reference builds them with
networkx; `shortest_paths` below is a dependency-free Dijkstra over an edge list).
"""
from __future__ import annotations

import heapq
import math
from collections import defaultdict
from typing import Dict, Hashable, Iterable, List, Mapping, Sequence, Tuple

Node = Hashable
Dist = Mapping[Node, Mapping[Node, float]]


def shortest_paths(edges: Iterable[Tuple[Node, Node, float]]) -> Dict[Node, Dict[Node, float]]:
    """All-pairs shortest path lengths of an undirected weighted graph (what evaluator.py:45-47 asks networkx for)."""
    adj: Dict[Node, List[Tuple[Node, float]]] = defaultdict(list)
    for u, v, w in edges:
        adj[u].append((v, float(w)))
        adj[v].append((u, float(w)))
    out = {}
    for src in adj:
        best = {src: 0.0}
        heap = [(0.0, 0, src)]
        tie = 1
        while heap:
            d, _, u = heapq.heappop(heap)
            if d > best.get(u, math.inf):
                continue
            for v, w in adj[u]:
                nd = d + w
                if nd < best.get(v, math.inf):
                    best[v] = nd
                    heapq.heappush(heap, (nd, tie, v))
                    tie += 1
        out[src] = best
    return out


def dtw_scores(prediction: Sequence[Node], reference: Sequence[Node], dist: Dist, threshold: float = 3.0):
    """(dtw, ndtw, sdtw) -- dtw.py:66-80: classic O(P*R) table over graph distances; ndtw = exp(-dtw / (threshold*|ref|)),
    sdtw = ndtw if the last predicted node is within `threshold` of the last reference node, else 0."""
    P, R = len(prediction), len(reference)
    prev = [math.inf] * (R + 1)
    prev[0] = 0.0
    for i in range(1, P + 1):
        cur = [math.inf] * (R + 1)
        du = dist[prediction[i - 1]]
        for j in range(1, R + 1):
            cur[j] = du[reference[j - 1]] + min(prev[j], cur[j - 1], prev[j - 1])
        prev = cur
    dtw = prev[R]
    ndtw = math.exp(-dtw / (threshold * R))
    success = dist[prediction[-1]][reference[-1]] <= threshold
    return dtw, ndtw, (ndtw if success else 0.0)


def cls_score(prediction: Sequence[Node], reference: Sequence[Node], dist: Dist, threshold: float = 3.0) -> float:
    """cls.py:63-90 with its argument roles: coverage of `reference` by `prediction`, weighted by a length score."""
    def length(nodes):
        return float(sum(dist[a][b] for a, b in zip(nodes[:-1], nodes[1:])))

    coverage = sum(math.exp(-min(dist[u][v] for v in prediction) / threshold) for u in reference) / len(reference)
    expected = coverage * length(reference)
    return coverage * expected / (expected + abs(expected - length(prediction)))


def score_trajectories(results: Sequence[Mapping], gt: Mapping[str, Mapping], distances: Mapping[str, Dist],
                       error_margin: float = 3.0, test_split: bool = False):
    """evaluator.py:49-146.  results: [{'instr_id', 'trajectory': [(viewpoint, heading, elevation), ...]}];
    gt[instr_id] = {'scan', 'path': [viewpoints]}.  Returns (summary dict, per-item score lists)."""
    sc = defaultdict(list)
    seen = set()
    for item in results:
        iid = item["instr_id"]
        if iid not in gt or iid in seen:
            continue
        seen.add(iid)
        g = gt[iid]
        d = distances[g["scan"]]
        path = [p[0] for p in item["trajectory"]]
        start, goal = g["path"][0], g["path"][-1]
        if path[0] != start:
            raise ValueError("Result trajectories should include the start position")
        final = path[-1]
        nearest = min(path, key=lambda v: d[v][goal])                  # first minimiser, like evaluator.py:49-57
        sc["nav_errors"].append(d[final][goal])
        sc["oracle_errors"].append(d[nearest][goal])
        sc["trajectory_steps"].append(len(path) - 1)
        _, ndtw, sdtw = dtw_scores(path, g["path"], d, error_margin)
        sc["ndtws"].append(ndtw)
        sc["sdtws"].append(sdtw)
        sc["clss"].append(cls_score(path, g["path"], d, error_margin))
        dist_m = sum(d[a][b] for a, b in zip(path[:-1], path[1:]))
        sc["trajectory_lengths"].append(dist_m)
        ok = d[final][goal] < error_margin
        sc["success_path_length"].append(0.0 if test_split else ok * d[start][goal] / max(d[start][goal], dist_m, 1e-12))
    missing = set(gt) - seen
    if missing:
        raise ValueError(f"Missing {len(missing)} of {len(gt)} instruction ids")
    n = len(sc["nav_errors"])
    mean = lambda k: float(sum(sc[k]) / n)
    summary = {"nav_error": mean("nav_errors"), "oracle_error": mean("oracle_errors"), "steps": mean("trajectory_steps"),
               "lengths": mean("trajectory_lengths"), "spl": mean("success_path_length"), "ndtw": mean("ndtws"),
               "sdtw": mean("sdtws"), "cls": mean("clss"),
               "success_rate": sum(e < error_margin for e in sc["nav_errors"]) / n,
               "oracle_rate": sum(e < error_margin for e in sc["oracle_errors"]) / n}
    return summary, dict(sc)
