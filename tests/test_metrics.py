"""Evaluation-path bookkeeping, host side: known-answer constants the reference carries in its own
doctests (src/utils/dtw.py:27-35, src/utils/cls.py:33-39 -- the only known-answer values in the reference) and a
golden captured from the reference's `Evaluation.score` (oracle/make_goldens.py::gen_eval_scores)."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vln_amd as vln  # noqa: E402

M = vln.metrics


def grid(w, h):
    """nx.grid_graph([3, 4]) of the doctests: nodes (x, y), x < 4, y < 3, unit edges."""
    e = []
    for x in range(w):
        for y in range(h):
            if x + 1 < w:
                e.append(((x, y), (x + 1, y), 1.0))
            if y + 1 < h:
                e.append(((x, y), (x, y + 1), 1.0))
    return M.shortest_paths(e)


def test_dtw_known_answers():
    d = grid(4, 3)
    pred = [(0, 0), (1, 0), (2, 0), (3, 0)]
    ref = [(0, 0), (1, 0), (2, 1), (3, 2)]
    dtw, ndtw, sdtw = M.dtw_scores(pred, ref, d)
    assert math.isclose(dtw, 3.0)
    assert math.isclose(ndtw, 0.77880078307140488, rel_tol=1e-12)
    assert math.isclose(sdtw, 0.77880078307140488, rel_tol=1e-12)
    assert M.dtw_scores(pred[:2], ref, d)[2] == 0.0


def test_cls_known_answers():
    d = grid(4, 3)
    ref = [(0, 0), (1, 0), (1, 1), (2, 1), (2, 2), (3, 2)]
    assert math.isclose(M.cls_score(ref, ref, d), 1.0)
    assert math.isclose(M.cls_score(ref, [(0, 0), (0, 1), (1, 1), (2, 1), (3, 1), (3, 2)], d), 0.81994915125863865, rel_tol=1e-12)
    assert math.isclose(M.cls_score(ref, [(0, 1), (1, 1), (2, 1), (3, 1)], d), 0.44197196102702557, rel_tol=1e-12)


def test_score_trajectories_matches_reference_evaluator():
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "eval_scores.json")))
    dist = {"s": M.shortest_paths([tuple(e) for e in G["edges"]])}
    summary, scores = M.score_trajectories(G["results"], G["gt"], dist)
    for k, v in G["summary"].items():
        assert math.isclose(summary[k], v, rel_tol=1e-9, abs_tol=1e-12), k
    for k, v in G["scores"].items():
        assert all(math.isclose(a, b, rel_tol=1e-9, abs_tol=1e-12) for a, b in zip(scores[k], v)), k
    with pytest.raises(ValueError):
        M.score_trajectories(G["results"][:-1], G["gt"], dist)         # a missing instruction id is an error there too
