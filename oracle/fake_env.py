"""Deterministic stand-in for the Matterport3D batch environment (TEST INFRASTRUCTURE).

Emits the observation-dict schema of the reference's `R2RBatch.observe()` (environ/common_env.py:310-329) from a
seeded synthetic script, so that (a) `oracle/make_goldens.py` can drive the reference's own agents
(`src/agent/envdrop.py`, ...) without the simulator and (b) the tests can drive this repo's rollout with the SAME
inputs.  Only the interface the agents touch is provided: `reset(restart=...)`, `observe()`,
`step(actions, obs, traj)`, `.batch`, `.batch_size`.
"""
from __future__ import annotations

import math
from typing import List

import numpy as np

IMG, ANG, VIEWS = 2048, 128, 36


def angle_feat(heading: float, elevation: float) -> np.ndarray:
    return np.array([math.sin(heading), math.cos(heading), math.sin(elevation), math.cos(elevation)],
                    np.float32).repeat(ANG // 4)


class FakeR2REnv:
    """B scripted episodes.  Episode i has a path of n_i moves then STOP; at node k the teacher's next viewpoint is
    candidate `teach[i][k]`; any move (teacher or not) advances one node.  Distance to goal = 2 m per remaining move."""

    def __init__(self, batch_size=4, max_len=8, vocab=40, seed=0, max_cands=3, max_moves=3, img=IMG):
        self.batch_size = batch_size
        self.rng = np.random.default_rng(seed)
        r = self.rng
        lens = np.sort(r.integers(3, max_len + 1, batch_size))[::-1].copy()
        lens[0] = max_len
        self.batch = []
        for i in range(batch_size):
            enc = np.zeros(max_len, np.int64)
            enc[:lens[i]] = r.integers(4, vocab, lens[i])
            self.batch.append({"instr_id": f"ep{i}_0", "path_id": i, "instr_encoding": enc, "instr_length": int(lens[i]),
                               "instructions": "", "scan": "s"})
        self.n_moves = r.integers(1, max_moves + 1, batch_size)
        n_nodes = max_moves + 2 + 40                                   # sampled rollouts may wander: plenty of nodes
        self.ncand = r.integers(1, max_cands + 1, (batch_size, n_nodes))
        self.teach = np.array([[r.integers(0, self.ncand[i, k]) for k in range(n_nodes)] for i in range(batch_size)])
        # low-entropy but distinct features: a few base vectors mixed per (episode, node, view)
        self.img = img                                                 # width of the visual part (2048 in R2R)
        self.base = np.abs(r.standard_normal((8, img))).astype(np.float32) * 0.5
        self.mix = r.random((batch_size, n_nodes, VIEWS, 8)).astype(np.float32)
        self.cmix = r.random((batch_size, n_nodes, max_cands, 8)).astype(np.float32)
        self.chead = ((r.random((batch_size, n_nodes, max_cands)) - 0.5) * 6).astype(np.float32)
        self.celev = ((r.random((batch_size, n_nodes, max_cands)) - 0.5)).astype(np.float32)
        self.heading = ((r.random((batch_size, n_nodes)) - 0.5) * 6).astype(np.float32)
        self.view_index = r.integers(0, VIEWS, (batch_size, n_nodes))
        self.node = np.zeros(batch_size, np.int64)
        self.actions_log: List[np.ndarray] = []

    # -- reference interface --------------------------------------------------------------------------------------
    def reset(self, restart=False, batch=None):
        self.node[:] = 0
        self.actions_log = []
        return self.observe()

    def reset_epoch(self, shuffle=False):
        pass

    def _remaining(self, i):
        return max(int(self.n_moves[i]) - int(self.node[i]), 0)

    def observe(self):
        obs = []
        for i in range(self.batch_size):
            k = int(self.node[i])
            vis = self.mix[i, k] @ self.base / 8.0                                   # [36, IMG]
            ang = np.stack([angle_feat((v % 12) * math.pi / 6, (v // 12 - 1) * math.pi / 6) for v in range(VIEWS)])
            feature = np.concatenate((vis, ang), -1).astype(np.float32)
            cands = []
            at_goal = self._remaining(i) == 0
            for c in range(int(self.ncand[i, k])):
                cf = np.concatenate((self.cmix[i, k, c] @ self.base / 8.0, angle_feat(self.chead[i, k, c], self.celev[i, k, c])))
                cands.append({"feature": cf.astype(np.float32), "nextViewpointId": f"vp{i}_{k + 1}_{c}", "viewpointId": f"vp{i}_{k + 1}_{c}",
                              "absViewIndex": int(c), "heading": float(self.chead[i, k, c]), "elevation": float(self.celev[i, k, c])})
            vp = f"vp{i}_{k}"
            teacher = vp if at_goal else cands[int(self.teach[i, k])]["nextViewpointId"]
            b = self.batch[i]
            obs.append({"instr_id": b["instr_id"], "scan": "s", "viewpointId": vp, "viewIndex": int(self.view_index[i, k]),
                        "heading": float(self.heading[i, k]), "elevation": 0.0, "feature": feature, "candidates": cands,
                        "teacher": teacher, "path_id": b["path_id"], "instr_encoding": b["instr_encoding"],
                        "instr_length": b["instr_length"], "distance": float(2.0 * self._remaining(i) + (0.0 if at_goal else 1.5))})
        return obs

    def step(self, actions, obs=None, traj=None):
        a = np.asarray(actions).copy()
        self.actions_log.append(a)
        for i, x in enumerate(a):
            if x >= 0:
                self.node[i] += 1
                if traj is not None:
                    traj[i]["path"].append((f"vp{i}_{int(self.node[i])}", 0.0, 0.0))
        return self.observe()
