"""Synthetic episode batches of BASELINE.md section 3 / SURVEY.md section 8(d): what `bench.py`, the parity tests and the
measurement scripts feed the iterations of `trainers.py` with (there is no Matterport simulator and no R2R data on the box).

A TAPE is one episode batch the way the reference's environment hands it to an agent (common_env.py:310-329, marshalled by
agent/base.py:114-178): instruction tokens sorted by length, and per decoder step the viewpoint rows, the agent's view index,
the candidates (panorama view + relative heading / elevation), the candidate mask, the previous action's angle feature and
the teacher action.  Features are defined through a per-viewpoint ResNet table, like `ImageFeatures` (utils/misc.py:253-279)."""
from __future__ import annotations

import torch

from .staging import DeviceFeatureStore, loc_embedding_table

N_VIEWPOINTS = 10567      # panoramas in the R2R ResNet-152 feature TSV (ImageFeatures.read_in, utils/misc.py:253-279)
N_TAPES = 8               # distinct episode batches a timed loop rotates through


def make_tape(B, L, T, C_max, seed, vocab=992, V=36, IMG=2048, ANG=128, n_rows=None):
    """Synthetic episode batch (BASELINE.md §3 / SURVEY.md §8d), CPU tensors.  Features are defined the way the
    reference's environment builds them (common_env.py:272,287-291,307-308): a per-viewpoint ResNet table
    [viewpoints, 36 views, 2048] (post-ReLU, non-negative), the agent's viewIndex selecting the static angle
    table, and each candidate = (view of the current panorama, its relative heading/elevation).
    n_rows=None: the tape brings its own compact table (T*B viewpoints) and the explicit img/cand tensors of every step
    (tensor path, CPU baseline, tests).  n_rows=N: INDEX-ONLY tape over a resident table of N viewpoints (the bench's
    DeviceFeatureStore): every step visits B random viewpoints, one row of every step has all C_max candidate slots in use
    so the padded candidate width is the same for every tape."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values
    lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, 0] = 3                                        # <BOS>
        tokens[i, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g)
        tokens[i, n - 1] = 2                                    # <EOS>
    seq_mask = tokens == 0
    table = None if n_rows is not None else torch.randn(T * B, V, IMG, generator=g).abs() * 0.5
    T_i = torch.randint(min(4, T), T + 1, (B,), generator=g)
    T_i[0] = T
    steps = []
    for t in range(T):
        rows = torch.arange(B) + t * B if n_rows is None else torch.randint(0, n_rows, (B,), generator=g)
        vidx = torch.randint(0, V, (B,), generator=g).int()
        ncand = torch.randint(3, C_max + 1, (B,), generator=g)      # candidates incl. the STOP slot
        if n_rows is not None:
            ncand[int(torch.randint(0, B, (1,), generator=g))] = C_max
        Ct = int(ncand.max())
        cmask = torch.arange(Ct)[None, :] >= ncand[:, None]
        real = torch.arange(Ct)[None, :] < (ncand - 1)[:, None]     # STOP slot + padding are all-zero rows
        crow = torch.where(real, rows[:, None].expand(B, Ct), torch.full((B, Ct), -1))
        cview = torch.randint(0, V, (B, Ct), generator=g).int()
        chead = (torch.rand(B, Ct, generator=g) - 0.5) * 6.0
        celev = (torch.rand(B, Ct, generator=g) - 0.5) * 1.04
        ended = t >= T_i
        tgt = torch.where(t == T_i - 1, ncand - 1, (torch.rand(B, generator=g) * (ncand - 1).float()).long())
        tgt = torch.where(ended, torch.full_like(tgt, -1), tgt)
        ah = torch.rand(B, generator=g) * 6.283 - 3.1415
        st = dict(cand_mask=cmask, angle=angle_feat(ah, torch.zeros(B), ANG), target=tgt, rows=rows,
                  vidx=vidx, crow=crow, cview=cview, chead=chead, celev=celev)
        if table is not None:
            st.update(materialize_step(st, table, ANG))
        steps.append(st)
    return dict(tokens=tokens, lengths=lens, seq_mask=seq_mask, steps=steps, table=table, B=B, L=L, T=T, IMG=IMG, ANG=ANG)


def angle_feat(h, e, ANG=128):                                  # utils/misc.py:285-293
    return torch.stack([h.sin(), h.cos(), e.sin(), e.cos()], -1).repeat_interleave(ANG // 4, dim=-1)


def materialize_step(st, table, ANG=128):
    """The explicit img [B,36,F] / cand [B,C,F] tensors of a step from the ResNet table (any device), built the way the
    reference's marshalling does (agent/base.py:141-157): what the tensor / host feature modes and the CPU baseline consume."""
    dev = table.device
    V = table.shape[1]
    loc_table = loc_embedding_table(ANG, V).to(dev)            # [V, V, ANG], misc.py:296-317
    rows, crow = st["rows"].to(dev), st["crow"].to(dev)
    real = (crow >= 0)
    img = torch.cat((table[rows].float(), loc_table[st["vidx"].to(dev).long()]), -1)
    cand = torch.cat((table[crow.clamp_min(0), st["cview"].to(dev).long()].float(),
                      angle_feat(st["chead"].to(dev), st["celev"].to(dev), ANG)), -1) * real[..., None]
    return dict(img=img, cand=cand)


def tape_to(tape, dev, store_dtype=None, host_dtype=None, store=None):
    """Device copy.  With `store_dtype` the tape's own ResNet table becomes a resident DeviceFeatureStore (or `store` = an
    existing one, for index-only tapes) and the per-step img/cand tensors are NOT uploaded (a step only needs its index
    vectors).  With `host_dtype` the per-step img/cand tensors stay on the HOST, pinned, in that dtype (fp32 = what the
    reference's ImageFeatures holds, utils/misc.py:253-279; bf16 = converted once at load time): every step then pays its
    H2D copy (PCIe-inclusive mode, never the headline value)."""
    skip = ("steps", "table")
    out = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tape.items() if k not in skip}
    out["lengths32"] = tape["lengths"].to(dev, torch.int32)
    drop = ("img", "cand") if (store_dtype is not None or host_dtype is not None or store is not None) else ()
    out["steps"] = [{k: v.to(dev) for k, v in s.items() if k not in drop} for s in tape["steps"]]
    if host_dtype is not None:
        for so, si in zip(out["steps"], tape["steps"]):
            so["img_host"] = si["img"].to("cpu", host_dtype).contiguous().pin_memory()
            so["cand_host"] = si["cand"].to("cpu", host_dtype).contiguous().pin_memory()
    if store is not None:
        out["store"] = store
    elif store_dtype is not None:
        out["store"] = DeviceFeatureStore(tape["table"], device=dev, dtype=store_dtype, angle_size=tape["ANG"])
    return out


def build_store(dev, dtype, n_rows=N_VIEWPOINTS, V=36, IMG=2048, ANG=128, seed=2020):
    """The full-size resident feature table: n_rows x 36 x 2048 (1.56 GB in bf16, 3.1 GB in fp32 -- the reference keeps 2.9 GB
    of fp32 on the host), generated on the device chunk by chunk.  Post-ReLU statistics: |N(0,1)| * 0.5."""
    g = torch.Generator(device=dev).manual_seed(seed)
    table = torch.empty(n_rows, V, IMG, dtype=dtype, device=dev)
    for r0 in range(0, n_rows, 512):
        r1 = min(n_rows, r0 + 512)
        table[r0:r1] = (torch.randn(r1 - r0, V, IMG, generator=g, device=dev).abs_() * 0.5).to(dtype)
    return DeviceFeatureStore(table, device=dev, dtype=dtype, angle_size=ANG)


class GoalGridWorld:
    """A LEARNABLE stand-in for R2R (there is no Matterport simulator or R2R data on the box): a G x G torus of viewpoints, four
    neighbours each (headings 0, pi/2, pi, 3 pi/2; edges `edge_m` metres long), a fixed ResNet-like panorama per viewpoint.  Goal
    viewpoints (x % 3 == 0 and y % 3 == 0) carry a visual landmark in every view; an episode starts 1-3 moves away from a goal in a
    straight line, its instruction names the direction (one of four direction words repeated, BOS / EOS around them) and the teacher
    is the shortest path: that direction until the landmark is in view, then STOP.  An agent that learns (direction word -> the
    candidate with that heading; landmark -> STOP) reaches success rate 1; a random policy about 0.1.

    Used by the learnability tests (loss curve of the HIP path against the oracle, greedy success rate via metrics.py) -- the
    stand-in for north_star's SR / SPL clause, which needs the real simulator.  Observations follow the reference's environment:
    candidates = (a view of the CURRENT panorama, relative heading / elevation), the last real slot is STOP (all-zero row,
    agent/base.py:152-153); ended episodes stay where they are and carry target -1 (envdrop.py:199-203)."""

    HEADINGS = (0.0, 1.5707963, 3.1415927, 4.712389)      # N, E, S, W
    MOVES = ((0, 1), (1, 0), (0, -1), (-1, 0))
    VIEW_OF = (12, 15, 18, 21)                             # the horizontal views that look along the four headings
    DIR_TOKENS = (4, 5, 6, 7)

    def __init__(self, G=9, IMG=2048, ANG=128, V=36, seed=0, edge_m=4.0, landmark=2.0):
        assert G % 3 == 0
        self.G, self.IMG, self.ANG, self.V, self.edge_m = G, IMG, ANG, V, edge_m
        g = torch.Generator().manual_seed(seed)
        self.N = G * G
        table = torch.randn(self.N, V, IMG, generator=g).abs() * 0.5
        self.goal = torch.tensor([(n // G) % 3 == 0 and (n % G) % 3 == 0 for n in range(self.N)])
        table[self.goal, :, :64] += landmark               # the landmark: a bump in the first 64 channels of every view
        self.table = table
        # candidate order per viewpoint: a fixed permutation of the four directions (the agent cannot learn "slot 0 = north")
        self.perm = torch.stack([torch.randperm(4, generator=g) for _ in range(self.N)])       # [N, 4]: slot -> direction

    def node(self, x, y):
        return (x % self.G) * self.G + (y % self.G)

    def neighbour(self, n, d):
        x, y = n // self.G, n % self.G
        dx, dy = self.MOVES[d]
        return self.node(x + dx, y + dy)

    def edges(self):
        return [(f"v{n}", f"v{self.neighbour(n, d)}", self.edge_m) for n in range(self.N) for d in range(2)]

    def episodes(self, B, seed, L=12):
        """B episodes: (start node, direction, moves to the goal), instruction tokens sorted by length (descending, like
        common_env.py:204-205)."""
        g = torch.Generator().manual_seed(seed)
        eps = []
        goals = [n for n in range(self.N) if bool(self.goal[n])]
        for _ in range(B):
            goal = goals[int(torch.randint(0, len(goals), (1,), generator=g))]
            d = int(torch.randint(0, 4, (1,), generator=g))
            n_moves = int(torch.randint(1, 3, (1,), generator=g))           # 1 or 2: the next goal along a line is 3 away
            start = goal
            for _ in range(n_moves):
                start = self.neighbour(start, (d + 2) % 4)
            length = int(torch.randint(4, L + 1, (1,), generator=g))
            eps.append(dict(start=start, goal=goal, d=d, n=n_moves, length=length))
        eps.sort(key=lambda e: -e["length"])
        eps[0]["length"] = L
        tokens = torch.zeros(B, L, dtype=torch.long)
        for i, e in enumerate(eps):
            n = e["length"]
            tokens[i, 0] = 3
            tokens[i, 1:n - 1] = self.DIR_TOKENS[e["d"]]
            tokens[i, n - 1] = 2
        return eps, tokens, torch.tensor([e["length"] for e in eps])

    def observe(self, nodes):
        """Index vectors of one step for the episodes standing at `nodes` [B] (what agent/base.py:141-157 marshals): 5 candidate
        slots = the four neighbours in the viewpoint's own order + STOP (zero row, crow = -1)."""
        B = len(nodes)
        nodes = torch.as_tensor(nodes, dtype=torch.long)
        dirs = self.perm[nodes]                                              # [B, 4]
        crow = torch.cat((nodes[:, None].expand(B, 4), torch.full((B, 1), -1, dtype=torch.long)), 1)
        cview = torch.cat((torch.tensor(self.VIEW_OF)[dirs], torch.zeros(B, 1, dtype=torch.long)), 1).int()
        chead = torch.cat((torch.tensor(self.HEADINGS)[dirs], torch.zeros(B, 1)), 1)
        return dict(rows=nodes.clone(), vidx=torch.full((B,), 12, dtype=torch.int32), crow=crow, cview=cview, chead=chead,
                    celev=torch.zeros(B, 5), cand_mask=torch.zeros(B, 5, dtype=torch.bool),
                    angle=angle_feat(torch.zeros(B), torch.zeros(B), self.ANG))

    def teacher(self, nodes, eps):
        """Shortest-path action per episode: the slot of its direction, or STOP (slot 4) at the goal."""
        out = []
        for n, e in zip(torch.as_tensor(nodes).tolist(), eps):
            out.append(4 if n == e["goal"] else int((self.perm[n] == e["d"]).nonzero()[0]))
        return torch.tensor(out)

    def step(self, nodes, actions, ended):
        """env.step: every running episode moves to the chosen neighbour; STOP (slot 4) or -1 ends / keeps it."""
        nxt, end = [], []
        for n, a, e in zip(torch.as_tensor(nodes).tolist(), torch.as_tensor(actions).tolist(), torch.as_tensor(ended).tolist()):
            if e or a < 0 or a == 4:
                nxt.append(n); end.append(True)
            else:
                nxt.append(self.neighbour(n, int(self.perm[n, a]))); end.append(False)
        return torch.tensor(nxt), torch.tensor(end)

    def tape(self, B, seed, L=12, T=3):
        """A teacher-forced episode batch in make_tape()'s format (index-only: the table is this world's), T steps."""
        eps, tokens, lens = self.episodes(B, seed, L)
        nodes = torch.tensor([e["start"] for e in eps])
        ended = torch.zeros(B, dtype=torch.bool)
        steps = []
        for _ in range(T):
            st = self.observe(nodes)
            tgt = self.teacher(nodes, eps)
            st["target"] = torch.where(ended, torch.full_like(tgt, -1), tgt)
            steps.append(st)
            nodes, ended = self.step(nodes, st["target"], ended)
        return dict(tokens=tokens, lengths=lens, seq_mask=tokens == 0, steps=steps, table=None, B=B, L=L, T=T, IMG=self.IMG,
                    ANG=self.ANG, episodes=eps)
