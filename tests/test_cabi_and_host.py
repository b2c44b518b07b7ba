"""CPU: the C-ABI library loads and exports every symbol include/vln_hip.h declares (no compute calls -- there
is no GPU here), the ctypes table covers exactly that set, and the host-side runtime logic (stash arena,
shadow keys, synthetic tape) behaves."""
import os
import re
import subprocess
import weakref

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vln_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vln_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    if not os.path.exists(vln_amd.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return vln_amd


def test_header_symbols_are_exported_and_typed(vln):
    syms = declared_symbols()
    assert len(syms) >= 25 and "vln_envdrop_step_fwd" in syms and "vln_lstm_seq_bwd" in syms
    lib = vln._lib.load()
    out = subprocess.run(["nm", "-D", "--defined-only", vln.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vln_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared in vln_hip.h but not exported: {missing}"
    untyped = [s for s in syms if s not in vln._lib.SIGNATURES]
    assert not untyped, f"declared but absent from the ctypes table: {untyped}"
    stale = [s for s in vln._lib.SIGNATURES if s not in syms]
    assert not stale, f"ctypes table lists undeclared symbols: {stale}"
    assert lib.vln_abi_version() == vln._lib.EXPECTED_ABI
    src = open(os.path.join(ROOT, "curriculum-learning-for-vln_amd", "csrc", "api.hip")).read()
    assert f"vln_abi_version(void) {{ return {vln._lib.EXPECTED_ABI}; }}" in src     # binding and sources move together
    assert lib.vln_prof_kernel_name(0) == b"gemm_nt"


def test_struct_layouts_match_header(vln):
    """The ctypes structs must list the header's fields in the header's order."""
    txt = open(os.path.join(ROOT, "include", "vln_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)

    def fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = re.sub(r"^(const\s+)?[a-z0-9_]+\s*\**", "", decl, count=1)
            out += [re.sub(r"\[\w+\]$", "", n.strip().lstrip("*").strip()) for n in names.split(",")]
        return out

    L = vln._lib
    assert fields("vln_envdrop_dims") == [f for f, _ in L.EnvDropDims._fields_]
    assert fields("vln_envdrop_weights") == [f for f, _ in L.EnvDropWeights._fields_]
    assert fields("vln_envdrop_step") == [f for f, _ in L.EnvDropStep._fields_]
    assert fields("vln_envdrop_grads") == [f for f, _ in L.EnvDropGrads._fields_]
    assert fields("vln_monitor_dims") == [f for f, _ in L.MonitorDims._fields_]
    assert fields("vln_monitor_weights") == [f for f, _ in L.MonitorWeights._fields_]
    assert fields("vln_monitor_step") == [f for f, _ in L.MonitorStep._fields_]
    assert fields("vln_monitor_grads") == [f for f, _ in L.MonitorGrads._fields_]
    assert fields("vln_follower_dims") == [f for f, _ in L.FollowerDims._fields_]
    assert fields("vln_follower_weights") == [f for f, _ in L.FollowerWeights._fields_]
    assert fields("vln_follower_step") == [f for f, _ in L.FollowerStep._fields_]
    assert fields("vln_follower_grads") == [f for f, _ in L.FollowerGrads._fields_]
    assert fields("vln_bn_affine") == [f for f, _ in L.BnAffine._fields_]
    assert fields("vln_bn_mlp_layer") == [f for f, _ in L.BnMlpLayer._fields_]
    assert fields("vln_bn_mlp") == [f for f, _ in L.BnMlp._fields_]
    assert fields("vln_bn_mlp_grad_layer") == [f for f, _ in L.BnMlpGradLayer._fields_]
    assert fields("vln_bn_mlp_grads") == [f for f, _ in L.BnMlpGrads._fields_]
    assert fields("vln_cat_step") == [f for f, _ in L.CatStep._fields_]


def test_modules_fail_loudly_without_gpu(vln):
    dec = vln.EnvDropDecoder(64, 0.5, 0.3, 16, 32, 128)
    B = 2
    with pytest.raises(vln.VlnError):
        dec(torch.zeros(B, 32), torch.zeros(B, 36, 128), torch.zeros(B, 3, 128), torch.zeros(B, 64), None,
            torch.zeros(B, 64), torch.zeros(B, 5, 64))
    enc = vln.EncoderLSTM(50, 16, 32, 0, 0.5, True, 1)
    with pytest.raises(vln.VlnError):
        enc(torch.zeros(B, 5, dtype=torch.long), torch.tensor([5, 3]))


def test_state_dict_keys_match_reference_goldens(vln):
    from conftest import load_golden
    G = load_golden("envdrop_step")
    dec = vln.EnvDropDecoder(int(G["cfg"]["H"]), 0.5, 0.3, int(G["cfg"]["AE"]), int(G["cfg"]["ANG"]),
                             int(G["cfg"]["IMG"]) + int(G["cfg"]["ANG"]))
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == {k: tuple(v.shape) for k, v in G["param"].items()}
    for name in ("encoder_envdrop", "encoder_follower", "encoder_monitor"):
        G = load_golden(name)
        c = G["cfg"]
        enc = vln.EncoderLSTM(int(c["vocab"]), int(c["E"]), int(c["H"]), 0, 0.5, bool(c["bidir"]), int(c["layers"]))
        assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == {k: tuple(v.shape) for k, v in G["param"].items()}
    G = load_golden("critic")
    assert set(vln.Critic(64, 0.5).state_dict()) == set(G["param"])


class _Owner:
    pass


def test_stash_arena(vln):
    from importlib import import_module
    rt = vln.runtime
    st = rt.Stash({"x": 4, "dy": 2}, torch.device("cpu"))
    st.CHUNK_ROWS = 8
    o1, o2 = _Owner(), _Owner()
    r1, r2 = weakref.ref(o1), weakref.ref(o2)
    a = st.take(3, r1); b = st.take(3, r1); c = st.take(3, r2)      # third block does not fit -> new chunk
    assert a.chunk is b.chunk and c.chunk is not a.chunk and (a.r0, b.r0, c.r0) == (0, 3, 0)
    a.view("x").fill_(1.0); b.view("x").fill_(2.0)
    assert torch.equal(a.chunk.bufs["x"][:6, 0], torch.tensor([1., 1, 1, 2, 2, 2]))
    # only steps whose backward ran are contracted; adjacent ones merge into one run
    a.done = True; b.done = True
    runs = list(st.done_runs())
    assert [(r0, r1_) for _, r0, r1_ in runs] == [(0, 6)]
    assert list(st.done_runs()) == []                                # flags are consumed
    b.done = True; c.done = True
    assert sorted((r0, r1_) for _, r0, r1_ in st.done_runs()) == [(0, 3), (3, 6)]
    # chunks whose rollouts died are recycled instead of growing the arena (graphs built but never backpropagated)
    first = a.chunk
    del o1
    d = st.take(8, r2)                                               # forces a new chunk -> reclaims `first`
    e = st.take(8, r2)
    assert d.chunk is first or e.chunk is first


def test_shadow_key_tracks_optimizer_steps(vln):
    p = torch.nn.Parameter(torch.zeros(3))
    k0 = vln.runtime.ShadowSet.key_of([p], torch.float32)
    with torch.no_grad():
        p.add_(1.0)
    assert vln.runtime.ShadowSet.key_of([p], torch.float32) != k0
    assert vln.runtime.ShadowSet.key_of([p], torch.bfloat16) != vln.runtime.ShadowSet.key_of([p], torch.float32)


def test_synthetic_tape_follows_the_obs_contract():
    import bench
    import vln_amd as vln
    t = vln.synthetic.make_tape(8, 20, 5, 6, seed=1)
    assert t["tokens"].shape == (8, 20) and t["lengths"][0] == 20
    assert (t["lengths"][:-1] >= t["lengths"][1:]).all()             # sorted descending (common_env.py:204-205)
    assert ((t["tokens"] == 0) == t["seq_mask"]).all()
    for s in t["steps"]:
        B, C, F = s["cand"].shape
        assert F == 2176 and s["img"].shape == (8, 36, 2176)
        n = (~s["cand_mask"]).sum(1)                                 # candidates incl. STOP
        for i in range(B):
            assert s["cand"][i, n[i] - 1:].abs().sum() == 0          # STOP slot + padding are zero rows (base.py:152-153)
            assert s["target"][i] == -1 or 0 <= s["target"][i] < n[i]
        assert (s["img"][..., :2048] >= 0).all()
    assert bench.usable_cores() >= 1


def test_feature_tsv_reader_matches_reference_format(tmp_path):
    """Row N2: `ImageFeatures.read_in` file format (utils/misc.py:253-279): TSV, base64 float32 [36, 2048] per viewpoint."""
    import base64
    import numpy as np
    from vln_amd import staging
    rng = np.random.default_rng(0)
    feats = {("scanA", "vp1"): rng.random((36, 2048), dtype=np.float32), ("scanA", "vp2"): rng.random((36, 2048), dtype=np.float32),
             ("scanB", "vp9"): rng.random((36, 2048), dtype=np.float32)}
    p = tmp_path / "feats.tsv"
    with open(p, "w") as f:
        for (scan, vp), a in feats.items():
            f.write("\t".join([scan, vp, "640", "480", "60", base64.b64encode(a.tobytes()).decode("ascii")]) + "\n")
    table, ids = staging.read_feature_tsv(str(p))
    assert ids == ["scanA_vp1", "scanA_vp2", "scanB_vp9"] and table.shape == (3, 36, 2048)
    for i, a in enumerate(feats.values()):
        assert np.array_equal(table[i].numpy(), a)
    with open(p, "a") as f:
        f.write("\t".join(["scanB", "bad", "641", "480", "60", "AAAA"]) + "\n")
    with pytest.raises(ValueError):
        staging.read_feature_tsv(str(p))


def test_stale_library_is_refused(vln, monkeypatch):
    """A library whose ABI version differs from the binding's is refused at load time (its entry points keep their names
    while their argument lists change between versions)."""
    lib_mod = vln._lib
    monkeypatch.setattr(lib_mod, "_lib", None)
    monkeypatch.setattr(lib_mod, "EXPECTED_ABI", lib_mod.EXPECTED_ABI + 1)
    with pytest.raises(lib_mod.VlnError, match="ABI version"):
        lib_mod.load()


def test_every_public_struct_has_a_size_checked_mirror():
    """include/vln_hip.h's structs against the ctypes mirrors: every `typedef struct vln_*` is listed in _lib.STRUCT_MIRRORS,
    the library reports its own sizeof for each (vln_struct_size), load() compares them -- and refuses a mirror that is a field
    short (a silent mismatch would make the kernels read wild pointers)."""
    import ctypes as C
    import re
    import vln_amd
    L = vln_amd._lib
    header = open(os.path.join(ROOT, "include", "vln_hip.h")).read()
    declared = set(re.findall(r"typedef\s+struct\s+(vln_[a-z0-9_]+)\s*\{", header))
    assert declared == set(L.STRUCT_MIRRORS), declared ^ set(L.STRUCT_MIRRORS)
    lib = L.load()
    for name, mirror in L.STRUCT_MIRRORS.items():
        assert lib.vln_struct_size(name.encode()) == C.sizeof(mirror), name
    assert lib.vln_struct_size(b"vln_no_such_struct") == -1

    class Short(C.Structure):
        _fields_ = L.MonitorGrads._fields_[:-1]
    good = L.STRUCT_MIRRORS["vln_monitor_grads"]
    try:
        L.STRUCT_MIRRORS["vln_monitor_grads"] = Short
        L._lib = None
        with pytest.raises(L.VlnError, match="vln_monitor_grads"):
            L.load()
    finally:
        L.STRUCT_MIRRORS["vln_monitor_grads"] = good
        L._lib = None
        L.load()


def test_row_block_tiling_of_tall_products_covers_every_row_once(vln):
    """csrc/gemm_rows.h's plan (host arithmetic, `vln_gemm_rows_tiling`): for every tall shape the row tiles -- n_big of rb_big
    16-row blocks, the rest of rb_big - 1 -- cover exactly ceil(M / 16) blocks, no tile is empty or taller than the 8 blocks the
    kernels are instantiated for, the workgroup count is (row tiles) x ceil(N / 64), and the tiling is only taken when it needs
    fewer block-rounds over the CUs than 64-row tiles would.  A wrong tiling would skip or repeat rows silently."""
    import ctypes as C
    lib = vln._lib.load()
    n_big, rb_big, tiles = C.c_int(), C.c_int(), C.c_int()
    taken = 0
    for cus in (256, 304, 64):
        for N in (64, 96, 256, 512, 1000, 1024, 2176):
            nb = (N + 63) // 64
            for M in list(range(256, 1400, 7)) + [1152, 2304, 4096, 5000, 5120, 8064, 65536]:
                ok = lib.vln_gemm_rows_tiling(M, N, cus, C.byref(n_big), C.byref(rb_big), C.byref(tiles))
                if not ok:
                    continue
                taken += 1
                mbk = (M + 15) // 16
                rb, nbig = rb_big.value, n_big.value
                assert 2 <= rb <= 8 and tiles.value % nb == 0, (M, N, cus)
                R = tiles.value // nb
                assert 1 <= nbig <= R, (M, N, cus, nbig, R)
                assert nbig * rb + (R - nbig) * (rb - 1) == mbk, (M, N, cus, nbig, rb, R)
                assert rb > 1 or nbig == R                                      # no tile of zero blocks
                rounds = lambda t: (t + cus - 1) // cus
                assert rounds(tiles.value) * rb < rounds(((M + 63) // 64) * nb) * 4, (M, N, cus)
    assert taken > 100
    # the BN-MLP forward at BASELINE config 2 (B 128, C 8): 256 workgroups of 80 / 64 rows
    assert lib.vln_gemm_rows_tiling(1152, 1024, 256, C.byref(n_big), C.byref(rb_big), C.byref(tiles)) == 1
    assert (n_big.value, rb_big.value, tiles.value) == (8, 5, 256)
    assert lib.vln_gemm_rows_tiling(128, 1024, 256, C.byref(n_big), C.byref(rb_big), C.byref(tiles)) == 0      # skinny: gemm_nt's own tiles
