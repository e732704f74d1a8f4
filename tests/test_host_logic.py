"""CPU: host-side logic added in round 3 -- the patch schedule of the tiled EdgeConv (graph_sched.tile_schedule), the batch buckets
of the eval runtime and the RANSAC stopping rule shared by the PnP oracle and kernel."""
import numpy as np
import torch

from checkerpose_amd.graph_sched import schedule_block, tile_schedule
from checkerpose_amd.model.init import knn
from tests.common import lm_p3d


def test_tile_schedule_invariants_on_lm_graphs():
    """The renumbering is a permutation, every patch's halo list holds exactly its out-of-patch neighbours, the slot lists rebuild
    the ORIGINAL graph (as sets per keypoint: the max over a neighbourhood ignores order), and the internal-numbering table matches."""
    P = lm_p3d(4096)[[1, 4]]                                       # LM objects 2 and 5 (5: the largest halo of the 15)
    idx = knn(P, 20).numpy()
    sc = tile_schedule(idx, P.numpy())
    assert sc is not None and sc["NB"] == 8 and sc["HPAD"] % 64 == 0 and sc["HPAD"] >= int(sc["halo_rows"].max())
    for g in range(2):
        perm, inv = sc["perm"][g].astype(np.int64), sc["inv"][g].astype(np.int64)
        assert sorted(perm.tolist()) == list(range(4096)) and (inv[perm] == np.arange(4096)).all()
        assert (sc["idx_internal"][g] == inv[idx[g][perm]]).all()
        for t in range(8):
            own = np.arange(512) + 512 * t
            nbrs_int = sc["idx_internal"][g][own]                  # (512, 20) internal ids
            outside = np.unique(nbrs_int[(nbrs_int < 512 * t) | (nbrs_int >= 512 * (t + 1))])
            n = int(sc["halo_rows"][g, t])
            assert n == len(outside) and (sc["halo"][g, t, :n] == outside).all()
            assert (sc["halo"][g, t, n:] == 512 * t).all()          # padding: a valid row of the patch
            table = np.concatenate([own, sc["halo"][g, t]])         # slot -> internal row
            rebuilt = perm[table[sc["nbr"][g, t].astype(np.int64)]]            # (512, 20) ORIGINAL keypoint ids
            want = idx[g][perm[own]]
            assert (np.sort(rebuilt, 1) == np.sort(want, 1)).all()
            assert int(sc["nbr"][g, t].max()) < 512 + n
    assert tile_schedule(idx[:, :512], P.numpy()[:, :, :512]) is None       # one patch: the N = 512 kernel's case
    assert tile_schedule(idx, P.numpy(), hpad_max=64) is None               # table would not fit: caller falls back to the L2 gather


def test_schedule_block_is_a_permutation_of_every_list():
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 896, size=(16, 20))
    out, clashes = schedule_block(rows.copy())
    assert (np.sort(out, 1) == np.sort(rows, 1)).all() and 0 <= clashes <= 16 * 20


def test_batch_buckets():
    from checkerpose_amd.model._runtime import batch_bucket
    got = [batch_bucket(b) for b in range(1, 66)]
    assert all(g >= b for g, b in zip(got, range(1, 66)))
    assert all(2 * g < 3 * b for g, b in zip(got, range(1, 66)))              # the bucket is under 1.5x the request
    assert sorted(set(got)) == [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96]
    assert [batch_bucket(b) for b in (100, 128, 129, 200, 256, 257, 1000)] == [128, 128, 192, 256, 256, 384, 1024]


def test_ransac_stopping_rule_known_values():
    """OpenCV's RANSACUpdateNumIters at confidence 0.99: log(0.01) / log(1 - w^m)"""
    from oracle.pnp_oracle import needed_iterations
    assert needed_iterations(70, 100, 5, 150) == 25                 # 70 % inliers, 5-point samples
    assert needed_iterations(100, 100, 5, 150) == 0                 # all inliers: done
    assert needed_iterations(4, 100, 5, 150) == 150                 # fewer inliers than a sample: keep going
    assert needed_iterations(30, 100, 5, 150) == 150                # 0.3^5: would need 1893 > max
    assert needed_iterations(50, 100, 4, 150) == 71


def test_optimizers_refuse_cpu_tensors_and_keep_torch_defaults():
    """checkerpose_amd.optim.Adam / SGD are HIP launches: parameters on the CPU raise (no torch fallback); the constructor takes the
    reference's arguments (train.py:244-246) and the param_groups carry torch's keys so state_dicts travel both ways"""
    import pytest
    from checkerpose_amd import optim as O
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    for opt in (O.Adam([p], lr=1e-3), O.SGD([p], lr=0.1, momentum=0.9)):
        with pytest.raises(RuntimeError, match="one GPU only"):
            opt.step()
        assert torch.equal(p.detach(), torch.zeros(8))                        # untouched
        opt.zero_grad()
        assert p.grad is None
        p.grad = torch.ones(8)
    ga, gt = O.Adam([p], lr=2e-4).param_groups[0], torch.optim.Adam([p], lr=2e-4).param_groups[0]
    assert {k: ga[k] for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize")} == {k: gt[k] for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize")}
    gs, gu = O.SGD([p], lr=0.1, momentum=0.9).param_groups[0], torch.optim.SGD([p], lr=0.1, momentum=0.9).param_groups[0]
    assert {k: gs[k] for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov")} == {k: gu[k] for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov")}
    with pytest.raises(ValueError):
        O.Adam([p], lr=-1.0)


def test_margin_aware_agreement_statistics():
    """checkerpose_amd/agreement.py on synthetic logits: flips are bucketed by the REFERENCE's margin, flips above tau are counted,
    and a free-running id mismatch is explained by (a) the keypoint's own near-tie flip or (b) an earlier flip in its graph
    neighbourhood -- an unexplained one (a confident bit flipped with no upstream cause) makes the contract fail."""
    import torch
    from checkerpose_amd.agreement import logit_agreement, margin_contract_violations, row_stages
    assert row_stages(6, 6) == [0, 0, 0, 0, 1, 2, 3, 0, 0, 0, 1, 2, 3]
    B, N = 1, 32
    g = torch.Generator().manual_seed(0)
    z = torch.randn(B, 13, N, generator=g) * 2.0
    z[z.abs() < 0.5] = 1.0                                   # confident reference decisions everywhere ...
    z[0, 2, 5] = 0.01                                        # ... except one near-tie: stage-0 bit x4 of keypoint 5
    ref = (z[:, 0:1].clone(), z[:, 1:7].clone(), z[:, 7:13].clone(), torch.randn(B, 2, 8, 8, generator=g),
           torch.zeros(B, N, dtype=torch.int64), torch.zeros(B, N, dtype=torch.int64))
    knn = torch.stack([torch.arange(N).roll(-k) for k in range(1, 4)], 1)[None]          # ring graph: neighbours n + 1 .. n + 3
    # (1) the near-tie flips; keypoint 5's later bit and its neighbour 3's stage-2 bit follow (3 -> 4 -> 5 within 3 hops)
    o = z.clone()
    o[0, 2, 5] = -0.01
    o[0, 4, 5] = -z[0, 4, 5]                                 # keypoint 5, stage 1: its own input moved
    o[0, 5, 3] = -z[0, 5, 3]                                 # keypoint 3, stage 2: neighbour of 5 within 2 hops
    xid = ref[4].clone(); xid[0, 5] = 7; xid[0, 3] = 9
    out = (o[:, 0:1], o[:, 1:7], o[:, 7:13], ref[3], xid, ref[5])
    tf = logit_agreement(out, ref, tau=0.05)
    assert tf["flips"] == 3 and tf["flips_above_margin"] == 2 and tf["flip_rate_by_margin"]["0-0.05"]["flips"] == 1
    fr = logit_agreement(out, ref, tau=0.05, explain=True, knn_idx=knn)
    assert fr["id_mismatches"] == 2 and fr["id_mismatches_explained"] == 2 and fr["id_mismatches_from_subtau_self_flip"] == 1
    # (2) a confident stage-0 bit of a far-away keypoint flips with nothing upstream: unexplained
    o2 = z.clone()
    o2[0, 1, 20] = -z[0, 1, 20]
    xid2 = ref[4].clone(); xid2[0, 20] = 3
    fr2 = logit_agreement((o2[:, 0:1], o2[:, 1:7], o2[:, 7:13], ref[3], xid2, ref[5]), ref, tau=0.05, explain=True, knn_idx=knn)
    assert fr2["id_mismatches"] == 1 and fr2["id_mismatches_explained"] == 0
    # (3) ... and such an unexplained flip explains NOTHING downstream: its neighbour's later mismatch stays unexplained too (round 5;
    #     before, any earlier flip within 3 hops "explained" every later mismatch)
    o3 = o2.clone()
    o3[0, 4, 19] = -z[0, 4, 19]                              # keypoint 19 (20 is its neighbour 19 + 1), stage 1
    xid3 = xid2.clone(); xid3[0, 19] = 5
    fr3 = logit_agreement((o3[:, 0:1], o3[:, 1:7], o3[:, 7:13], ref[3], xid3, ref[5]), ref, tau=0.05, explain=True, knn_idx=knn)
    assert fr3["id_mismatches"] == 2 and fr3["id_mismatches_explained"] == 0 and fr3["perturbed_coverage_by_stage"] == [0.0] * 4
    assert fr["perturbed_coverage_by_stage"][0] == 0.0 and 0.0 < fr["perturbed_coverage_by_stage"][1] < 1.0
    # tau never exceeds 10 bf16 epsilons of the logit RMS, however large the run's own error is
    noisy = logit_agreement((o[:, 0:1] + 0.9, o[:, 1:7] - 0.7, o[:, 7:13] + 0.8, ref[3], xid, ref[5]), ref)
    assert noisy["tau"] == noisy["tau_cap"] and abs(noisy["tau_cap"] - 10 * 2.0 ** -8 * noisy["logit_rms"]) < 1e-4
    assert any("bf16 epsilons" in v for v in margin_contract_violations(noisy))
    clean = logit_agreement(ref, ref)
    assert margin_contract_violations(clean, clean) == []
    assert any("explained" in v for v in margin_contract_violations(clean, fr2))
    bad_tf = logit_agreement((o2[:, 0:1], o2[:, 1:7], o2[:, 7:13], ref[3], ref[4], ref[5]), ref)
    assert any("reference margin" in v for v in margin_contract_violations(bad_tf))


def test_bench_prices_a_multi_launch_call_as_the_set(tmp_path, monkeypatch):
    """bench.committed_profile on a "a + b" symbol (one entry point, two launches -- cp_edgeconv_tiled): durations and traffic of the
    parts are summed; a part that the summary lacks voids the figure instead of silently pricing the rest."""
    import bench
    f = tmp_path / "r09_unit_kernel_summary.csv"
    f.write_text("kernel,calls,avg_us,avg_hbm_read_MB(FETCH_SIZE*2),avg_hbm_write_MB\n"
                 "ptable<256>,18,248.7,1155.5,766.7\n"
                 "tiled2<256>,18,458.0,546.7,536.9\n"
                 "other,3,10.0,,\n")
    monkeypatch.setattr(bench, "_summary_files", lambda tag: [str(f)])
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    tr, us, src = bench.committed_profile("ptable<256> + tiled2<256>", "unit")
    assert abs(tr - 3005.8) < 0.1 and abs(us - 706.7) < 0.05 and src.endswith("r09_unit_kernel_summary.csv")
    tr1, us1, _ = bench.committed_profile("tiled2<256>", "unit")
    assert abs(tr1 - 1083.6) < 0.1 and abs(us1 - 458.0) < 0.05
    assert bench.committed_profile("ptable<256> + missing<1>", "unit")[:2] == (None, None)
    assert bench.committed_profile("other", "unit")[:2] == (None, 10.0)


def test_kernel_log_collects_every_symbol_since_begin():
    """cp_kernel_log: "" after begin (no launch on the CPU box); cp_last_kernel is untouched by begin"""
    from checkerpose_amd import _abi
    lib = _abi.load()
    lib.cp_kernel_log_begin()
    assert lib.cp_kernel_log() == b""


def test_struct_epoch_is_scoped_to_the_model():
    """model/_runtime.py: torch's module registration hooks are process-wide, the staleness epoch they move is per model -- a
    registration inside a tracked model's tree bumps THAT model only, unrelated modules cost nothing, and the hooks go away with
    the last tracked model."""
    import gc
    import torch
    from checkerpose_amd.model import _runtime as RT
    from tests.common import build_net
    a, b = build_net(full=False), build_net(full=False, seed=2)
    RT._track_tree(a)
    RT._track_tree(b)
    assert len(RT._HOOKS) == 3
    ea, eb = a._struct_epoch, b._struct_epoch
    torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.BatchNorm1d(3))           # somebody else's modules: nobody's epoch moves
    assert (a._struct_epoch, b._struct_epoch) == (ea, eb)
    a.mlp.bias = torch.nn.Parameter(torch.zeros(7))                                # a child's parameter reassigned
    assert (a._struct_epoch, b._struct_epoch) == (ea + 1, eb)
    b.pre_query_block[0].register_buffer("extra", torch.zeros(1))
    a.mlp = torch.nn.Linear(64, 7)                                                 # a submodule replaced
    assert (a._struct_epoch, b._struct_epoch) == (ea + 2, eb + 1)
    del a, b
    gc.collect()
    assert not RT._HOOKS and not len(RT._TRACKED)


def test_struct_epoch_reaches_every_model_that_shares_a_submodule():
    """a submodule that sits in TWO drop-in models (a shared head, an init net reachable from two PoseNets): a registration inside it
    moves both models' epochs (round 5 kept one root per module: the model that looked last), and both forget it when they die"""
    import gc
    import torch
    from checkerpose_amd.model import _runtime as RT
    from tests.common import build_net
    a, b = build_net(full=False), build_net(full=False, seed=2)
    b.mlp = a.mlp                                                                  # shared
    RT._track_tree(a)
    RT._track_tree(b)
    ea, eb = a._struct_epoch, b._struct_epoch
    a.mlp.bias = torch.nn.Parameter(torch.zeros(7))
    assert (a._struct_epoch, b._struct_epoch) == (ea + 1, eb + 1)
    del a
    gc.collect()
    b.mlp.weight = torch.nn.Parameter(torch.zeros(7, 64))
    assert b._struct_epoch == eb + 2 and len(RT._HOOKS) == 3
    del b
    gc.collect()
    assert not RT._HOOKS and not len(RT._TRACKED)


def test_committed_sq_counters_cover_the_bench_lines_kernels():
    """profiles/summarize.py `check`: every kernel symbol the newest committed default bench line spends its time in has a row in the
    newest committed SQ counter file (round 5 committed counters of a build that was no longer the launched one: its file fails this
    check for `hr_chain0p_kernel` and the chain tails)."""
    import glob
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("summarize", os.path.join(root, "profiles", "summarize.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bench = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_bf16_b256.json")))[-1]
    sq = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r*_sq_counters.csv")))[-1]
    assert mod.sq_missing(bench, sq) == [], (bench, sq, mod.sq_missing(bench, sq))
    bench5 = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_lm13_n4096_bf16_b256.json")))[-1]     # config #5: a composite symbol
    sq5 = sorted(glob.glob(os.path.join(root, "profiles", "r*_sq_counters_lm13_n4096.csv")))[-1]            # "(ptable) + (gather)"
    assert mod.sq_missing(bench5, sq5) == [], (bench5, sq5, mod.sq_missing(bench5, sq5))
    stale = os.path.join(root, "profiles", "r05_sq_counters_PRE_FINAL_BUILD.csv")
    if os.path.exists(stale):
        assert "hr_chain0p_kernel" in mod.sq_missing(bench, stale)
