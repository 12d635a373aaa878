"""GPU: batches BY INDEX (cmlpl_batch.d_lab_idx / d_unl_idx -- the kernels read the rows of the resident split where they
lie, hsi_loader.py:109-133 without the gathered batch tensor) and the step replayed from a captured hipGraph
(cmlpl_step_graph_*, per-step scalars in a device table): both must be BIT-identical to the plain eager step on the
gathered rows, for every shape (fused per-sample kernels and the general path alike)."""
import pytest
import torch

from cmlpl_amd import HyperParams, NetShape, TrainEngine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = {"B2": (103, 11, 11, 103, 9), "B5": (48, 15, 15, 48, 20), "P": (60, 20, 20, 103, 9)}


def _data(shape, n_lab, n_unl, seed):
    C, H, W, bands, K = shape
    g = torch.Generator().manual_seed(seed)
    XP = torch.randn(n_lab, C, H, W, generator=g)
    X = torch.randn(n_lab, bands, generator=g)
    Y = torch.randint(0, K, (n_lab,), generator=g)
    XPu = torch.randn(n_unl, C, H, W, generator=g)
    Xu = torch.randn(n_unl, bands, generator=g)
    return [t.to(DEV) for t in (XP, X, Y, XPu, Xu)]


def _engine(shape, bt, btu, hist_rows=4):
    eng = TrainEngine(NetShape(*shape), bt, btu, HyperParams(), device=DEV, seed=1088, hist_rows=hist_rows)
    eng.init_params_default(1088)
    return eng


def _state(eng):
    return [eng.params.clone(), eng.m.clone(), eng.v.clone(), eng.bank_feats.clone(), eng.bank_probs.clone(),
            eng.scalar_hist.clone(), eng.grads.clone()]


def _same(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"{what}: state tensor {i} differs, max |d| = {(x - y).abs().max().item():.3e}"


@pytest.mark.parametrize("name,bt,btu", [("B2", 128, 128), ("B2", 24, 40), ("B5", 8, 64), ("P", 16, 16)])
def test_indexed_step_is_bit_identical_to_the_gathered_step(name, bt, btu):
    shape = SHAPES[name]
    XP, X, Y, XPu, Xu = _data(shape, 300, 500, 7)
    g = torch.Generator().manual_seed(3)
    ea, eb = _engine(shape, bt, btu), _engine(shape, bt, btu)
    for s in range(3):
        li = torch.randperm(300, generator=g)[:bt].to(DEV)
        ui = torch.randperm(500, generator=g)[:btu].to(DEV)
        ea.step(XP[li].contiguous(), X[li].contiguous(), Y[li].contiguous(), XPu[ui].contiguous(), Xu[ui].contiguous(),
                1, s)
        eb.step(XP, X, Y, XPu, Xu, 1, s, lab_idx=li, unl_idx=ui)
        _same(_state(ea), _state(eb), f"{name} step {s}")
        la, fa = ea.outputs()
        lb, fb = eb.outputs()
        assert torch.equal(la, lb) and torch.equal(fa, fb)
    assert torch.isfinite(ea.scalar_hist).all()


@pytest.mark.parametrize("name,bt,btu", [("B2", 128, 128), ("B5", 8, 64), ("P", 16, 16)])
def test_graph_replay_is_bit_identical_to_eager_steps(name, bt, btu):
    """8 replays of the captured step against 8 eager steps: parameters, Adam moments, banks, logged rows, pointers.
    The schedule crosses the smoothing gate (train.py:212, batch 17 -> 18 of epoch 0) and an epoch boundary (the
    adaptive threshold changes, train.py:147-148), so every table field is exercised."""
    shape = SHAPES[name]
    XP, X, Y, XPu, Xu = _data(shape, 4 * bt + 5, 4 * btu + 3, 11)
    g = torch.Generator().manual_seed(5)
    lab_perm = torch.randperm(XP.shape[0], generator=g).to(DEV)
    unl_perm = torch.randperm(XPu.shape[0], generator=g).to(DEV)
    sched = [(0, 15), (0, 16), (0, 17), (0, 18), (0, 19), (1, 0), (1, 1), (2, 0), (2, 1)]   # (epoch, batch_index)
    offs = [(k % 4) * bt for k in range(len(sched))], [(k % 4) * btu for k in range(len(sched))]
    ea, eb = _engine(shape, bt, btu, 16), _engine(shape, bt, btu, 16)
    # eager reference: every step by index
    for k, (ep, bi) in enumerate(sched):
        ea.step(XP, X, Y, XPu, Xu, ep, bi, lab_idx=lab_perm[offs[0][k]:offs[0][k] + bt],
                unl_idx=unl_perm[offs[1][k]:offs[1][k] + btu])
    # replayed: first step eager (warm-up), the rest from the graph
    ep, bi = sched[0]
    eb.step(XP, X, Y, XPu, Xu, ep, bi, lab_idx=lab_perm[:bt], unl_idx=unl_perm[:btu])
    graph = eb.capture(XP, X, Y, XPu, Xu, lab_perm, unl_perm, bt, btu, capacity=16)
    graph.program([(e, b, offs[0][k], offs[1][k]) for k, (e, b) in enumerate(sched)][1:5])
    for _ in range(4):
        graph.launch()
    graph.program([(e, b, offs[0][k], offs[1][k]) for k, (e, b) in enumerate(sched)][5:])
    for _ in range(4):
        graph.launch()
    torch.cuda.synchronize()
    assert ea.ptr == eb.ptr and ea.adam_t == eb.adam_t and ea.step_count == eb.step_count
    _same(_state(ea), _state(eb), f"{name} after {len(sched)} steps")
    assert torch.isfinite(eb.scalar_hist[:len(sched)]).all()
    with pytest.raises(RuntimeError):
        graph.launch()                       # nothing programmed
    graph.close()


def test_graph_needs_a_warm_engine_and_index_buffers():
    shape = SHAPES["B2"]
    XP, X, Y, XPu, Xu = _data(shape, 40, 40, 1)
    eng = _engine(shape, 16, 16)
    idx = torch.arange(40, device=DEV)
    with pytest.raises(RuntimeError):
        eng.capture(XP, X, Y, XPu, Xu, idx, idx, 16, 16)
    eng.step(XP, X, Y, XPu, Xu, 0, 0, lab_idx=idx[:16], unl_idx=idx[:16])
    with pytest.raises(ValueError):
        eng.capture(XP, X, Y, XPu, Xu, None, None, 16, 16)
    with pytest.raises(ValueError):
        eng.step(XP, X, Y, XPu, Xu, 0, 1, lab_idx=idx[:16])          # one index list without the other


def test_index_lists_are_checked_and_pending_replays_block_eager_steps():
    """(1) the fast path of the row check (same resident splits as last call) refuses a strided index view
    like the slow path does; (2) index values are range-checked where lists are filled (capture, validate_indices) and
    on request (check_index_range); (3) an eager step while programmed replays are pending raises -- their table rows were
    formed from the state before it; (4) launch() refreshes the packed weights after load_state_dict."""
    shape = SHAPES["B2"]
    XP, X, Y, XPu, Xu = _data(shape, 64, 64, 2)
    eng = _engine(shape, 16, 16)
    idx = torch.arange(64, device=DEV)
    eng.step(XP, X, Y, XPu, Xu, 0, 0, lab_idx=idx[:16], unl_idx=idx[:16])
    with pytest.raises(ValueError):
        eng.step(XP, X, Y, XPu, Xu, 0, 1, lab_idx=idx[::2][:16], unl_idx=idx[:16])      # strided view, fast path
    with pytest.raises(ValueError):
        TrainEngine.check_index_range(torch.tensor([0, 64], device=DEV), 64, "lab_idx")
    with pytest.raises(ValueError):
        TrainEngine.check_index_range(torch.tensor([-1, 3], device=DEV), 64, "lab_idx")
    bad = idx.clone(); bad[5] = 999
    with pytest.raises(ValueError):
        eng.capture(XP, X, Y, XPu, Xu, bad, idx, 16, 16)
    graph = eng.capture(XP, X, Y, XPu, Xu, idx, idx, 16, 16, capacity=4)
    graph.program([(0, 1, 0, 0), (0, 2, 16, 16)])
    with pytest.raises(RuntimeError):
        eng.step(XP, X, Y, XPu, Xu, 0, 1, lab_idx=idx[:16], unl_idx=idx[:16])             # replays pending
    graph.launch()
    eng.load_state_dict(0, {k: v * 1.0 for k, v in eng.state_dict(0).items()})               # marks the packed copies stale
    assert eng._packed_dirty
    graph.launch()
    assert not eng._packed_dirty                                                             # launch() re-packed them
    eng.step(XP, X, Y, XPu, Xu, 0, 3, lab_idx=idx[:16], unl_idx=idx[:16])                 # nothing pending: fine
    torch.cuda.synchronize()
    assert torch.isfinite(eng.scalar_hist).all()
    graph.close()
