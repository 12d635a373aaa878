"""Seeded inputs for the loss_helper.py row, shared by the golden generator, the CPU oracle test
and the GPU parity test."""
import numpy as np
import torch

CASES_UNSUP = {
    "a": dict(B=64, K=9, percent=80.0, seed=301, ignored=0),
    "b": dict(B=300, K=16, percent=50.0, seed=302, ignored=17),
    "c": dict(B=1024, K=9, percent=100.0, seed=303, ignored=5),       # percentile 100: only the maximum is dropped
    "d": dict(B=37, K=4, percent=12.5, seed=304, ignored=3),
    "e": dict(B=4096, K=9, percent=20.0, seed=305, ignored=100),      # eight rows per thread of the one-workgroup kernel
    "f": dict(B=9000, K=9, percent=70.0, seed=306, ignored=11),       # beyond it: the rank-counting kernels
    "g": dict(B=512, K=20, percent=35.0, seed=307, ignored=9),        # more classes than the register-row variant holds
}
GRAD_ROWS = 256   # fixtures of the big cases keep the gradient's first rows and its norm

CASES_CONTRA = {
    # N = Nl + Nu rows, D features, K classes, bank capacity per class, rows already in each bank
    "small": dict(Nl=32, Nu=64, D=32, K=6, cap=500, prefill=0, seed=401),
    "k9": dict(Nl=128, Nu=128, D=256, K=9, cap=3000, prefill=200, seed=402),
    "k16_wrap": dict(Nl=96, Nu=160, D=64, K=16, cap=52, prefill=50, seed=403),      # banks overflow: sliding window
    "gap": dict(Nl=40, Nu=40, D=48, K=9, cap=400, prefill=30, seed=404, absent=(0, 3)),   # valid classes != positions
    "one_class": dict(Nl=16, Nu=16, D=16, K=5, cap=100, prefill=10, seed=405, only=2),    # <= 1 valid class: zero loss
    "ema": dict(Nl=64, Nu=64, D=64, K=9, cap=800, prefill=100, seed=406, momentum=True, i_iter=7),
}


def unsup_inputs(cfg):
    rng = np.random.Generator(np.random.PCG64(cfg["seed"]))
    B, K = cfg["B"], cfg["K"]
    predict = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32) * 2)
    teacher = torch.from_numpy((rng.standard_normal((B, K)) * rng.uniform(0.2, 4.0, (B, 1))).astype(np.float32))
    target = torch.from_numpy(rng.integers(0, K, B, dtype=np.int64))
    if cfg["ignored"]:
        target[torch.from_numpy(rng.choice(B, cfg["ignored"], replace=False))] = 255
    return predict, target, teacher


def contra_inputs(cfg):
    rng = np.random.Generator(np.random.PCG64(cfg["seed"]))
    Nl, Nu, D, K = cfg["Nl"], cfg["Nu"], cfg["D"], cfg["K"]
    N = Nl + Nu

    def f32(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32))

    classes = [c for c in range(K) if c not in cfg.get("absent", ())]
    if "only" in cfg:
        classes = [cfg["only"]]
    y_l = rng.choice(classes, Nl)
    y_u = rng.choice(classes, Nu)
    onehot = np.eye(K, dtype=np.float32)

    def probs(y, sharp):
        z = rng.standard_normal((len(y), K)) + sharp * onehot[y]
        e = np.exp(z - z.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)

    inp = dict(
        rep=f32(rng.standard_normal((N, D))), rep_teacher=f32(rng.standard_normal((N, D)) * 0.7 + 0.1),
        label_l=f32(onehot[y_l]), label_u=f32(onehot[y_u]),
        prob_l=f32(probs(y_l, 2.0)), prob_u=f32(probs(y_u, 1.0)),
        low_mask=f32(rng.random((N, 1)) < 0.7), high_mask=f32(rng.random((N, 1)) < 0.6),
        bank=[f32(rng.standard_normal((cfg["prefill"], D))) for _ in range(K)],
        ptrs=[cfg["prefill"] % cfg["cap"]] * K, sizes=[cfg["cap"]] * K)
    if cfg.get("momentum"):
        inp["momentum"] = f32(rng.standard_normal((K, 256, 1, D)) * 0.5)
    return inp


def regenerate_draws(cfg, plan):
    """the indices the reference drew for this case (golden generator: PCG64(seed + 1000), one anchor draw and one
    negative draw per loop position that is not skipped); plan = oracle.contra_draw_plan(...)"""
    rng = np.random.Generator(np.random.PCG64(cfg["seed"] + 1000))
    anchor_idx, neg_idx, highs = {}, {}, []
    for i, pool, rows in plan:
        anchor_idx[i] = torch.from_numpy(rng.integers(0, pool, size=(256,), dtype=np.int64))
        neg_idx[i] = torch.from_numpy(rng.integers(0, rows, size=(256 * 50,), dtype=np.int64))
        highs += [(pool, 256), (rows, 256 * 50)]
    return anchor_idx, neg_idx, np.array(highs, dtype=np.int64).reshape(-1, 2)
