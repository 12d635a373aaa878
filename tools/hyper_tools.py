"""Caller-side helpers of the reference's tools/hyper_tools.py that the training driver uses around
the hot path: whole-image inference (hyper_tools.py:416-437) and OA / Kappa / per-class accuracy
(hyper_tools.py:208-223).  Host-side glue only; the forward runs on the HIP kernels."""
import numpy as np
import torch


@torch.no_grad()
def test_whole(model, data_loader, print_per_batches=10):
    """argmax prediction for every pixel of the scene (reference hyper_tools.py:416-437).  ``data_loader`` yields
    (XP, X) batches of materialised patches, as in the reference -- or is a ``cmlpl_amd.infer.CubeSource`` (the scene
    cube [rows, cols, C] and the spectra, resident in HBM): the windows are then gathered on the device inside the
    forward kernel (cmlpl_infer_cube), no patch tensor and no DataLoader.  (The reference forgets no_grad here; the result
    is the same.)"""
    model.eval()
    from cmlpl_amd.infer import CubeSource, infer_cube
    if isinstance(data_loader, CubeSource):
        return infer_cube(model, data_loader.cube, data_loader.spectra).cpu().numpy()
    out = []
    for batch_idx, (XP, X) in enumerate(data_loader):
        logits, _ = model(XP.cuda(non_blocking=True), X.cuda(non_blocking=True))
        out.append(logits.argmax(1).cpu().numpy())
        if (batch_idx + 1) % print_per_batches == 0:
            print('---------------------Testing the whole set-[%d/%d]---------------------'
                  % (batch_idx + 1, len(data_loader)))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def CalAccuracy(predict, label):
    """OA, Kappa, producer's accuracy per class (confusion-matrix statistics)."""
    predict = np.asarray(predict).astype(np.int64)
    label = np.asarray(label).astype(np.int64)
    n = int(label.max()) + 1
    cm = np.zeros((n, n), dtype=np.float64)
    np.add.at(cm, (label, np.clip(predict, 0, n - 1)), 1.0)
    total = cm.sum()
    OA = np.trace(cm) / total
    pe = float((cm.sum(0) * cm.sum(1)).sum()) / (total * total)
    Kappa = (OA - pe) / (1.0 - pe) if pe < 1.0 else 0.0
    producerA = np.diag(cm) / np.maximum(cm.sum(1), 1.0)
    return OA, Kappa, producerA
