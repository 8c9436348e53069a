"""Pairwise-verification ROC of a set of embeddings (reference roc_cuda.py): the pair histogram on the GPU
(``fedfr_roc_histogram``: fp64 MFMA + LDS-private histogram instead of the reference's one-thread-per-pair numba kernel with
fp64 global atomics), the TPR-at-FPR read-out on the host exactly as ``plot_ROC`` (roc_cuda.py:61-78)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _C


@torch.no_grad()
def roc_histogram(features: torch.Tensor, labels: torch.Tensor, target_size: int) -> torch.Tensor:
    """int64 [2001, 2]: for every pair a < b with a < target_size, bin int((<f_a, f_b> + 1) * 1000); column 0 counts pairs with
    equal labels, column 1 the others (roc_cuda.py:14-30).  ``features`` [N, D] fp32 on the GPU, target rows first."""
    features = _C.require_gpu_tensor(features.contiguous(), torch.float32, "features")
    labels = _C.require_gpu_tensor(labels.to(torch.int64).contiguous(), torch.int64, "labels")
    n, d = features.shape
    if labels.shape[0] != n or not 0 < target_size <= n:
        raise RuntimeError("roc_histogram: labels must match features and 0 < target_size <= N")
    hist = torch.zeros(2001 * 2, dtype=torch.int64, device=features.device)
    _C.call("fedfr_roc_histogram", features.data_ptr(), labels.data_ptr(), n, d, int(target_size), hist.data_ptr(), _C.stream())
    return hist.view(2001, 2)


def order_targets(features: torch.Tensor, labels: torch.Tensor, target_label: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """rows of the target identities first (roc_cuda.py:129-136)."""
    t = torch.zeros_like(labels, dtype=torch.bool)
    for l in target_label:
        t |= labels == l
    return torch.cat([features[t], features[~t]], dim=0), torch.cat([labels[t], labels[~t]]), int(t.sum())


def tpr_at_fpr(hist) -> List[float]:
    """TPR (%) at FPR = 1e-1 ... 1e-6, as plot_ROC prints it (roc_cuda.py:61-78)."""
    from scipy.interpolate import interp1d
    data = np.cumsum(np.asarray(hist.cpu() if torch.is_tensor(hist) else hist, dtype=np.int64), axis=0)
    tpr, fpr = [1.0], [1.0]
    for i in range(data.shape[0]):
        tpr.append((data[-1, 0] - data[i, 0]) / data[-1, 0])
        fpr.append((data[-1, 1] - data[i, 1]) / data[-1, 1])
    tpr, fpr = np.array(tpr), np.array(fpr)
    idx = np.argsort(fpr)
    roc = interp1d(fpr[idx], tpr[idx])
    return [float("%.2f" % (100 * roc(10 ** i))) for i in range(-1, -7, -1)]
