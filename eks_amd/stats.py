"""Host-side statistics around the multicam smoother (reference eks/stats.py).

compute_pca: one small PCA per keypoint on the low-variance frames; stays on the host (it is
set-up for the observation matrices C, not part of the Kalman path)."""
from __future__ import annotations

import numpy as np
from sklearn.decomposition import PCA

from .marker_array import MarkerArray, mA_to_stacked_array


def compute_pca(valid_frames_mask: np.ndarray, emA_centered_preds: MarkerArray,
                emA_good_centered_preds: MarkerArray, n_components: int = 3,
                pca_object: PCA | None = None):
    """Per keypoint: fit PCA on the variance-filtered frames (stacked views (n_good, 2V)),
    transform all frames, keep the components of the frames in `valid_frames_mask`
    (reference eks/stats.py:9-64).  Returns (list of PCA, list of (n_valid_k, n_components))."""
    M, V, T, K, _ = emA_centered_preds.shape
    assert M == 1, 'MarkerArray should have n_models = 1 after ensembling.'
    models, good_pcs = [], []
    for k in range(K):
        all_frames = mA_to_stacked_array(emA_centered_preds, k)
        model = pca_object if pca_object is not None else \
            PCA(n_components=n_components).fit(mA_to_stacked_array(emA_good_centered_preds, k))
        models.append(model)
        good_pcs.append(model.transform(all_frames)[np.flatnonzero(valid_frames_mask[:, k])])
    return models, good_pcs
