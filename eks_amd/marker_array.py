"""MarkerArray - the reference's boundary container, re-implemented (eks/marker_array.py:15-355).

A thin named-axis view over a 5-D array with the fixed axis order
    (n_models, n_cameras, n_frames, n_keypoints, n_fields)
and a list of field names for the last axis.  Same public surface as upstream (`array`,
`data_fields`, `shape`, `n_*`, `get_array`, `slice`, `slice_fields`, `stack`, `stack_fields`,
`reorder_data_fields`) so the reference's drivers, CLI and tests can use it unchanged; objects of
the reference's own class are accepted anywhere by duck typing (`.array`, `.data_fields`).
"""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np
import pandas as pd

__all__ = ['MarkerArray', 'input_dfs_to_markerArray', 'mA_to_stacked_array', 'stacked_array_to_mA']

_AXES = ('models', 'cameras', 'frames', 'keypoints', 'fields')


def _is_arraylike(a) -> bool:
    return hasattr(a, 'shape') and hasattr(a, 'ndim') and hasattr(a, 'dtype')


class MarkerArray:
    axis_map = {name: i for i, name in enumerate(_AXES)}

    def __init__(self, array=None, shape=None, data_fields=None, marker_array=None, dtype=np.float32):
        if marker_array is not None:
            assert hasattr(marker_array, 'array') and hasattr(marker_array, 'data_fields'), \
                'marker_array must be a MarkerArray.'
            src = marker_array.array if array is None else array
            self.array = np.array(src, dtype=dtype)
            self.data_fields = list(marker_array.data_fields) if data_fields is None else data_fields
        elif array is not None:
            assert _is_arraylike(array), 'Input must be a NumPy or JAX array.'
            assert array.ndim == 5, \
                'Expected shape (n_models, n_cameras, n_frames, n_keypoints, n_fields).'
            self.array = array
            self.data_fields = data_fields
        elif shape is not None:
            assert len(shape) == 5, \
                'Shape must be (n_models, n_cameras, n_frames, n_keypoints, n_fields).'
            self.array = np.zeros(shape, dtype=dtype)
            self.data_fields = data_fields
        else:
            raise AssertionError('Provide either `array`, `shape`, or `marker_array`.')
        (self.n_models, self.n_cameras, self.n_frames, self.n_keypoints,
         self.n_fields) = self.array.shape

    # ------------------------------------------------------------------ basic accessors
    @property
    def shape(self):
        return tuple(self.array.shape)

    def get_array(self, squeeze: bool = False):
        return np.squeeze(self.array) if squeeze else self.array

    def __repr__(self) -> str:
        dims = ', '.join(f'{n}={s}' for n, s in zip(_AXES, self.array.shape))
        kind = 'NumPy' if isinstance(self.array, np.ndarray) else type(self.array).__module__
        return f'MarkerArray({dims}, data_fields={self.data_fields}, type={kind})'

    # ------------------------------------------------------------------ slicing (copies)
    def slice(self, axis: str, indices) -> 'MarkerArray':
        assert axis in self.axis_map, \
            f'Invalid slice axis: {axis}. Must be one of {list(self.axis_map.keys())}.'
        if isinstance(indices, (int, np.integer)):
            indices = [int(indices)]
        return MarkerArray(np.take(np.asarray(self.array), indices, axis=self.axis_map[axis]),
                           data_fields=self.data_fields)

    def _field_indices(self, fields: Iterable[str]) -> list[int]:
        idx = []
        for f in fields:
            assert f in self.data_fields, f"Field '{f}' not found in data_fields: {self.data_fields}"
            idx.append(self.data_fields.index(f))
        return idx

    def slice_fields(self, *fields: str) -> 'MarkerArray':
        return MarkerArray(np.take(np.asarray(self.array), self._field_indices(fields), axis=4),
                           data_fields=list(fields))

    def reorder_data_fields(self, new_order: Sequence[str]) -> 'MarkerArray':
        assert set(new_order) == set(self.data_fields), \
            f'Mismatch in data fields: Expected {self.data_fields}, but got {new_order}'
        arr = np.take(np.asarray(self.array), self._field_indices(new_order), axis=4)
        return MarkerArray(marker_array=self, data_fields=list(new_order), array=arr)

    # ------------------------------------------------------------------ stacking
    @staticmethod
    def stack(others: Sequence['MarkerArray'], axis: str) -> 'MarkerArray':
        assert len(others) > 0, 'At least one MarkerArray must be provided for stacking.'
        first = others[0]
        assert axis in first.axis_map, \
            f'Invalid stack axis: {axis}. Must be one of {list(first.axis_map.keys())}.'
        ax = first.axis_map[axis]
        ref = first.array.shape[:ax] + first.array.shape[ax + 1:]
        for o in others[1:]:
            assert hasattr(o, 'array'), "All elements in 'others' must be MarkerArray instances."
            assert o.array.shape[:ax] + o.array.shape[ax + 1:] == ref, \
                f"Shape mismatch: Cannot stack along '{axis}' due to differing dimensions."
        return MarkerArray(np.concatenate([np.asarray(o.array) for o in others], axis=ax),
                           data_fields=first.data_fields)

    def stack_fields(*marker_arrays: 'MarkerArray') -> 'MarkerArray':  # noqa: N805 (upstream API)
        assert len(marker_arrays) > 0, 'At least one MarkerArray must be provided for stacking.'
        first = marker_arrays[0]
        names: list[str] = []
        for o in marker_arrays:
            assert hasattr(o, 'array'), 'All inputs must be MarkerArray instances.'
            assert o.array.shape[:4] == first.array.shape[:4], \
                "Shape mismatch: Cannot stack along 'fields' due to differing dimensions."
            assert o.data_fields is not None, 'All MarkerArrays must have data_fields defined.'
            names.extend(o.data_fields)
        return MarkerArray(np.concatenate([np.asarray(o.array) for o in marker_arrays], axis=4),
                           data_fields=names)


def input_dfs_to_markerArray(input_dfs_list, bodypart_list, camera_names,
                             data_fields=('x', 'y', 'likelihood')) -> MarkerArray:
    """(cameras x models) nested list of flat `{keypoint}_{field}` DataFrames -> MarkerArray
    float64 (M, V, T, K, F).  Reference: eks/marker_array.py:269-299."""
    data_fields = list(data_fields)
    V, M = len(camera_names), len(input_dfs_list[0])
    T = input_dfs_list[0][0].shape[0]
    cols = [f'{kp}_{f}' for kp in bodypart_list for f in data_fields]
    out = np.zeros((M, V, T, len(bodypart_list), len(data_fields)))
    for c in range(V):
        for m in range(M):
            df: pd.DataFrame = input_dfs_list[c][m]
            out[m, c] = df[cols].to_numpy().reshape(T, len(bodypart_list), len(data_fields))
    return MarkerArray(out, data_fields=data_fields)


def mA_to_stacked_array(marker_array: MarkerArray, keypoint_idx: int) -> np.ndarray:
    """(1, V, T, K, F) -> (T, V*F) for one keypoint, ordered [cam0 f0, cam0 f1, cam1 f0, ...]
    (reference eks/marker_array.py:302-324)."""
    _, V, T, K, F = marker_array.shape
    assert 0 <= keypoint_idx < K, f'keypoint_idx {keypoint_idx} is out of range (0-{K - 1})'
    sel = np.asarray(marker_array.array)[0, :, :, keypoint_idx, :]           # (V, T, F)
    return np.transpose(sel, (1, 0, 2)).reshape(T, V * F)


def stacked_array_to_mA(reshaped_x: np.ndarray, n_cameras: int, data_fields) -> MarkerArray:
    """Inverse of `mA_to_stacked_array` for one keypoint: (T, V*F) -> (1, V, T, 1, F)."""
    T, total = reshaped_x.shape
    assert total % n_cameras == 0, \
        'Input shape mismatch: total fields must be divisible by n_cameras.'
    F = total // n_cameras
    arr = np.transpose(reshaped_x.reshape(T, n_cameras, F), (1, 0, 2))[None, :, :, None, :]
    return MarkerArray(arr, data_fields=data_fields)
