"""Sampler / filter argument objects of the operator surface.

The native module extracts them by attribute (reference: src/python.rs:107-168): an object with
`.with_replacement` is a uniform sampler, one with `.weights` a weighted sampler, and a filter is the
tuple `(obj with .window/.timestamps/.forward/.mode, initial_state)`.  These dataclasses have the
field names of the reference's tch_geometric/utils.py:26-67 and do not need torch_geometric."""
from dataclasses import dataclass
from typing import Dict, Tuple, Union

import torch
from torch import Tensor

MixedData = Union[Tensor, Dict[str, Tensor]]

TEMPORAL_SAMPLE_STATIC: int = 0    # neighbor_sampling.rs:32-34
TEMPORAL_SAMPLE_RELATIVE: int = 1
TEMPORAL_SAMPLE_DYNAMIC: int = 2


def _check(data: MixedData, hetero: bool, dtype: torch.dtype) -> None:
    tensors = list(data.values()) if hetero else [data]
    if hetero and not isinstance(data, dict):
        raise TypeError("heterogeneous data must be a dict of tensors keyed 'src__rel__dst'")
    for t in tensors:
        if t.dtype != dtype:
            raise TypeError("expected %s, got %s" % (dtype, t.dtype))


@dataclass
class UniformEdgeSampler:
    with_replacement: bool = False

    def validate(self, hetero: bool = False) -> None:
        return None


@dataclass
class WeightedEdgeSampler:
    weights: MixedData

    def validate(self, hetero: bool = False) -> None:
        _check(self.weights, hetero, torch.float64)


@dataclass
class TemporalEdgeFilter:
    window: Tuple[int, int]
    timestamps: MixedData
    forward: bool = False
    mode: int = TEMPORAL_SAMPLE_STATIC

    def validate(self, hetero: bool = False) -> None:
        _check(self.timestamps, hetero, torch.int64)
