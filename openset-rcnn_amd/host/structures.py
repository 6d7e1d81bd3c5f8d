"""Minimal stand-ins for the detectron2 structures the hot path's public signatures use:
ShapeSpec, Boxes, Instances, ImageList. Written for this repo (detectron2 is not installed on the GPU box);
only the members the reference touches on the path are provided."""
from __future__ import annotations

from collections import namedtuple
from typing import Any, Dict, List, Sequence, Tuple

import torch

ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None, None, None, None))


class Boxes:
    """(N,4) XYXY boxes."""

    def __init__(self, tensor: torch.Tensor):
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self) -> "Boxes":
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs) -> "Boxes":
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self) -> torch.Tensor:
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        b = self.tensor
        self.tensor = torch.stack((b[:, 0].clamp(min=0, max=w), b[:, 1].clamp(min=0, max=h),
                                   b[:, 2].clamp(min=0, max=w), b[:, 3].clamp(min=0, max=h)), dim=-1)

    def nonempty(self, threshold: float = 0.0) -> torch.Tensor:
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, sx: float, sy: float) -> None:
        self.tensor = self.tensor * torch.tensor([sx, sy, sx, sy], dtype=self.tensor.dtype, device=self.tensor.device)

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self) -> int:
        return self.tensor.shape[0]

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list: Sequence["Boxes"]) -> "Boxes":
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"


class Instances:
    """Per-image container of equally long fields (pred_boxes, scores, pred_classes, proposal_boxes, ...)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        if len(self._fields):
            assert len(self) == len(value), f"Adding a field of length {len(value)} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args, **kwargs) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(*args, **kwargs) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields.keys():
            vals = [i.get(k) for i in instance_lists]
            v0 = vals[0]
            if isinstance(v0, torch.Tensor):
                vals = torch.cat(vals, dim=0)
            elif hasattr(type(v0), "cat"):
                vals = type(v0).cat(vals)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, vals)
        return ret

    def __repr__(self):
        return f"Instances(num_instances={len(self) if self._fields else 0}, image_size={self._image_size}, fields={list(self._fields)})"


class ImageList:
    """Batched images padded to one size + the original (h, w) of each."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self) -> int:
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0) -> "ImageList":
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        hm, wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            hm, wm = (hm + d - 1) // d * d, (wm + d - 1) // d * d
        batch = tensors[0].new_full((len(tensors), tensors[0].shape[0], hm, wm), pad_value)
        for i, t in enumerate(tensors):
            batch[i, :, : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(batch, sizes)
