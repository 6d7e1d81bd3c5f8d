"""Input side of the path: what [d2] DatasetMapper / build_detection_{test,train}_loader do for the
reference (train.py:84-104 builds them through DefaultTrainer): read an image file to BGR uint8, ResizeShortestEdge
(MIN_SIZE_TEST 800 / MAX_SIZE_TEST 1333; training: a random choice of MIN_SIZE_TRAIN + horizontal flip), carry the annotations
through the same transform, and hand `list[dict{image (3,H,W) uint8 BGR, height, width, image_id, instances?}]` batches to the
model. Host code, as in the reference (PIL); normalisation, padding and batching happen on the GPU (osr_preprocess).

Sharding: the test loader gives rank r the contiguous slice shard_range(N, r, world) ([d2] InferenceSampler); the train loader
draws an infinite seeded permutation stream and gives rank r every world-th element ([d2] TrainingSampler)."""
from __future__ import annotations

from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

from .parallel import shard_range, world_info
from .structures import Boxes, Instances


def read_image(file_name: str, format: str = "BGR") -> np.ndarray:
    """[d2] detection_utils.read_image: PIL decode, EXIF orientation applied, RGB -> BGR; (H, W, 3) uint8."""
    from PIL import Image, ImageOps
    with Image.open(file_name) as im:
        im = ImageOps.exif_transpose(im).convert("RGB")
        a = np.asarray(im)
    return a[:, :, ::-1].copy() if format == "BGR" else a.copy()


def shortest_edge_size(h: int, w: int, size: int, max_size: int) -> Tuple[int, int]:
    """[d2] ResizeShortestEdge.get_output_shape."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def resize_image(img: np.ndarray, new_hw: Tuple[int, int]) -> np.ndarray:
    """[d2] ResizeTransform.apply_image for uint8: PIL bilinear."""
    from PIL import Image
    if img.shape[:2] == tuple(new_hw):
        return img
    return np.asarray(Image.fromarray(img).resize((new_hw[1], new_hw[0]), Image.BILINEAR))


# ---- ResizeShortestEdge on the device (osr_resize_bilinear_u8): Pillow's resampling tables, computed on the host -----------------
_PIL_PRECISION_BITS = 22  # Pillow Resample.c: 32 - 8 - 2


def pil_resample_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the BILINEAR (triangle, support 1) filter
    and the full box (0, in_size): per output index its first input index and tap count (bounds, (out,2) int32) and the taps'
    22-bit fixed-point weights (coef, (out,ksize) int32). Python floats are C doubles, so every intermediate rounds as in C."""
    import math
    scale = in_size / out_size  # (in1 - in0) / outSize with in0 = 0
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)  # (int) truncates toward zero, like C
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w = 1.0 - t if t < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            coef[xx, x] = int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coef


def pil_resize_emulated(img: np.ndarray, new_hw: Tuple[int, int]) -> np.ndarray:
    """The two passes of Pillow's ImagingResample on the host in numpy integers (the CPU statement of what the device kernels do;
    tests pin it to PIL.Image.resize bit for bit): horizontal pass over the rows the vertical pass reads, 8-bit intermediate,
    vertical pass."""
    h, w = img.shape[:2]
    nh, nw = new_hw
    xb, xc = pil_resample_coeffs(w, nw)
    yb, yc = pil_resample_coeffs(h, nh)
    y_first, y_last = int(yb[0, 0]), int(yb[-1, 0] + yb[-1, 1])
    src = img[y_first:y_last].astype(np.int64)
    half = 1 << (_PIL_PRECISION_BITS - 1)
    tmp = np.empty((y_last - y_first, nw, 3), dtype=np.uint8)
    for xx in range(nw):
        x0, n = int(xb[xx, 0]), int(xb[xx, 1])
        acc = (src[:, x0:x0 + n, :] * xc[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1) + half
        tmp[:, xx, :] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    t64 = tmp.astype(np.int64)
    out = np.empty((nh, nw, 3), dtype=np.uint8)
    for yy in range(nh):
        y0, n = int(yb[yy, 0]) - y_first, int(yb[yy, 1])
        acc = (t64[y0:y0 + n] * yc[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0) + half
        out[yy] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


class DeviceResizer:
    """ResizeShortestEdge of decoded frames ON THE GPU: the (H, W, 3) uint8 frame goes through a pinned staging buffer to the
    device on a copy stream of its own and is resampled there (osr_resize_bilinear_u8: Pillow's algorithm, bit-exact), so the host
    spends no time on the resize and the upload of the next frames overlaps the detector. Returns the (3, nh, nw) uint8 CUDA tensor
    a model input dict carries as "image"; the caller's stream is made to wait for the copy stream's work (an event, no host
    sync). Coefficient tables are cached per (input size, output size) pair."""

    def __init__(self, device, max_pixels: int = 4096 * 4096):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.tables = {}
        self.staging = []  # ring of pinned host buffers: one frame is being copied while the next is filled
        self.turn = 0
        self.max_bytes = max_pixels * 3
        self.done = torch.cuda.Event()  # recorded after every resize; unrecorded until the first call (waiting on it is a no-op)
        self._pid = os.getpid()         # a forked DataLoader worker must not touch the parent's GPU context (use num_workers=0)

    def _axis(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self.tables:
            b, c = pil_resample_coeffs(n_in, n_out)
            self.tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), int(b[0, 0]), int(b[-1, 0] + b[-1, 1]), c.shape[1])
        return self.tables[key]

    def __call__(self, img: np.ndarray, new_hw: Tuple[int, int], out: Optional[torch.Tensor] = None, wait: bool = True) -> torch.Tensor:
        """out: a (3, nh, nw) uint8 CUDA tensor to write into (a slice of a batch buffer). wait=False: the caller's stream is NOT made
        to wait here -- it waits once for `self.done` after a whole batch (one event per batch instead of one per frame)."""
        from . import ops
        if os.getpid() != self._pid:
            raise RuntimeError("DeviceResizer used in a forked worker process: resize on the device runs in the process that owns the GPU "
                               "context (DataLoader num_workers=0 with DatasetMapper(device_resize=...))")
        if not wait and out is None:
            raise ValueError("DeviceResizer(wait=False) needs out=: a result allocated on the copy stream could be reused by the caching "
                             "allocator before the consumer stream has read it")
        h, w = img.shape[:2]
        nh, nw = int(new_hw[0]), int(new_hw[1])
        nbytes = h * w * 3
        if len(self.staging) < 2:
            self.staging.append((torch.empty((max(nbytes, 1 << 22),), dtype=torch.uint8).pin_memory(), torch.cuda.Event()))
        slot = self.turn % len(self.staging)
        self.turn += 1
        buf, ev = self.staging[slot]
        if buf.numel() < nbytes:
            buf = torch.empty((nbytes,), dtype=torch.uint8).pin_memory()
            self.staging[slot] = (buf, ev)
        ev.synchronize()  # the previous copy out of this pinned buffer has finished (no-op for an unrecorded event)
        buf[:nbytes].view(h, w, 3).numpy()[...] = img  # (one host pass: gathers a flipped / strided view into the pinned buffer)
        xb, xc, _, _, kx = self._axis(w, nw)
        yb, yc, y_first, y_last, ky = self._axis(h, nh)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            dsrc = buf[:nbytes].to(self.device, non_blocking=True)
            ev.record(self.stream)
            out = ops.resize_bilinear_u8(dsrc.view(h, w, 3), xb, xc, kx, yb, yc, ky, y_first, y_last - y_first, nh, nw, out=out)
            self.done = self.stream.record_event()
        if wait:
            cur.wait_event(self.done)
        out.record_stream(cur)  # always: the consumer is the caller's stream (with wait=False it waits for self.done once per batch)
        return out


class DatasetMapper:
    """dataset dict -> model input dict ([d2] DatasetMapper with the default augmentations of cfg.INPUT)."""

    def __init__(self, cfg, is_train: bool, seed: int = 0, device_resize: Optional[DeviceResizer] = None):
        """device_resize: a DeviceResizer -- the frame is resized on the GPU and "image" is a CUDA tensor (the host then only
        decodes and flips); None: PIL on the host, as the reference's mapper."""
        self.device_resize = device_resize
        self.is_train = is_train
        self.format = cfg.INPUT.FORMAT
        if is_train:
            self.min_sizes, self.max_size = tuple(cfg.INPUT.MIN_SIZE_TRAIN), cfg.INPUT.MAX_SIZE_TRAIN
            self.flip = cfg.INPUT.RANDOM_FLIP == "horizontal"
        else:
            self.min_sizes, self.max_size, self.flip = (cfg.INPUT.MIN_SIZE_TEST,), cfg.INPUT.MAX_SIZE_TEST, False
        self.rng = np.random.RandomState(seed)

    def __call__(self, d: dict, sample_seed: Optional[int] = None) -> dict:
        """sample_seed: draw this sample's augmentation (flip, training size) from its own seeded stream instead of the mapper's
        running one -- the train loader passes f(seed, position in the sample stream), which makes every sample's augmentation
        independent of how many samples were mapped before it (a resumed run skips positions without decoding images)."""
        rng = self.rng if sample_seed is None else np.random.RandomState(sample_seed % (2 ** 31))
        img = read_image(d["file_name"], self.format)
        h, w = img.shape[:2]
        out = {k: v for k, v in d.items() if k != "annotations"}
        out.setdefault("height", h)
        out.setdefault("width", w)
        do_flip = self.flip and rng.rand() < 0.5
        if do_flip:
            img = img[:, ::-1]
        size = int(self.min_sizes[rng.randint(len(self.min_sizes))]) if self.is_train else int(self.min_sizes[0])
        nh, nw = shortest_edge_size(h, w, size, self.max_size) if size > 0 else (h, w)
        if self.device_resize is not None:
            out["image"] = self.device_resize(img, (nh, nw))  # (3, nh, nw) uint8 on the GPU, bit-identical to the PIL result
        else:
            img = resize_image(np.ascontiguousarray(img), (nh, nw))
            out["image"] = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        if self.is_train and "annotations" in d:
            boxes = np.array([a["bbox"] for a in d["annotations"] if not a.get("iscrowd", 0)], dtype=np.float32).reshape(-1, 4)
            classes = [a["category_id"] for a in d["annotations"] if not a.get("iscrowd", 0)]
            if do_flip:
                boxes = np.stack((w - boxes[:, 2], boxes[:, 1], w - boxes[:, 0], boxes[:, 3]), axis=1) if len(boxes) else boxes
            boxes = boxes * np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
            boxes[:, 0::2] = boxes[:, 0::2].clip(0, nw)
            boxes[:, 1::2] = boxes[:, 1::2].clip(0, nh)
            inst = Instances((nh, nw))
            inst.gt_boxes = Boxes(torch.from_numpy(boxes))
            inst.gt_classes = torch.tensor(classes, dtype=torch.int64)
            keep = inst.gt_boxes.nonempty()
            out["instances"] = inst[keep]
        return out


def build_detection_test_loader(dataset_dicts: Sequence[dict], mapper: Callable[[dict], dict], batch_size: int = 1,
                                rank: Optional[int] = None, world: Optional[int] = None) -> Iterator[List[dict]]:
    """This rank's contiguous shard of the dataset in batches ([d2] build_detection_test_loader + InferenceSampler)."""
    if rank is None or world is None:
        rank, world = world_info()
    lo, hi = shard_range(len(dataset_dicts), rank, world)
    for i in range(lo, hi, batch_size):
        yield [mapper(dataset_dicts[j]) for j in range(i, min(i + batch_size, hi))]


def build_detection_train_loader(dataset_dicts: Sequence[dict], mapper: Callable[[dict], dict], images_per_batch: int, seed: int = 0,
                                 rank: Optional[int] = None, world: Optional[int] = None, filter_empty: bool = True,
                                 start_iter: int = 0) -> Iterator[List[dict]]:
    """Infinite stream of per-rank batches (images_per_batch is the GLOBAL batch, SOLVER.IMS_PER_BATCH): a seeded permutation per
    epoch shared by all ranks, rank r takes elements r, r+world, ... ([d2] TrainingSampler); images without annotations are dropped
    first (DATALOADER.FILTER_EMPTY_ANNOTATIONS). start_iter: the first batch yielded is the one iteration `start_iter` of an
    uninterrupted run would have seen -- the skipped positions only advance the permutation (no image is read), and every
    sample's augmentation is seeded by (seed, its position), so a resumed run continues the exact data stream."""
    if rank is None or world is None:
        rank, world = world_info()
    assert images_per_batch % world == 0, "IMS_PER_BATCH must be divisible by the number of ranks"
    per_rank = images_per_batch // world
    dicts = [d for d in dataset_dicts if not filter_empty or len(d.get("annotations", [])) > 0]
    g = torch.Generator().manual_seed(seed)
    batch: List[dict] = []
    pos, skip = 0, start_iter * images_per_batch  # positions of the global sample stream consumed by iterations < start_iter
    seeded = isinstance(mapper, DatasetMapper)
    while True:
        perm = torch.randperm(len(dicts), generator=g).tolist()
        if skip - pos >= len(perm):  # a whole epoch lies before the resume point
            pos += len(perm)
            continue
        for idx in perm:
            if pos >= skip and pos % world == rank:
                batch.append(mapper(dicts[idx], sample_seed=seed * 1000003 + pos) if seeded else mapper(dicts[idx]))
                if len(batch) == per_rank:
                    yield batch
                    batch = []
            pos += 1
