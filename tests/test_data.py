"""Input pipeline (host/data.py): resize rule, mapper, sharded loaders -- on synthetic PNG files."""
import pytest
import numpy as np
import torch


def _write_images(tmp_path, sizes):
    from PIL import Image
    rng = np.random.RandomState(0)
    files = []
    for i, (h, w) in enumerate(sizes):
        f = tmp_path / f"im{i}.png"
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(f)
        files.append(str(f))
    return files


def _cfg(osr):
    from openset_rcnn_amd.host import config as Cfg
    cfg = Cfg.get_cfg()
    Cfg.add_openset_rcnn_config(cfg)
    return cfg


def test_shortest_edge_rule(osr):
    from openset_rcnn_amd.host.data import shortest_edge_size
    assert shortest_edge_size(480, 640, 800, 1333) == (800, 1067)
    assert shortest_edge_size(375, 500, 800, 1333) == (800, 1067)
    assert shortest_edge_size(720, 1280, 800, 1333) == (750, 1333)   # GraspNet frames hit the max-size cap
    assert shortest_edge_size(1000, 300, 800, 1333) == (1333, 400)


def test_mapper_and_loaders(osr, tmp_path):
    from openset_rcnn_amd.host import data as D
    cfg = _cfg(osr)
    files = _write_images(tmp_path, [(60, 80), (90, 50), (64, 64), (40, 100), (70, 70)])
    dicts = [dict(file_name=f, image_id=str(i), annotations=[dict(bbox=[5.0, 6.0, 30.0, 40.0], category_id=i % 20)] if i != 2 else [])
             for i, f in enumerate(files)]
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 120, 180
    m = D.DatasetMapper(cfg, is_train=False)
    x = m(dicts[0])
    assert x["image"].dtype == torch.uint8 and tuple(x["image"].shape) == (3, 120, 160) and (x["height"], x["width"]) == (60, 80)
    # BGR: channel 0 of the mapped image is the blue channel of the file
    from PIL import Image
    rgb = np.asarray(Image.open(files[2]))
    cfg.INPUT.MIN_SIZE_TEST = 64
    x2 = D.DatasetMapper(cfg, is_train=False)(dicts[2])
    assert np.array_equal(x2["image"].numpy()[0], rgb[:, :, 2])
    # test loader: 2 ranks cover every image exactly once, in order
    got = []
    for r in range(2):
        for batch in D.build_detection_test_loader(dicts, m, batch_size=2, rank=r, world=2):
            got += [b["image_id"] for b in batch]
    assert got == ["0", "1", "2", "3", "4"]
    # train mapper: flip + resize carry the boxes along; the image without annotations is filtered
    cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = (120,), 400
    tm = D.DatasetMapper(cfg, is_train=True, seed=1)
    seen_flip = seen_plain = False
    for _ in range(12):
        y = tm(dicts[0])
        b = y["instances"].gt_boxes.tensor[0]
        assert tuple(y["image"].shape) == (3, 120, 160)
        if abs(float(b[0]) - 10.0) < 1e-4:      # x1 = 5 * 2
            seen_plain = True
            assert torch.allclose(b, torch.tensor([10.0, 12.0, 60.0, 80.0]))
        else:                                    # flipped: x1 = (80 - 30) * 2
            seen_flip = True
            assert torch.allclose(b, torch.tensor([100.0, 12.0, 150.0, 80.0]))
    assert seen_flip and seen_plain
    it0 = D.build_detection_train_loader(dicts, tm, images_per_batch=4, seed=3, rank=0, world=2)
    it1 = D.build_detection_train_loader(dicts, tm, images_per_batch=4, seed=3, rank=1, world=2)
    ids = []
    for _ in range(2):
        b0, b1 = next(it0), next(it1)
        assert len(b0) == len(b1) == 2
        ids += [x["image_id"] for x in b0 + b1]
    assert "2" not in ids and len(set(ids[:4])) == 4  # one epoch of the 4 annotated images is a permutation split over the ranks


def test_pillow_resampling_restatement_matches_pil_bit_for_bit(osr):
    """host/data.py restates Pillow's BILINEAR resampling (Resample.c: precompute_coeffs, normalize_coeffs_8bpc, horizontal then
    vertical pass with an 8-bit intermediate) for the device resize. Pinned here against PIL itself, which IS what the reference's
    loader calls ([d2] ResizeTransform.apply_image): down-scaling (support > 1), up-scaling, one axis unchanged, odd sizes."""
    import numpy as np
    from PIL import Image
    from openset_rcnn_amd.host.data import pil_resample_coeffs, pil_resize_emulated, shortest_edge_size
    rng = np.random.RandomState(0)
    cases = [((37, 53), (80, 115)), ((120, 90), (48, 36)), ((64, 64), (64, 100)), ((51, 77), (51, 40)), ((600, 1000), (800, 1333)),
             ((1080, 1920), shortest_edge_size(1080, 1920, 800, 1333)), ((9, 7), (3, 2)), ((5, 5), (5, 5))]
    for (h, w), (nh, nw) in cases:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        got = pil_resize_emulated(img, (nh, nw))
        assert np.array_equal(got, want), ((h, w), (nh, nw), int(np.abs(got.astype(int) - want.astype(int)).max()))
    b, c = pil_resample_coeffs(10, 10)  # scale 1: the identity (Pillow skips such a pass; the tables reproduce it exactly)
    assert b[:, 0].tolist() == list(range(10)) and (c[:, 0] == 1 << 22).all() and (c[:, 1:] == 0).all()
    b, c = pil_resample_coeffs(100, 25)  # 4x down: triangle of half-width 4 -> up to 9 taps, weights sum to 1 in 22-bit fixed point
    assert c.shape[1] == 9 and int(b[:, 1].max()) <= 9 and all(abs(int(r.sum()) - (1 << 22)) <= 4 for r in c)


@pytest.mark.gpu
def test_device_resize_is_the_pil_image_bit_for_bit(osr, tmp_path):
    """osr_resize_bilinear_u8 (second half): the frame resized ON THE GPU equals PIL's Image.resize(BILINEAR) -- what
    [d2] ResizeShortestEdge does on the host -- in every byte, and a DatasetMapper with a DeviceResizer hands the model the same
    "image" (as a CUDA tensor) as the host mapper."""
    import numpy as np
    from PIL import Image
    from openset_rcnn_amd.host.data import DatasetMapper, DeviceResizer, shortest_edge_size
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    osr._lib.load()
    rz = DeviceResizer("cuda:0")
    rng = np.random.RandomState(1)
    for (h, w), (nh, nw) in [((375, 500), shortest_edge_size(375, 500, 800, 1333)), ((720, 1280), shortest_edge_size(720, 1280, 800, 1333)),
                             ((600, 1000), (800, 1333)), ((97, 61), (40, 25)), ((33, 47), (33, 90)), ((5, 5), (5, 5))]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)).transpose(2, 0, 1)
        got = rz(img, (nh, nw))
        torch.cuda.synchronize()
        assert tuple(got.shape) == (3, nh, nw) and got.dtype == torch.uint8 and got.is_cuda
        assert np.array_equal(got.cpu().numpy(), want), ((h, w), (nh, nw))
    # flipped, non-contiguous source (the training augmentation hands the resizer a reversed view)
    img = rng.randint(0, 256, (120, 200, 3)).astype(np.uint8)
    want = np.asarray(Image.fromarray(np.ascontiguousarray(img[:, ::-1])).resize((300, 180), Image.BILINEAR)).transpose(2, 0, 1)
    assert np.array_equal(rz(img[:, ::-1], (180, 300)).cpu().numpy(), want)
    # through the mapper
    from openset_rcnn_amd.host.config import add_openset_rcnn_config, get_cfg
    cfg = get_cfg()
    add_openset_rcnn_config(cfg)
    f = tmp_path / "a.png"
    Image.fromarray(rng.randint(0, 256, (240, 320, 3)).astype(np.uint8)).save(f)
    d = {"file_name": str(f), "image_id": 1}
    host, dev = DatasetMapper(cfg, False)(d), DatasetMapper(cfg, False, device_resize=rz)(d)
    torch.cuda.synchronize()
    assert dev["image"].is_cuda and torch.equal(dev["image"].cpu(), host["image"]) and dev["height"] == host["height"] == 240
