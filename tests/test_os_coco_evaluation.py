"""Open-set COCO-style evaluator (host/os_coco_evaluation.py): hand-computed case.
Reference protocol: openset_rcnn/evaluation/os_cocoeval.py + os_coco_evaluation.py (no fixtures exist there; this is our own KAT)."""
import pytest
import torch


def _dataset():
    cats = [dict(id=1, name="a"), dict(id=2, name="b"), dict(id=3, name="c")]
    images = [dict(id=1, height=300, width=300), dict(id=2, height=300, width=300)]
    anns = [dict(id=1, image_id=1, category_id=1, bbox=[10, 10, 50, 50], area=2500, iscrowd=0),
            dict(id=2, image_id=1, category_id=3, bbox=[100, 100, 40, 40], area=1600, iscrowd=0),   # not a known name -> unknown
            dict(id=3, image_id=2, category_id=2, bbox=[20, 20, 100, 100], area=10000, iscrowd=0),
            dict(id=4, image_id=2, category_id=1, bbox=[200, 50, 20, 20], area=400, iscrowd=0)]
    return dict(images=images, annotations=anns, categories=cats)


def _dets():
    return [dict(image_id=1, category_id=1, bbox=[10, 10, 50, 50], score=0.9),         # a: true positive
            dict(image_id=1, category_id=1, bbox=[100, 100, 40, 40], score=0.8),       # a on the unknown object: open-set error
            dict(image_id=1, category_id=1000, bbox=[100, 100, 40, 40], score=0.7),    # unknown: true positive
            dict(image_id=2, category_id=2, bbox=[20, 20, 100, 100], score=0.95),      # b: true positive
            dict(image_id=2, category_id=1, bbox=[20, 20, 100, 100], score=0.6),       # a on the b object: other-known confusion
            dict(image_id=2, category_id=1000, bbox=[200, 50, 20, 20], score=0.5)]     # unknown on a known object


def test_box_iou_and_crowd(osr):
    from openset_rcnn_amd.host.os_coco_evaluation import box_iou_xywh
    d = [[0, 0, 10, 10], [5, 5, 10, 10]]
    g = [[0, 0, 10, 10], [0, 0, 20, 20]]
    iou = box_iou_xywh(d, g, [0, 1])
    assert iou[0, 0] == pytest.approx(1.0) and iou[1, 0] == pytest.approx(25 / 175)
    assert iou[0, 1] == pytest.approx(1.0) and iou[1, 1] == pytest.approx(1.0)  # crowd: intersection / detection area


def test_openset_coco_eval_kat(osr):
    from openset_rcnn_amd.host.os_coco_evaluation import OpensetCOCOEval, derive_results
    ev = OpensetCOCOEval(_dataset(), _dets(), known_cat_ids=[1, 2])
    ev.evaluate()
    ev.accumulate()
    st = ev.summarize()
    ap_a = 51 / 101  # precision 1 up to recall 0.5 (one of the two "a" objects is found), 0 beyond
    assert st[0] == pytest.approx((ap_a + 1.0) / 2) and st[1] == pytest.approx((ap_a + 1.0) / 2) and st[2] == pytest.approx((ap_a + 1.0) / 2)
    assert st[3] == pytest.approx(0.0)          # small: only the missed "a" object counts
    assert st[4] == pytest.approx(1.0)          # medium: the found "a" object (its two false positives are out of range -> ignored)
    assert st[5] == pytest.approx(1.0)          # large: the "b" object
    assert st[10] == pytest.approx(0.75)        # AR@100 = mean(0.5, 1.0)
    assert st[14] == pytest.approx(0.25)        # WI at recall 0.8: mean(open-set FP [1, 0]) / mean(TP+FP [3, 1])
    assert st[15] == 1.0                        # A-OSE: one known detection sits on an unknown object
    assert st[16] == pytest.approx(1.0) and st[26] == pytest.approx(1.0)  # unknown AP / AR@100
    assert ev.k_det_as_unk == 1.0               # one unknown detection sits on a known object
    assert float(ev.eval_kdt["ok_det_as_known"][0, 0, 0, -1]) == 1.0      # "a" detection matched to the "b" object
    res = derive_results(st)
    assert res["bbox"]["AP"] == pytest.approx(100 * (ap_a + 1.0) / 2) and res["bbox"]["AOSE"] == 1.0 and res["bbox_unknown"]["AP"] == pytest.approx(100.0)


def test_evaluator_wrapper(osr):
    from openset_rcnn_amd.host.os_coco_evaluation import OpensetCOCOEvaluator
    from openset_rcnn_amd.host.structures import Boxes, Instances
    ev = OpensetCOCOEvaluator(_dataset(), known_names=["a", "b"], contiguous_to_dataset_id={0: 1, 1: 2})
    for img in (1, 2):
        ds = [d for d in _dets() if d["image_id"] == img]
        inst = Instances((300, 300))
        inst.pred_boxes = Boxes(torch.tensor([[d["bbox"][0], d["bbox"][1], d["bbox"][0] + d["bbox"][2], d["bbox"][1] + d["bbox"][3]] for d in ds], dtype=torch.float32))
        inst.scores = torch.tensor([d["score"] for d in ds])
        inst.pred_classes = torch.tensor([d["category_id"] - 1 if d["category_id"] != 1000 else 1000 for d in ds])
        ev.process([{"image_id": img}], [{"instances": inst}])
    res = ev.evaluate()
    assert res["bbox"]["AOSE"] == 1.0 and res["bbox"]["WI"] == pytest.approx(0.25) and res["bbox_unknown"]["AR100"] == pytest.approx(100.0)


def test_graspnet_registration_and_loader(osr, tmp_path):
    import json
    from openset_rcnn_amd.host import datasets as D
    from openset_rcnn_amd.host import config as Cfg
    root = tmp_path / "graspnet_os"
    (root / "annotations").mkdir(parents=True)
    ds = _dataset()
    ds["categories"] = [dict(id=5, name="banana"), dict(id=9, name="mug"), dict(id=2, name="sugar_box")]  # two known names, one unknown
    for a, c in zip(ds["annotations"], (5, 2, 9, 5)):
        a["category_id"] = c
    for im in ds["images"]:
        im["file_name"] = f"{im['id']}.png"
    (root / "annotations" / "toy.json").write_text(json.dumps(ds))
    if "graspnet_toy" not in D.DatasetCatalog:
        D.register_graspnet_instances("graspnet_toy", str(root / "annotations" / "toy.json"), str(root / "images"))
    recs = D.DatasetCatalog["graspnet_toy"]()
    meta = D.MetadataCatalog.get("graspnet_toy")
    assert meta.thing_classes == ["sugar_box", "banana", "mug"] and meta.thing_dataset_id_to_contiguous_id == {2: 0, 5: 1, 9: 2}
    assert recs[0]["annotations"][0] == dict(bbox=[10, 10, 60, 60], bbox_mode="XYXY_ABS", category_id=1, iscrowd=0)
    assert D.graspnet_class_map(meta.thing_classes).tolist() == [1, 2]
    cfg = Cfg.get_cfg()
    Cfg.add_openset_rcnn_config(cfg)
    ev = D.get_evaluator(cfg, "graspnet_toy")
    assert ev.known_ids == [5, 9] and ev.reverse_id_map == {0: 2, 1: 5, 2: 9}
