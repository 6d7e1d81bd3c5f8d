"""Open-set VOC evaluator (host/evaluation.py): hand-computed cases on a synthetic VOC tree written to tmp_path.
Reference protocol: openset_rcnn/evaluation/pascal_voc_evaluation.py (no fixtures exist there; these are our own KATs)."""
import os

import numpy as np
import pytest
import torch


def _xml(objs, h=100, w=200):
    s = f"<annotation><size><width>{w}</width><height>{h}</height><depth>3</depth></size>"
    for name, box, diff in objs:
        s += (f"<object><name>{name}</name><difficult>{diff}</difficult><bndbox><xmin>{box[0]}</xmin><ymin>{box[1]}</ymin>"
              f"<xmax>{box[2]}</xmax><ymax>{box[3]}</ymax></bndbox></object>")
    return s + "</annotation>"


@pytest.fixture()
def voc(tmp_path, osr):
    d = tmp_path / "voc_coco"
    (d / "Annotations").mkdir(parents=True)
    (d / "ImageSets" / "Main").mkdir(parents=True)
    annos = {
        "a": [("aeroplane", (11, 11, 60, 60), 0), ("zebra", (101, 11, 150, 60), 0)],      # zebra is outside the known set -> unknown
        "b": [("aeroplane", (21, 21, 80, 80), 0), ("aeroplane", (101, 21, 160, 80), 1)],  # second one is "difficult"
        "c": [("bicycle", (11, 11, 50, 90), 0)],
    }
    for k, v in annos.items():
        (d / "Annotations" / f"{k}.xml").write_text(_xml(v))
    (d / "ImageSets" / "Main" / "toy.txt").write_text("a\nb\nc\n")
    return str(d)


def _inst(osr, boxes, scores, classes):
    from openset_rcnn_amd.host.structures import Boxes, Instances
    i = Instances((100, 200))
    i.pred_boxes = Boxes(torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4))
    i.scores = torch.tensor(scores, dtype=torch.float32)
    i.pred_classes = torch.tensor(classes, dtype=torch.int64)
    return {"instances": i}


def test_voc_ap_closed_forms(osr):
    from openset_rcnn_amd.host.evaluation import voc_ap
    assert voc_ap(np.array([]), np.array([])) == 0.0
    assert voc_ap(np.array([0.5, 1.0]), np.array([1.0, 1.0])) == pytest.approx(1.0)
    # envelope: precision dips to 0.5 at recall 0.5 and recovers to 2/3 at recall 1 -> 0.5*1 ... (0..0.5 at p=1, 0.5..1 at p=2/3)
    assert voc_ap(np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2 / 3])) == pytest.approx(0.5 * 1.0 + 0.5 * 2 / 3)
    assert voc_ap(np.array([0.5, 1.0]), np.array([1.0, 1.0]), use_07_metric=True) == pytest.approx(1.0)


def test_openset_voc_evaluator_kat(osr, voc):
    from openset_rcnn_amd.host.datasets import VOC_COCO_CATEGORIES
    from openset_rcnn_amd.host.evaluation import PascalVOCDetectionEvaluator
    names = VOC_COCO_CATEGORIES[:2] + ["zebra", "unknown"]  # 2 known classes, one unseen category, unknown last
    ev = PascalVOCDetectionEvaluator(voc, "toy", names, 2, output_dir=os.path.dirname(voc))
    ev.reset()
    # boxes are 0-based (loader convention): GT (11,11,60,60) is predicted as (10,10,60,60)
    ev.process([{"image_id": "a"}], [_inst(osr, [[10, 10, 60, 60], [100, 10, 150, 60], [100, 10, 150, 60]], [0.9, 0.8, 0.7], [0, 0, 3])])
    #   a: aeroplane TP (0.9); aeroplane on the zebra = FP that overlaps an unknown GT (A-OSE 1); unknown det on the zebra = unknown TP
    ev.process([{"image_id": "b"}], [_inst(osr, [[20, 20, 80, 80], [20, 20, 80, 80], [100, 20, 160, 80]], [0.95, 0.6, 0.5], [0, 0, 0])])
    #   b: TP (0.95), duplicate = FP (0.6), the difficult GT match is ignored (neither TP nor FP)
    ev.process([{"image_id": "c"}], [_inst(osr, [[150, 50, 190, 90]], [0.4], [1])])  # bicycle detection far from the GT: FP, recall 0
    res = ev.evaluate()
    # aeroplane: sorted conf .95 TP, .9 TP, .8 FP, .6 FP, .5 ignored; npos = 2 -> rec [.5,1,1,1,1], prec [1,1,2/3,.5,.5] -> AP 1.0
    # bicycle: AP 0; zebra / unknown class names: zebra has no GT under its own name (renamed) and no dets -> 0; unknown: AP 1
    assert res["AP@K"] == pytest.approx(50.0) and res["AP@U"] == pytest.approx(100.0)
    assert res["mAP"] == pytest.approx((100 + 0 + 0 + 100) / 4)
    assert res["AOSE"] == 1.0
    assert res["R@K"] == pytest.approx((100 + 0) / 2) and res["P@K"] == pytest.approx((50 + 0) / 2)
    assert res["R@U"] == pytest.approx(100.0) and res["P@U"] == pytest.approx(100.0)
    # WI at recall 0.8: aeroplane's closest recall is index 1 (rec 1.0; |0.5-.8| = .3 > .2): TP+FP = 2, open-set FP = 0;
    # bicycle: one det, rec 0: TP+FP = 1, open-set FP = 0 -> WI = 0
    assert res["WI"] == 0.0
    assert os.path.exists(os.path.join(os.path.dirname(voc), "pascal_voc_eval", "aeroplane.txt"))
    # text rounding is part of the protocol: the stored record carries 3 / 1 decimals and the +1 shift
    assert ev._predictions[0][0] == "a 0.900 11.0 11.0 60.0 60.0"


def test_wi_counts_open_set_false_positives(osr, voc):
    from openset_rcnn_amd.host.evaluation import PascalVOCDetectionEvaluator
    ev = PascalVOCDetectionEvaluator(voc, "toy", ["aeroplane", "bicycle", "unknown"], 2)
    # the aeroplane-on-zebra false positive now outranks the second true positive, so it sits inside the recall-0.8 prefix
    ev.process([{"image_id": "a"}], [_inst(osr, [[10, 10, 60, 60], [100, 10, 150, 60]], [0.9, 0.85], [0, 0])])
    ev.process([{"image_id": "b"}], [_inst(osr, [[20, 20, 80, 80]], [0.8], [0])])
    res = ev.evaluate()
    # aeroplane: rec [.5,.5,1.0] -> closest to 0.8 is index 2 (|1-.8| = .2 < .3): TP+FP = 3, open-set FP = 1; bicycle: no dets (skipped)
    assert res["WI"] == pytest.approx(100.0 / 3, abs=0.01) and res["AOSE"] == 1.0
    assert res["AP@K"] == pytest.approx((100 * (0.5 * 1.0 + 0.5 * 2 / 3) + 0) / 2, abs=0.01)


def test_dataset_registration_and_loader(osr, voc):
    from openset_rcnn_amd.host import datasets as D
    if "toy_reg" not in D.DatasetCatalog:
        D.register_voc_coco("toy_reg", voc, "toy", 2012)
    recs = D.DatasetCatalog["toy_reg"]()
    assert [r["image_id"] for r in recs] == ["a", "b", "c"] and recs[0]["height"] == 100 and recs[0]["width"] == 200
    assert recs[0]["annotations"][0] == dict(category_id=0, bbox=[10.0, 10.0, 60.0, 60.0], bbox_mode="XYXY_ABS")
    assert recs[0]["annotations"][1]["category_id"] == D.VOC_COCO_CATEGORIES.index("zebra")
    meta = D.MetadataCatalog.get("toy_reg")
    assert meta.evaluator_type == "pascal_voc" and len(meta.thing_classes) == 81 and meta.thing_classes[-1] == "unknown"
    from openset_rcnn_amd.host import config as Cfg
    cfg = Cfg.get_cfg()
    Cfg.add_openset_rcnn_config(cfg)
    ev = D.get_evaluator(cfg, "toy_reg")
    assert ev.num_known_classes == 20 and ev.unknown_class_index == 80


@pytest.mark.gpu
def test_inference_on_dataset_with_the_hip_model(osr, voc, tmp_path):
    """The glue train.py:96 exercises: GeneralizedRCNN (HIP path) -> inference_on_dataset -> open-set VOC evaluator, on a synthetic
    VOC tree with random images. Random weights detect nothing meaningful; the point is that the pipeline runs end to end and the
    evaluator returns the full metric dict."""
    from openset_rcnn_amd.host import config as Cfg, datasets as D, modeling as M
    from openset_rcnn_amd.host.evaluation import inference_on_dataset
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    cfg = Cfg.get_cfg()
    Cfg.add_openset_rcnn_config(cfg)
    y = tmp_path / "m.yaml"
    y.write_text("MODEL:\n  META_ARCHITECTURE: GeneralizedRCNN\n  DEVICE: cuda\n  BACKBONE:\n    NAME: build_resnet_fpn_backbone\n"
                 "  RESNETS:\n    OUT_FEATURES: [res2, res3, res4, res5]\n  FPN:\n    IN_FEATURES: [res2, res3, res4, res5]\n"
                 "  ANCHOR_GENERATOR:\n    SIZES: [[32], [64], [128], [256], [512]]\n    ASPECT_RATIOS: [[1.0]]\n"
                 "  PROPOSAL_GENERATOR:\n    NAME: ClsFreeRPN\n  RPN:\n    HEAD_NAME: ClsFreeRPNHead\n    IN_FEATURES: [p2, p3, p4, p5, p6]\n"
                 "    PRE_NMS_TOPK_TRAIN: 2000\n    PRE_NMS_TOPK_TEST: 1000\n"
                 "  ROI_HEADS:\n    NAME: OpensetROIHeads\n    IN_FEATURES: [p2, p3, p4, p5]\n    NUM_CLASSES: 81\n    KNOWN_TOPK: 50\n    UNKNOWN_TOPK: 50\n"
                 "    UNKNOWN_SCORE_THRESH: 0.0\n"
                 "  ROI_BOX_HEAD:\n    NAME: FastRCNNConvFCHead\n    NUM_FC: 2\n    POOLER_RESOLUTION: 7\n    CLS_AGNOSTIC_BBOX_REG: True\n"
                 "TEST:\n  DETECTIONS_PER_IMAGE: 1000\nOPENDET_BENCHMARK: True\n")
    cfg.merge_from_file(str(y))
    model = M.build_model(cfg).eval()
    if "toy_gpu" not in D.DatasetCatalog:
        D.register_voc_coco("toy_gpu", voc, "toy", 2012)
    ev = D.get_evaluator(cfg, "toy_gpu", str(tmp_path))
    g = torch.Generator().manual_seed(0)
    recs = D.DatasetCatalog["toy_gpu"]()
    batches = [[{"image": torch.randint(0, 256, (3, r["height"], r["width"]), generator=g, dtype=torch.uint8), "image_id": r["image_id"],
                 "height": r["height"], "width": r["width"]}] for r in recs]
    res = inference_on_dataset(model, batches, ev)
    assert set(res) == {"mAP", "WI", "AOSE", "AP@K", "P@K", "R@K", "AP@U", "P@U", "R@U"}
    assert sum(len(v) for v in ev._predictions.values()) > 0  # the detector produced (meaningless) detections that were scored
