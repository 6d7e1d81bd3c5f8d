"""The command-line driver (run_net.py; interface of the reference's train.py:211-306): argument rules, config set-up and
the --resume_test path on CPU; a tiny train -> checkpoint -> resume -> evaluate run on the GPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import run_net  # noqa: E402


def test_flags_mirror_the_reference_driver(osr):
    a = run_net.parse_args(["--config-file", "x.yaml", "--eval-only", "--num-gpus", "4", "--opendet-benchmark", "MODEL.WEIGHTS", "w.pth"])
    assert a.config_file == "x.yaml" and a.eval_only and a.num_gpus == 4 and a.opendet_benchmark and a.opts == ["MODEL.WEIGHTS", "w.pth"]
    assert not a.resume and not a.resume_test and a.test_iter == 0 and a.eval_type == "openset"
    with pytest.raises(SystemExit):  # train.py:292-293
        run_net.parse_args(["--resume_test", "--opendet-benchmark"])
    with pytest.raises(SystemExit):
        run_net.parse_args(["--test_iter", "5", "--opendet-benchmark"])
    with pytest.raises(SystemExit):  # one node only
        run_net.parse_args(["--num-machines", "2"])


def test_setup_merges_yaml_overrides_and_benchmark_flag(osr, tmp_path):
    a = run_net.parse_args(["--config-file", os.path.join(ROOT, "configs", "graspnet.yaml"), "--opendet-benchmark",
                            "OUTPUT_DIR", str(tmp_path / "out"), "SOLVER.BASE_LR", "0.01"])
    cfg = run_net.setup(a)
    assert cfg.is_frozen() and cfg.OPENDET_BENCHMARK is True and cfg.SOLVER.BASE_LR == 0.01
    assert cfg.MODEL.ROI_HEADS.NUM_KNOWN_CLASSES == 28 and cfg.MODEL.PLN.UNK_THR == 0.09 and cfg.MODEL.RPN.HEAD_NAME == "ClsFreeRPNHead"
    assert cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS == [[1.0]] and cfg.DATASETS.TRAIN == ("graspnet_train",)
    assert os.path.exists(tmp_path / "out" / "config.json")
    with pytest.raises(AttributeError):
        cfg.SOLVER.BASE_LR = 1.0


def test_num_gpus_relaunches_one_process_per_gpu(osr, monkeypatch):
    """--num-gpus N without a launcher: the driver becomes the parent of a torch.distributed.run job (one rank per GPU,
    rendezvous on 127.0.0.1) before it imports anything that touches the GPU, and returns the job's exit code."""
    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(run_net.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["run_net.py", "--num-gpus", "4", "--eval-only", "--config-file", "c.yaml", "MODEL.WEIGHTS", "w.pth"])
    assert run_net.main(["--num-gpus", "4", "--eval-only", "--config-file", "c.yaml", "MODEL.WEIGHTS", "w.pth"]) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == ["--num-gpus", "4", "--eval-only", "--config-file", "c.yaml", "MODEL.WEIGHTS", "w.pth"]  # the ranks see the same flags
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "run_net.py"


def _coco_gt():
    cats = [{"id": 1, "name": "banana"}, {"id": 2, "name": "mug"}, {"id": 7, "name": "novel_thing"}]
    imgs = [{"id": 10, "height": 100, "width": 200, "file_name": "a.jpg"}, {"id": 11, "height": 100, "width": 200, "file_name": "b.jpg"}]
    anns = [{"id": 1, "image_id": 10, "category_id": 1, "bbox": [10, 10, 40, 40], "area": 1600, "iscrowd": 0},
            {"id": 2, "image_id": 10, "category_id": 7, "bbox": [100, 10, 50, 50], "area": 2500, "iscrowd": 0},
            {"id": 3, "image_id": 11, "category_id": 2, "bbox": [20, 20, 60, 60], "area": 3600, "iscrowd": 0}]
    return {"images": imgs, "annotations": anns, "categories": cats}


def test_coco_evaluator_writes_detections_and_rescoring_them_gives_the_same_result(osr, tmp_path):
    """--resume_test (train.py:97-98; os_coco_evaluation.py:156-190): the detections file a run leaves behind scores identically."""
    from openset_rcnn_amd.host.os_coco_evaluation import OpensetCOCOEvaluator
    from openset_rcnn_amd.host.structures import Boxes, Instances
    out = str(tmp_path / "inference")
    ev = OpensetCOCOEvaluator(_coco_gt(), ["banana", "mug"], {0: 1, 1: 2}, output_dir=out)

    def inst(boxes, scores, classes):
        i = Instances((100, 200))
        i.pred_boxes, i.scores, i.pred_classes = Boxes(torch.tensor(boxes, dtype=torch.float32)), torch.tensor(scores), torch.tensor(classes)
        return {"instances": i}

    ev.process([{"image_id": 10}], [inst([[10, 10, 50, 50], [100, 10, 150, 60]], [0.9, 0.8], [0, 1000])])
    ev.process([{"image_id": 11}], [inst([[20, 20, 80, 80]], [0.7], [1])])
    first = ev.evaluate()
    dets = json.load(open(os.path.join(out, "coco_instances_results.json")))
    assert len(dets) == 3 and {d["image_id"] for d in dets} == {10, 11}
    for tag in ("known", "unknown"):
        assert np.load(os.path.join(out, f"{tag}_precision_bbox.npy")).ndim >= 4
    again = OpensetCOCOEvaluator(_coco_gt(), ["banana", "mug"], {0: 1, 1: 2}, output_dir=out).evaluate(resume=True)
    assert again.keys() == first.keys()
    for grp in first:
        assert again[grp].keys() == first[grp].keys()
        for m, v in first[grp].items():
            assert (np.isnan(v) and np.isnan(again[grp][m])) or again[grp][m] == v, (grp, m)
    with pytest.raises(ValueError):
        OpensetCOCOEvaluator(_coco_gt(), ["banana", "mug"]).evaluate(resume=True)


def _xml(objs, h, w):
    s = f"<annotation><size><width>{w}</width><height>{h}</height><depth>3</depth></size>"
    for name, box in objs:
        s += (f"<object><name>{name}</name><difficult>0</difficult><bndbox><xmin>{box[0]}</xmin><ymin>{box[1]}</ymin>"
              f"<xmax>{box[2]}</xmax><ymax>{box[3]}</ymax></bndbox></object>")
    return s + "</annotation>"


@pytest.fixture()
def toy_voc_root(tmp_path):
    """Four 96x128 JPEGs in detectron2's datasets/VOC2007 layout; the same ids serve as train and test split."""
    from PIL import Image
    d = tmp_path / "datasets" / "VOC2007"
    for sub in ("Annotations", "ImageSets/Main", "JPEGImages"):
        (d / sub).mkdir(parents=True)
    g = np.random.default_rng(0)
    objs = {"i0": [("aeroplane", (9, 9, 60, 70)), ("sheep", (70, 20, 120, 90))], "i1": [("bicycle", (20, 10, 100, 80))],
            "i2": [("cat", (5, 5, 50, 50)), ("dog", (60, 30, 125, 90))], "i3": [("person", (30, 8, 90, 92))]}
    for k, v in objs.items():
        Image.fromarray(g.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(d / "JPEGImages" / f"{k}.jpg")
        (d / "Annotations" / f"{k}.xml").write_text(_xml(v, 96, 128))
    for split in ("train", "test"):
        (d / "ImageSets" / "Main" / f"{split}.txt").write_text("\n".join(objs) + "\n")
    return str(tmp_path / "datasets")


@pytest.mark.gpu
def test_train_checkpoint_resume_and_evaluate_on_a_toy_dataset(osr, toy_voc_root, tmp_path, monkeypatch):
    monkeypatch.setenv("DETECTRON2_DATASETS", toy_voc_root)
    out = str(tmp_path / "out")
    common = ["--config-file", os.path.join(ROOT, "configs", "voc_coco.yaml"), "--opendet-benchmark", "--test-batch", "2"]
    opts = ["OUTPUT_DIR", out, "SEED", "3", "DATASETS.TRAIN", "('voc_2007_train',)", "DATASETS.TEST", "('voc_2007_test',)",
            "SOLVER.IMS_PER_BATCH", "2", "SOLVER.BASE_LR", "0.00005", "SOLVER.WARMUP_ITERS", "0", "SOLVER.CHECKPOINT_PERIOD", "2",
            "INPUT.MIN_SIZE_TRAIN", "(96,)", "INPUT.MAX_SIZE_TRAIN", "128", "INPUT.MIN_SIZE_TEST", "96", "INPUT.MAX_SIZE_TEST", "128"]
    assert run_net.main(common + opts + ["SOLVER.MAX_ITER", "3"]) == 0
    assert os.path.exists(os.path.join(out, "model_0000001.pth")) and os.path.exists(os.path.join(out, "model_final.pth"))
    assert open(os.path.join(out, "last_checkpoint")).read().strip() == "model_final.pth"
    blob = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert blob["iteration"] == 2 and any(k.startswith("roi_heads.") for k in blob["model"]) and len(blob["momentum"]) > 50
    assert all(torch.isfinite(v).all() for v in blob["model"].values() if v.is_floating_point())
    # the per-class detection files of the evaluator are left in the inference folder (pascal_voc_evaluation.py:121-136)
    assert os.path.isdir(os.path.join(out, "inference", "voc_2007_test", "Final"))
    # resume: continues at iteration 3 from the saved weights and momentum, runs to 4
    assert run_net.main(common + ["--resume"] + opts + ["SOLVER.MAX_ITER", "4"]) == 0
    blob2 = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert blob2["iteration"] == 3
    k = "roi_heads.box_head.fc2.weight"
    assert not torch.equal(blob["model"][k], blob2["model"][k])
    # evaluation only, from the checkpoint file
    assert run_net.main(common + ["--eval-only"] + opts + ["MODEL.WEIGHTS", os.path.join(out, "model_final.pth")]) == 0


@pytest.fixture()
def toy_graspnet_root(tmp_path):
    """GraspNet layout of the reference (datasets/graspnet_os/{annotations,images}): four 96x128 JPEGs, the 88-category table with
    the reference's known names among them, annotations of known and unknown categories; the same images serve every split."""
    import json
    from PIL import Image
    from openset_rcnn_amd.host.datasets import GRASPNET_KNOWN_CATEGORIES, GRASPNET_SPLITS
    root = tmp_path / "datasets" / "graspnet_os"
    (root / "annotations").mkdir(parents=True)
    (root / "images").mkdir(parents=True)
    g = np.random.default_rng(1)
    names = []
    known = iter(GRASPNET_KNOWN_CATEGORIES)
    for i in range(88):  # known categories at every third slot, so that dataset ids and known indices differ (class_map is not identity)
        names.append(next(known) if i % 3 == 1 and len([n for n in names if n in GRASPNET_KNOWN_CATEGORIES]) < 28 else f"other_{i}")
    cats = [dict(id=i + 1, name=n) for i, n in enumerate(names)]
    kid = [c["id"] for c in cats if c["name"] in GRASPNET_KNOWN_CATEGORIES]
    uid = [c["id"] for c in cats if c["name"] not in GRASPNET_KNOWN_CATEGORIES]
    images, anns = [], []
    for i in range(4):
        Image.fromarray(g.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(root / "images" / f"{i}.jpg")
        images.append(dict(id=i + 1, file_name=f"{i}.jpg", height=96, width=128))
        for j, (cid, box) in enumerate(((kid[i], [8, 10, 50, 60]), (kid[i + 4], [64, 20, 56, 64]), (uid[i], [30, 40, 40, 40]))):
            anns.append(dict(id=len(anns) + 1, image_id=i + 1, category_id=cid, bbox=box, area=box[2] * box[3], iscrowd=0))
    blob = json.dumps(dict(images=images, annotations=anns, categories=cats))
    for jf in GRASPNET_SPLITS.values():
        (root / "annotations" / jf).write_text(blob)
    return str(tmp_path / "datasets")


@pytest.mark.gpu
def test_graspnet_configuration_trains_and_evaluates_on_a_coco_layout_toy_set(osr, toy_graspnet_root, tmp_path, monkeypatch):
    """SURVEY.md 8f rank 4 end to end: configs/graspnet.yaml (28 known of 88 classes, the sparse class_id map of
    prototype_learning_network.py:80-95, COCO-layout registration, the open-set COCO-style evaluator) through run_net.py."""
    monkeypatch.setenv("DETECTRON2_DATASETS", toy_graspnet_root)
    out = str(tmp_path / "out")
    common = ["--config-file", os.path.join(ROOT, "configs", "graspnet.yaml"), "--test-batch", "2"]
    opts = ["OUTPUT_DIR", out, "SEED", "5", "DATASETS.TEST", "('graspnet_test_1',)", "SOLVER.IMS_PER_BATCH", "2", "SOLVER.BASE_LR", "0.00005",
            "SOLVER.WARMUP_ITERS", "0", "SOLVER.CHECKPOINT_PERIOD", "0", "SOLVER.MAX_ITER", "2", "INPUT.MIN_SIZE_TRAIN", "(96,)",
            "INPUT.MAX_SIZE_TRAIN", "128", "INPUT.MIN_SIZE_TEST", "96", "INPUT.MAX_SIZE_TEST", "128"]
    assert run_net.main(common + opts) == 0
    blob = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert blob["iteration"] == 1 and blob["model"]["roi_heads.dml.representatives"].shape == (28, 256)
    assert blob["model"]["roi_heads.softmaxcls.cls_score.weight"].shape == (29, 1024)
    assert all(torch.isfinite(v).all() for v in blob["model"].values() if v.is_floating_point())
    # the evaluator kept the detections file the reference's --resume_test re-scores (os_coco_evaluation.py)
    inf = os.path.join(out, "inference", "graspnet_test_1", "Final")
    assert os.path.isdir(inf) and any(f.endswith(".json") for f in os.listdir(inf))
    assert run_net.main(common + ["--resume_test"] + opts) == 0


@pytest.mark.gpu
def test_two_rank_training_rehearsal_over_gloo(osr, toy_voc_root, tmp_path):
    """The N > 1 training path end to end with two processes on this one GPU (gloo instead of RCCL, which needs one GPU per rank):
    torchrun launch, sharded train loader, the bucketed gradient all-reduce issued from inside the HIP backward (async_op on the
    device buffer), the loss reduce of train.py:139, rank-0 checkpoints, sharded evaluation. Both ranks must finish and agree."""
    import subprocess
    out = str(tmp_path / "out2")
    env = dict(os.environ, DETECTRON2_DATASETS=toy_voc_root, OSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "run_net.py"), "--config-file", os.path.join(ROOT, "configs", "voc_coco.yaml"), "--opendet-benchmark", "--test-batch", "2",
           "OUTPUT_DIR", out, "SEED", "3", "DATASETS.TRAIN", "('voc_2007_train',)", "DATASETS.TEST", "('voc_2007_test',)", "SOLVER.IMS_PER_BATCH", "2",
           "SOLVER.BASE_LR", "0.00005", "SOLVER.WARMUP_ITERS", "0", "SOLVER.CHECKPOINT_PERIOD", "0", "SOLVER.MAX_ITER", "3", "INPUT.MIN_SIZE_TRAIN", "(96,)",
           "INPUT.MAX_SIZE_TRAIN", "128", "INPUT.MIN_SIZE_TEST", "96", "INPUT.MAX_SIZE_TEST", "128"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    blob = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert blob["iteration"] == 2 and all(torch.isfinite(v).all() for v in blob["model"].values() if v.is_floating_point())
    assert os.path.isdir(os.path.join(out, "inference", "voc_2007_test", "Final"))
