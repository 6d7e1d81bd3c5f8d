"""Experiment driver: bench.py's config-4 leg (GraspNet train step, 8 frames) with the multi-level launches on / off (LEVELS=0 disables them)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
import bench
from openset_rcnn_amd.host import engine
if os.environ.get("LEVELS", "1") == "0":
    engine.OpensetRCNNEngine._fpn_outputs_one_launch = lambda self, lats: None
    engine.OpensetRCNNEngine._rpn_levels_fused = lambda self, *a: False
out = bench.config4_leg(torch.float16, "cuda:0", int(os.environ.get("STEPS", 8)), 3)
print(json.dumps({k: out[k] for k in ("ms_per_iter", "images_per_sec") if k in out} | {"err": out.get("error")}))
