"""Experiment driver: bench.py's inference pass with the shortcut + conv1 pair launch on / off (PAIR=0 disables it), same box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
import bench
from openset_rcnn_amd.host import engine
if os.environ.get("PAIR", "1") == "0":
    engine.OpensetRCNNEngine._shortcut_conv1_one_launch = lambda self, x, pre, stride: None
sys.exit(bench.main(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-parity", "--no-pcie", "--no-pmc", "--no-train-step", "--layers"]))
