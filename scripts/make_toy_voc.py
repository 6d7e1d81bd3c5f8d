"""Writes a four-image PASCAL-VOC-layout toy dataset (random pixels, a few boxes) under <root>/VOC2007, for rehearsing
run_net.py end to end without the real data: python scripts/make_toy_voc.py /tmp/toy && DETECTRON2_DATASETS=/tmp/toy python run_net.py ..."""
import os
import sys

import numpy as np
from PIL import Image

root = sys.argv[1]
d = os.path.join(root, "VOC2007")
for sub in ("Annotations", "ImageSets/Main", "JPEGImages"):
    os.makedirs(os.path.join(d, sub), exist_ok=True)
g = np.random.default_rng(0)
objs = {"i0": [("aeroplane", (9, 9, 60, 70)), ("sheep", (70, 20, 120, 90))], "i1": [("bicycle", (20, 10, 100, 80))],
        "i2": [("cat", (5, 5, 50, 50)), ("dog", (60, 30, 125, 90))], "i3": [("person", (30, 8, 90, 92))]}
for k, v in objs.items():
    Image.fromarray(g.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(os.path.join(d, "JPEGImages", k + ".jpg"))
    s = "<annotation><size><width>128</width><height>96</height><depth>3</depth></size>"
    for name, b in v:
        s += (f"<object><name>{name}</name><difficult>0</difficult><bndbox><xmin>{b[0]}</xmin><ymin>{b[1]}</ymin><xmax>{b[2]}</xmax>"
              f"<ymax>{b[3]}</ymax></bndbox></object>")
    open(os.path.join(d, "Annotations", k + ".xml"), "w").write(s + "</annotation>")
for split in ("train", "test"):
    open(os.path.join(d, "ImageSets", "Main", split + ".txt"), "w").write("\n".join(objs) + "\n")
