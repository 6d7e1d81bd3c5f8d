"""Diagnostic: does splitting the batch over streams change the detections, with and without the conv2 -> conv3 chain?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine  # noqa: E402
from openset_rcnn_amd.host.weights import random_params  # noqa: E402

DEV = "cuda:0"
params = random_params(0)
g = torch.Generator().manual_seed(7)
images = torch.randint(0, 256, (2, 3, 250, 330), generator=g, dtype=torch.uint8)
sizes = [(250, 330), (240, 300)]
imgs = torch.cat([images, images.flip(0)]).to(DEV)
hw = torch.tensor(sizes + sizes[::-1], dtype=torch.int32, device=DEV)
for dt in (torch.float16, torch.bfloat16):
    eng = OpensetRCNNEngine(params, dtype=dt, device=DEV)
    for chain in (True, False):
        eng.chain_res3 = chain
        bad = {2: 0, 4: 0}
        badres3 = 0
        for rep in range(10):
            a = eng.forward_device(imgs, hw, 256, 352)
            torch.cuda.synchronize()
            for ns in (2, 4):
                b = eng.forward_device_streams(imgs, hw, 256, 352, nstreams=ns)
                torch.cuda.synchronize()
                bad[ns] += int(not all(torch.equal(x, y) for x, y in zip(a, b)))
            # res3 of the whole batch against one image at a time (backbone only)
            k4, k1 = {}, {}
            eng._backbone(imgs, 256, 352, k4)
            r1 = []
            for i in range(4):
                k1 = {}
                eng._backbone(imgs[i:i + 1], 256, 352, k1)
                r1.append(k1["res3"])
            torch.cuda.synchronize()
            badres3 += int(not torch.equal(k4["res3"], torch.cat(r1)))
        print(dt, "chain" if chain else "separate", "mismatching runs of 10: 2 streams", bad[2], " 4 streams", bad[4], " res3 batch-vs-single", badres3, flush=True)
