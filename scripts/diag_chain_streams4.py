"""Diagnostic: full forward on two streams; for every res3 block keep the chain's operands (o, sc) and its output y, then
recompute y from the kept operands after a sync: are the operands intact, and is a recomputation from them right?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine  # noqa: E402
from openset_rcnn_amd.host.weights import random_params  # noqa: E402
from openset_rcnn_amd.host import ops  # noqa: E402

DEV = "cuda:0"
params = random_params(0)
g = torch.Generator().manual_seed(7)
images = torch.randint(0, 256, (2, 3, 250, 330), generator=g, dtype=torch.uint8)
sizes = [(250, 330), (240, 300)]
imgs = torch.cat([images, images.flip(0)]).to(DEV)
hw = torch.tensor(sizes + sizes[::-1], dtype=torch.int32, device=DEV)
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float16
eng = OpensetRCNNEngine(params, dtype=dt, device=DEV)
orig_chain = ops.conv2d_chain
clog = []


def spy_chain(x, w2, b2, w3, b3, res, stride=1, pad=0):
    y = orig_chain(x, w2, b2, w3, b3, res, stride, pad)
    clog.append((x, w2, b2, w3, b3, res, y))
    return y


ops.conv2d_chain = spy_chain
nbad = 0
for rep in range(40):
    clog.clear()
    eng.forward_device_streams(imgs, hw, 256, 352, nstreams=2)
    torch.cuda.synchronize()
    for k, (x, w2, b2, w3, b3, res, y) in enumerate(clog):
        o2 = ops.conv2d(x, w2, b2, 1, 1, relu=True)
        sep = ops.conv2d(o2, w3, b3, relu=True, residual=res, res_mode=1)
        again = orig_chain(x, w2, b2, w3, b3, res, 1, 1)
        torch.cuda.synchronize()
        if not torch.equal(y, sep):
            nbad += 1
            d = (y.float() - sep.float()).abs().view(-1, 512)
            rows = torch.nonzero(d.amax(1) > 0).flatten()
            print(f"rep {rep} chain launch {k} (n={x.shape[0]}): y != recomputation from the kept operands; {len(rows)} rows {rows.tolist()[:12]}; "
                  f"chain again == separate: {torch.equal(again, sep)}", flush=True)
            r0 = int(rows[0])
            yy, ss, rr = y.view(-1, 512), sep.view(-1, 512), res.view(-1, 512)
            bad_c = torch.nonzero(d[r0] > 0).flatten().tolist()
            print(f"    row {r0}: {len(bad_c)} bad cols; y[:6]={yy[r0, :6].float().tolist()} want {ss[r0, :6].float().tolist()} res {rr[r0, :6].float().tolist()}")
print("bad chain launches:", nbad)
