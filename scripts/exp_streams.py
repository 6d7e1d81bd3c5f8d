"""Experiment: does splitting the 16-image batch over 2-4 HIP streams (concurrent kernels) raise throughput?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine
from openset_rcnn_amd.host.weights import random_params
eng = OpensetRCNNEngine(random_params(0), device="cuda:0")
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (16, 3, 800, 1333), generator=g, dtype=torch.uint8).cuda()
for ns in (1, 2, 4):
    per = 16 // ns
    streams = [torch.cuda.Stream() for _ in range(ns)]
    chunks = [images[i * per:(i + 1) * per].contiguous() for i in range(ns)]
    hws = [torch.tensor([(800, 1333)] * per, dtype=torch.int32, device="cuda") for _ in range(ns)]
    def step():
        for s, c, h in zip(streams, chunks, hws):
            with torch.cuda.stream(s):
                eng.forward_device(c, h, 800, 1344)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"streams={ns} batch/stream={per}: {dt*1e3:.2f} ms/step  {16/dt:.1f} img/s")
