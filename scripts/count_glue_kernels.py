"""How many tensor-library / runtime kernels (copies, fills, cat, ...) ride in one inference pass? Run under
rocprofv3 --kernel-trace --stats with PASSES=1 and PASSES=5 and divide the difference of the call counts by 4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg._lib.load()
from openset_rcnn_amd.host.engine import OpensetRCNNEngine  # noqa: E402
from openset_rcnn_amd.host.weights import random_params  # noqa: E402

dev = "cuda:0"
eng = OpensetRCNNEngine(random_params(0), dtype=torch.float16, device=dev)
g = torch.Generator().manual_seed(1)
images = torch.randint(0, 256, (4, 3, 256, 352), generator=g, dtype=torch.uint8).to(dev)
hw = torch.tensor([(256, 352)] * 4, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("PASSES", "1"))):
    eng.forward_device(images, hw, 256, 352)
torch.cuda.synchronize()
