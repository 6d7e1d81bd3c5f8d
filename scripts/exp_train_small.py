"""Times the small kernels of the training step's head / loss section at the step's shapes (16 images, 89 523 anchors per image,
8 192 sampled RoI rows, 20 known classes, 256-d embeddings): run on the GPU box.  python scripts/exp_train_small.py"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops  # noqa: E402


def timed(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:44s} {a.elapsed_time(b) / reps * 1e3:9.1f} us")


def main():
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    rows = 16 * 89523
    t = torch.randn(rows, 256, generator=g).relu().half().to(dev)
    w_tail = (torch.randn(5, 256, generator=g) * 0.05).to(dev)
    d5 = torch.zeros(rows, 5)
    idx = torch.randperm(rows, generator=g)[:16 * 256]
    d5[idx] = torch.randn(len(idx), 5, generator=g) * 1e-3
    d5 = d5.to(dev)
    timed("cfrpn_tail_bwd (kernel + reduce)", lambda: ops.cfrpn_tail_bwd(t, w_tail, d5))

    m, d, k = 8192, 256, 20
    emb = torch.randn(m, d, generator=g).to(dev)
    protos = torch.randn(k, d, generator=g).to(dev)
    pn = torch.nn.functional.normalize(protos, dim=1)
    cls = torch.full((m,), 81, dtype=torch.int64)
    fg = (torch.arange(m) % 512) < 128  # the sampled lists: per image 512 rows, the foreground quarter first
    cls[fg] = torch.randint(0, k, (int(fg.sum()),), generator=g)
    ious = torch.rand(m, generator=g)
    cls, ious = cls.to(dev), ious.to(dev)
    timed("pln_loss_fwd", lambda: ops.pln_loss_fwd(emb, pn, cls, ious, 0.5, 0.3, 0.6, 1.0))
    timed("pln_loss_bwd (count + rows + protos)", lambda: ops.pln_loss_bwd(emb, protos, cls, ious, 0.5, 0.3, 0.6, 1.0))
    logits = torch.randn(m, k + 1, generator=g).to(dev)
    timed("softmax_ce_loss_fwd", lambda: ops.softmax_ce_loss_fwd(logits, cls, 81, 1.0))
    timed("softmax_ce_loss_bwd", lambda: ops.softmax_ce_loss_bwd(logits, cls, 81, 1.0))
    lab = torch.randint(-1, 2, (16, 89523), generator=g, dtype=torch.int8).to(dev)
    keys = torch.rand(16, 89523, generator=g).to(dev)
    timed("subsample_labels (256 of 89 523, 16 images)", lambda: ops.subsample_labels_(lab.clone(), keys, 256, 0.5))
    dy = torch.randn(16 * 200 * 336, 256, generator=g).half().to(dev)
    timed("bias_grad (1.07 M rows x 256)", lambda: ops.bias_grad(dy))


if __name__ == "__main__":
    main()
